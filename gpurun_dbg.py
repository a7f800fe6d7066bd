import sys, os, time
sys.path.insert(0, '.')
import torch
from gvfdiffusion_amd.ops import dit_ops
cuda = torch.device('cuda:0')
def run(M,N,K,epi,iters=50):
    a = torch.randn((M,K),device=cuda).to(torch.bfloat16); w = torch.randn((N,K),device=cuda).to(torch.bfloat16)
    bias = torch.randn((N,),device=cuda)
    out = torch.zeros((M,N),device=cuda,dtype=torch.float32 if epi>=2 else torch.bfloat16)
    for _ in range(5): dit_ops.gemm_bf16(a,w,bias,out,epi)
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): dit_ops.gemm_bf16(a,w,bias,out,epi)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1)/iters*1e3
    return us, 2.0*M*N*K/us/1e6
for abl in ("0","1","2","3","4","6","7"):
    os.environ["GVF_GEMM_ABLATE"]=abl
    res=[]
    for (M,N,K,epi) in [(12288,1536,512,0),(12288,512,512,3),(12288,2048,512,1),(12288,512,2048,3)]:
        us,tf=run(M,N,K,epi); res.append("%dx%dx%d e%d: %.1fus %.0fTF"%(M,N,K,epi,us,tf))
    print("ablate",abl," | ".join(res))
