// pingpong.hip -- two waves per SIMD, each alternating an MFMA part (8 x 32x32x16 + 8 x 4x4x4) and a VALU part (32 v_exp_f32 + 16 v_cvt_pk)
// with a workgroup barrier after every part.  mode 0: both waves of a SIMD run the same part at the same time (lockstep);
// mode 1: waves 4-7 run one part behind waves 0-3 (ping-pong: one wave of a SIMD on the matrix pipe while the other is on the VALU);
// mode 2: no barriers at all (free running, the attention kernel's situation); mode 3: MFMA part only; mode 4: VALU part only.
// hipcc --offload-arch=gfx950 -O3 pingpong.hip -o pingpong.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ void part_mfma(f32x16& c0, f32x16& c1, f32x4& l0, f32x4& l1, bf16x8 a, bf16x8 b, bf16x4 a4, bf16x4 b4) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        l0 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a4, b4, l0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        l1 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a4, b4, l1, 0, 0, 0);
    }
}
__device__ __forceinline__ void part_valu(float (&x)[32], unsigned (&pk)[16]) {
#pragma unroll
    for (int i = 0; i < 32; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[i]) : "v"(x[2 * i]), "v"(x[2 * i + 1]));
}

__global__ __launch_bounds__(512, 1) void k(float* out, int iters, int mode) {
    const int wave = threadIdx.x >> 6, grp = wave >> 2;
    bf16x8 a, b; bf16x4 a4, b4;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.37f + 0.001f * (threadIdx.x % 61)); b[i] = (__bf16)(0.11f * (i + 1)); }
    for (int i = 0; i < 4; ++i) { a4[i] = (__bf16)1.0f; b4[i] = (__bf16)(0.01f * i); }
    f32x16 c0 = {0}, c1 = {0}; f32x4 l0 = {0}, l1 = {0};
    float x[32]; unsigned pk[16];
    for (int i = 0; i < 32; ++i) x[i] = -0.001f * (threadIdx.x + i);
    for (int i = 0; i < 16; ++i) pk[i] = 0;
    if (mode == 0) {
        for (int it = 0; it < iters; ++it) { part_mfma(c0, c1, l0, l1, a, b, a4, b4); __builtin_amdgcn_s_barrier(); part_valu(x, pk); __builtin_amdgcn_s_barrier(); }
    } else if (mode == 1) {
        if (grp == 0) {
            for (int it = 0; it < iters; ++it) { part_mfma(c0, c1, l0, l1, a, b, a4, b4); __builtin_amdgcn_s_barrier(); part_valu(x, pk); __builtin_amdgcn_s_barrier(); }
        } else {
            for (int it = 0; it < iters; ++it) { part_valu(x, pk); __builtin_amdgcn_s_barrier(); part_mfma(c0, c1, l0, l1, a, b, a4, b4); __builtin_amdgcn_s_barrier(); }
        }
    } else if (mode == 2) {
        for (int it = 0; it < iters; ++it) { part_mfma(c0, c1, l0, l1, a, b, a4, b4); part_valu(x, pk); }
    } else if (mode == 3) {
        for (int it = 0; it < iters; ++it) { part_mfma(c0, c1, l0, l1, a, b, a4, b4); }
    } else if (mode == 4) {
        for (int it = 0; it < iters; ++it) { part_valu(x, pk); }
    } else if (mode == 5) {          // free running, the groups start half a period apart
        if (grp == 1) part_valu(x, pk);
        for (int it = 0; it < iters; ++it) { part_mfma(c0, c1, l0, l1, a, b, a4, b4); part_valu(x, pk); }
    } else if (mode == 6) {          // one wave per SIMD does ONLY MFMA parts, the other ONLY VALU parts (the B test of mfma_valu_overlap)
        if (grp == 0) for (int it = 0; it < iters; ++it) part_mfma(c0, c1, l0, l1, a, b, a4, b4);
        else for (int it = 0; it < iters; ++it) part_valu(x, pk);
    }
    // fine-grained orders of the same mix (per iteration and wave: 8 big + 8 small MFMAs, 32 exp, 16 cvt), free running
#define BIG(c_) c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c_, 0, 0, 0);
#define SML(l_) l_ = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a4, b4, l_, 0, 0, 0);
#define EX(i_) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i_]));
#define CV(i_) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[i_]) : "v"(x[2 * (i_)]), "v"(x[2 * (i_) + 1]));
#define FENCE __builtin_amdgcn_sched_barrier(0);
    else if (mode == 7) {            // the attention kernel's order: [8 exp] big [4 cvt, 2 small] big
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                EX(8 * u) EX(8 * u + 1) EX(8 * u + 2) EX(8 * u + 3) EX(8 * u + 4) EX(8 * u + 5) EX(8 * u + 6) EX(8 * u + 7) FENCE
                BIG(c0) FENCE CV(4 * u) CV(4 * u + 1) CV(4 * u + 2) CV(4 * u + 3) SML(l0) SML(l1) FENCE BIG(c1) FENCE
            }
        }
    } else if (mode == 8) {          // uniform: big [4 exp, 2 cvt] small  (every MFMA followed by VALU work of about its own length)
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (u & 1) { BIG(c1) } else { BIG(c0) } FENCE
                EX(4 * u) EX(4 * u + 1) EX(4 * u + 2) EX(4 * u + 3) FENCE
                if (u & 1) { SML(l1) } else { SML(l0) } FENCE
                CV(2 * u) CV(2 * u + 1) FENCE
            }
        }
    } else if (mode == 9) {          // big [4 exp] big-less: small rides directly behind its big
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (u & 1) { BIG(c1) SML(l1) } else { BIG(c0) SML(l0) } FENCE
                EX(4 * u) EX(4 * u + 1) EX(4 * u + 2) EX(4 * u + 3) CV(2 * u) CV(2 * u + 1) FENCE
            }
        }
    } else if (mode == 10) {         // two big back to back, then 8 exp + 4 cvt, smalls in front of the bigs
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                SML(l0) SML(l1) BIG(c0) BIG(c1) FENCE
                EX(8 * u) EX(8 * u + 1) EX(8 * u + 2) EX(8 * u + 3) EX(8 * u + 4) EX(8 * u + 5) EX(8 * u + 6) EX(8 * u + 7)
                CV(4 * u) CV(4 * u + 1) CV(4 * u + 2) CV(4 * u + 3) FENCE
            }
        }
    } else if (mode == 11) {         // as 7 without the small MFMAs
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                EX(8 * u) EX(8 * u + 1) EX(8 * u + 2) EX(8 * u + 3) EX(8 * u + 4) EX(8 * u + 5) EX(8 * u + 6) EX(8 * u + 7) FENCE
                BIG(c0) FENCE CV(4 * u) CV(4 * u + 1) CV(4 * u + 2) CV(4 * u + 3) FENCE BIG(c1) FENCE
            }
        }
    } else if (mode == 12) {         // as 8 without the small MFMAs
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (u & 1) { BIG(c1) } else { BIG(c0) } FENCE
                EX(4 * u) EX(4 * u + 1) EX(4 * u + 2) EX(4 * u + 3) CV(2 * u) CV(2 * u + 1) FENCE
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 32; ++i) s += x[i];
    for (int i = 0; i < 16; ++i) s += (float)pk[i];
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
    for (int i = 0; i < 4; ++i) s += l0[i] + l1[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    float* out; (void)hipMalloc(&out, 256 * 512 * sizeof(float));
    const int N = 20000;
    const char* names[] = {"lockstep (barriers, same part)", "ping-pong (barriers, offset by one part)", "free running", "MFMA part only", "VALU part only",
                           "free running, started half a period apart", "one wave all-MFMA, one wave all-VALU",
                           "[8 exp] big [4 cvt 2 small] big  (attn_xt)", "big [4 exp] small [2 cvt]", "big small [4 exp 2 cvt]", "small small big big [8 exp 4 cvt]",
                           "[8 exp] big [4 cvt] big  (no small)", "big [4 exp 2 cvt]  (no small)"};
    for (int mode = 0; mode < 13; ++mode) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        k<<<256, 512>>>(out, 200, mode); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        k<<<256, 512>>>(out, N, mode);
        (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-50s %8.1f us = %6.1f ns per (MFMA part + VALU part) per wave\n", names[mode], ms * 1e3f, ms * 1e6f / N);
    }
    return 0;
}
