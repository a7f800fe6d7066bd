// x64_bench.hip -- standalone check + timing of csrc/attn_xt64.hip (pre-tiled, LDS-resident K/V, head_dim 64) on the motion-VAE decoder's cross
// attention: n Gaussians (queries shared by the T frames, inner stride 0) x L latents per frame, 12 heads of 64.  Same shapes, data and host
// reference as kvres_bench.hip (csrc/attn.hip's kernel), so the two print comparable lines.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 <flags> [-DX64_ORDER=..] [-DX64_PASSES=..] x64_bench.hip -o x64_<tag>.bin
//   x64_<tag>.bin [iters] [dtype 0 bf16 / 1 fp16] [n] [L] [gain] [force_exact]
#include "../../gvfdiffusion_amd/csrc/attn_xt64.hip"
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

static unsigned short h_f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); }
static float h_bf2f(unsigned short h) { unsigned u = ((unsigned)h) << 16; float f; memcpy(&f, &u, 4); return f; }
static unsigned short h_f2h(float f) { _Float16 h = (_Float16)f; unsigned short u; memcpy(&u, &h, 2); return u; }
static float h_h2f(unsigned short u) { _Float16 h; memcpy(&h, &u, 2); return (float)h; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 5, dt = argc > 2 ? atoi(argv[2]) : 0;
    const int n = argc > 3 ? atoi(argv[3]) : 43648, T = 24, L = argc > 4 ? atoi(argv[4]) : 512, H = 12, d = 64, C = H * d;
    const float gain = argc > 5 ? (float)atof(argv[5]) : 1.f; const int force_exact = argc > 6 ? atoi(argv[6]) : 0;
    auto enc = [&](float f) { return dt ? h_f2h(f) : h_f2bf(f); };
    auto dec = [&](unsigned short u) { return dt ? h_h2f(u) : h_bf2f(u); };
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    // kv rows as the decoder's to_kv GEMM leaves them: row (t * L + key), K at column h * 64, V at column C + h * 64
    std::vector<unsigned short> hq((size_t)n * C), hkv((size_t)T * L * 2 * C);
    for (auto& v : hq) v = enc(nd(rng));
    for (size_t i = 0; i < hkv.size(); ++i) hkv[i] = enc(nd(rng) * ((i % (2 * C)) < (size_t)C && ((i / (2 * C)) & 1) == 0 ? gain : 1.f));
    unsigned short *dq, *dkv, *dout; uint4 *dkt, *dvt; int* dfb;
    const size_t out_elems = (size_t)T * n * C;
    const int n_tiles = (L + 63) / 64;
    const size_t img = (size_t)T * H * n_tiles * 8192;
    CK(hipMalloc(&dq, hq.size() * 2)); CK(hipMalloc(&dkv, hkv.size() * 2)); CK(hipMalloc(&dout, out_elems * 2));
    CK(hipMalloc(&dkt, img)); CK(hipMalloc(&dvt, img)); CK(hipMalloc(&dfb, 4)); CK(hipMemset(dfb, 0, 4));
    CK(hipMemcpy(dq, hq.data(), hq.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dkv, hkv.data(), hkv.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemset(dout, 0xff, out_elems * 2));
    const int64_t qs[4] = {(int64_t)n * C, 0, C, d}, os[4] = {(int64_t)T * n * C, (int64_t)n * C, C, d};
    const float scale = 1.0f / sqrtf((float)d), ksc = scale * 1.4426950408889634f;
#ifdef X64_TIMING
    CK(hipMalloc(&g_x64_dbg, 128 * 8)); CK(hipMemset(g_x64_dbg, 0, 128 * 8));
#endif
    int rc = gvf_attn_pack_kv64(dt, dkv, 0, 2 * C, 0, C, T, L, H, ksc, dkt, dvt, nullptr);
    if (rc) { printf("pack rc %d\n", rc); return 1; }
    auto run = [&]() { return gvf_attn_tiled64_fwd(dt, dq, dkt, dvt, dout, 1, T, n, L, H, qs, os, T, 1, force_exact, dfb, nullptr); };
    rc = run();
    if (rc) { printf("rc %d\n", rc); return 1; }
    CK(hipDeviceSynchronize());
    std::vector<unsigned short> hout(out_elems);
    CK(hipMemcpy(hout.data(), dout, out_elems * 2, hipMemcpyDeviceToHost));
    double num = 0, den = 0, maxabs = 0; int nan_count = 0;
    for (int sidx = 0; sidx < 64; ++sidx) {
        const int t = (sidx * 7) % T, h = (sidx * 5) % H;
        const int qi = (int)(((long long)sidx * 7919 + (sidx % 3 == 0 ? n - 1 - sidx : 0)) % n);
        std::vector<double> s(L); double m = -1e300;
        for (int k = 0; k < L; ++k) {
            double acc = 0;
            for (int e = 0; e < d; ++e) acc += (double)dec(hq[(size_t)qi * C + h * d + e]) * dec(enc(dec(hkv[((size_t)t * L + k) * 2 * C + h * d + e]) * ksc));
            s[k] = acc; m = s[k] > m ? s[k] : m;
        }
        double l = 0; std::vector<double> ov(d, 0.0);
        for (int k = 0; k < L; ++k) { const double p = exp2(s[k] - m); l += p; for (int e = 0; e < d; ++e) ov[e] += p * dec(hkv[((size_t)t * L + k) * 2 * C + C + h * d + e]); }
        for (int e = 0; e < d; ++e) {
            const double ref = ov[e] / l, got = dec(hout[((size_t)t * n + qi) * C + h * d + e]);
            if (!(got == got)) ++nan_count;
            num += (got - ref) * (got - ref); den += ref * ref; maxabs = fabs(got - ref) > maxabs ? fabs(got - ref) : maxabs;
        }
    }
#ifdef X64_TIMING
    {
        long long st[8];
        CK(hipMemcpy(st, g_x64_dbg, 64, hipMemcpyDeviceToHost));
        printf("pass %d of workgroup 0, s_memtime ticks: first phase %lld | tile loop %lld | last two phases %lld | guard %lld | normalise + LDS %lld | take q + stores %lld | next pass starts +%lld\n", X64_TIMING,
               st[1] - st[0], st[2] - st[1], st[3] - st[2], st[4] - st[3], st[5] - st[4], st[6] - st[5], st[6] - st[0]);
    }
#endif
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) (void)run();
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) (void)run();
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / iters, tf = 4.0 * (double)T * n * L * C / us / 1e6;
    const double rel = sqrt(num / (den + 1e-300));
    int fb = 0; CK(hipMemcpy(&fb, dfb, 4, hipMemcpyDeviceToHost));
    printf("vae decoder cross attention (tiled64) dt=%d n=%d L=%d gain=%.0f exact=%d order=%d passes=%d: rel_l2 %.3e max_abs %.3e nan %d fallback_waves %d | %8.1f us  %7.1f TFLOP/s (%.1f%% of 2.5 PF)\n",
           dt, n, L, gain, force_exact, X64_ORDER, X64_PASSES, rel, maxabs, nan_count, fb, us, tf, tf / 25.0);
    return (rel < 1e-2 && nan_count == 0) ? 0 : 2;
}
