// kvres_bench.hip -- standalone check + timing of csrc/attn.hip's K/V-resident kernel on the motion-VAE decoder's cross attention:
// n Gaussians (queries shared by the T frames, inner stride 0) x 512 latents per frame, 12 heads of 64, V^T head-major.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 <attn.hip flags> [-DRES_WAVES=..] [-DRES_NQ=..] [-DRES_SCHED=..] kvres_bench.hip -o kvres_<tag>.bin
//   kvres_<tag>.bin [iters] [dtype 0 bf16 / 1 fp16] [n]
#include "../../gvfdiffusion_amd/csrc/attn.hip"
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
    const int n = argc > 3 ? atoi(argv[3]) : 43648, T = 24, L = 512, H = 12, d = 64, C = H * d, Lp = 512;
    auto enc = [&](float f) { return dt ? h_f2h(f) : h_f2bf(f); };
    auto dec = [&](unsigned short u) { return dt ? h_h2f(u) : h_bf2f(u); };
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<unsigned short> hq((size_t)n * C), hk((size_t)T * H * L * d), hv((size_t)T * H * d * Lp);
    for (auto& v : hq) v = enc(nd(rng));
    for (auto& v : hk) v = enc(nd(rng));
    for (auto& v : hv) v = enc(nd(rng));
    unsigned short *dq, *dk, *dv, *dout;
    const size_t out_elems = (size_t)T * n * C;
    CK(hipMalloc(&dq, hq.size() * 2)); CK(hipMalloc(&dk, hk.size() * 2)); CK(hipMalloc(&dv, hv.size() * 2)); CK(hipMalloc(&dout, out_elems * 2));
    CK(hipMemcpy(dq, hq.data(), hq.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dk, hk.data(), hk.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dv, hv.data(), hv.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemset(dout, 0xff, out_elems * 2));
    const int64_t qs[4] = {(int64_t)n * C, 0, C, d}, ks[4] = {(int64_t)T * H * L * d, (int64_t)H * L * d, d, (int64_t)L * d};
    const int64_t vs[4] = {(int64_t)T * H * d * Lp, (int64_t)H * d * Lp, Lp, (int64_t)d * Lp}, os[4] = {(int64_t)T * n * C, (int64_t)n * C, C, d};
    const float scale = 1.0f / sqrtf((float)d);
    auto run = [&]() { return gvf_attn_fwd(dt, dq, dk, dv, dout, 1, T, n, L, H, d, qs, ks, vs, os, 1, nullptr, nullptr, scale, nullptr); };
    int rc = run();
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
            for (int e = 0; e < d; ++e) acc += (double)dec(hq[(size_t)qi * C + h * d + e]) * dec(hk[(((size_t)t * H + h) * L + k) * d + e]);
            s[k] = acc * scale; m = s[k] > m ? s[k] : m;
        }
        double l = 0; std::vector<double> ov(d, 0.0);
        for (int k = 0; k < L; ++k) { const double p = exp(s[k] - m); l += p; for (int e = 0; e < d; ++e) ov[e] += p * dec(hv[(((size_t)t * H + h) * d + e) * Lp + k]); }
        for (int e = 0; e < d; ++e) {
            const double ref = ov[e] / l, got = dec(hout[((size_t)t * n + qi) * C + h * d + e]);
            if (!(got == got)) ++nan_count;
            num += (got - ref) * (got - ref); den += ref * ref; maxabs = fabs(got - ref) > maxabs ? fabs(got - ref) : maxabs;
        }
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) (void)run();
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) (void)run();
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / iters, tf = 4.0 * (double)T * n * L * C / us / 1e6;
    const double rel = sqrt(num / (den + 1e-300));
    printf("vae decoder cross attention dt=%d n=%d waves=%d NQ=%d sched=%d: rel_l2 %.3e max_abs %.3e nan %d | %8.1f us  %7.1f TFLOP/s (%.1f%% of 2.5 PF)\n",
           dt, n, RES_WAVES, RES_NQ, RES_SCHED, rel, maxabs, nan_count, us, tf, tf / 25.0);
    return (rel < 1e-2 && nan_count == 0) ? 0 : 2;
}
