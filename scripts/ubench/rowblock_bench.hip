// rowblock_bench.hip -- standalone check + timing of csrc/rowblock.hip on the DiT's shapes (M = 12288 rows, C = 512).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 rowblock_bench.hip -o rowblock_bench.bin     (scripts/ubench/build_rb.sh)
// Checks sampled rows of the updated stream and of the last projection against a float64 host reference that rounds where the
// kernel rounds (bf16 operands, bf16 normalised rows, bf16 hidden units), then times the launch.
#include "../../gvfdiffusion_amd/csrc/rowblock.hip"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

static unsigned short h_f2bf(float f) {
    unsigned u; memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
static float h_bf2f(unsigned short h) { unsigned u = ((unsigned)h) << 16; float f; memcpy(&f, &u, 4); return f; }
static double rbf(double v) { return (double)h_bf2f(h_f2bf((float)v)); }
static double gelu(double x) { const double u = 0.7978845608028654 * (x + 0.044715 * x * x * x); return x / (1.0 + exp(-2.0 * u)); }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <class T> static T* dev(const std::vector<T>& h) {
    T* d = nullptr;
    if (hipMalloc(&d, h.size() * sizeof(T)) != hipSuccess) return nullptr;
    (void)hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    return d;
}

// reads one dword per 128-byte line of [p, p + bytes): `copies` > 1 = every XCD (block % 8) reads the whole range
__global__ void rb_touch_kernel(const char* p, size_t bytes, int copies, unsigned* sink) {
    const size_t per = (bytes / 128 + 255) / 256;                    // blocks per copy
    const size_t blk = copies > 1 ? blockIdx.x / 8 : blockIdx.x;    // (block b runs on XCD b % 8)
    const size_t off = (blk * 256 + threadIdx.x) * 128;
    if (blk < per && off < bytes) {
        const unsigned v = *reinterpret_cast<const unsigned*>(p + off);
        if (v == 0x12345678u) sink[threadIdx.x] = v;
    }
}

static int run_case(const char* name, int B, int TN, int K1, int hidden, int N3, bool adaln, int iters) {
    const int C = 512, M = B * TN, mod_ld = 4 * C;
    std::mt19937 rng(99);
    std::normal_distribution<float> nd(0.f, 1.f);
    auto randbf = [&](size_t n, float sc) { std::vector<unsigned short> v(n); for (auto& e : v) e = h_f2bf(nd(rng) * sc); return v; };
    auto randf = [&](size_t n, float sc, float off = 0.f) { std::vector<float> v(n); for (auto& e : v) e = nd(rng) * sc + off; return v; };
    std::vector<unsigned short> hA = randbf((size_t)M * K1, 1.f), hW1 = randbf((size_t)C * K1, 1.f / sqrtf((float)K1));
    std::vector<unsigned short> hWf1 = randbf((size_t)(hidden ? hidden : 1) * C, 1.f / sqrtf(512.f)), hWf2 = randbf((size_t)C * (hidden ? hidden : 1), 1.f / sqrtf((float)(hidden ? hidden : 1)));
    std::vector<unsigned short> hW3 = randbf((size_t)(N3 ? N3 : 1) * C, 1.f / sqrtf(512.f));
    std::vector<float> hb1 = randf(C, 0.1f), hbf1 = randf(hidden ? hidden : 1, 0.1f), hbf2 = randf(C, 0.1f), hb3 = randf(N3 ? N3 : 1, 0.1f);
    std::vector<float> hx = randf((size_t)M * C, 1.f, 0.3f), hmod = randf((size_t)B * mod_ld, 0.3f);
    std::vector<float> hlnw = randf(C, 0.2f, 1.f), hlnb = randf(C, 0.2f);

    unsigned short *dA = dev(hA), *dW1 = dev(hW1), *dWf1 = dev(hWf1), *dWf2 = dev(hWf2), *dW3 = dev(hW3);
    float *db1 = dev(hb1), *dbf1 = dev(hbf1), *dbf2 = dev(hbf2), *db3 = dev(hb3), *dx = dev(hx), *dmod = dev(hmod), *dlnw = dev(hlnw), *dlnb = dev(hlnb);
    const long long bytes1 = gvf_rowblock_packed_bytes(C, K1), bytesm = hidden ? 2LL * hidden * C * 2 : 0, bytes3 = N3 ? gvf_rowblock_packed_bytes(N3, C) : 0;
    char* dW; CK(hipMalloc(&dW, bytes1 + bytesm + bytes3));
    int rc = gvf_rowblock_pack_weight(dW1, K1, C, K1, dW, nullptr);
    if (!rc && hidden) rc = gvf_rowblock_pack_mlp(dWf1, dWf2, hidden, dW + bytes1, nullptr);
    if (!rc && N3) rc = gvf_rowblock_pack_weight(dW3, C, N3, C, dW + bytes1 + bytesm, nullptr);
    if (rc) { printf("pack rc %d\n", rc); return 1; }
    unsigned short *dout, *dhb;
    CK(hipMalloc(&dout, (size_t)M * (N3 ? N3 : 1) * 2)); CK(hipMalloc(&dhb, (size_t)M * C * 2));

    gvf_rowblock_args a;
    memset(&a, 0, sizeof(a));
    a.a = dA; a.lda = K1; a.K1 = K1; a.w = dW; a.b1 = db1; a.x = dx; a.M = M; a.C = C;
    a.mod_ld = mod_ld; a.rows_per_group = TN; a.eps = 1e-6f;
    if (adaln) { a.gate1 = dmod; a.ln1.shift = dmod + C; a.ln1.scale = dmod + 2 * C; }
    else { a.ln1.ln_w = dlnw; a.ln1.ln_b = dlnb; }
    if (hidden) { a.b_fc1 = dbf1; a.b_fc2 = dbf2; a.hidden = hidden; a.gate_m = dmod + 3 * C; a.ln2.shift = dmod + 2 * C; a.ln2.scale = dmod + C; }
    a.b3 = db3; a.out3 = dout; a.N3 = N3; a.epi3 = GVF_EPI_STORE_BF16; a.hb_out = N3 ? nullptr : dhb;
    rc = gvf_rowblock_fused_bf16(&a, nullptr);
    if (rc) { printf("launch rc %d\n", rc); return 1; }
    CK(hipDeviceSynchronize());
    std::vector<float> gx((size_t)M * C);
    std::vector<unsigned short> gout((size_t)M * (N3 ? N3 : 1)), ghb((size_t)M * C);
    CK(hipMemcpy(gx.data(), dx, gx.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(gout.data(), dout, gout.size() * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(ghb.data(), dhb, ghb.size() * 2, hipMemcpyDeviceToHost));

    double ex = 0, nx = 0, eo = 0, no = 0;
    const int n_samples = 24;
    for (int si = 0; si < n_samples; ++si) {
        const int row = (int)(((long long)si * 7919 + (si % 3) * 47) % M), g = row / TN;
        const float* md = &hmod[(size_t)g * mod_ld];
        std::vector<double> x1(C), hbv(C);
        auto layer_norm = [&](const std::vector<double>& x, bool ada, const float* sh, const float* sc, std::vector<double>& out) {
            double mean = 0, var = 0;
            for (int c = 0; c < C; ++c) mean += x[c];
            mean /= C;
            for (int c = 0; c < C; ++c) var += (x[c] - mean) * (x[c] - mean);
            var /= C;
            const double rstd = 1.0 / sqrt(var + 1e-6);
            for (int c = 0; c < C; ++c) {
                double y = (x[c] - mean) * rstd;
                y = ada ? y * (1.0 + sc[c]) + sh[c] : y * hlnw[c] + hlnb[c];
                out[c] = rbf(y);
            }
        };
        for (int c = 0; c < C; ++c) {
            double acc = 0;
            for (int k = 0; k < K1; ++k) acc += (double)h_bf2f(hA[(size_t)row * K1 + k]) * h_bf2f(hW1[(size_t)c * K1 + k]);
            x1[c] = hx[(size_t)row * C + c] + (adaln ? md[c] : 1.0) * (acc + hb1[c]);
        }
        layer_norm(x1, adaln, md + C, md + 2 * C, hbv);
        std::vector<double> xf = x1;
        if (hidden) {
            std::vector<double> hid(hidden);
            for (int j = 0; j < hidden; ++j) {
                double acc = 0;
                for (int k = 0; k < C; ++k) acc += hbv[k] * h_bf2f(hWf1[(size_t)j * C + k]);
                hid[j] = rbf(gelu(acc + hbf1[j]));
            }
            for (int c = 0; c < C; ++c) {
                double acc = 0;
                for (int j = 0; j < hidden; ++j) acc += hid[j] * h_bf2f(hWf2[(size_t)c * hidden + j]);
                xf[c] = x1[c] + md[3 * C + c] * (acc + hbf2[c]);
            }
            layer_norm(xf, true, md + 2 * C, md + C, hbv);
        }
        for (int c = 0; c < C; ++c) { const double d = gx[(size_t)row * C + c] - xf[c]; ex += d * d; nx += xf[c] * xf[c]; }
        if (N3) {
            for (int n = 0; n < N3; ++n) {
                double acc = 0;
                for (int k = 0; k < C; ++k) acc += hbv[k] * h_bf2f(hW3[(size_t)n * C + k]);
                const double ref = acc + hb3[n], d = h_bf2f(gout[(size_t)row * N3 + n]) - ref;
                eo += d * d; no += ref * ref;
            }
        } else {
            for (int c = 0; c < C; ++c) { const double d = h_bf2f(ghb[(size_t)row * C + c]) - hbv[c]; eo += d * d; no += hbv[c] * hbv[c]; }
        }
    }
    const double rx = sqrt(ex / nx), ro = sqrt(eo / no);

#ifdef RB_TIMING
    {   // phase stamps (100 MHz s_memrealtime) of one launch, averaged over the workgroups
        const int nwg = M / 48;
        long long* dd; CK(hipMalloc(&dd, (size_t)nwg * 16 * 8)); CK(hipMemset(dd, 0, (size_t)nwg * 16 * 8));
        g_rb_dbg = dd;
        (void)gvf_rowblock_fused_bf16(&a, nullptr);
        CK(hipDeviceSynchronize());
        g_rb_dbg = nullptr;
        std::vector<long long> hd((size_t)nwg * 16);
        CK(hipMemcpy(hd.data(), dd, hd.size() * 8, hipMemcpyDeviceToHost));
        long long t0 = hd[0], tend = 0;
        for (int w = 0; w < nwg; ++w) t0 = hd[w * 16] < t0 ? hd[w * 16] : t0;
        printf("    phase ends, us after the first workgroup's entry (mean over %d workgroups):", nwg);
        for (int i = 0; i < 16; ++i) {
            double sum = 0; int cnt = 0;
            for (int w = 0; w < nwg; ++w) if (hd[w * 16 + i]) { sum += (hd[w * 16 + i] - t0) * 0.01; ++cnt; tend = hd[w * 16 + i] > tend ? hd[w * 16 + i] : tend; }
            if (cnt) printf(" %.1f", sum / cnt);
        }
        printf("  | last stamp %.1f\n", (tend - t0) * 0.01);
        (void)hipFree(dd);
    }
#endif
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) (void)gvf_rowblock_fused_bf16(&a, nullptr);
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) (void)gvf_rowblock_fused_bf16(&a, nullptr);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / iters;
    // the same launch with everything it reads COLD in the caches, as inside the denoise step (every launch there has its own weights,
    // 115 MB per step, and its inputs were written by the launch before): rotate over copies of the weight stream, the activations and
    // the residual stream that together exceed the 256 MB Infinity Cache
    // RB_COLD=1: everything cold; 2: weights warm (same stream every launch), activations + residual stream cold; 3: weights cold only;
    // 4: everything cold, but a prefetch launch reads the weight stream ONCE just before (-> Infinity Cache, one XCD's L2 per line);
    // 5: as 4 with every XCD reading the whole stream (-> every L2).  The prefetch launches of 4 / 5 are timed separately and subtracted.
    double us_cold = 0, us_pref = 0;
    if (getenv("RB_COLD")) {
        const int cm = atoi(getenv("RB_COLD"));
        const int n_rot = 24;
        const size_t wbytes_ = (size_t)(bytes1 + bytesm + bytes3);
        char* dWr; unsigned short* dAr; float* dxr; unsigned* dsink;
        CK(hipMalloc(&dWr, wbytes_ * n_rot)); CK(hipMalloc(&dAr, hA.size() * 2 * n_rot)); CK(hipMalloc(&dxr, hx.size() * 4 * n_rot)); CK(hipMalloc(&dsink, 4096 * 4));
        for (int r = 0; r < n_rot; ++r) {
            CK(hipMemcpy(dWr + wbytes_ * r, dW, wbytes_, hipMemcpyDeviceToDevice));
            CK(hipMemcpy(dAr + hA.size() * r, dA, hA.size() * 2, hipMemcpyDeviceToDevice));
            CK(hipMemcpy(dxr + hx.size() * r, dx, hx.size() * 4, hipMemcpyDeviceToDevice));
        }
        CK(hipDeviceSynchronize());
        const bool w_rot = cm != 2, ax_rot = cm != 3;
        for (int pass = 0; pass < 2; ++pass) {          // pass 0: the prefetch launches alone (modes 4, 5), pass 1: everything
            if (pass == 0 && cm < 4) continue;
            CK(hipEventRecord(e0));
            for (int i = 0; i < iters; ++i) {
                gvf_rowblock_args b = a;
                if (w_rot) b.w = dWr + wbytes_ * (i % n_rot);
                if (ax_rot) { b.a = dAr + hA.size() * (i % n_rot); b.x = dxr + hx.size() * (i % n_rot); }
                if (cm == 4) rb_touch_kernel<<<dim3((unsigned)((wbytes_ / 128 + 255) / 256)), dim3(256)>>>((const char*)b.w, wbytes_, 1, dsink);
                if (cm == 5) rb_touch_kernel<<<dim3(8 * (unsigned)((wbytes_ / 128 + 255) / 256)), dim3(256)>>>((const char*)b.w, wbytes_, 8, dsink);
                if (pass == 1) (void)gvf_rowblock_fused_bf16(&b, nullptr);
            }
            CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (pass == 0) us_pref = ms * 1e3 / iters; else us_cold = ms * 1e3 / iters;
        }
        (void)hipFree(dWr); (void)hipFree(dAr); (void)hipFree(dxr); (void)hipFree(dsink);
    }
    const double flops = 2.0 * M * C * ((double)K1 + 2.0 * hidden + N3);
    const double wbytes = (double)(bytes1 + bytesm + bytes3) * (M / 48);
    printf("%-44s: stream rel_l2 %.2e  out rel_l2 %.2e | %7.1f us  %6.1f TFLOP/s  weight stream %5.1f TB/s (L2)", name, rx, ro, us, flops / us / 1e6,
           wbytes / us / 1e6);
    if (us_cold > 0) printf("  | RB_COLD=%s %7.1f us", getenv("RB_COLD"), us_cold);
    if (us_pref > 0) printf(" (of which the prefetch launch alone: %5.1f)", us_pref);
    printf("\n");
    (void)hipFree(dA); (void)hipFree(dW1); (void)hipFree(dWf1); (void)hipFree(dWf2); (void)hipFree(dW3); (void)hipFree(dW); (void)hipFree(dout); (void)hipFree(dhb);
    (void)hipFree(dx); (void)hipFree(dmod);
    return (rx < (hidden ? 2e-4 : 1e-5) && ro < 6e-3) ? 0 : 2;      // MLP: a bf16 rounding flip of one hidden unit moves a stream element by ~1e-4
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20;
    const int only = argc > 2 ? atoi(argv[2]) : -1;      // run one case (PMC passes)
    int bad = 0;
    if (only < 0 || only == 0) bad |= run_case("to_out + adaLN + to_qkv (N3 1536)", 1, 12288, 512, 0, 1536, true, iters);
    if (only < 0 || only == 1) bad |= run_case("to_out + affine LN + to_q (N3 512)", 1, 12288, 512, 0, 512, false, iters);
    if (only < 0 || only == 2) bad |= run_case("input_layer (K 128) + adaLN + to_qkv", 1, 12288, 128, 0, 1536, true, iters);
    if (only < 0 || only == 3) bad |= run_case("to_out + LN + MLP 2048 + adaLN + to_qkv", 1, 12288, 512, 2048, 1536, false, iters);
    if (only < 0 || only == 4) bad |= run_case("to_out + LN + MLP 2048 + adaLN -> hb_out", 1, 12288, 512, 2048, 0, false, iters);
    if (only < 0 || only == 5) bad |= run_case("B=3: to_out + LN + MLP + adaLN + to_qkv", 3, 12288, 512, 2048, 1536, false, iters);
    if (only < 0 || only == 6) bad |= run_case("small: M 96, MLP 512, N3 512", 2, 48, 128, 512, 512, true, 3);
    printf(bad ? "FAILED\n" : "ALL OK\n");
    return bad;
}
