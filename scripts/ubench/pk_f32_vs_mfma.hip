// pk_f32_vs_mfma.hip -- minimal, library-free reproducer: on gfx950 (MI355X, ROCm 7.x) packed-fp32 VALU arithmetic of one wave returns wrong
// values while another wave on the same CU issues MFMAs.
//   victim   : every lane runs a chain of v_pk_fma_f32 (inline asm) and the same chain as two scalar v_fma_f32; stores both pairs
//   aggressor: a kernel of back-to-back v_mfma_f32_16x16x32_bf16 on another stream (128 VGPRs free for the victim's waves: they share SIMDs)
// The scalar chain is the reference (bitwise: fused multiply-add either way).  Expected output without the hazard: 0 mismatches everywhere.
//   hipcc --offload-arch=gfx950 -O3 pk_f32_vs_mfma.hip -o pk_f32_vs_mfma.bin && ./pk_f32_vs_mfma.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstring>

__global__ __launch_bounds__(256) void victim(const float* __restrict__ in, float4* __restrict__ out, int n, int chain) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float a0 = in[2 * i], a1 = in[2 * i + 1];
    double pa = __builtin_bit_cast(double, make_float2(a0, a1));
    double pb = __builtin_bit_cast(double, make_float2(0.999f, 1.001f));
    double pacc = __builtin_bit_cast(double, make_float2(0.f, 0.f));
    float s0 = 0.f, s1 = 0.f;
    for (int k = 0; k < chain; ++k) {
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(pacc) : "v"(pa), "v"(pb));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s0) : "v"(a0), "v"(0.999f));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s1) : "v"(a1), "v"(1.001f));
    }
    const float2 p = __builtin_bit_cast(float2, pacc);
    out[i] = make_float4(p.x, p.y, s0, s1);
}

__global__ __launch_bounds__(256) void aggressor(float* sink, int iters) {
    typedef __attribute__((ext_vector_type(4))) float f4; typedef __attribute__((ext_vector_type(8))) __bf16 b8;
    f4 c = {0, 0, 0, 0}; b8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x + e); b[e] = (__bf16)1.0f; }
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int j = 0; j < 16; ++j) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    if (c[0] == 12345.f) sink[threadIdx.x] = c[0];
}

__global__ __launch_bounds__(256) void aggressor_valu(float* sink, int iters) {
    float f = 1.0f + threadIdx.x * 1e-6f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int j = 0; j < 64; ++j) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f));
    if (f == 12345.f) sink[threadIdx.x] = f;
}

int main() {
    const int n = 1 << 20, chain = 64;
    std::vector<float> hin(2 * n);
    for (int i = 0; i < 2 * n; ++i) hin[i] = 0.5f + (float)(i % 977) * 1e-3f;
    float *din, *sink; float4* dout;
    (void)hipMalloc(&din, hin.size() * 4); (void)hipMalloc(&dout, (size_t)n * 16); (void)hipMalloc(&sink, 4096);
    (void)hipMemcpy(din, hin.data(), hin.size() * 4, hipMemcpyHostToDevice);
    hipStream_t sa, sb; (void)hipStreamCreate(&sa); (void)hipStreamCreate(&sb);
    std::vector<float4> h(n);
    const char* names[] = {"no other kernel", "v_fma_f32 kernel on another stream", "MFMA kernel on another stream"};
    for (int mode = 0; mode < 3; ++mode) {
        long long bad_launch = 0, bad_elem = 0;
        for (int rep = 0; rep < 40; ++rep) {
            if (mode == 1) aggressor_valu<<<2048, 256, 0, sb>>>(sink, 3000);
            if (mode == 2) aggressor<<<2048, 256, 0, sb>>>(sink, 3000);
            victim<<<n / 256, 256, 0, sa>>>(din, dout, n, chain);
            (void)hipStreamSynchronize(sa);
            (void)hipMemcpy(h.data(), dout, (size_t)n * 16, hipMemcpyDeviceToHost);
            (void)hipStreamSynchronize(sb);
            long long e = 0;
            for (int i = 0; i < n; ++i) e += (memcmp(&h[i].x, &h[i].z, 4) != 0) + (memcmp(&h[i].y, &h[i].w, 4) != 0);
            bad_launch += e > 0; bad_elem += e;
        }
        printf("%-40s: %lld of 40 victim launches with v_pk_fma_f32 != v_fma_f32 (%lld of %lld values)\n", names[mode], bad_launch, bad_elem, 80LL * n);
    }
    return 0;
}
