// f32mfma_overlap.hip -- does v_mfma_f32_32x32x2_f32 (the f32-input matrix instruction, which runs at the f32 VECTOR rate) overlap with plain
// VALU work of another wave on the same SIMD, the way the 16-bit MFMAs do?  8-wave workgroups (2 waves per SIMD), one per CU:
//   A: every wave issues f32 MFMAs only;  B: every wave issues v_fma_f32 only;  C: waves 0-3 MFMA, waves 4-7 VALU (one of each per SIMD);
//   D/E/F: the same with v_mfma_f32_32x32x16_bf16.  If C ~ max(A, B) / 2-ish the pipes overlap; if C ~ (A + B) / 2 they share hardware.
//   hipcc --offload-arch=gfx950 -O3 f32mfma_overlap.hip -o f32mfma_overlap.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

template <int MODE>
__global__ __launch_bounds__(512, 1) void k(float* out, int iters) {
    const int wave = threadIdx.x >> 6;
    f32x16 c0 = {0}, c1 = {0};
    float a = 0.001f * threadIdx.x, b = 1.0f + 0.0001f * threadIdx.x;
    bf16x8 ha, hb;
    for (int i = 0; i < 8; ++i) { ha[i] = (__bf16)(0.37f + 0.001f * (threadIdx.x % 61)); hb[i] = (__bf16)(0.11f * (i + 1)); }
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = 0.5f + 0.01f * i;
    const bool do_mfma = MODE == 0 || MODE == 3 || ((MODE == 2 || MODE == 5) && wave < 4);
    const bool do_valu = MODE == 1 || MODE == 4 || ((MODE == 2 || MODE == 5) && wave >= 4);
    const bool f32 = MODE < 3;
    for (int it = 0; it < iters; ++it) {
        if (do_mfma) {
            if (f32) {
#pragma unroll
                for (int u = 0; u < 4; ++u) { c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, c1, 0, 0, 0); }
            } else {
#pragma unroll
                for (int u = 0; u < 8; ++u) { c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha, hb, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hb, ha, c1, 0, 0, 0); }
            }
        }
        if (do_valu) {
#pragma unroll
            for (int u = 0; u < 12; ++u)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(b), "v"(a));
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += x[i] + c0[i] + c1[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int M> static float run(float* out, int N) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<M><<<256, 512>>>(out, 100); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<M><<<256, 512>>>(out, N);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6f / N;
}

int main() {
    float* out; (void)hipMalloc(&out, 256 * 512 * sizeof(float));
    const int N = 20000;
    printf("# ns per iteration; per wave and iteration: 8 v_mfma_f32_32x32x2_f32 (512 matrix cycles) | 16 v_mfma_f32_32x32x16_bf16 (512) | 192 v_fma_f32\n");
    printf("f32 MFMA in all 8 waves                       %8.1f ns\n", run<0>(out, N));
    printf("v_fma_f32 in all 8 waves                      %8.1f ns\n", run<1>(out, N));
    printf("f32 MFMA in waves 0-3, v_fma_f32 in waves 4-7 %8.1f ns\n", run<2>(out, N));
    printf("bf16 MFMA in all 8 waves                      %8.1f ns\n", run<3>(out, N));
    printf("v_fma_f32 in all 8 waves                      %8.1f ns\n", run<4>(out, N));
    printf("bf16 MFMA in waves 0-3, v_fma_f32 in waves 4-7 %7.1f ns\n", run<5>(out, N));
    return 0;
}
