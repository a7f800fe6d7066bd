// mfma_valu_overlap.hip -- do the matrix pipe and the VALU of ONE SIMD run concurrently on gfx950?
//   mode A: every wave runs MFMA only / VALU only / exp only            (baselines, 1..4 waves per SIMD)
//   mode B: half of the waves of each SIMD run MFMA, the other half VALU (inter-wave overlap)
//   mode C: each wave interleaves 1 MFMA with K VALU instructions       (intra-wave overlap)
// hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip -o mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define MF(acc) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
#define FMA4 asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(kb), "v"(kc));
#define EXP4 asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));

// role: 0 = MFMA only (8 per iter), 1 = fma only (64 per iter), 2 = exp only (32 per iter),
//       3 = 1 MFMA + 4 fma (x8), 4 = 1 MFMA + 8 fma (x8), 5 = 1 MFMA + 4 exp (x8), 6 = 1 MFMA + 2 exp + 4 fma (x8)
__device__ __forceinline__ void body(int role, int iters, float* out) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * threadIdx.x); b[i] = (__bf16)(0.002f * i); }
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, kb = 0.999f, kc = 1e-6f;
    if (role == 0) for (int it = 0; it < iters; ++it) { MF(c0) MF(c1) MF(c2) MF(c3) MF(c0) MF(c1) MF(c2) MF(c3) }
    else if (role == 1) for (int it = 0; it < iters; ++it) { FMA4 FMA4 FMA4 FMA4 FMA4 FMA4 FMA4 FMA4 FMA4 FMA4 FMA4 FMA4 FMA4 FMA4 FMA4 FMA4 }
    else if (role == 2) for (int it = 0; it < iters; ++it) { EXP4 EXP4 EXP4 EXP4 EXP4 EXP4 EXP4 EXP4 }
    else if (role == 3) for (int it = 0; it < iters; ++it) { MF(c0) FMA4 MF(c1) FMA4 MF(c2) FMA4 MF(c3) FMA4 MF(c0) FMA4 MF(c1) FMA4 MF(c2) FMA4 MF(c3) FMA4 }
    else if (role == 4) for (int it = 0; it < iters; ++it) { MF(c0) FMA4 FMA4 MF(c1) FMA4 FMA4 MF(c2) FMA4 FMA4 MF(c3) FMA4 FMA4 MF(c0) FMA4 FMA4 MF(c1) FMA4 FMA4 MF(c2) FMA4 FMA4 MF(c3) FMA4 FMA4 }
    else if (role == 5) for (int it = 0; it < iters; ++it) { MF(c0) EXP4 MF(c1) EXP4 MF(c2) EXP4 MF(c3) EXP4 MF(c0) EXP4 MF(c1) EXP4 MF(c2) EXP4 MF(c3) EXP4 }
    else if (role == 7) for (int it = 0; it < iters; ++it) { MF(c0) MF(c0) MF(c0) MF(c0) MF(c0) MF(c0) MF(c0) MF(c0) }
    else if (role == 8) for (int it = 0; it < iters; ++it) { MF(c0) MF(c1) MF(c0) MF(c1) MF(c0) MF(c1) MF(c0) MF(c1) }
    else if (role == 9) for (int it = 0; it < iters; ++it) { MF(c0) EXP4 MF(c1) EXP4 MF(c0) EXP4 MF(c1) EXP4 MF(c0) EXP4 MF(c1) EXP4 MF(c0) EXP4 MF(c1) EXP4 }
    else if (role == 10) for (int it = 0; it < iters; ++it) { MF(c0) MF(c0) EXP4 EXP4 MF(c1) MF(c1) EXP4 EXP4 MF(c2) MF(c2) EXP4 EXP4 MF(c3) MF(c3) EXP4 EXP4 }
    else if (role == 6) for (int it = 0; it < iters; ++it) { MF(c0) EXP4 FMA4 MF(c1) EXP4 FMA4 MF(c2) EXP4 FMA4 MF(c3) EXP4 FMA4 MF(c0) EXP4 FMA4 MF(c1) EXP4 FMA4 MF(c2) EXP4 FMA4 MF(c3) EXP4 FMA4 }
    float s = x0 + x1 + x2 + x3;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// waves [0, split) of the block take role_a, the rest role_b.  Waves of a workgroup go to SIMDs round-robin, so
// with 8 waves and split = 4 every SIMD hosts one wave of each role.
__global__ void k(float* out, int iters_a, int iters_b, int role_a, int role_b, int split) {
    const int wave = threadIdx.x >> 6;
    if (wave < split) body(role_a, iters_a, out); else body(role_b, iters_b, out);
}

float run(int threads, int ia, int ib, int ra, int rb, int split, float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<<<256, threads>>>(out, ia / 50 + 1, ib / 50 + 1, ra, rb, split);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<<<256, threads>>>(out, ia, ib, ra, rb, split);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f;
}

int main() {
    float* out; hipMalloc(&out, 256 * 1024 * sizeof(float));
    const int N = 20000;
    const char* names[] = {"8 MFMA", "64 fma", "32 exp", "8x(MFMA+4fma)", "8x(MFMA+8fma)", "8x(MFMA+4exp)", "8x(MFMA+4exp+4fma)", "8 MFMA 1 chain", "8 MFMA 2 chains", "8x(MFMA+4exp) 2 chains", "4x(2 dep MFMA + 8 exp)"};
    printf("one block per CU; time for %d iterations (us).  MFMA 32x32x16 bf16 = 32 cycles of matrix pipe\n", N);
    for (int role = 0; role < 11; ++role)
        for (int wps = 1; wps <= 4; wps *= 2) {
            float us = run(256 * wps, N, N, role, role, 99, out);
            printf("A  all waves %-20s waves/SIMD %d : %8.1f us  = %6.1f ns per iteration per SIMD-wave-slot\n", names[role], wps, us, us * 1e3 / N / wps);
        }
    // B: inter-wave.  8 MFMA per iter (256 pipe cycles) against 64 fma per iter
    for (int wps = 2; wps <= 4; wps *= 2) {
        float us_m = run(256 * wps, N, N, 0, 0, 99, out), us_v = run(256 * wps, N, N, 1, 1, 99, out), us_e = run(256 * wps, N, N, 2, 2, 99, out);
        float us_mv = run(256 * wps, N, N, 0, 1, 4 * wps / 2, out), us_me = run(256 * wps, N, N, 0, 2, 4 * wps / 2, out);
        float us_mh = run(128 * wps, N, N, 0, 0, 99, out), us_vh = run(128 * wps, N, N, 1, 1, 99, out), us_eh = run(128 * wps, N, N, 2, 2, 99, out);
        printf("B  waves/SIMD %d: all-MFMA %.1f  all-fma %.1f  all-exp %.1f | half the waves alone: MFMA %.1f fma %.1f exp %.1f | half MFMA + half fma %.1f | half MFMA + half exp %.1f\n",
               wps, us_m, us_v, us_e, us_mh, us_vh, us_eh, us_mv, us_me);
    }
    return 0;
}
