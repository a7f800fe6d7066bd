// solo_wave.hip -- the instruction mix of ONE attn_xt wave-iteration (64 queries x 64 keys, head_dim 32:
// 16 v_mfma_f32_32x32x16_bf16 + 16 v_mfma_f32_4x4x4 + 64 v_exp_f32 + 32 v_cvt_pk) issued by ONE wave per SIMD (4-wave workgroups, one per
// CU) against the same mix split over TWO waves per SIMD (8-wave workgroups), in several fine-grained orders.  No memory, no dependencies
// between the pipes.  Reports ns AND shader cycles (s_memtime) per mix per SIMD, i.e. the clock the chip held.
//   hipcc --offload-arch=gfx950 -O3 solo_wave.hip -o solo_wave.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define BIG(c_) c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c_, 0, 0, 0);
#define SML(l_) l_ = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a4, b4, l_, 0, 0, 0);
#define EX(i_) asm volatile("v_exp_f32 %0, %0" : "+v"(x[(i_) & 31]));
#define CV(i_) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[(i_) & 15]) : "v"(x[(2 * (i_)) & 31]), "v"(x[(2 * (i_) + 1) & 31]));
#define ADD2(i_) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(ls[(i_) & 3]) : "v"(*(const double*)&x[(2 * (i_)) & 30]));
#define FENCE __builtin_amdgcn_sched_barrier(0);

// HALVES = how many "half mixes" (8 big, 8 small, 32 exp, 16 cvt) a wave issues per iteration: 2 for one wave per SIMD, 1 for two
template <int HALVES, int mode>
__global__ __launch_bounds__(HALVES == 2 ? 256 : 512, 1) void k(float* out, long long* cyc, int iters) {
    bf16x8 a, b; bf16x4 a4, b4;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.37f + 0.001f * (threadIdx.x % 61)); b[i] = (__bf16)(0.11f * (i + 1)); }
    for (int i = 0; i < 4; ++i) { a4[i] = (__bf16)1.0f; b4[i] = (__bf16)(0.01f * i); }
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0}; f32x4 l0 = {0}, l1 = {0};
    float x[32]; unsigned pk[16]; double ls[4] = {0, 0, 0, 0};
    for (int i = 0; i < 32; ++i) x[i] = -0.001f * (threadIdx.x + i);
    for (int i = 0; i < 16; ++i) pk[i] = 0;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int hv = 0; hv < HALVES; ++hv) {
            if (mode == 0) {             // attn_xt's order: [8 exp] big [4 cvt, 2 small] big
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    EX(8 * u) EX(8 * u + 1) EX(8 * u + 2) EX(8 * u + 3) EX(8 * u + 4) EX(8 * u + 5) EX(8 * u + 6) EX(8 * u + 7) FENCE
                    BIG(c0) FENCE CV(4 * u) CV(4 * u + 1) CV(4 * u + 2) CV(4 * u + 3) SML(l0) SML(l1) FENCE BIG(c1) FENCE
                }
            } else if (mode == 1) {      // balanced units: big [4 exp] small [2 cvt]
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (u & 1) { BIG(c1) } else { BIG(c0) } FENCE
                    EX(4 * u) EX(4 * u + 1) EX(4 * u + 2) EX(4 * u + 3) FENCE
                    if (u & 1) { SML(l1) } else { SML(l0) } FENCE
                    CV(2 * u) CV(2 * u + 1) FENCE
                }
            } else if (mode == 2) {      // big [4 exp 2 cvt] small
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (u & 1) { BIG(c1) } else { BIG(c0) } FENCE
                    EX(4 * u) EX(4 * u + 1) EX(4 * u + 2) EX(4 * u + 3) CV(2 * u) CV(2 * u + 1) FENCE
                    if (u & 1) { SML(l1) } else { SML(l0) } FENCE
                }
            } else if (mode == 3) {      // no small MFMAs: big [4 exp 2 cvt]
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (u & 1) { BIG(c1) } else { BIG(c0) } FENCE
                    EX(4 * u) EX(4 * u + 1) EX(4 * u + 2) EX(4 * u + 3) CV(2 * u) CV(2 * u + 1) FENCE
                }
            } else if (mode == 4) {      // no small MFMAs, row sums as packed fp32 adds: big [4 exp 2 cvt 2 pk_add]
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (u & 1) { BIG(c1) } else { BIG(c0) } FENCE
                    EX(4 * u) EX(4 * u + 1) EX(4 * u + 2) EX(4 * u + 3) CV(2 * u) CV(2 * u + 1) ADD2(2 * u) ADD2(2 * u + 1) FENCE
                }
            } else if (mode == 5) {      // four independent accumulators (no back-to-back MFMA on one accumulator): big [4 exp] small [2 cvt]
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if ((u & 3) == 0) { BIG(c0) } else if ((u & 3) == 1) { BIG(c1) } else if ((u & 3) == 2) { BIG(c2) } else { BIG(c3) } FENCE
                    EX(4 * u) EX(4 * u + 1) EX(4 * u + 2) EX(4 * u + 3) FENCE
                    if (u & 1) { SML(l1) } else { SML(l0) } FENCE
                    CV(2 * u) CV(2 * u + 1) FENCE
                }
            } else if (mode == 6) {      // MFMAs only
#pragma unroll
                for (int u = 0; u < 8; ++u) { if (u & 1) { BIG(c1) SML(l1) } else { BIG(c0) SML(l0) } FENCE }
            } else if (mode == 7) {      // VALU only
#pragma unroll
                for (int u = 0; u < 8; ++u) { EX(4 * u) EX(4 * u + 1) EX(4 * u + 2) EX(4 * u + 3) CV(2 * u) CV(2 * u + 1) FENCE }
            } else if (mode == 8) {      // big MFMAs only
#pragma unroll
                for (int u = 0; u < 8; ++u) { if (u & 1) { BIG(c1) } else { BIG(c0) } FENCE }
            } else if (mode == 9) {      // exp only
#pragma unroll
                for (int u = 0; u < 8; ++u) { EX(4 * u) EX(4 * u + 1) EX(4 * u + 2) EX(4 * u + 3) FENCE }
            } else if (mode == 10) {     // big [2 exp] x2 ... : finer: big [2 exp] [1 cvt] small? -> big [2 exp 1 cvt] [2 exp 1 cvt] small
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (u & 1) { BIG(c1) } else { BIG(c0) } FENCE
                    EX(4 * u) EX(4 * u + 1) CV(2 * u) EX(4 * u + 2) EX(4 * u + 3) CV(2 * u + 1) FENCE
                    if (u & 1) { SML(l1) } else { SML(l0) } FENCE
                }
            } else if (mode == 11) {     // two bigs back to back then 8 exp 4 cvt 2 small
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    BIG(c0) BIG(c1) FENCE
                    EX(8 * u) EX(8 * u + 1) EX(8 * u + 2) EX(8 * u + 3) EX(8 * u + 4) EX(8 * u + 5) EX(8 * u + 6) EX(8 * u + 7)
                    CV(4 * u) CV(4 * u + 1) CV(4 * u + 2) CV(4 * u + 3) FENCE SML(l0) SML(l1) FENCE
                }
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 32; ++i) s += x[i];
    for (int i = 0; i < 16; ++i) s += (float)pk[i];
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    for (int i = 0; i < 4; ++i) s += l0[i] + l1[i] + (float)ls[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int M>
static void launch_m(int waves, float* out, long long* cyc, int iters) {
    if (waves == 1) k<2, M><<<256, 256>>>(out, cyc, iters); else k<1, M><<<256, 512>>>(out, cyc, iters);
}
static void launch(int waves, int mode, float* out, long long* cyc, int iters) {
    switch (mode) {
        case 0: launch_m<0>(waves, out, cyc, iters); break; case 1: launch_m<1>(waves, out, cyc, iters); break;
        case 2: launch_m<2>(waves, out, cyc, iters); break; case 3: launch_m<3>(waves, out, cyc, iters); break;
        case 4: launch_m<4>(waves, out, cyc, iters); break; case 5: launch_m<5>(waves, out, cyc, iters); break;
        case 6: launch_m<6>(waves, out, cyc, iters); break; case 7: launch_m<7>(waves, out, cyc, iters); break;
        case 8: launch_m<8>(waves, out, cyc, iters); break; case 9: launch_m<9>(waves, out, cyc, iters); break;
        case 10: launch_m<10>(waves, out, cyc, iters); break; case 11: launch_m<11>(waves, out, cyc, iters); break;
    }
}

int main() {
    float* out; (void)hipMalloc(&out, 256 * 512 * sizeof(float));
    long long* cyc; (void)hipMalloc(&cyc, 8);
    const int N = 20000;
    const char* names[] = {"[8 exp] big [4 cvt 2 small] big (attn_xt)", "big [4 exp] small [2 cvt]", "big [4 exp 2 cvt] small", "big [4 exp 2 cvt] (no small)",
                           "big [4 exp 2 cvt 2 pk_add] (no small)", "as 1, four accumulators", "MFMAs only (16 big + 16 small)", "VALU only (64 exp + 32 cvt)",
                           "big MFMAs only", "exp only", "big [2 exp cvt 2 exp cvt] small", "big big [8 exp 4 cvt] small small"};
    printf("# per SIMD and per full mix (16 big + 16 small MFMAs, 64 v_exp_f32, 32 v_cvt_pk); s_memtime ticks are 100 MHz on this part if the clock column reads ~0.1\n");
    for (int waves = 1; waves <= 2; ++waves) {
        for (int mode = 0; mode < 12; ++mode) {
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            launch(waves, mode, out, cyc, 200);
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0);
            launch(waves, mode, out, cyc, N);
            (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            printf("%d wave(s)/SIMD  %-46s %7.1f ns  %8.1f ticks  (%.3f ticks/ns)\n", waves, names[mode], ms * 1e6f / N, (double)c / N, (double)c / N / (ms * 1e6f / N));
        }
    }
    return 0;
}
