// gemm8_bench.hip -- development bench of an 8-wave 256 x 256 x 64 bf16 GEMM for gfx950 (`gvf_gemm` >= 850 TFLOP/s on the
// motion VAE's shapes).  C[M][N] = bf16(A[M][K] W[N][K]^T + bias), fp32 accumulation.  Standalone: own timing, own correctness check against a
// naive kernel.  Variants by -D:
//   G8_PREFETCH=1   the fragments of k-step ks + 1 are requested before the MFMAs of k-step ks (two register sets)
//   G8_PHASES=1     the k-tile is walked as four C-quadrants of 16 MFMAs (one A-half x one B-half, both k-steps) with s_setprio around the MFMAs
//   G8_STAGGER=1    (with G8_PHASES) the two M-halves of the workgroup run one barrier apart: one group in its MFMAs while the other reads / stages
// Build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form [-DG8_...] gemm8_bench.hip -o gemm8_<name>.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <cstring>

typedef __attribute__((ext_vector_type(8))) __bf16 x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#ifndef G8_PREFETCH
#define G8_PREFETCH 0
#endif
#ifndef G8_PHASES
#define G8_PHASES 0
#endif
#ifndef G8_STAGGER
#define G8_STAGGER 0
#endif

constexpr int G8_T = 256, G8_BK = 64, G8_THREADS = 512;
constexpr int G8_OP = G8_T * 8;            // 16-byte chunks of one operand tile (256 rows x 64 k)
constexpr int G8_STAGE = 2 * G8_OP;        // A tile, then W tile

__device__ __forceinline__ void g8_dma16(const unsigned short* g, uint4* l) {
    __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
__device__ __forceinline__ unsigned g8_pack(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 x2;
    x2 v; v[0] = (__bf16)lo; v[1] = (__bf16)hi;
    return __builtin_bit_cast(unsigned, v);
}

__global__ __launch_bounds__(G8_THREADS, 1) void gemm8_kernel(const unsigned short* __restrict__ A, int lda, const unsigned short* __restrict__ W, int ldw,
                                                              const float* __restrict__ bias, unsigned short* __restrict__ C, int ldc, int K,
                                                              int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];          // [2 stages][A: 2048 chunks | W: 2048 chunks] = 128 KiB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3, l15 = lane & 15, lq = lane >> 4;

    // workgroup b runs on XCD b % 8: an XCD keeps whole N-tiles of W (or whole bands of tile rows) in its L2
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    int tile_m, tile_n;
    if ((tiles_n & 7) == 0) { const int npx = tiles_n >> 3; tile_n = xcd * npx + slot % npx; tile_m = slot / npx; }
    else if ((tiles_m & 7) == 0) { const int mpx = tiles_m >> 3; tile_m = xcd * mpx + slot / tiles_n; tile_n = slot % tiles_n; }
    else { tile_m = (int)blockIdx.x / tiles_n; tile_n = (int)blockIdx.x % tiles_n; }
    const int bm = tile_m * G8_T, bn = tile_n * G8_T;

    // staging: instruction i (0..3) of wave w fills LDS slots (i * 8 + w) * 64 + lane = tile rows (i * 8 + w) * 8 + lane / 8, chunk slot lane % 8;
    // the source chunk is (lane % 8) ^ (row & 7) with row & 7 = lane / 8 (swizzle on the source side, LDS side linear)
    const int st_row = wave * 8 + (lane >> 3);
    const int st_chunk = (lane & 7) ^ ((lane >> 3) & 7);
    const unsigned a_off = (unsigned)(st_row * lda + st_chunk * 8), w_off = (unsigned)(st_row * ldw + st_chunk * 8);
    const unsigned short* a_tile = A + (size_t)bm * lda;
    const unsigned short* w_tile = W + (size_t)bn * ldw;
#define G8_STAGE_A(kt_, buf_, i_) g8_dma16(a_tile + ((size_t)((i_) * 64) * lda + (size_t)(kt_) * G8_BK) + a_off, &smem[(buf_) * G8_STAGE + ((i_) * 8 + wave) * 64]);
#define G8_STAGE_W(kt_, buf_, i_) g8_dma16(w_tile + ((size_t)((i_) * 64) * ldw + (size_t)(kt_) * G8_BK) + w_off, &smem[(buf_) * G8_STAGE + G8_OP + ((i_) * 8 + wave) * 64]);
#define G8_STAGE_IN(kt_, buf_) _Pragma("unroll") for (int i = 0; i < 4; ++i) { G8_STAGE_A(kt_, buf_, i) G8_STAGE_W(kt_, buf_, i) }

    f32x4 acc[4][8];                      // [column fragment of the wave][row fragment]: acc[i][j][r] = C[wm*128 + 16 j + l15][wn*64 + 16 i + 4 lq + r]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int a_row0 = wm * 128 + l15, w_row0 = wn * 64 + l15;            // (+ 16 j keeps row & 7 only if ... it does not: recomputed per fragment)
    const int KT = K / G8_BK;
#define G8_AF(buf_, j_, ks_) __builtin_bit_cast(x8, smem[(buf_) * G8_STAGE + (a_row0 + 16 * (j_)) * 8 + ((4 * (ks_) + lq) ^ ((a_row0 + 16 * (j_)) & 7))])
#define G8_WF(buf_, i_, ks_) __builtin_bit_cast(x8, smem[(buf_) * G8_STAGE + G8_OP + (w_row0 + 16 * (i_)) * 8 + ((4 * (ks_) + lq) ^ ((w_row0 + 16 * (i_)) & 7))])

    G8_STAGE_IN(0, 0)
    __syncthreads();
#if G8_PHASES
#if G8_STAGGER
    if (wm == 1) __builtin_amdgcn_s_barrier();          // the lower half of the tile runs one barrier behind the upper half
#endif
    for (int kt = 0; kt < KT; ++kt) {
        const int buf = kt & 1;
        const bool more = kt + 1 < KT;
        // quadrant q = (a half ah, b half bh): rows 64 ah .. of the wave's 128, columns 32 bh .. of its 64; 4 x 2 fragments x 2 k-steps = 16 MFMAs
        x8 af[2][4], wf[2][2];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ah = (q == 0 || q == 1) ? 0 : 1, bh = (q == 0 || q == 3) ? 0 : 1;      // (0,0) (0,1) (1,1) (1,0): one operand half changes per phase
            if (q == 0 || q == 2) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int j = 0; j < 4; ++j) af[ks][j] = G8_AF(buf, 4 * ah + j, ks);
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 2; ++i) wf[ks][i] = G8_WF(buf, 2 * bh + i, ks);
            if (more) { G8_STAGE_A(kt + 1, buf ^ 1, q) G8_STAGE_W(kt + 1, buf ^ 1, q) }       // a quarter of the next tile per phase
            __builtin_amdgcn_sched_barrier(0);
            // this wave's fragment reads have landed BEFORE the barrier (the other group, one barrier ahead, may restage what they read right
            // behind it), and (stagger) its DMA of the earlier phases too: what it staged in phase q is published by its barrier of phase q + 1
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#if G8_STAGGER
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            __builtin_amdgcn_s_barrier();
#endif
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[2 * bh + i][4 * ah + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][i], af[ks][j], acc[2 * bh + i][4 * ah + j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
#if G8_STAGGER
            if (q < 3) __builtin_amdgcn_s_barrier();
#endif
        }
        __syncthreads();                  // drains this wave's DMA (vmcnt(0)) and publishes the next stage
    }
#if G8_STAGGER
    if (wm == 0) __builtin_amdgcn_s_barrier();          // re-align the two halves
#endif
#else
    for (int kt = 0; kt < KT; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < KT) { G8_STAGE_IN(kt + 1, buf ^ 1) }
#if G8_PREFETCH
        x8 af[2][8], wf[2][4];
#pragma unroll
        for (int j = 0; j < 8; ++j) af[0][j] = G8_AF(buf, j, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) wf[0][i] = G8_WF(buf, i, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (ks == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) af[1][j] = G8_AF(buf, j, 1);
#pragma unroll
                for (int i = 0; i < 4; ++i) wf[1][i] = G8_WF(buf, i, 1);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][i], af[ks][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#else
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            x8 af[8], wf[4];
#pragma unroll
            for (int j = 0; j < 8; ++j) af[j] = G8_AF(buf, j, ks);
#pragma unroll
            for (int i = 0; i < 4; ++i) wf[i] = G8_WF(buf, i, ks);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af[j], acc[i][j], 0, 0, 0);
        }
#endif
        __syncthreads();
    }
#endif

    // ---- epilogue: the wave's 128 x 64 tile through its own 16 KiB of LDS ([row][8 chunks of 8 columns], chunk c of row r at slot c ^ (r & 7)), out as 128-byte row pieces
    uint4* so = &smem[wave * 1024];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float4 b4 = bias != nullptr ? *reinterpret_cast<const float4*>(bias + bn + wn * 64 + 16 * i + 4 * lq) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int r = 16 * j + l15;
            uint2 w2;
            w2.x = g8_pack(acc[i][j][0] + b4.x, acc[i][j][1] + b4.y);
            w2.y = g8_pack(acc[i][j][2] + b4.z, acc[i][j][3] + b4.w);
            *reinterpret_cast<uint2*>(reinterpret_cast<unsigned char*>(so) + r * 128 + (((2 * i + (lq >> 1)) ^ (r & 7)) * 16) + 8 * (lq & 1)) = w2;
        }
    }
    __builtin_amdgcn_wave_barrier();
    unsigned short* crow = C + (size_t)(bm + wm * 128) * ldc + bn + wn * 64 + 8 * (lane & 7);
#pragma unroll 8
    for (int k = 0; k < 16; ++k) {
        const int row = 8 * k + (lane >> 3);
        const uint4 v = so[row * 8 + ((lane & 7) ^ (row & 7))];
        *reinterpret_cast<uint4*>(crow + (size_t)row * ldc) = v;
    }
}

__global__ void naive_kernel(const unsigned short* A, const unsigned short* W, const float* bias, float* out, int N, int K, int nrows, const int* rows) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, ri = blockIdx.y;
    if (n >= N || ri >= nrows) return;
    const int m = rows[ri];
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += __uint_as_float(((unsigned)A[(size_t)m * K + k]) << 16) * __uint_as_float(((unsigned)W[(size_t)n * K + k]) << 16);
    out[(size_t)ri * N + n] = s + bias[n];
}

static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); }
static float bf2f(unsigned short h) { unsigned u = ((unsigned)h) << 16; float f; memcpy(&f, &u, 4); return f; }

int main(int argc, char** argv) {
    const char* variant = argc > 1 ? argv[1] : "";
    struct Shape { const char* name; int M, N, K; } shapes[] = {{"to_qkv", 12288, 2304, 768}, {"fc1", 12288, 6144, 768}, {"dec to_q", 262144, 768, 768}, {"big", 8192, 8192, 4096}};
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm8_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * G8_STAGE * 16);
    for (auto& sh : shapes) {
        const int M = sh.M, N = sh.N, K = sh.K;
        std::vector<unsigned short> hA((size_t)M * K), hW((size_t)N * K);
        std::vector<float> hb(N);
        unsigned s = 12345u;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
        for (auto& v : hA) v = f2bf(rnd());
        for (auto& v : hW) v = f2bf(rnd() * 0.05f);
        for (auto& v : hb) v = rnd();
        unsigned short *dA, *dW, *dC; float* db;
        (void)hipMalloc(&dA, hA.size() * 2); (void)hipMalloc(&dW, hW.size() * 2); (void)hipMalloc(&dC, (size_t)M * N * 2); (void)hipMalloc(&db, N * 4);
        (void)hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice); (void)hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice);
        (void)hipMemcpy(db, hb.data(), N * 4, hipMemcpyHostToDevice);
        const int tm = M / 256, tn = N / 256;
        auto launch = [&]() { gemm8_kernel<<<dim3(tm * tn), dim3(G8_THREADS), 2 * G8_STAGE * 16>>>(dA, K, dW, K, db, dC, N, K, tm, tn); };
        for (int i = 0; i < 3; ++i) launch();
        (void)hipDeviceSynchronize();
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        const int reps = 20;
        (void)hipEventRecord(e0);
        for (int i = 0; i < reps; ++i) launch();
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / reps, tf = 2.0 * M * N * K / us / 1e6;
        // correctness on 8 rows spread over the matrix, several launches (race screen)
        const int nr = 8;
        int hrows[nr] = {0, 1, 255, 256, M / 2 + 17, M - 257, M - 2, M - 1};
        int* drows; float* dref;
        (void)hipMalloc(&drows, sizeof(hrows)); (void)hipMalloc(&dref, (size_t)nr * N * 4);
        (void)hipMemcpy(drows, hrows, sizeof(hrows), hipMemcpyHostToDevice);
        naive_kernel<<<dim3((N + 255) / 256, nr), 256>>>(dA, dW, db, dref, N, K, nr, drows);
        std::vector<float> href((size_t)nr * N);
        (void)hipMemcpy(href.data(), dref, href.size() * 4, hipMemcpyDeviceToHost);
        double worst = 0; long bad = 0;
        std::vector<unsigned short> hrow(N);
        for (int rep = 0; rep < 3; ++rep) {
            launch();
            (void)hipDeviceSynchronize();
            for (int r = 0; r < nr; ++r) {
                (void)hipMemcpy(hrow.data(), dC + (size_t)hrows[r] * N, N * 2, hipMemcpyDeviceToHost);
                for (int n = 0; n < N; ++n) {
                    const double ref = href[(size_t)r * N + n], got = bf2f(hrow[n]);
                    const double e = std::fabs(got - ref) / (std::fabs(ref) + 1.0);
                    worst = e > worst ? e : worst;
                    if (e > 1e-2) ++bad;
                }
            }
        }
        printf("%-10s %-9s M=%6d N=%5d K=%5d: %8.1f us %7.1f TF/s   check: worst rel %.2e, %ld bad\n", variant, sh.name, M, N, K, us, tf, worst, bad);
        (void)hipFree(dA); (void)hipFree(dW); (void)hipFree(dC); (void)hipFree(db); (void)hipFree(drows); (void)hipFree(dref);
    }
    return 0;
}
