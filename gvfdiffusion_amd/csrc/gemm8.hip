// gemm8.hip -- the plain projection  C[M][N] = epi( A[M][K] W[N][K]^T + bias )  for LARGE outputs, 16-bit operands, gfx950 (MI355X), round 6.
//
// 256 x 256 x 64 workgroup tile, EIGHT waves = two per SIMD (2 along M x 4 along N, wave tile 128 x 64 = 8 x 4 accumulators of v_mfma_f32_16x16x32:
// 128 accumulator registers, ~195 in all), operands by LDS-DMA into two 64 KiB stages (chunk swizzle on the SOURCE side, 128-byte tile rows:
// slot = chunk ^ (row & 7), conflict-free ds_read_b128 fragments), ONE barrier per k-tile, the k-loop left to the compiler's scheduler: while one
// wave of a SIMD waits for its 24 fragment reads the other one issues its 64 MFMAs -- the ping-pong MI355X_MICROARCH.md describes falls out of
// the residency, no barrier choreography needed.  Measured in isolation (scripts/ubench/gemm8_bench.hip, profiles/r06_gemm8.txt; bf16):
//     to_qkv  12288 x 2304 x 768   868 TFLOP/s   (gemm.hip's 128 x 128 x 32 tiles at four workgroups per CU: 614-640; hipBLASLt 957-978)
//     fc1     12288 x 6144 x 768   904           (557-604; 908-934)
//     to_q   262144 x  768 x 768   793           (618-637; 1040-1063)
//     8192 x 8192 x 4096          1154           (gemm256.hip, the same tile on FOUR waves: 980)
// Three more elaborate schedules of the same tile were measured and lost: fragments of the next k-step requested ahead in a second register
// set (-1 %), the k-tile as four 16-MFMA quadrants with s_setprio around them (-5 %), and that with the two M-halves one barrier apart (-5 .. -12 %).
// What the 128-wide kernel lacks is LDS bandwidth: four workgroups of 4 waves read one fragment per two MFMAs and stage 64 KiB per round, ~75 % of
// the LDS cycles of an MFMA-bound loop; the 128 x 64 wave tile reads 0.375 fragments per MFMA and stages half as much per flop.
//
// The N = 768 residual projections of a latent block (to_out, mlp.2: 12 288 x 768, x += a w^T + b on the fp32 stream) have 144 tiles of 256^2 for 256 CUs;
// they run the same loop on 192 x 192 tiles (T = 192: wave tile 96 x 48, 72 accumulator registers, 96 KiB of LDS): 64 x 4 = exactly one tile per CU,
// each XCD 8 tile rows = its own band of A and all of W.  to_out 410 -> 501 TFLOP/s (hipBLASLt 515), mlp.2 (K = 3072) 596 -> 850 (929).
//
// Epilogues: GVF_EPI_RESID_F32 without a gate (T = 192: 16-byte read-modify-writes straight from the accumulators), GVF_EPI_STORE_BF16 (16-bit store, bias) and GVF_EPI_GEGLU_16 (the wave's 64 columns are 32 value columns then their 32 gate columns,
// csrc/gemm.hip's convention: value and gate of one output sit in ONE lane here, so the GEGLU is register arithmetic; both rounded to the operand
// type first = bit-identical to the store epilogue + gvf_geglu).  The tile leaves through the (free) operand stages as whole row pieces.
// GVF_EPI_STORE_F32 (fp32 store + bias from the accumulators, ANY M: the last row tile stages row M - 1 for the rows past the end and skips their
// stores -- the hoisted condition projections of DiT.prepare_conditions, 24 x 1370 image tokens: 646 -> ~850 TFLOP/s).
// gvf_gemm routes eligible calls here (M, N multiples of the tile, K of 64, at least one tile per CU); GVF_GEMM8=0 switches it off.
// In place (motion-VAE decode, cold operands): latent blocks 5.65 -> 4.84 ms, decode 17.07 -> 16.65 ms bf16 (profiles/r06_gemm8.txt).
#include <cstdlib>
#include "gvf_common.h"
#include "gvf_lp.h"
#include "../../include/gvf_rast.h"
#include "../../include/gvf_dit.h"

namespace {

typedef gvf_f32x4 f32x4;

constexpr int G8_THREADS = 512;
constexpr int G8_BK = 64;
// T = tile rows (M) and columns (N): 256, or 192 -- the square that gives the N = 768 projections of a 12 288-row latent block exactly one tile per CU
// (64 x 4 tiles, each XCD 8 tile rows = its own band of A and all of W; 256-wide tiles leave 144 workgroups for 256 CUs).  Wave tile T/2 x T/4.

__device__ __forceinline__ void g8_dma16(const unsigned short* g, uint4* l) {
    __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

__device__ __forceinline__ float g8_gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }   // as csrc/gemm.hip / csrc/vae.hip

template <int DT, int EPI, int T>
__global__ __launch_bounds__(G8_THREADS, 1) void gemm8_kernel(const unsigned short* __restrict__ A, int lda, const unsigned short* __restrict__ W, int ldw,
                                                              const float* __restrict__ bias, void* __restrict__ Cv, int ldc, int M, int K,
                                                              int tiles_m, int tiles_n) {
    typedef GvfLp<DT> LP;
    typedef typename LP::x8 x8;
    constexpr int G8_T = T;
    constexpr int G8_OP = G8_T * 8;           // 16-byte chunks of one operand tile (T rows x 64 k)
    constexpr int G8_STAGE = 2 * G8_OP;       // A tile, then W tile
    constexpr int FM = T / 32, FN = T / 64;   // 16-row / 16-column fragments of the wave tile (8 x 4 or 6 x 3)
    constexpr int NDMA = T / 64;              // DMA instructions per wave, operand and k-tile (64 lanes x 16 B = 8 tile rows each)
    unsigned short* C = reinterpret_cast<unsigned short*>(Cv);
    extern __shared__ __attribute__((aligned(16))) uint4 g8_smem[];          // [2 stages][A | W]: 128 KiB (T = 256) / 96 KiB (T = 192)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3, l15 = lane & 15, lq = lane >> 4;

    // workgroup b runs on XCD b % 8 (round-robin dispatch): an XCD keeps whole N-tiles of W, or whole bands of tile rows, in its L2
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    int tile_m, tile_n;
    if ((tiles_n & 7) == 0) { const int npx = tiles_n >> 3; tile_n = xcd * npx + slot % npx; tile_m = slot / npx; }
    else if ((tiles_m & 7) == 0) { const int mpx = tiles_m >> 3; tile_m = xcd * mpx + slot / tiles_n; tile_n = slot % tiles_n; }
    else { tile_m = (int)blockIdx.x / tiles_n; tile_n = (int)blockIdx.x % tiles_n; }
    const int bm = tile_m * G8_T, bn = tile_n * G8_T;

    // staging: instruction i (0..3) of wave w fills LDS slots (8 i + w) * 64 + lane = tile rows (8 i + w) * 8 + lane / 8, chunk slot lane % 8; the
    // source chunk is (lane % 8) ^ (row & 7) with row & 7 = lane / 8 (swizzle on the source side, the LDS side is linear as the DMA requires)
    const int st_row = wave * 8 + (lane >> 3);
    const int st_chunk = (lane & 7) ^ ((lane >> 3) & 7);
    const unsigned a_off = (unsigned)(st_row * lda + st_chunk * 8), w_off = (unsigned)(st_row * ldw + st_chunk * 8);
    const unsigned short* a_tile = A + (size_t)bm * lda;
    const unsigned short* w_tile = W + (size_t)bn * ldw;
    // GVF_EPI_STORE_F32 takes any M: the last row tile stages row M - 1 in place of the rows past the end (never stored)
    constexpr bool RAGGED = EPI == GVF_EPI_STORE_F32;
    size_t a_rag[NDMA];
    if constexpr (RAGGED) {
#pragma unroll
        for (int i = 0; i < NDMA; ++i) { const int r = bm + i * 64 + st_row; a_rag[i] = (size_t)(r < M ? r : M - 1) * lda + st_chunk * 8; }
    }
#define G8_A_SRC(i_, kt_) (RAGGED ? A + a_rag[i_] + (size_t)(kt_) * G8_BK : a_tile + ((size_t)((i_) * 64) * lda + (size_t)(kt_) * G8_BK) + a_off)
#define G8_STAGE_IN(kt_, buf_)                                                                                       \
    _Pragma("unroll") for (int i = 0; i < NDMA; ++i) {                                                               \
        g8_dma16(G8_A_SRC(i, kt_), &g8_smem[(buf_) * G8_STAGE + (i * 8 + wave) * 64]);                                                            \
        g8_dma16(w_tile + ((size_t)(i * 64) * ldw + (size_t)(kt_) * G8_BK) + w_off, &g8_smem[(buf_) * G8_STAGE + G8_OP + (i * 8 + wave) * 64]);  \
    }

    f32x4 acc[FN][FM];                    // [column fragment][row fragment]: acc[i][j][r] = C[bm + (T/2) wm + 16 j + l15][bn + (T/4) wn + 16 i + 4 lq + r]
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int a_row0 = wm * (T / 2) + l15, w_row0 = wn * (T / 4) + l15;   // (+ 16 j keeps row & 7)
    const int a_sw = a_row0 & 7, w_sw = w_row0 & 7;
    const int KT = K / G8_BK;
    G8_STAGE_IN(0, 0)
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < KT) { G8_STAGE_IN(kt + 1, buf ^ 1) }              // lands while this tile is multiplied
        const uint4* sA = &g8_smem[buf * G8_STAGE], *sW = sA + G8_OP;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            x8 af[FM], wf[FN];
#pragma unroll
            for (int j = 0; j < FM; ++j) af[j] = __builtin_bit_cast(x8, sA[(a_row0 + 16 * j) * 8 + ((4 * ks + lq) ^ a_sw)]);
#pragma unroll
            for (int i = 0; i < FN; ++i) wf[i] = __builtin_bit_cast(x8, sW[(w_row0 + 16 * i) * 8 + ((4 * ks + lq) ^ w_sw)]);
#pragma unroll
            for (int i = 0; i < FN; ++i)
#pragma unroll
                for (int j = 0; j < FM; ++j) acc[i][j] = LP::mfma16(wf[i], af[j], acc[i][j]);     // D[n][m]: a lane gets 4 consecutive columns of one row
        }
        __syncthreads();                  // drains this wave's DMA (vmcnt(0)), publishes the next stage; everybody is done with this one
    }
#undef G8_STAGE_IN
#undef G8_A_SRC

    if constexpr (EPI == GVF_EPI_STORE_F32) {
        // C = acc + bias in fp32, straight from the accumulators (a lane holds 4 consecutive columns of a row: one 16-byte store); rows past M skipped
        const int row0 = bm + wm * (T / 2) + l15;
        float* Cf = reinterpret_cast<float*>(Cv) + (size_t)row0 * ldc + bn + wn * (T / 4) + 4 * lq;
#pragma unroll
        for (int i = 0; i < FN; ++i) {
            const float4 b4 = bias != nullptr ? *reinterpret_cast<const float4*>(bias + bn + wn * (T / 4) + 16 * i + 4 * lq) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int j = 0; j < FM; ++j)
                if (row0 + 16 * j < M)
                    *reinterpret_cast<float4*>(Cf + (size_t)(16 * j) * ldc + 16 * i) =
                        make_float4(acc[i][j][0] + b4.x, acc[i][j][1] + b4.y, acc[i][j][2] + b4.z, acc[i][j][3] + b4.w);
        }
    } else if constexpr (EPI == GVF_EPI_RESID_F32) {
        // x += acc + bias on the fp32 stream, straight from the accumulators: a lane holds 4 consecutive columns of a row (one 16-byte
        // read-modify-write; a wave instruction covers 16 rows x 64 bytes)
        float* Cf = reinterpret_cast<float*>(Cv) + (size_t)(bm + wm * (T / 2) + l15) * ldc + bn + wn * (T / 4) + 4 * lq;
#pragma unroll
        for (int i = 0; i < FN; ++i) {
            const float4 b4 = bias != nullptr ? *reinterpret_cast<const float4*>(bias + bn + wn * (T / 4) + 16 * i + 4 * lq) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int j = 0; j < FM; ++j) {
                float4* c = reinterpret_cast<float4*>(Cf + (size_t)(16 * j) * ldc + 16 * i);
                float4 x = *c;
                x.x += acc[i][j][0] + b4.x; x.y += acc[i][j][1] + b4.y; x.z += acc[i][j][2] + b4.z; x.w += acc[i][j][3] + b4.w;
                *c = x;
            }
        }
    } else {
    // ---- 16-bit epilogues: the wave's tile through its own slice of the (now free) stages, out as whole row pieces
    static_assert(T == 256, "the 16-bit epilogues are written for the 128 x 64 wave tile");
    uint4* so = &g8_smem[wave * 1024];
    if constexpr (EPI == GVF_EPI_GEGLU_16) {
        // columns 16 i + 4 lq + r of the wave's 64: i = 0, 1 are values, i = 2, 3 their gates -> 32 output columns, [row][4 chunks of 8 columns],
        // chunk c of row m at slot c ^ ((m >> 1) & 3)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), bg = bv;
            if (bias != nullptr) {
                bv = *reinterpret_cast<const float4*>(bias + bn + wn * 64 + 16 * i + 4 * lq);
                bg = *reinterpret_cast<const float4*>(bias + bn + wn * 64 + 32 + 16 * i + 4 * lq);
            }
            const float bvv[4] = {bv.x, bv.y, bv.z, bv.w}, bgg[4] = {bg.x, bg.y, bg.z, bg.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int m = 16 * j + l15;
                float o4[4];
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    o4[r] = LP::from16(LP::to16(acc[i][j][r] + bvv[r])) * g8_gelu_erf(LP::from16(LP::to16(acc[i + 2][j][r] + bgg[r])));
                uint2 w2;
                w2.x = LP::pack(o4[0], o4[1]); w2.y = LP::pack(o4[2], o4[3]);
                *reinterpret_cast<uint2*>(reinterpret_cast<unsigned char*>(so) + m * 64 + (((2 * i + (lq >> 1)) ^ ((m >> 1) & 3)) * 16) + 8 * (lq & 1)) = w2;
            }
        }
        __builtin_amdgcn_wave_barrier();
        unsigned short* crow = C + (size_t)(bm + wm * 128) * ldc + ((bn + wn * 64) >> 1) + 8 * (lane & 3);
#pragma unroll 8
        for (int k = 0; k < 8; ++k) {
            const int row = 16 * k + (lane >> 2);
            const uint4 v = so[row * 4 + ((lane & 3) ^ ((row >> 1) & 3))];
            *reinterpret_cast<uint4*>(crow + (size_t)row * ldc) = v;
        }
        return;
    }
    if constexpr (EPI == GVF_EPI_GEGLU_16) return;          // (the store form below indexes the full wave tile)
    // store: [row][8 chunks of 8 columns], chunk c of row m at slot c ^ (m & 7)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float4 b4 = bias != nullptr ? *reinterpret_cast<const float4*>(bias + bn + wn * 64 + 16 * i + 4 * lq) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int m = 16 * j + l15;
            uint2 w2;
            w2.x = LP::pack(acc[i][j][0] + b4.x, acc[i][j][1] + b4.y);
            w2.y = LP::pack(acc[i][j][2] + b4.z, acc[i][j][3] + b4.w);
            *reinterpret_cast<uint2*>(reinterpret_cast<unsigned char*>(so) + m * 128 + (((2 * i + (lq >> 1)) ^ (m & 7)) * 16) + 8 * (lq & 1)) = w2;
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
}

template <int DT, int EPI, int T>
int g8_launch(const void* A, int lda, const void* W, int ldw, const float* bias, void* C, int ldc, int M, int N, int K, hipStream_t stream) {
    static GvfPerDeviceOnce once;                                           // per instantiation and per device (gvf_common.h)
    constexpr int lds_bytes = 2 * (2 * T * 8) * 16;
    if (!gvf_once_per_device(once, [] {
            return hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm8_kernel<DT, EPI, T>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) ==
                   hipSuccess;
        }))
        return GVF_ELAUNCH;
    const int tiles_m = (M + T - 1) / T, tiles_n = N / T;            // (only GVF_EPI_STORE_F32 is let in with a ragged M: g8_tile)
    gemm8_kernel<DT, EPI, T><<<dim3((unsigned)(tiles_m * tiles_n)), dim3(G8_THREADS), lds_bytes, stream>>>(
        (const unsigned short*)A, lda, (const unsigned short*)W, ldw, bias, C, ldc, M, K, tiles_m, tiles_n);
    return GVF_OK;
}

// the tile this call would run on: 256, 192 (residual epilogue only) or 0 = not eligible
int g8_tile(int M, int N, int K, int lda, int ldw, int ldc, int epilogue) {
    if (epilogue != GVF_EPI_STORE_BF16 && epilogue != GVF_EPI_GEGLU_16 && epilogue != GVF_EPI_RESID_F32 && epilogue != GVF_EPI_STORE_F32) return 0;
    const int n_out = epilogue == GVF_EPI_GEGLU_16 ? N / 2 : N;
    const int cal = (epilogue == GVF_EPI_RESID_F32 || epilogue == GVF_EPI_STORE_F32) ? 4 : 8;                  // C rows: 16-byte aligned pieces
    if (!(M > 0 && N > 0 && K > 0 && (K % G8_BK) == 0 && (lda % 8) == 0 && (ldw % 8) == 0 && (ldc % cal) == 0 && lda >= K && ldw >= K && ldc >= n_out)) return 0;
    if (epilogue == GVF_EPI_RESID_F32) {
        // the fp32 residual epilogue exists for the shapes it was built for: squares of 192 (the motion VAE's to_out / mlp.2: 12 288 x 768)
        return ((M % 192) == 0 && (N % 192) == 0 && (long long)(M / 192) * (N / 192) <= 0x7fffffffLL) ? 192 : 0;
    }
    if (epilogue == GVF_EPI_STORE_F32)       // any M (the hoisted condition projections: 24 x 1370 image tokens); the last row tile is guarded
        return ((N % 256) == 0 && (long long)((M + 255) / 256) * (N / 256) <= 0x7fffffffLL) ? 256 : 0;
    return ((M % 256) == 0 && (N % 256) == 0 && (long long)(M / 256) * (N / 256) <= 0x7fffffffLL) ? 256 : 0;
}

}  // namespace

extern "C" int gvf_gemm8_eligible(int M, int N, int K, int lda, int ldw, int ldc, int epilogue) { return g8_tile(M, N, K, lda, ldw, ldc, epilogue); }

extern "C" int gvf_gemm8(int dtype, const void* A, int lda, const void* W, int ldw, const float* bias, void* C, int ldc, int M, int N, int K,
                         int epilogue, void* stream_) {
    if (dtype != GVF_DT_BF16 && dtype != GVF_DT_F16) return GVF_EINVAL;
    const int tile = g8_tile(M, N, K, lda, ldw, ldc, epilogue);
    if (tile == 0) return GVF_EINVAL;
    if (!A || !W || !C) return GVF_EINVAL;
    if ((((uintptr_t)A) & 15) || (((uintptr_t)W) & 15) || (((uintptr_t)C) & 15) || (bias != nullptr && (((uintptr_t)bias) & 15))) return GVF_EINVAL;
    (void)hipGetLastError();
    int rc = GVF_OK;
    if (epilogue == GVF_EPI_GEGLU_16) {
        GVF_LP_DISPATCH(dtype, rc = (g8_launch<DT, GVF_EPI_GEGLU_16, 256>(A, lda, W, ldw, bias, C, ldc, M, N, K, (hipStream_t)stream_)));
    } else if (epilogue == GVF_EPI_STORE_F32) {
        GVF_LP_DISPATCH(dtype, rc = (g8_launch<DT, GVF_EPI_STORE_F32, 256>(A, lda, W, ldw, bias, C, ldc, M, N, K, (hipStream_t)stream_)));
    } else if (epilogue == GVF_EPI_RESID_F32) {
        GVF_LP_DISPATCH(dtype, rc = (g8_launch<DT, GVF_EPI_RESID_F32, 192>(A, lda, W, ldw, bias, C, ldc, M, N, K, (hipStream_t)stream_)));
    } else {
        GVF_LP_DISPATCH(dtype, rc = (g8_launch<DT, GVF_EPI_STORE_BF16, 256>(A, lda, W, ldw, bias, C, ldc, M, N, K, (hipStream_t)stream_)));
    }
    if (rc != GVF_OK) return rc;
    GVF_CHECK_LAUNCH();
    return GVF_OK;
}
