// gemm.hip -- bf16 MFMA GEMM  C = A W^T (+bias) with fused epilogues, for gfx950 (MI355X).
//
// Replaces the nn.Linear launches of the reference's DiT block (model/dit.py:227-278 via
// model/attention/modules.py:112-146 and model/dit.py:128-138): to_qkv / to_q / to_kv / to_out,
// mlp.0 (+GELU-tanh), mlp.2 / to_out fused with the adaLN gate and the fp32 residual add.
// W is the nn.Linear weight as stored ([N][K], K contiguous) -- exactly the "B^T" operand MFMA wants.
//
// Structure (first correct version; tuning notes in DESIGN.md): 128x128x64 block tile, 4 waves as
// 2x2, each wave 64x64 = 4x4 fragments of v_mfma_f32_16x16x32_bf16; operands staged
// global -> LDS by LDS-DMA (global_load_lds_dwordx4; 16-byte chunks, rows XOR-swizzled on the source side so
// the per-fragment ds_read_b128 of a 16-lane group hits 8 distinct 16-byte slots), the next k-tile's DMA
// issued before the MFMAs of the current one, one __syncthreads per k-tile, two LDS buffers.
#include "gvf_common.h"
#include "../../include/gvf_rast.h"
#include "../../include/gvf_dit.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int THREADS = 256;
constexpr int CHUNKS_PER_ROW = BK / 8;              // 16-byte chunks per tile row
constexpr int LOADS = BM * CHUNKS_PER_ROW / THREADS;  // 4 chunks of A (and of W) per thread per k-tile

__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);  // quiet NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

__device__ __forceinline__ float gelu_tanh(float x) {
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    float u = k0 * (x + k1 * x * x * x);
    return 0.5f * x * (1.0f + tanhf(u));
}

template <int EPI>
__global__ __launch_bounds__(THREADS) void gemm_bf16_kernel(const unsigned short* __restrict__ A, int lda,
                                                            const unsigned short* __restrict__ W, int ldw,
                                                            const float* __restrict__ bias, void* __restrict__ Cv,
                                                            int ldc, int M, int N, int K,
                                                            const float* __restrict__ gate, int gate_ld, int rpg,
                                                            int tiles_n) {
    __shared__ uint4 sA[2][BM * CHUNKS_PER_ROW];
    __shared__ uint4 sB[2][BN * CHUNKS_PER_ROW];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x % tiles_n;
    const int bm = tile_m * BM, bn = tile_n * BN;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int KT = K / BK;

    // Staging: global -> LDS directly (global_load_lds_dwordx4, no VGPR round trip, no ds_write pass).  One
    // wave-instruction fills 64 consecutive 16-byte slots = 8 tile rows; the XOR swizzle that makes the fragment
    // reads conflict-free is applied to the per-lane SOURCE chunk (LDS side stays linear, as the DMA requires).
    // Rows past M / N are clamped in-bounds; their products land in accumulator rows / columns the epilogue drops.
    const int st_row = lane >> 3, st_c = lane & 7;
    const unsigned short* a_src[LOADS];
    const unsigned short* w_src[LOADS];
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
        const int row = (i * 4 + wave) * 8 + st_row;          // tile row this lane fills with load i
        const int gr = bm + row < M ? bm + row : M - 1;
        const int gn = bn + row < N ? bn + row : N - 1;
        a_src[i] = A + (size_t)gr * lda + ((st_c ^ (row & 7)) * 8);
        w_src[i] = W + (size_t)gn * ldw + ((st_c ^ (row & 7)) * 8);
    }
#define GVF_GEMM_STAGE(kt_, buf_)                                                                          \
    _Pragma("unroll") for (int i = 0; i < LOADS; ++i) {                                                      \
        __builtin_amdgcn_global_load_lds(a_src[i] + (size_t)(kt_) * BK, &sA[buf_][(i * 4 + wave) * 64], 16, 0, 0); \
        __builtin_amdgcn_global_load_lds(w_src[i] + (size_t)(kt_) * BK, &sB[buf_][(i * 4 + wave) * 64], 16, 0, 0); \
    }

    GVF_GEMM_STAGE(0, 0)
    __syncthreads();          // drains the DMA (vmcnt(0)) and publishes the tile

    for (int kt = 0; kt < KT; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < KT) { GVF_GEMM_STAGE(kt + 1, buf ^ 1) }   // lands while this tile is multiplied
#pragma unroll
        for (int ks = 0; ks < BK / 32; ++ks) {
            bf16x8 af[4], bfr[4];
            const int kc = ks * 4 + (lane >> 4);
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const int ar = wm * 64 + f * 16 + (lane & 15);
                const int br = wn * 64 + f * 16 + (lane & 15);
                af[f] = __builtin_bit_cast(bf16x8, sA[buf][ar * CHUNKS_PER_ROW + (kc ^ (ar & 7))]);
                bfr[f] = __builtin_bit_cast(bf16x8, sB[buf][br * CHUNKS_PER_ROW + (kc ^ (br & 7))]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
#undef GVF_GEMM_STAGE

    // epilogue.  16x16 accumulator fragment: column = lane & 15, rows = (lane >> 4) * 4 + r
    const int col_l = lane & 15, row_l = (lane >> 4) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = bn + wn * 64 + j * 16 + col_l;
            if (col >= N) continue;
            const float bv = bias != nullptr ? bias[col] : 0.0f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = bm + wm * 64 + i * 16 + row_l + r;
                if (row >= M) continue;
                float v = acc[i][j][r] + bv;
                const size_t o = (size_t)row * ldc + col;
                if (EPI == GVF_EPI_STORE_BF16) {
                    reinterpret_cast<unsigned short*>(Cv)[o] = f32_to_bf16(v);
                } else if (EPI == GVF_EPI_GELU_BF16) {
                    reinterpret_cast<unsigned short*>(Cv)[o] = f32_to_bf16(gelu_tanh(v));
                } else if (EPI == GVF_EPI_STORE_F32) {
                    reinterpret_cast<float*>(Cv)[o] = v;
                } else {
                    float g = gate != nullptr ? gate[(size_t)(row / rpg) * gate_ld + col] : 1.0f;
                    float* c = reinterpret_cast<float*>(Cv) + o;
                    *c = *c + g * v;
                }
            }
        }
    }
}

}  // namespace

extern "C" int gvf_gemm_bf16(const void* A, int lda, const void* W, int ldw, const float* bias, void* C, int ldc,
                             int M, int N, int K, int epilogue, const float* gate, int gate_ld, int rows_per_group,
                             void* stream_) {
    if (M < 0 || N <= 0 || K <= 0 || (K % BK) != 0 || (lda % 8) != 0 || (ldw % 8) != 0 || lda < K || ldw < K || ldc < N)
        return GVF_EINVAL;
    if (M == 0) return GVF_OK;
    if (!A || !W || !C) return GVF_EINVAL;
    if ((((uintptr_t)A) & 15) != 0 || (((uintptr_t)W) & 15) != 0) return GVF_EINVAL;
    if (epilogue == GVF_EPI_RESID_F32 && gate != nullptr && rows_per_group <= 0) return GVF_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    (void)hipGetLastError();
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    const dim3 grid(tiles_m * tiles_n), block(THREADS);
    const unsigned short* a = (const unsigned short*)A;
    const unsigned short* w = (const unsigned short*)W;
    const int rpg = rows_per_group > 0 ? rows_per_group : 1;
    switch (epilogue) {
        case GVF_EPI_STORE_BF16:
            hipLaunchKernelGGL(gemm_bf16_kernel<GVF_EPI_STORE_BF16>, grid, block, 0, stream, a, lda, w, ldw, bias, C, ldc, M, N, K, gate, gate_ld, rpg, tiles_n);
            break;
        case GVF_EPI_GELU_BF16:
            hipLaunchKernelGGL(gemm_bf16_kernel<GVF_EPI_GELU_BF16>, grid, block, 0, stream, a, lda, w, ldw, bias, C, ldc, M, N, K, gate, gate_ld, rpg, tiles_n);
            break;
        case GVF_EPI_STORE_F32:
            hipLaunchKernelGGL(gemm_bf16_kernel<GVF_EPI_STORE_F32>, grid, block, 0, stream, a, lda, w, ldw, bias, C, ldc, M, N, K, gate, gate_ld, rpg, tiles_n);
            break;
        case GVF_EPI_RESID_F32:
            hipLaunchKernelGGL(gemm_bf16_kernel<GVF_EPI_RESID_F32>, grid, block, 0, stream, a, lda, w, ldw, bias, C, ldc, M, N, K, gate, gate_ld, rpg, tiles_n);
            break;
        default:
            return GVF_EINVAL;
    }
    GVF_CHECK_LAUNCH();
    return GVF_OK;
}
