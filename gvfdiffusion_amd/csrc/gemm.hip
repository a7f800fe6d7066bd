// gemm.hip -- bf16 MFMA GEMM  C = A W^T (+bias) with fused epilogues, for gfx950 (MI355X).
//
// Replaces the nn.Linear launches of the reference's DiT block (model/dit.py:227-278 via
// model/attention/modules.py:112-146 and model/dit.py:128-138): to_qkv / to_q / to_kv / to_out,
// mlp.0 (+GELU-tanh), mlp.2 / to_out fused with the adaLN gate and the fp32 residual add.
// W is the nn.Linear weight as stored ([N][K], K contiguous) -- exactly the "B^T" operand MFMA wants.
//
// Structure: 128x128x32 block tile (32 KiB of LDS -> 4 workgroups per CU: the k-loop is short, K = 512 for
// most of the DiT's projections, so DMA latency is hidden by co-resident workgroups), 4 waves as
// 2x2, each wave 64x64 = 4x4 fragments of v_mfma_f32_16x16x32_bf16; operands staged
// global -> LDS by LDS-DMA (global_load_lds_dwordx4; 16-byte chunks, rows XOR-swizzled on the source side so
// the per-fragment ds_read_b128 of a 16-lane group hits 8 distinct 16-byte slots), the next k-tile's DMA
// issued before the MFMAs of the current one, one __syncthreads per k-tile, two LDS buffers.
#include <cstdlib>
#include "gvf_common.h"
#include "gvf_lp.h"
#include "../../include/gvf_rast.h"
#include "../../include/gvf_dit.h"

namespace {

typedef gvf_f32x4 f32x4;

constexpr int BN_DEFAULT = 128;                     // BM (128 or 64), BK (32 or 64) and BN (128 or 192) are template parameters
constexpr int THREADS = 256;
// chunk swizzle, slot = chunk ^ swz(row): conflict-free for the ds_read_b128 lane groups ({0-3,12-15,20-27},
// {4-11,16-19,28-31}, ...).  64-byte rows (BK = 32): 3 for rows 8..15 of each 16-row group, else 0; 128-byte rows
// (BK = 64): row & 7 (the 16 lanes of a group then hit 16 distinct 16-byte slots of the 256-byte bank row).
template <int CPR>
__device__ __forceinline__ int swz(int row) { return CPR == 4 ? ((row & 8) ? 3 : 0) : (row & 7); }

// GELU (tanh form): 0.5 x (1 + tanh(u)) = x * sigmoid(2u), u = sqrt(2/pi) (x + 0.044715 x^3): one exp, one rcp
__device__ __forceinline__ float gelu_tanh(float x) {
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    const float u = k0 * (x + k1 * x * x * x);
    return x / (1.0f + __expf(-2.0f * u));
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }   // as csrc/vae.hip's geglu kernel

// LDS-DMA of one 16-byte chunk per lane.  Kept out of the kernel template on purpose: with template-dependent
// arguments clang checks the builtin only at instantiation, where the HOST pass rejects it silently and then emits
// no stub for the kernel at all (undefined symbol at dlopen).
__device__ __forceinline__ void dma16(const unsigned short* g, uint4* l) {
    __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// BM = 128: 2x2 waves of 64x64 (4x4 fragments).  BM = 64: 2x2 waves of 32x64 (2x4 fragments), used when the
// 128-row tiling would leave the chip with fewer than two workgroups per CU (the DiT's N = 512 projections:
// 384 tiles on 256 CUs) -- twice the workgroups, so twice the DMA tiles in flight to hide the load latency.
// BK = 64 halves the number of k-tiles (barriers, DMA issue points) for the long-K projection (mlp.2, K = 2048).
// ALN: the A operand is LayerNorm(X) * s + t computed on the fly from the FP32 residual stream X (lda in floats): the row
// statistics come as `n_part` partial (sum, sum of squares) pairs per row, written by the residual epilogue of the GEMM that
// produced X (STATS below); s, t = per-column gain / shift of the row group (affine LayerNorm and / or adaLN modulate).  The
// LayerNorm pass -- a 37.7 MB read + write and a launch per sub-layer of the DiT block -- disappears; the rounding point is
// unchanged (the bf16 operand is the rounded normalised value, exactly what gvf_layernorm_modulate_bf16 writes).
struct GemmLnArgs {
    const float* stats;      // [M][n_part][2]
    int n_part;
    float eps;
    const float* ln_w; const float* ln_b;       // [K] or null
    const float* shift; const float* scale;     // row g of leading dimension mod_ld, or null
    int mod_ld, rpg;
};
constexpr int ALN_MAX_K = 1024;

// BN = 192 (2x2 waves of 64x96): for shapes whose 128-wide tiling leaves a mostly empty last round of workgroups -- the
// DiT's to_qkv (M = 12288, N = 1536): 1152 tiles of 128x128 on 1024 resident slots against 768 tiles of 128x192 on 768.
template <int DT, int EPI, int BM, int BK, bool ALN = false, int BN = BN_DEFAULT>
__global__ __launch_bounds__(THREADS, BN == 192 ? 3 : (BK == 32 ? 4 : (BM == 64 ? 3 : 2))) void gemm_lp_kernel(const unsigned short* __restrict__ A, int lda,
                                                            const unsigned short* __restrict__ W, int ldw,
                                                            const float* __restrict__ bias, void* __restrict__ Cv,
                                                            int ldc, int M, int N, int K,
                                                            const float* __restrict__ gate, int gate_ld, int rpg,
                                                            int tiles_n, float* __restrict__ stats_out, GemmLnArgs ln) {
    typedef GvfLp<DT> LP;
    typedef typename LP::x8 x8;
    constexpr int MI = BM / 32;                          // 16-row fragments per wave along M
    constexpr int NJ = BN / 32;                          // 16-column fragments per wave along N (wave tile = BM/2 x BN/2)
    constexpr int CHUNKS_PER_ROW = BK / 8;               // 16-byte chunks per tile row
    constexpr int ROWS_PER_DMA = 64 / CHUNKS_PER_ROW;    // tile rows filled by one wave-wide DMA instruction
    constexpr int LOADS_A = BM * CHUNKS_PER_ROW / THREADS, LOADS_B = BN * CHUNKS_PER_ROW / THREADS;
    // one LDS block: A buffers, then W buffers; indexed through the array itself so that the address space stays
    // LDS for the DMA builtin (a pointer variable would be generic)
    __shared__ uint4 smem[2 * (BM + BN) * CHUNKS_PER_ROW + (ALN ? 2 * ALN_MAX_K / 4 : 0)];
    float* sS = reinterpret_cast<float*>(&smem[2 * (BM + BN) * CHUNKS_PER_ROW]);      // ALN: column gain / shift of this tile's row group
    float* sT = sS + ALN_MAX_K;
#define SA(buf_, idx_) smem[(buf_) * (BM * CHUNKS_PER_ROW) + (idx_)]
#define SB(buf_, idx_) smem[2 * BM * CHUNKS_PER_ROW + (buf_) * (BN * CHUNKS_PER_ROW) + (idx_)]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware order: an XCD walks whole rows of tiles, so the A panel of a tile row is fetched over the
    // fabric once per chip (not once per XCD) and W (<= a few MiB) stays resident in every L2.
    const unsigned tile = gvf_xcd_remap(blockIdx.x, gridDim.x);
    const int tile_m = tile / tiles_n, tile_n = tile % tiles_n;
    const int bm = tile_m * BM, bn = tile_n * BN;

    f32x4 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int KT = K / BK;

    // Staging: global -> LDS directly (global_load_lds_dwordx4, no VGPR round trip, no ds_write pass).  One
    // wave-instruction fills 64 consecutive 16-byte slots = 8 tile rows; the XOR swizzle that makes the fragment
    // reads conflict-free is applied to the per-lane SOURCE chunk (LDS side stays linear, as the DMA requires).
    // Rows past M / N are clamped in-bounds; their products land in accumulator rows / columns the epilogue drops.
    const int st_row = lane / CHUNKS_PER_ROW, st_c = lane % CHUNKS_PER_ROW;
    const unsigned short* a_src[LOADS_A];
    const unsigned short* w_src[LOADS_B];
#pragma unroll
    for (int i = 0; i < LOADS_A; ++i) {
        const int row = (i * 4 + wave) * ROWS_PER_DMA + st_row;   // tile row this lane fills with load i
        const int gr = bm + row < M ? bm + row : M - 1;
        a_src[i] = A + (size_t)gr * lda + ((st_c ^ swz<CHUNKS_PER_ROW>(row)) * 8);
    }
#pragma unroll
    for (int i = 0; i < LOADS_B; ++i) {
        const int row = (i * 4 + wave) * ROWS_PER_DMA + st_row;
        const int gn = bn + row < N ? bn + row : N - 1;
        w_src[i] = W + (size_t)gn * ldw + ((st_c ^ swz<CHUNKS_PER_ROW>(row)) * 8);
    }
#define GVF_GEMM_STAGE(kt_, buf_)                                                                          \
    if (!ALN) {                                                                                            \
        _Pragma("unroll") for (int i = 0; i < LOADS_A; ++i)                                                  \
            dma16(a_src[i] + (size_t)(kt_) * BK, &SA(buf_, (i * 4 + wave) * 64));                          \
    }                                                                                                      \
    _Pragma("unroll") for (int i = 0; i < LOADS_B; ++i)                                                      \
        dma16(w_src[i] + (size_t)(kt_) * BK, &SB(buf_, (i * 4 + wave) * 64));

    // ---- ALN: register-staged A.  Thread owns chunk (st_c ^ swz) of LOADS_A rows per k-tile (the slots the DMA would fill)
    const float* x_src[ALN ? LOADS_A : 1];
    float ln_a[ALN ? LOADS_A : 1], ln_b2[ALN ? LOADS_A : 1];
    float4 araw[ALN ? LOADS_A : 1][2];
    int a_col[ALN ? LOADS_A : 1];
    if (ALN) {
        const float* X = reinterpret_cast<const float*>(A);
        const int g = bm / ln.rpg;                              // the tile lies inside one row group (checked by the host)
        for (int k = tid; k < K; k += THREADS) {
            float sv = ln.ln_w != nullptr ? ln.ln_w[k] : 1.0f, tv = ln.ln_b != nullptr ? ln.ln_b[k] : 0.0f;
            if (ln.scale != nullptr) {
                const float sc = 1.0f + ln.scale[(size_t)g * ln.mod_ld + k];
                sv *= sc; tv = tv * sc + ln.shift[(size_t)g * ln.mod_ld + k];
            }
            sS[k] = sv; sT[k] = tv;
        }
#pragma unroll
        for (int i = 0; i < LOADS_A; ++i) {
            const int row = (i * 4 + wave) * ROWS_PER_DMA + st_row;
            const int gr = bm + row < M ? bm + row : M - 1;
            a_col[i] = (st_c ^ swz<CHUNKS_PER_ROW>(row)) * 8;
            x_src[i] = X + (size_t)gr * lda + a_col[i];
            float sum = 0.f, sq = 0.f;
            for (int pt = 0; pt < ln.n_part; ++pt) {
                const float2 st2 = *reinterpret_cast<const float2*>(ln.stats + ((size_t)gr * ln.n_part + pt) * 2);
                sum += st2.x; sq += st2.y;
            }
            const float mean = sum / (float)K;
            const float var = fmaxf(sq / (float)K - mean * mean, 0.0f);
            ln_a[i] = rsqrtf(var + ln.eps);
            ln_b2[i] = -mean * ln_a[i];
        }
    }
#define GVF_GEMM_LOADA(kt_)                                                                                \
    if (ALN) {                                                                                             \
        _Pragma("unroll") for (int i = 0; i < LOADS_A; ++i) {                                                \
            araw[i][0] = *reinterpret_cast<const float4*>(x_src[i] + (size_t)(kt_) * BK);                  \
            araw[i][1] = *reinterpret_cast<const float4*>(x_src[i] + (size_t)(kt_) * BK + 4);              \
        }                                                                                                  \
    }
#define GVF_GEMM_STOREA(kt_, buf_)                                                                         \
    if (ALN) {                                                                                             \
        _Pragma("unroll") for (int i = 0; i < LOADS_A; ++i) {                                                \
            const int kc_ = (kt_) * BK + a_col[i];                                                         \
            const float4 s0 = *reinterpret_cast<const float4*>(sS + kc_), s1 = *reinterpret_cast<const float4*>(sS + kc_ + 4);   \
            const float4 t0 = *reinterpret_cast<const float4*>(sT + kc_), t1 = *reinterpret_cast<const float4*>(sT + kc_ + 4);   \
            const float4 x0 = araw[i][0], x1 = araw[i][1];                                                 \
            const float a_ = ln_a[i], b_ = ln_b2[i];                                                       \
            uint4 pk;                                                                                      \
            pk.x = LP::pack(__builtin_fmaf(__builtin_fmaf(x0.x, a_, b_), s0.x, t0.x), __builtin_fmaf(__builtin_fmaf(x0.y, a_, b_), s0.y, t0.y)); \
            pk.y = LP::pack(__builtin_fmaf(__builtin_fmaf(x0.z, a_, b_), s0.z, t0.z), __builtin_fmaf(__builtin_fmaf(x0.w, a_, b_), s0.w, t0.w)); \
            pk.z = LP::pack(__builtin_fmaf(__builtin_fmaf(x1.x, a_, b_), s1.x, t1.x), __builtin_fmaf(__builtin_fmaf(x1.y, a_, b_), s1.y, t1.y)); \
            pk.w = LP::pack(__builtin_fmaf(__builtin_fmaf(x1.z, a_, b_), s1.z, t1.z), __builtin_fmaf(__builtin_fmaf(x1.w, a_, b_), s1.w, t1.w)); \
            SA(buf_, (i * 4 + wave) * 64 + lane) = pk;                                                     \
        }                                                                                                  \
    }

    GVF_GEMM_LOADA(0)
    GVF_GEMM_STAGE(0, 0)
    // Residual epilogue (x += gate * acc on the fp32 stream): fetch this thread's 4-float pieces of the C tile NOW, so that the
    // 25 MB read of the stream overlaps the k-loop instead of sitting, latency-exposed, between the last MFMA and the store
    // (64-row tiles only: 8 float4 = 32 VGPRs; the 128-row variant has no registers to spare at 4 waves per SIMD).
    constexpr bool PRE_C = (EPI == GVF_EPI_RESID_F32) && BM == 64 && BN == 128;
    float4 cpre[PRE_C ? MI * 4 : 1];
    const bool pre_ok = PRE_C && (N % 4 == 0) && (ldc % 4 == 0) && (gate == nullptr || gate_ld % 4 == 0) &&
                        (bn + wn * 64 + (lane & 15) * 4 + 3 < N);
    if (PRE_C && pre_ok) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int step = 0; step < 4; ++step) {
                const int row = bm + wm * (BM / 2) + i * 16 + step * 4 + (lane >> 4);
                cpre[i * 4 + step] = row < M ? *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(Cv) + (size_t)row * ldc + bn + wn * 64 + (lane & 15) * 4)
                                             : make_float4(0.f, 0.f, 0.f, 0.f);
            }
    }
    if (ALN) {
        __syncthreads();      // sS / sT complete
        GVF_GEMM_STOREA(0, 0)
    }
    __syncthreads();          // drains the DMA (vmcnt(0)) and publishes the tile

    for (int kt = 0; kt < KT; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < KT) { GVF_GEMM_STAGE(kt + 1, buf ^ 1) GVF_GEMM_LOADA(kt + 1) }   // land while this tile is multiplied
#pragma unroll
        for (int ks = 0; ks < BK / 32; ++ks) {
            x8 af[MI], bfr[NJ];
            const int kc = ks * 4 + (lane >> 4);
#pragma unroll
            for (int f = 0; f < MI; ++f) {
                const int ar = wm * (BM / 2) + f * 16 + (lane & 15);
                af[f] = __builtin_bit_cast(x8, SA(buf, ar * CHUNKS_PER_ROW + (kc ^ swz<CHUNKS_PER_ROW>(ar))));
            }
#pragma unroll
            for (int f = 0; f < NJ; ++f) {
                const int br = wn * (BN / 2) + f * 16 + (lane & 15);
                bfr[f] = __builtin_bit_cast(x8, SB(buf, br * CHUNKS_PER_ROW + (kc ^ swz<CHUNKS_PER_ROW>(br))));
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = LP::mfma16(af[i], bfr[j], acc[i][j]);
        }
        if (kt + 1 < KT) { GVF_GEMM_STOREA(kt + 1, buf ^ 1) }  // buffer buf ^ 1 was last read in iteration kt - 1 (barrier in between)
        __syncthreads();
    }
#undef GVF_GEMM_STAGE
#undef GVF_GEMM_LOADA
#undef GVF_GEMM_STOREA
#undef SA
#undef SB

    // ---- epilogue.  A 16x16 accumulator fragment holds column (lane & 15), rows (lane >> 4) * 4 + r: writing it out
    // directly costs 64 scattered 2/4-byte stores per lane (measured: half of the kernel's time).  Instead each
    // wave bounces one 16-row x 64-column slab at a time through its private LDS region and emits whole 16-byte
    // vectors: one wave-instruction then covers 4 rows x 256 B (fp32) / 128 B (bf16), and the read-modify-write of
    // the fp32 residual stream, the bias and the gate are float4 accesses.
    constexpr int EP_LD = 68;                                  // floats per staged row (64 + 4 pad)
    static_assert(4 * 16 * EP_LD * 4 <= (int)sizeof(smem), "epilogue slabs must fit the operand buffers");
    float* ep = reinterpret_cast<float*>(smem) + wave * (16 * EP_LD);       // 4 x 4352 B inside the operand buffers
    __syncthreads();                                           // every wave is done reading the operand tiles
    const int col_l = lane & 15, row_l = (lane >> 4) * 4;
    const int er = lane >> 4, ec = (lane & 15) * 4;            // read-back role: row er (+4 per step), columns ec..ec+3
    const bool vec_ok = (N % 4 == 0) && (ldc % 4 == 0) && (EPI != GVF_EPI_RESID_F32 || gate == nullptr || gate_ld % 4 == 0);
    constexpr int NCG = (NJ * 16 + 63) / 64;                   // 64-column groups of the wave tile (BN = 192: 64 + 32 columns)
#pragma unroll
  for (int cg = 0; cg < NCG; ++cg) {
    constexpr int dummy_ = 0; (void)dummy_;
    const int gcols = (NJ * 16 - cg * 64) < 64 ? (NJ * 16 - cg * 64) : 64;
    const bool lane_on = ec < gcols;                            // the narrow last group leaves the upper lanes idle
    const int col0 = lane_on ? bn + wn * (BN / 2) + cg * 64 + ec : N;
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias != nullptr) {
        if (vec_ok && col0 + 3 < N) bias4 = *reinterpret_cast<const float4*>(bias + col0);
        else {
            bias4.x = col0 < N ? bias[col0] : 0.f; bias4.y = col0 + 1 < N ? bias[col0 + 1] : 0.f;
            bias4.z = col0 + 2 < N ? bias[col0 + 2] : 0.f; bias4.w = col0 + 3 < N ? bias[col0 + 3] : 0.f;
        }
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (cg * 4 + j < NJ) ep[(row_l + r) * EP_LD + j * 16 + col_l] = acc[i][cg * 4 + j][r];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int step = 0; step < 4; ++step) {
            const int lr = step * 4 + er;
            const int row = bm + wm * (BM / 2) + i * 16 + lr;
            if (EPI == GVF_EPI_GEGLU_16) {
                // the wave's 64 columns are 32 value columns then their 32 gate columns: lanes 0-7 of every 16 pair them up.  Both are rounded to
                // the operand type first -- the rounding the stored projection would have had -- so the fused result equals gemm + gvf_geglu bit for bit
                if (ec >= 32 || row >= M || col0 >= N) continue;
                const float4 va = *reinterpret_cast<const float4*>(&ep[lr * EP_LD + ec]), vg = *reinterpret_cast<const float4*>(&ep[lr * EP_LD + 32 + ec]);
                float4 bg = make_float4(0.f, 0.f, 0.f, 0.f);
                if (bias != nullptr) bg = *reinterpret_cast<const float4*>(bias + col0 + 32);
                const float a4[4] = {va.x + bias4.x, va.y + bias4.y, va.z + bias4.z, va.w + bias4.w};
                const float g4[4] = {vg.x + bg.x, vg.y + bg.y, vg.z + bg.z, vg.w + bg.w};
                float o4[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o4[e] = LP::from16(LP::to16(a4[e])) * gelu_erf(LP::from16(LP::to16(g4[e])));
                uint2 w2;
                w2.x = LP::pack(o4[0], o4[1]); w2.y = LP::pack(o4[2], o4[3]);
                *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(Cv) + (size_t)row * ldc + ((bn + wn * (BN / 2) + cg * 64) >> 1) + ec) = w2;
                continue;
            }
            float4 v = *reinterpret_cast<const float4*>(&ep[lr * EP_LD + ec]);
            v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
            if (row >= M || col0 >= N) continue;
            if (EPI == GVF_EPI_GELU_BF16) { v.x = gelu_tanh(v.x); v.y = gelu_tanh(v.y); v.z = gelu_tanh(v.z); v.w = gelu_tanh(v.w); }
            const size_t o = (size_t)row * ldc + col0;
            if (vec_ok && col0 + 3 < N) {
                if (EPI == GVF_EPI_STORE_BF16 || EPI == GVF_EPI_GELU_BF16) {
                    uint2 w2;
                    w2.x = LP::pack(v.x, v.y); w2.y = LP::pack(v.z, v.w);
                    *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(Cv) + o) = w2;
                } else if (EPI == GVF_EPI_STORE_F32) {
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(Cv) + o) = v;
                } else {
                    float4 g = make_float4(1.f, 1.f, 1.f, 1.f);
                    if (gate != nullptr) g = *reinterpret_cast<const float4*>(gate + (size_t)(row / rpg) * gate_ld + col0);
                    float4* c = reinterpret_cast<float4*>(reinterpret_cast<float*>(Cv) + o);
                    float4 x;
                    if (PRE_C && pre_ok) x = cpre[i * 4 + step];
                    else x = *c;
                    x.x += g.x * v.x; x.y += g.y * v.y; x.z += g.z * v.z; x.w += g.w * v.w;
                    *c = x;
                    if (stats_out != nullptr) {     // LayerNorm statistics of the UPDATED stream: this wave's 64 columns of the row
                        float rs = (x.x + x.y) + (x.z + x.w), rq = (x.x * x.x + x.y * x.y) + (x.z * x.z + x.w * x.w);
#pragma unroll
                        for (int d = 1; d < 16; d <<= 1) { rs += __shfl_xor(rs, d, 64); rq += __shfl_xor(rq, d, 64); }
                        if ((lane & 15) == 0)
                            *reinterpret_cast<float2*>(stats_out + ((size_t)row * (2 * tiles_n) + tile_n * 2 + wn) * 2) = make_float2(rs, rq);
                    }
                }
            } else {
                const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (col0 + e >= N) break;
                    if (EPI == GVF_EPI_STORE_BF16 || EPI == GVF_EPI_GELU_BF16) {
                        reinterpret_cast<unsigned short*>(Cv)[o + e] = LP::to16(vv[e]);
                    } else if (EPI == GVF_EPI_STORE_F32) {
                        reinterpret_cast<float*>(Cv)[o + e] = vv[e];
                    } else {
                        const float g = gate != nullptr ? gate[(size_t)(row / rpg) * gate_ld + col0 + e] : 1.0f;
                        reinterpret_cast<float*>(Cv)[o + e] += g * vv[e];
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
  }
}

}  // namespace

namespace {

template <int DT>
int launch_gemm(const void* A, int lda, const void* W, int ldw, const float* bias, void* C, int ldc, int M, int N, int K, int epilogue,
                const float* gate, int gate_ld, int rows_per_group, float* stats_out, const GemmLnArgs* ln, hipStream_t stream) {
    const bool aln = ln != nullptr;
    const bool geglu = epilogue == GVF_EPI_GEGLU_16;
    if (M < 0 || N <= 0 || K <= 0 || (K % 32) != 0 || (lda % 8) != 0 || (ldw % 8) != 0 || lda < K || ldw < K || ldc < (geglu ? N / 2 : N))
        return GVF_EINVAL;
    if (geglu && ((N % 64) != 0 || (ldc % 4) != 0 || ln != nullptr || (bias != nullptr && (((uintptr_t)bias) & 15) != 0))) return GVF_EINVAL;
    if (M == 0) return GVF_OK;
    if (!A || !W || !C) return GVF_EINVAL;
    if ((((uintptr_t)A) & 15) != 0 || (((uintptr_t)W) & 15) != 0) return GVF_EINVAL;
    if (epilogue == GVF_EPI_RESID_F32 && gate != nullptr && rows_per_group <= 0) return GVF_EINVAL;
    if (stats_out != nullptr && (epilogue != GVF_EPI_RESID_F32 || (N % 128) != 0 || (ldc % 4) != 0 || (gate != nullptr && (gate_ld % 4) != 0)))
        return GVF_EINVAL;
    (void)hipGetLastError();
    const int tiles_n = (N + BN_DEFAULT - 1) / BN_DEFAULT;
    static const int bm_override = []() { const char* e = getenv("GVF_GEMM_BM"); return e ? atoi(e) : 0; }();   // tuning aid
    const bool small = bm_override ? bm_override == 64 : ((M + 127) / 128) * tiles_n < 512;      // fewer than two 128-row workgroups per CU: use 64-row tiles
    const int bm = small ? 64 : 128;
    const int tiles_m = (M + bm - 1) / bm;
    const dim3 grid(tiles_m * tiles_n), block(THREADS);
    const unsigned short* a = (const unsigned short*)A;
    const unsigned short* w = (const unsigned short*)W;
    const int rpg = rows_per_group > 0 ? rows_per_group : 1;
    GemmLnArgs lnv = {};
    if (aln) {
        lnv = *ln;
        // every tile must lie inside one row group (one (shift, scale) row), and s / t are staged for K <= ALN_MAX_K columns
        if (K > ALN_MAX_K || lnv.n_part <= 0 || !lnv.stats || ((lnv.shift == nullptr) != (lnv.scale == nullptr)) ||
            ((lnv.ln_w == nullptr) != (lnv.ln_b == nullptr)) || (lnv.scale != nullptr && (lnv.rpg <= 0 || (lnv.rpg % bm) != 0)))
            return GVF_EINVAL;
        if (lnv.rpg <= 0) lnv.rpg = 0x7fffffff;
    }
    // (<<<>>> rather than hipLaunchKernelGGL: a parenthesised two-argument template-id inside the macro is only an
    // address-of expression and does not make clang emit the host stub of the instantiation)
    static const int bk_override = []() { const char* e = getenv("GVF_GEMM_BK"); return e ? atoi(e) : 0; }();   // tuning aid
    // 64-deep k-tiles only for the long-K projection (mlp.2, K = 2048: 47 -> 40 us).  On the K = 512 shapes they win
    // 4-7 % in a back-to-back micro-benchmark (scripts/bench_gemm.py, operands cache-hot) and LOSE 3 % of the whole
    // denoise step in place (8.65 -> 8.92 ms / NFE, A/B in one process): fewer resident workgroups per CU.
    const bool bk64 = !aln && (bk_override ? bk_override == 64 : K >= 1024);
#define GVF_GEMM_ARGS a, lda, w, ldw, bias, C, ldc, M, N, K, gate, gate_ld, rpg, tiles_n, stats_out, lnv
    // 192-wide tiles: when N is a multiple of 192 and the 128-wide grid would spill a small tail round over the resident slots
    // (measured on the denoise step: to_qkv at B = 1 -- 1152 tiles = one full round of the 1024 resident slots + a 12 % tail --
    // gains 1.4 % of the step; at B = 3 -- 3456 tiles, tail 37 % of a round -- the wider tile loses 1.7 %.)
    static const int bn192_mode = [] { const char* e = getenv("GVF_GEMM_BN192"); return e == nullptr ? 1 : atoi(e); }();   // 0 off, 1 auto, 2 always
    const int tiles128 = tiles_m * tiles_n, tiles192 = tiles_m * (N / 192), tail128 = tiles128 % 1024, tail192 = tiles192 % 768;
    const bool wide = !aln && !small && !bk64 && !geglu && stats_out == nullptr && (N % 192) == 0 && epilogue != GVF_EPI_RESID_F32 &&
                      bn192_mode != 0 && (bn192_mode == 2 || (tail128 > 0 && tail128 <= 256 && (tail192 == 0 || tail192 > 384)));
    const dim3 grid_w(tiles_m * (N / 192 > 0 ? N / 192 : 1));
    const int tiles_n_w = N / 192;
#define GVF_GEMM_LAUNCH(EPI_)                                                                                     \
    if (wide) {                                                                                                    \
        gemm_lp_kernel<DT, EPI_, 128, 32, false, 192><<<grid_w, block, 0, stream>>>(a, lda, w, ldw, bias, C, ldc, M, N, K, gate, gate_ld, rpg, tiles_n_w, stats_out, lnv); \
    } else if (aln) {                                                                                                     \
        if (small) gemm_lp_kernel<DT, EPI_, 64, 32, true><<<grid, block, 0, stream>>>(GVF_GEMM_ARGS);               \
        else gemm_lp_kernel<DT, EPI_, 128, 32, true><<<grid, block, 0, stream>>>(GVF_GEMM_ARGS);                    \
    } else if (bk64 && (K % 64) == 0) {                                                                            \
        if (small) gemm_lp_kernel<DT, EPI_, 64, 64><<<grid, block, 0, stream>>>(GVF_GEMM_ARGS);                     \
        else gemm_lp_kernel<DT, EPI_, 128, 64><<<grid, block, 0, stream>>>(GVF_GEMM_ARGS);                          \
    } else {                                                                                                       \
        if (small) gemm_lp_kernel<DT, EPI_, 64, 32><<<grid, block, 0, stream>>>(GVF_GEMM_ARGS);                     \
        else gemm_lp_kernel<DT, EPI_, 128, 32><<<grid, block, 0, stream>>>(GVF_GEMM_ARGS);                          \
    }
    switch (epilogue) {
        case GVF_EPI_STORE_BF16: GVF_GEMM_LAUNCH(GVF_EPI_STORE_BF16) break;
        case GVF_EPI_GELU_BF16: GVF_GEMM_LAUNCH(GVF_EPI_GELU_BF16) break;
        case GVF_EPI_GEGLU_16: GVF_GEMM_LAUNCH(GVF_EPI_GEGLU_16) break;
        case GVF_EPI_STORE_F32: GVF_GEMM_LAUNCH(GVF_EPI_STORE_F32) break;
        case GVF_EPI_RESID_F32: GVF_GEMM_LAUNCH(GVF_EPI_RESID_F32) break;
        default: return GVF_EINVAL;
    }
#undef GVF_GEMM_LAUNCH
#undef GVF_GEMM_ARGS
    GVF_CHECK_LAUNCH();
    return GVF_OK;
}

}  // namespace

#define GVF_GEMM_DT(dtype_, ...) ((dtype_) == GVF_DT_BF16 ? launch_gemm<0>(__VA_ARGS__) : (dtype_) == GVF_DT_F16 ? launch_gemm<1>(__VA_ARGS__) : GVF_EINVAL)

extern "C" int gvf_gemm(int dtype, const void* A, int lda, const void* W, int ldw, const float* bias, void* C, int ldc,
                        int M, int N, int K, int epilogue, const float* gate, int gate_ld, int rows_per_group, void* stream_) {
    // GVF_GEMM256=1: large plain projections on the 256-wide, one-wave-per-SIMD kernel (csrc/gemm256.hip) once there is a tile per CU.  Off by
    // default: faster in isolation, not inside the VAE decode (see that file)
    // round 6: large plain / GEGLU projections on the 256-wide EIGHT-wave kernel (csrc/gemm8.hip) once there is a tile per CU: 790-900 TFLOP/s where
    // the 128-wide one below reaches 560-640 (GVF_GEMM8=0: off -- the A/B switch; =2: from one tile on)
    static const int g8_mode = [] { const char* e = getenv("GVF_GEMM8"); return e == nullptr ? 1 : atoi(e); }();
    const int g8_tile = (g8_mode != 0 && gate == nullptr && (dtype == GVF_DT_BF16 || dtype == GVF_DT_F16)) ? gvf_gemm8_eligible(M, N, K, lda, ldw, ldc, epilogue) : 0;
    if (g8_tile != 0 && (long long)(M / g8_tile) * (N / g8_tile) >= (g8_mode == 2 ? 1 : 256) && A && W && C &&
        ((((uintptr_t)A) | ((uintptr_t)W) | ((uintptr_t)C) | ((uintptr_t)bias)) & 15) == 0)
        return gvf_gemm8(dtype, A, lda, W, ldw, bias, C, ldc, M, N, K, epilogue, stream_);
    static const int big_mode = [] { const char* e = getenv("GVF_GEMM256"); return e == nullptr ? 0 : atoi(e); }();
    if (big_mode != 0 && epilogue == GVF_EPI_STORE_BF16 && (dtype == GVF_DT_BF16 || dtype == GVF_DT_F16) && gvf_gemm256_eligible(M, N, K, lda, ldw, ldc) &&
        (long long)(M / 256) * (N / 256) >= 256 && A && W && C && ((((uintptr_t)A) | ((uintptr_t)W) | ((uintptr_t)C) | ((uintptr_t)bias)) & 15) == 0)
        return gvf_gemm256(dtype, A, lda, W, ldw, bias, C, ldc, M, N, K, stream_);
    return GVF_GEMM_DT(dtype, A, lda, W, ldw, bias, C, ldc, M, N, K, epilogue, gate, gate_ld, rows_per_group, nullptr, nullptr, (hipStream_t)stream_);
}

extern "C" int gvf_gemm_bf16(const void* A, int lda, const void* W, int ldw, const float* bias, void* C, int ldc,
                             int M, int N, int K, int epilogue, const float* gate, int gate_ld, int rows_per_group,
                             void* stream_) {
    return gvf_gemm(GVF_DT_BF16, A, lda, W, ldw, bias, C, ldc, M, N, K, epilogue, gate, gate_ld, rows_per_group, stream_);
}

extern "C" int gvf_gemm_stats_parts(int N) { return 2 * ((N + BN_DEFAULT - 1) / BN_DEFAULT); }

extern "C" int gvf_gemm_resid_stats(int dtype, const void* A, int lda, const void* W, int ldw, const float* bias, float* C, int ldc, int M, int N,
                                    int K, const float* gate, int gate_ld, int rows_per_group, float* row_stats, void* stream_) {
    if (!row_stats) return GVF_EINVAL;
    return GVF_GEMM_DT(dtype, A, lda, W, ldw, bias, C, ldc, M, N, K, GVF_EPI_RESID_F32, gate, gate_ld, rows_per_group, row_stats, nullptr,
                       (hipStream_t)stream_);
}

extern "C" int gvf_gemm_bf16_resid_stats(const void* A, int lda, const void* W, int ldw, const float* bias, float* C, int ldc, int M, int N,
                                         int K, const float* gate, int gate_ld, int rows_per_group, float* row_stats, void* stream_) {
    return gvf_gemm_resid_stats(GVF_DT_BF16, A, lda, W, ldw, bias, C, ldc, M, N, K, gate, gate_ld, rows_per_group, row_stats, stream_);
}

extern "C" int gvf_gemm_ln(int dtype, const float* X, int ldx, const float* row_stats, int n_part, float eps, const float* ln_w, const float* ln_b,
                           const float* shift, const float* scale, int mod_ld, int rows_per_group, const void* W, int ldw,
                           const float* bias, void* C, int ldc, int M, int N, int K, int epilogue, void* stream_) {
    if (epilogue == GVF_EPI_RESID_F32) return GVF_EINVAL;
    GemmLnArgs ln;
    ln.stats = row_stats; ln.n_part = n_part; ln.eps = eps; ln.ln_w = ln_w; ln.ln_b = ln_b; ln.shift = shift; ln.scale = scale;
    ln.mod_ld = mod_ld; ln.rpg = rows_per_group;
    if ((ldx % 4) != 0) return GVF_EINVAL;
    // launch_gemm's operand checks are written for bf16 rows (lda % 8): the fp32 rows need 16-byte alignment only
    return GVF_GEMM_DT(dtype, X, (ldx % 8) == 0 ? ldx : -1, W, ldw, bias, C, ldc, M, N, K, epilogue, nullptr, 0, 0, nullptr, &ln, (hipStream_t)stream_);
}

extern "C" int gvf_gemm_ln_bf16(const float* X, int ldx, const float* row_stats, int n_part, float eps, const float* ln_w, const float* ln_b,
                                const float* shift, const float* scale, int mod_ld, int rows_per_group, const void* W, int ldw,
                                const float* bias, void* C, int ldc, int M, int N, int K, int epilogue, void* stream_) {
    return gvf_gemm_ln(GVF_DT_BF16, X, ldx, row_stats, n_part, eps, ln_w, ln_b, shift, scale, mod_ld, rows_per_group, W, ldw, bias, C, ldc, M, N, K,
                       epilogue, stream_);
}
