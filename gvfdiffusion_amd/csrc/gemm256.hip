// gemm256.hip -- the plain projection  C[M][N] = r16( A[M][K] W[N][K]^T + bias )  for LARGE outputs, 16-bit operands, gfx950 (MI355X).
//
// An OPT-IN kernel (gvf_gemm256 directly, or GVF_GEMM256=1 for gvf_gemm's store epilogue) -- the other corner of the design space from
// gemm.hip (128 x 128 x 32 tiles, 64 x 64 wave tiles, four workgroups per CU, 31 bytes of LDS per kflop; 580-640 TFLOP/s on the motion VAE's
// latent-block shapes where hipBLASLt reaches 930-1050, profiles/r04b_gemm_vae_shapes.txt): ONE wave per SIMD with the whole register file, as
// in attn_xt64.hip --
//   * 256 x 256 x 64 workgroup tile, 4 waves (2 x 2), wave tile 128 x 128 = 4 x 4 accumulators of v_mfma_f32_32x32x16 (256 accumulator
//     registers); 16 bytes of LDS and 7.8 bytes of L2 per kflop, 2048 MFMA cycles per wave between two barriers;
//   * operands by LDS-DMA (global_load_lds_dwordx4) into two 64 KiB stages, the chunk swizzle on the SOURCE side ((row >> 1) & 7: the
//     32-row fragment reads are conflict-free ds_read_b128, see attn_xt64.hip); fragments of the next k-step requested before the MFMAs of
//     this one (fenced: hipcc on its own waits lgkmcnt(0) in front of every four MFMAs);
//   * D[n][m] = W-fragment x A-fragment: a lane ends up with 4 consecutive columns of one row of C; the tile leaves through the (now free)
//     128 KiB of LDS as whole 256-byte row pieces;
//   * workgroups are dealt to the XCDs so that an XCD's L2 holds what its workgroups share: whole N-tiles of W per XCD when the number of
//     N-tiles divides by 8 (mlp.0 of the VAE: 9.4 MB of W, 1.2 MB per XCD), else whole bands of tile rows per XCD.
// Measured (profiles/r04_gemm256.txt): 690-720 TFLOP/s on the VAE's three large projections in isolation (gemm.hip 580-640), 980 at
// M = N = 8192, K = 4096 (hipBLASLt 1490) -- at one workgroup per CU every CU pulls its own 64 KiB per k-tile from L2, 7.6 TB/s chip-wide at
// that rate, and a ring of four 32-deep stages requested three tiles ahead (counted vmcnt) is 5-10 % SLOWER: the loop waits on the L2, not on
// latency.  Inside the decode, where the operands are not cache-hot, it does not beat gemm.hip (latent blocks 5.87 vs 5.86 ms; the 262 144-row
// to_q + 0.3 ms), so gvf_gemm keeps its 128-wide kernel unless asked.
// Requirements (checked by the launcher): M, N multiples of 256, K of 64, 16-byte aligned rows and pointers.
#include <cstdlib>
#include <mutex>
#include "gvf_common.h"
#include "gvf_lp.h"
#include "../../include/gvf_rast.h"
#include "../../include/gvf_dit.h"

namespace {

typedef gvf_f32x16 f32x16;

constexpr int G2_THREADS = 256;
constexpr int G2_T = 256;                 // tile rows (M) and columns (N)
constexpr int G2_BK = 64;
constexpr int G2_OP = G2_T * 8;           // 16-byte chunks of one operand tile (256 rows x 64 k)
constexpr int G2_STAGE = 2 * G2_OP;       // A tile, then W tile

__device__ __forceinline__ void g2_dma16(const unsigned short* g, uint4* l) {
    __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

template <int DT>
__global__ __launch_bounds__(G2_THREADS) void gemm256_kernel(const unsigned short* __restrict__ A, int lda, const unsigned short* __restrict__ W, int ldw,
                                                             const float* __restrict__ bias, unsigned short* __restrict__ C, int ldc, int K,
                                                             int tiles_m, int tiles_n) {
    typedef GvfLp<DT> LP;
    typedef typename LP::x8 x8;
    extern __shared__ __attribute__((aligned(16))) uint4 g2_smem[];          // [2 stages][A: 2048 chunks | W: 2048 chunks]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, half = lane >> 5;

    // ---- which tile: workgroup b runs on XCD b % 8 (round-robin dispatch)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    int tile_m, tile_n;
    if ((tiles_n & 7) == 0) { const int npx = tiles_n >> 3; tile_n = xcd * npx + slot % npx; tile_m = slot / npx; }
    else if ((tiles_m & 7) == 0) { const int mpx = tiles_m >> 3; tile_m = xcd * mpx + slot / tiles_n; tile_n = slot % tiles_n; }
    else { tile_m = (int)blockIdx.x / tiles_n; tile_n = (int)blockIdx.x % tiles_n; }
    const int bm = tile_m * G2_T, bn = tile_n * G2_T;

    // ---- staging: instruction i of wave w fills tile rows (4 i + w) * 8 + lane / 8 (LDS slot = row * 8 + lane % 8, linear in the lane); the source
    // chunk is (lane % 8) ^ ((row >> 1) & 7), and (row >> 1) & 7 = (4 w + lane / 16) & 7 does not depend on i: one base pointer per operand
    const int st_row = wave * 8 + (lane >> 3);
    const int st_chunk = (lane & 7) ^ ((st_row >> 1) & 7);
    // (addresses = wave-uniform base (SGPRs: tile row block, k-tile) + one 32-bit lane offset per operand: the sixteen requests of a k-tile share two
    // offset registers instead of holding sixteen 64-bit pointers)
    const unsigned a_off = (unsigned)(st_row * lda + st_chunk * 8), w_off = (unsigned)(st_row * ldw + st_chunk * 8);
    const unsigned short* a_tile = A + (size_t)bm * lda;
    const unsigned short* w_tile = W + (size_t)bn * ldw;
#define G2_STAGE_IN(kt_, buf_)                                                                             \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                        \
        g2_dma16(a_tile + ((size_t)(i * 32) * lda + (size_t)(kt_) * G2_BK) + a_off, &g2_smem[(buf_) * G2_STAGE + (i * 4 + wave) * 64]);            \
        g2_dma16(w_tile + ((size_t)(i * 32) * ldw + (size_t)(kt_) * G2_BK) + w_off, &g2_smem[(buf_) * G2_STAGE + G2_OP + (i * 4 + wave) * 64]);    \
    }

    f32x16 acc[4][4];                     // [n tile of the wave][m tile of the wave]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment (32 rows, k-step ks of 16): lane (row, half) reads chunk 2 ks + half of its row
    const int a_row = wm * 128 + l31, w_row = wn * 128 + l31;
    const int a_sw = (a_row >> 1) & 7, w_sw = (w_row >> 1) & 7;       // (+ 32 j keeps (row >> 1) & 7)

    const int KT = K / G2_BK;
    G2_STAGE_IN(0, 0)
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < KT) { G2_STAGE_IN(kt + 1, buf ^ 1) }            // lands while this tile is multiplied
        const uint4* sA = &g2_smem[buf * G2_STAGE], *sW = sA + G2_OP;
        // fragments of k-step ks + 1 are requested BEFORE the 16 MFMAs of k-step ks (two register sets, every group fenced: left to itself
        // hipcc sinks each read to its first use and waits lgkmcnt(0) in front of every group of four MFMAs -- 60 % of this loop's speed)
        x8 af[2][4], wf[2][4];
#define G2_LOAD(ks_, set_)                                                                                                              \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) af[set_][j] = __builtin_bit_cast(x8, sA[(a_row + 32 * j) * 8 + ((2 * (ks_) + half) ^ a_sw)]); \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) wf[set_][i] = __builtin_bit_cast(x8, sW[(w_row + 32 * i) * 8 + ((2 * (ks_) + half) ^ w_sw)]); \
        __builtin_amdgcn_sched_barrier(0);
        G2_LOAD(0, 0)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks + 1 < 4) { G2_LOAD(ks + 1, (ks + 1) & 1) }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = LP::mfma32(wf[ks & 1][i], af[ks & 1][j], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
        }
#undef G2_LOAD
        __syncthreads();                  // drains this wave's DMA (vmcnt(0)) and publishes the next stage; everybody is done with this one
    }
#undef G2_STAGE_IN

    // ---- epilogue: acc[i][j][r] = C[m = bm + 128 wm + 32 j + l31][n = bn + 128 wn + 32 i + 8 (r >> 2) + 4 half + (r & 3)].  The wave's 128 x 128
    // tile goes through its own 32 KiB of LDS ([row][16 chunks of 8 columns], chunk c of row m at slot c ^ (m & 15)) and leaves as 256-byte row pieces.
    uint4* so = &g2_smem[wave * 2048];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float4 b4[4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
            b4[g] = bias != nullptr ? *reinterpret_cast<const float4*>(bias + bn + wn * 128 + 32 * i + 8 * g + 4 * half) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = 32 * j + l31;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 w2;
                w2.x = LP::pack(acc[i][j][4 * g] + b4[g].x, acc[i][j][4 * g + 1] + b4[g].y);
                w2.y = LP::pack(acc[i][j][4 * g + 2] + b4[g].z, acc[i][j][4 * g + 3] + b4[g].w);
                *reinterpret_cast<uint2*>(reinterpret_cast<unsigned char*>(so) + m * 256 + (((4 * i + g) ^ (m & 15)) * 16) + 8 * half) = w2;
            }
        }
    }
    unsigned short* crow = C + (size_t)(bm + wm * 128) * ldc + bn + wn * 128 + 8 * (lane & 15);
#pragma unroll 8
    for (int k = 0; k < 32; ++k) {
        const int row = 4 * k + (lane >> 4);
        const uint4 v = so[row * 16 + ((lane & 15) ^ (row & 15))];
        *reinterpret_cast<uint4*>(crow + (size_t)row * ldc) = v;
    }
}

template <int DT>
int g2_launch(const void* A, int lda, const void* W, int ldw, const float* bias, void* C, int ldc, int M, int N, int K, hipStream_t stream) {
    static GvfPerDeviceOnce once;                                           // per instantiation and per device (gvf_common.h)
    if (!gvf_once_per_device(once, [] {
            return hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_kernel<DT>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * G2_STAGE * 16) ==
                   hipSuccess;
        }))
        return GVF_ELAUNCH;
    const int tiles_m = M / G2_T, tiles_n = N / G2_T;
    gemm256_kernel<DT><<<dim3((unsigned)(tiles_m * tiles_n)), dim3(G2_THREADS), 2 * G2_STAGE * 16, stream>>>(
        (const unsigned short*)A, lda, (const unsigned short*)W, ldw, bias, (unsigned short*)C, ldc, K, tiles_m, tiles_n);
    return GVF_OK;
}

}  // namespace

extern "C" int gvf_gemm256_eligible(int M, int N, int K, int lda, int ldw, int ldc) {
    return M > 0 && N > 0 && K > 0 && (M % G2_T) == 0 && (N % G2_T) == 0 && (K % G2_BK) == 0 && (lda % 8) == 0 && (ldw % 8) == 0 && (ldc % 8) == 0 &&
           lda >= K && ldw >= K && ldc >= N && (long long)(M / G2_T) * (N / G2_T) <= 0x7fffffffLL;
}

extern "C" int gvf_gemm256(int dtype, const void* A, int lda, const void* W, int ldw, const float* bias, void* C, int ldc, int M, int N, int K,
                           void* stream_) {
    if (dtype != GVF_DT_BF16 && dtype != GVF_DT_F16) return GVF_EINVAL;
    if (!gvf_gemm256_eligible(M, N, K, lda, ldw, ldc)) return GVF_EINVAL;
    if (!A || !W || !C) return GVF_EINVAL;
    if ((((uintptr_t)A) & 15) || (((uintptr_t)W) & 15) || (((uintptr_t)C) & 15) || (bias != nullptr && (((uintptr_t)bias) & 15))) return GVF_EINVAL;
    (void)hipGetLastError();
    int rc = GVF_OK;
    GVF_LP_DISPATCH(dtype, rc = g2_launch<DT>(A, lda, W, ldw, bias, C, ldc, M, N, K, (hipStream_t)stream_));
    if (rc != GVF_OK) return rc;
    GVF_CHECK_LAUNCH();
    return GVF_OK;
}
