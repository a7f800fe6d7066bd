// attn_xt64.hip -- cross attention against a PRE-TILED, LDS-RESIDENT K/V set; head_dim 64, <= 512 keys, bf16 / fp16 MFMA, gfx950 (MI355X).
//
// The decoder cross attention of the motion VAE (model/autoencoder.py:557-577 of the reference: every static Gaussian queries the 512
// latents of a frame; 24 x 262 144 queries x 512 keys x 12 heads of 64 = 9.9 TFLOP per decode, half of the decode) -- the head_dim-64 twin
// of attn_xt.hip.  What is the same: K pre-multiplied by softmax_scale * log2(e) before its one rounding, K / V^T stored in the image the
// MFMA fragments are read from (gvf_attn_pack_kv64), softmax without the running maximum behind a range guard with an exact fallback,
// row sums on the matrix pipe, per-query shift for fp16, the two-sub-tile software pipeline with the issue order written out.
// What differs:
//   * the whole key set of a (frame, head) fits in LDS (8 tiles x 16 KiB), so a workgroup copies it ONCE (linear LDS-DMA) and then walks
//     X64_PASSES x 256 queries over it with no barrier and no staging in the loop; the guard and its fallback are per WAVE;
//   * one wave per SIMD (4 waves, 256 threads, one workgroup per CU -- the key set fills the LDS anyway): at head_dim 64 two score tiles, two
//     output tiles, the query fragments and the operand fragments of a wave are ~280 registers, and a wave that interleaves its own MFMA and
//     VALU work is what overlaps the two pipes on this chip (scripts/ubench/mfma_valu_overlap.hip), a second wave adds little
//     (the compiler-scheduled kernel of attn.hip: 8 waves 2.32 ms, 4 waves 2.59 ms per launch);
//   * per phase 8 + 8 big MFMAs and 4 row-sum MFMAs (576 matrix cycles) against 32 v_exp_f32 + 16 v_cvt_pk (~450 vector cycles): the
//     matrix pipe is the longer one here, the order gives every MFMA a piece of vector work to cover.
// LDS images (per 64-key tile: 512 + 512 chunks of 16 bytes):
//   K   : chunk (key, c) at slot  key * 8 + (c ^ ((key >> 1) & 7))      -- c = 8 dims of the key row
//   V^T : row d, chunk j = 2 g + half at slot  d * 8 + (j ^ ((d >> 1) & 7)), its 8 key slots = keys 16 g + 4 half + (e & 3) + 8 (e >> 2)
// (a ds_read_b128 is served in lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} over 64 banks: with 128-byte rows and lane = row the
// (row >> 1) & 7 swizzle gives the 16 rows of a group 16 distinct 16-byte slots.)
#include <cstdlib>
#include <mutex>
#include "gvf_common.h"
#include "gvf_lp.h"
#include "../../include/gvf_rast.h"
#include "../../include/gvf_dit.h"

#ifndef X64_ORDER
#define X64_ORDER 0            // issue order of a phase (see x64_phase)
#endif
#ifndef X64_PASSES
#define X64_PASSES 8           // 256-query passes per workgroup = per copy of the key set into LDS
#endif

#ifdef X64_TIMING
static long long* g_x64_dbg;
#endif
namespace {

typedef gvf_f32x16 f32x16;
typedef gvf_f32x4 f32x4;

constexpr int X64_THREADS = 256;
constexpr int X64_KT = 64;
constexpr int X64_TILE = 1024;         // 16-byte chunks per staged tile: 512 K + 512 V^T
constexpr int X64_MAX_TILES = 8;

struct X64Params {
    const unsigned short* q;
    unsigned short* out;
    const uint4* kt;                   // [set][head][tile][512 chunks]
    const uint4* vt;
    int n_outer, n_inner, Lq, Lk, H, q_blocks, n_tiles;
    long long q_so, q_si, q_sl, q_sh, o_so, o_si, o_sl, o_sh;
    long long kv_so, kv_si;
    int* fallbacks;                    // optional: += 1 per WAVE PASS (64 queries) that took the exact path
    const uint4* fold_w;               // FOLD kernels: [head][4 fragments][64 lanes] of the (<= 16 x H*64) matrix applied to the attention output
    float* part;                       // FOLD kernels: [set = outer * n_inner + inner][head][Lq][16] f32 partial products (one per head)
#ifdef X64_TIMING
    long long* dbg;                    // timing builds (scripts/ubench/x64_bench.hip): s_memtime stamps of workgroup 0, wave 0
#endif
};
#ifdef X64_TIMING
// stamps of pass X64_TIMING (1 .. passes - 1) of workgroup 0 stay in registers and are written after the last pass: a store per stamp would sit in
// the very vmcnt queue whose waits are being measured
#define X64_STAMP(i_) do { if ((i_) >= 3 && ((i_) - 3) / 8 == X64_TIMING) x64_st[((i_) - 3) % 8] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define X64_STAMP(i_)
#endif

__device__ __forceinline__ void x64_dma16(const uint4* g, uint4* l) {
    __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
template <int DT>
__device__ __forceinline__ typename GvfLp<DT>::x8 x64_ld_k(const uint4* sK, int sub, int st, int l31, int half) {
    const int key = sub * 32 + l31;
    return __builtin_bit_cast(typename GvfLp<DT>::x8, sK[key * 8 + ((2 * st + half) ^ ((key >> 1) & 7))]);
}
template <int DT>
__device__ __forceinline__ typename GvfLp<DT>::x8 x64_ld_v(const uint4* sV, int g, int dt, int l31, int half) {
    const int d = dt * 32 + l31;
    return __builtin_bit_cast(typename GvfLp<DT>::x8, sV[d * 8 + ((2 * g + half) ^ ((d >> 1) & 7))]);
}

// ---------------------------------------------------------------------------------------------------------------
// One pipeline phase of a wave (attn_xt.hip's xt_phase at head_dim 64).
//   DO_QK: s_out[sub] = K'(32 keys of half `sub`) q^T for the 32 queries whose fragments are qf   (2 x 4 MFMAs), K fragments kf[sub][st]
//   DO_SM: P = exp2(s_in) (keys >= n_valid -> 0 when MASK), l += row sums, o[dt] += V^T P^T         (4 x 2 MFMAs), V^T fragments vf[g][dt]
// Fragments live in registers across the two phases that use them (sub-tile A, then B) and are refilled IN PLACE right behind the last MFMA that
// reads them: PF = 1 refills vf from sNext (the V^T image of the tile whose scores are computed in this phase), PF = 2 refills kf from sNext (the
// K image of the next tile).  Every group is fenced: source order = issue order.
// ---------------------------------------------------------------------------------------------------------------
template <int DT, bool DO_QK, bool DO_SM, bool MASK, int PF>
__device__ __forceinline__ void x64_phase(typename GvfLp<DT>::x8 (&kf)[2][4], typename GvfLp<DT>::x8 (&vf)[4][2], const uint4* __restrict__ sNext,
                                          const typename GvfLp<DT>::x8 (&qf)[4], f32x16 (&s_out)[2], const f32x16 (&s_in)[2], f32x16 (&o_acc)[2],
                                          f32x4& l4, int l31, int half, int n_valid, const f32x16& c0) {
    typedef GvfLp<DT> LP;
    typedef typename LP::x8 x8;
    float pe[8];
    unsigned pw[4][4];
    // row sums: the lane's 8 probabilities as the B operand of a 16x16x32 MFMA against a 0 / 1 selector (attn_xt.hip, XT_SUM_MFMA = 2)
    const unsigned selw = ((((unsigned)l31 >> 3) ^ ((unsigned)l31 >> 4)) & 1u) ? 0u : LP::ONE2;
    const x8 sel = __builtin_bit_cast(x8, make_uint4(selw, selw, selw, selw));
#define X_FENCE() __builtin_amdgcn_sched_barrier(0)
#define X_EH(g_, h_)                                                                                        \
    if (DO_SM) {                                                                                            \
        _Pragma("unroll") for (int e = 4 * (h_); e < 4 * (h_) + 4; ++e) {                                   \
            pe[e] = __builtin_amdgcn_exp2f(s_in[(g_) >> 1][8 * ((g_) & 1) + e]);                            \
            if (MASK) pe[e] = (16 * (g_) + 4 * half + (e & 3) + 8 * (e >> 2)) < n_valid ? pe[e] : 0.f;      \
        }                                                                                                   \
        X_FENCE();                                                                                          \
    }
#define X_CH(g_, h_)                                                                                        \
    if (DO_SM) {                                                                                            \
        pw[g_][2 * (h_)] = LP::pack(pe[4 * (h_)], pe[4 * (h_) + 1]);                                        \
        pw[g_][2 * (h_) + 1] = LP::pack(pe[4 * (h_) + 2], pe[4 * (h_) + 3]);                                \
        X_FENCE();                                                                                          \
    }
#define X_S(g_)                                                                                             \
    if (DO_SM) {                                                                                            \
        l4 = LP::mfma16(sel, __builtin_bit_cast(x8, make_uint4(pw[g_][0], pw[g_][1], pw[g_][2], pw[g_][3])), l4); \
        X_FENCE();                                                                                          \
    }
#define X_V(g_, dt_)                                                                                        \
    if (DO_SM) {                                                                                            \
        o_acc[dt_] = LP::mfma32(vf[g_][dt_], __builtin_bit_cast(x8, make_uint4(pw[g_][0], pw[g_][1], pw[g_][2], pw[g_][3])), o_acc[dt_]); \
        if (PF == 1) vf[g_][dt_] = x64_ld_v<DT>(sNext, g_, dt_, l31, half);                                 \
        X_FENCE();                                                                                          \
    }
#define X_Q(i_)                                                                                             \
    if (DO_QK) {                                                                                            \
        s_out[(i_) >> 2] = LP::mfma32(kf[(i_) >> 2][(i_) & 3], qf[(i_) & 3], ((i_) & 3) == 0 ? c0 : s_out[(i_) >> 2]); \
        if (PF == 2) kf[(i_) >> 2][(i_) & 3] = x64_ld_k<DT>(sNext, (i_) >> 2, (i_) & 3, l31, half);         \
        X_FENCE();                                                                                          \
    }
    X_FENCE();
#if X64_ORDER == 1
    // coarse (attn_xt's shape): a block of 8 exponentials, then MFMAs
    X_EH(0, 0) X_EH(0, 1) X_Q(0) X_Q(1) X_CH(0, 0) X_CH(0, 1) X_S(0) X_Q(2) X_Q(3)
    X_EH(1, 0) X_EH(1, 1) X_V(0, 0) X_V(0, 1) X_CH(1, 0) X_CH(1, 1) X_S(1) X_Q(4) X_Q(5)
    X_EH(2, 0) X_EH(2, 1) X_V(1, 0) X_V(1, 1) X_CH(2, 0) X_CH(2, 1) X_S(2) X_Q(6) X_Q(7)
    X_EH(3, 0) X_EH(3, 1) X_V(2, 0) X_V(2, 1) X_CH(3, 0) X_CH(3, 1) X_S(3) X_V(3, 0) X_V(3, 1)
#elif X64_ORDER == 2
    // the two score chains interleaved (no MFMA waits on the one before it), half a chunk of vector work behind every MFMA
    X_Q(0) X_EH(0, 0) X_Q(4) X_EH(0, 1) X_Q(1) X_CH(0, 0) X_CH(0, 1) X_S(0) X_Q(5) X_EH(1, 0)
    X_V(0, 0) X_EH(1, 1) X_V(0, 1) X_CH(1, 0) X_CH(1, 1) X_S(1) X_Q(2) X_EH(2, 0)
    X_V(1, 0) X_EH(2, 1) X_V(1, 1) X_CH(2, 0) X_CH(2, 1) X_S(2) X_Q(6) X_EH(3, 0)
    X_V(2, 0) X_EH(3, 1) X_V(2, 1) X_CH(3, 0) X_CH(3, 1) X_S(3) X_Q(3) X_V(3, 0) X_Q(7) X_V(3, 1)
#else
    // half a chunk of vector work (4 exponentials, or 4 conversions + the row-sum MFMA) behind every big MFMA
    X_Q(0) X_EH(0, 0) X_Q(1) X_EH(0, 1) X_Q(2) X_CH(0, 0) X_CH(0, 1) X_S(0) X_Q(3) X_EH(1, 0)
    X_V(0, 0) X_EH(1, 1) X_V(0, 1) X_CH(1, 0) X_CH(1, 1) X_S(1) X_Q(4) X_EH(2, 0)
    X_V(1, 0) X_EH(2, 1) X_V(1, 1) X_CH(2, 0) X_CH(2, 1) X_S(2) X_Q(5) X_EH(3, 0)
    X_V(2, 0) X_EH(3, 1) X_V(2, 1) X_CH(3, 0) X_CH(3, 1) X_S(3) X_Q(6) X_V(3, 0) X_Q(7) X_V(3, 1)
#endif
    if (PF == 1 && !DO_SM) {           // first phase of a pass: nothing to chase, load the V^T fragments now
#pragma unroll
        for (int g = 0; g < 4; ++g) { vf[g][0] = x64_ld_v<DT>(sNext, g, 0, l31, half); vf[g][1] = x64_ld_v<DT>(sNext, g, 1, l31, half); }
    }
#undef X_EH
#undef X_CH
#undef X_S
#undef X_V
#undef X_Q
#undef X_FENCE
}

// classic online softmax over one staged tile for ONE 32-query sub-tile (exact fallback; not pipelined)
template <int DT>
__device__ __forceinline__ void x64_safe_tile(const uint4* __restrict__ sK, const uint4* __restrict__ sV, const typename GvfLp<DT>::x8 (&qf)[4],
                                              f32x16 (&o_acc)[2], float& m_run, float& l_run, int l31, int half, int n_valid) {
    typedef GvfLp<DT> LP;
    typedef typename LP::x8 x8;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 s[2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
        s[sub] = zero;
#pragma unroll
        for (int st = 0; st < 4; ++st) s[sub] = LP::mfma32(x64_ld_k<DT>(sK, sub, st, l31, half), qf[st], s[sub]);
    }
    float mloc = -INFINITY;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int key = 16 * g + 4 * half + (e & 3) + 8 * (e >> 2);
            if (key >= n_valid) s[g >> 1][8 * (g & 1) + e] = -INFINITY;
            mloc = fmaxf(mloc, s[g >> 1][8 * (g & 1) + e]);
        }
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
    const float m_new = fmaxf(m_run, mloc);                    // finite: every tile holds >= 1 valid key
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new); // first tile: exp2(-inf) = 0
    l_run *= alpha;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o_acc[0][r] *= alpha; o_acc[1][r] *= alpha; }
    m_run = m_new;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        unsigned pw[4];
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            const float p0 = __builtin_amdgcn_exp2f(s[g >> 1][8 * (g & 1) + e] - m_run);
            const float p1 = __builtin_amdgcn_exp2f(s[g >> 1][8 * (g & 1) + e + 1] - m_run);
            l_run += p0 + p1;
            pw[e >> 1] = LP::pack(p0, p1);
        }
        const x8 pf = __builtin_bit_cast(x8, make_uint4(pw[0], pw[1], pw[2], pw[3]));
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) o_acc[dt] = LP::mfma32(x64_ld_v<DT>(sV, g, dt, l31, half), pf, o_acc[dt]);
    }
}

// fp16: the shift of a sub-tile's queries = the maximum of their scores against the first key tile (attn_xt.hip, xt_take_shift)
__device__ __forceinline__ void x64_take_shift(f32x16 (&s)[2], f32x16& c) {
    float m = fmaxf(s[0][0], s[1][0]);
#pragma unroll
    for (int r = 1; r < 16; ++r) m = fmaxf(fmaxf(m, s[0][r]), s[1][r]);
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    if (!(m > -3.0e38f && m < 3.0e38f)) m = 0.f;          // NaN / inf scores: no shift; the range guard sends the wave to the exact path
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[0][r] -= m; s[1][r] -= m; c[r] = -m; }
}

template <int DT, bool FOLD>
__global__ __launch_bounds__(X64_THREADS) void attn_xt64_kernel(X64Params p, int force_safe) {
    typedef GvfLp<DT> LP;
    typedef typename LP::x8 x8;
    constexpr bool SHIFT = LP::kNeedsShift;
    extern __shared__ __attribute__((aligned(16))) uint4 x64_smem[];           // [n_tiles][512 K chunks | 512 V^T chunks], then 8 KiB per wave (epilogue)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;

    int bid = (int)gvf_xcd_remap(blockIdx.x, gridDim.x);
    const int qb = bid % p.q_blocks; bid /= p.q_blocks;
    const int inner = bid % p.n_inner; bid /= p.n_inner;
    const int outer = bid % p.n_outer, head = bid / p.n_outer;

    const unsigned short* qp = p.q + outer * p.q_so + inner * p.q_si + head * p.q_sh;
    unsigned short* op = p.out + outer * p.o_so + inner * p.o_si + head * p.o_sh;
    const long long set = outer * p.kv_so + inner * p.kv_si;
    const uint4* kbase = p.kt + ((set * p.H + head) * p.n_tiles) * 512;
    const uint4* vbase = p.vt + ((set * p.H + head) * p.n_tiles) * 512;
    const int T = p.n_tiles;
    const int last_valid = p.Lk - (T - 1) * X64_KT;

    // ---- the first pass's query rows are requested first, then the whole key set (linear 1 KiB LDS-DMA pieces: no register, no ds_write)
    const int q_base = qb * (X64_PASSES * X64_THREADS) + wave * 64;
    uint4 qn[2][4];                       // raw rows of the NEXT pass (requested a pass ahead; rows past Lq read row 0 and are zeroed at use)
    auto load_q = [&](int pass) {
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int row = q_base + pass * X64_THREADS + a * 32 + l31;
#pragma unroll
            for (int st = 0; st < 4; ++st)
                qn[a][st] = *reinterpret_cast<const uint4*>(qp + (long long)(row < p.Lq ? row : 0) * p.q_sl + 16 * st + 8 * half);
        }
    };
#ifdef X64_TIMING
    long long x64_st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    load_q(0);
    for (int t = 0; t < T; ++t) {
        uint4* dst = &x64_smem[t * X64_TILE];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            x64_dma16(kbase + (long long)t * 512 + i * 256 + wave * 64 + lane, dst + i * 256 + wave * 64);
            x64_dma16(vbase + (long long)t * 512 + i * 256 + wave * 64 + lane, dst + 512 + i * 256 + wave * 64);
        }
    }
    // FOLD: the head's 64 columns of the output matrix as four A operands of the 32x32x16 shape (rows m < 16 of W^T, 8 values of the contraction
    // per lane in the order a lane's accumulator registers hold d: fragment (dt, j), lane (m, half), element e <-> d = 32 dt + 8 (2 j + (e >> 2)) + 4 half + (e & 3))
    x8 wf[4];
    if (FOLD) {
#pragma unroll
        for (int i = 0; i < 4; ++i) wf[i] = __builtin_bit_cast(x8, p.fold_w[(head * 4 + i) * 64 + lane]);
    }
    __syncthreads();
#define X_K(t_) (&x64_smem[(t_) * X64_TILE])
#define X_VT(t_) (&x64_smem[(t_) * X64_TILE + 512])

    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // the query fragments of a pass are taken out of qn BEFORE the previous pass's stores are issued, and the rows of the pass after it are
    // requested right then: the wait for them (a pass later) then covers stores that are a whole pass old.  Taken at the top of the pass
    // instead, that wait sits behind the stores just issued -- hipcc waits vmcnt(0) at the loop's back edge -- and every pass paid the
    // write latency (3.5 us per pass, a third of the kernel).
    x8 qf[2][4];
    auto take_q = [&](int pass) {
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const unsigned m = (q_base + pass * X64_THREADS + a * 32 + l31) < p.Lq ? 0xffffffffu : 0u;
#pragma unroll
            for (int st = 0; st < 4; ++st) qf[a][st] = __builtin_bit_cast(x8, make_uint4(qn[a][st].x & m, qn[a][st].y & m, qn[a][st].z & m, qn[a][st].w & m));
        }
#if !defined(X64_ABL_NOQ)
        if (pass + 1 < X64_PASSES && q_base + (pass + 1) * X64_THREADS < p.Lq) load_q(pass + 1);
#endif
    };
    take_q(0);
    for (int pass = 0; pass < X64_PASSES; ++pass) {
        const int row0 = q_base + pass * X64_THREADS;
        if (row0 >= p.Lq) break;                              // wave-uniform

        f32x16 oA[2] = {zero, zero}, oB[2] = {zero, zero};
        f32x4 l4A = {0.f, 0.f, 0.f, 0.f}, l4B = {0.f, 0.f, 0.f, 0.f};
        float lA = 0.f, lB = 0.f;
        bool bad = force_safe != 0;
        if (!bad) {
            f32x16 sA[2], sB[2];
            f32x16 cA = zero, cB = zero;
            x8 kf[2][4], vf[4][2];
#pragma unroll
            for (int i = 0; i < 8; ++i) kf[i >> 2][i & 3] = x64_ld_k<DT>(X_K(0), i >> 2, i & 3, l31, half);
            // tile t is consumed in iteration t: phase 1 = QK^T of sub-tile A on K(t) | softmax + PV of B on V(t-1) (refills vf with V(t)),
            // phase 2 = QK^T of B on K(t) | softmax + PV of A on V(t) (refills kf with K(t+1)); first and last tile peeled
            X64_STAMP(3 + 8 * pass);
            x64_phase<DT, true, false, false, 1>(kf, vf, X_VT(0), qf[0], sA, sB, oB, l4B, l31, half, X64_KT, cA);
            if (SHIFT) x64_take_shift(sA, cA);
            X64_STAMP(4 + 8 * pass);
            if (T > 1) {
                x64_phase<DT, true, true, false, 2>(kf, vf, X_K(1), qf[1], sB, sA, oA, l4A, l31, half, X64_KT, cB);
                if (SHIFT) x64_take_shift(sB, cB);
                for (int t = 1; t + 1 < T; ++t) {
                    x64_phase<DT, true, true, false, 1>(kf, vf, X_VT(t), qf[0], sA, sB, oB, l4B, l31, half, X64_KT, cA);
                    x64_phase<DT, true, true, false, 2>(kf, vf, X_K(t + 1), qf[1], sB, sA, oA, l4A, l31, half, X64_KT, cB);
                }
                x64_phase<DT, true, true, false, 1>(kf, vf, X_VT(T - 1), qf[0], sA, sB, oB, l4B, l31, half, X64_KT, cA);
            }
            X64_STAMP(5 + 8 * pass);
            x64_phase<DT, true, true, true, 0>(kf, vf, X_K(0), qf[1], sB, sA, oA, l4A, l31, half, last_valid, cB);
            if (SHIFT && T == 1) x64_take_shift(sB, cB);
            x64_phase<DT, false, true, true, 0>(kf, vf, X_K(0), qf[1], sA, sB, oB, l4B, l31, half, last_valid, cB);
            lA = __shfl(l4A[0], (l31 & 15) + 32 * (l31 >> 4), 64);      // the selector form holds query (l % 16) + 16 (l / 32) in lane l, both key halves
            lB = __shfl(l4B[0], (l31 & 15) + 32 * (l31 >> 4), 64);
            // range guard on the bit patterns (NaN-proof under -fno-honor-nans; a NaN is what the selector makes of an infinity)
            const float l_min = SHIFT ? 0.015625f : 7.8886e-31f, l_max = 1.2676e30f;
            const unsigned u_min = __float_as_uint(l_min), u_span = __float_as_uint(l_max) - __float_as_uint(l_min);
            const bool okA = (__float_as_uint(lA) - u_min - 1u) < (u_span - 1u), okB = (__float_as_uint(lB) - u_min - 1u) < (u_span - 1u);
            X64_STAMP(6 + 8 * pass);
            bad = __any(!(okA && okB)) != 0;
        }
        if (bad) {
            // exact path for this wave's 64 queries: classic online softmax over the resident tiles
            if (lane == 0 && p.fallbacks != nullptr) atomicAdd(p.fallbacks, 1);
            float mA = -INFINITY, mB = -INFINITY;
            oA[0] = zero; oA[1] = zero; oB[0] = zero; oB[1] = zero; lA = 0.f; lB = 0.f;
            for (int t = 0; t < T; ++t) {
                const int nv = t + 1 < T ? X64_KT : last_valid;
                x64_safe_tile<DT>(X_K(t), X_VT(t), qf[0], oA, mA, lA, l31, half, nv);
                x64_safe_tile<DT>(X_K(t), X_VT(t), qf[1], oB, mB, lB, l31, half, nv);
            }
            lA += __shfl_xor(lA, 32, 64);
            lB += __shfl_xor(lB, 32, 64);
        }
        X64_STAMP(7 + 8 * pass);
        // ---- epilogue: O[q][d] / l, d = 32 dt + (r & 3) + 8 (r >> 2) + 4 half.  A lane holds 4 consecutive d of ONE query per (dt, g): stored
        // from the accumulator layout that is 32 eight-byte pieces in 32 different cache lines per instruction.  The wave's 64 x 128-byte tile
        // goes through its own 8 KiB of LDS instead (chunk c of row q at slot c ^ (q & 7)) and leaves as 8 stores of 8 whole rows each.
        if (FOLD) {
            // y^T[m][q] = sum_d W[m][head * 64 + d] * bf16(o[q][d] / l): the output projection folded with to_out (<= 16 x 768) applied per head to the
            // rounded attention output while it is still in registers -- the packed pairs of a lane ARE a B operand (4 consecutive d per pair).
            // The 16 floats per (query, head) leave as two 16-byte stores per lane; gvf_attn_fold_reduce adds the heads and the bias.
            f32x16 yA = zero, yB = zero;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const float inv = 1.0f / (a == 0 ? lA : lB);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const f32x16& o = a == 0 ? oA[dt] : oB[dt];
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        unsigned w[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) w[u] = LP::pack(o[8 * j + 2 * u] * inv, o[8 * j + 2 * u + 1] * inv);
                        const x8 bfrag = __builtin_bit_cast(x8, make_uint4(w[0], w[1], w[2], w[3]));
                        if (a == 0) yA = LP::mfma32(wf[2 * dt + j], bfrag, yA); else yB = LP::mfma32(wf[2 * dt + j], bfrag, yB);
                    }
                }
            }
            if (pass + 1 < X64_PASSES && row0 + X64_THREADS < p.Lq) take_q(pass + 1);
            float* pbase = p.part + ((((long long)outer * p.n_inner + inner) * p.H + head) * p.Lq) * 16;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int qr = row0 + a * 32 + l31;
                const f32x16& y = a == 0 ? yA : yB;
                if (qr < p.Lq) {                 // rows m = 4 half + (r & 3) + 8 (r >> 2), r = 0 .. 7
                    float* dst = pbase + (long long)qr * 16 + 4 * half;
                    *reinterpret_cast<float4*>(dst) = make_float4(y[0], y[1], y[2], y[3]);
                    *reinterpret_cast<float4*>(dst + 8) = make_float4(y[4], y[5], y[6], y[7]);
                }
            }
        } else
        {
            uint4* so = &x64_smem[T * X64_TILE + wave * 512];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const float inv = 1.0f / (a == 0 ? lA : lB);
                const int q = a * 32 + l31;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const f32x16& o = a == 0 ? oA[dt] : oB[dt];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        uint2 w;
                        w.x = LP::pack(o[4 * g] * inv, o[4 * g + 1] * inv);
                        w.y = LP::pack(o[4 * g + 2] * inv, o[4 * g + 3] * inv);
                        *reinterpret_cast<uint2*>(reinterpret_cast<unsigned char*>(so) + q * 128 + (((4 * dt + g) ^ (q & 7)) * 16) + 8 * half) = w;
                    }
                }
            }
            uint4 ov[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int row = k * 8 + (lane >> 3), c = lane & 7;
                ov[k] = so[row * 8 + (c ^ (row & 7))];
            }
            // the next pass's fragments are taken (and the rows of the pass after it requested) between the tile's way through LDS and its
            // stores: the wait for the rows has the normalisation above in front of it and no store of this pass behind it
            X64_STAMP(8 + 8 * pass);
            if (pass + 1 < X64_PASSES && row0 + X64_THREADS < p.Lq) take_q(pass + 1);      // (qf is dead from here on in this pass)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int row = k * 8 + (lane >> 3), c = lane & 7;
#if !defined(X64_ABL_NOSTORE) && !defined(X64_PLAIN_STORE)
                // streaming stores: the 1.5 KiB rows of a 43 k-Gaussian chunk (1.6 GB per launch) are read once, by the GEMM behind this launch
                if (row0 + row < p.Lq) {
                    unsigned* dst = reinterpret_cast<unsigned*>(op + (long long)(row0 + row) * p.o_sl + 8 * c);
                    __builtin_nontemporal_store(ov[k].x, dst); __builtin_nontemporal_store(ov[k].y, dst + 1);
                    __builtin_nontemporal_store(ov[k].z, dst + 2); __builtin_nontemporal_store(ov[k].w, dst + 3);
                }
#elif !defined(X64_ABL_NOSTORE)
                if (row0 + row < p.Lq) *reinterpret_cast<uint4*>(op + (long long)(row0 + row) * p.o_sl + 8 * c) = ov[k];
#else
                if (row0 + row < p.Lq && ov[k].x == 0x12345678u) *reinterpret_cast<uint4*>(op + (long long)(row0 + row) * p.o_sl + 8 * c) = ov[k];
#endif
            }
        }
        X64_STAMP(9 + 8 * pass);
    }
#undef X_K
#undef X_VT
#ifdef X64_TIMING
    if (blockIdx.x == 0 && tid == 0)
        for (int i = 0; i < 8; ++i) p.dbg[i] = x64_st[i];
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// Image builder: kv rows (fp32 or 16-bit; K of head h at columns k_col0 + 64 h, V at v_col0 + 64 h of row set * L + key) -> tiled images.
// One workgroup = one (set, head, 64-key tile): K chunks go out directly (scale, one rounding), V is transposed through LDS.
// ---------------------------------------------------------------------------------------------------------------
template <int DT>
__device__ __forceinline__ void x64_ld8(const float* p, float (&v)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <int DT>
__device__ __forceinline__ void x64_ld8(const unsigned short* p, float (&v)[8]) {
    const uint4 r = *reinterpret_cast<const uint4*>(p);
    const unsigned w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = GvfLp<DT>::lo(w[i]); v[2 * i + 1] = GvfLp<DT>::hi(w[i]); }
}

constexpr int X64_PK_LD = 72;    // 16-bit pitch of the staged V rows (144 B)

template <typename TIn, int DT>
__global__ __launch_bounds__(256) void attn_pack_kv64_kernel(const TIn* __restrict__ kv, long long ld, int k_col0, int v_col0, int L, int H, int n_tiles,
                                                             float k_scale, uint4* __restrict__ kt, uint4* __restrict__ vt) {
    __shared__ unsigned short sV[X64_KT * X64_PK_LD];
    const int tid = threadIdx.x;
    long long rest = blockIdx.x;
    const int tile = (int)(rest % n_tiles); rest /= n_tiles;
    const int h = (int)(rest % H);
    const long long set = rest / H;
    const long long base = ((set * H + h) * n_tiles + tile) * 512;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + 256 * i, key_l = idx >> 3, c = idx & 7, key = tile * X64_KT + key_l;
        const bool valid = key < L;
        const TIn* row = kv + (set * L + (valid ? key : 0)) * ld + h * 64 + 8 * c;
        float k8[8], v8[8];
        x64_ld8<DT>(row + k_col0, k8);
        x64_ld8<DT>(row + v_col0, v8);
        unsigned kw[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) kw[j] = valid ? GvfLp<DT>::pack(k8[2 * j] * k_scale, k8[2 * j + 1] * k_scale) : 0u;
        kt[base + key_l * 8 + (c ^ ((key_l >> 1) & 7))] = make_uint4(kw[0], kw[1], kw[2], kw[3]);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *reinterpret_cast<unsigned*>(&sV[key_l * X64_PK_LD + 8 * c + 2 * j]) = valid ? GvfLp<DT>::pack(v8[2 * j], v8[2 * j + 1]) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + 256 * i, d = idx >> 3, pos = idx & 7, j = pos ^ ((d >> 1) & 7), g = j >> 1, hf = j & 1;
        unsigned vw[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e0 = 2 * u, e1 = 2 * u + 1;
            const unsigned lo = sV[(16 * g + 4 * hf + (e0 & 3) + 8 * (e0 >> 2)) * X64_PK_LD + d];
            const unsigned hi = sV[(16 * g + 4 * hf + (e1 & 3) + 8 * (e1 >> 2)) * X64_PK_LD + d];
            vw[u] = lo | (hi << 16);
        }
        vt[base + d * 8 + pos] = make_uint4(vw[0], vw[1], vw[2], vw[3]);
    }
}

}  // namespace

static bool x64_set_lds_limit();

extern "C" int gvf_attn_pack_kv64(int dtype, const void* kv, int kv_is_f32, int64_t ld, int k_col0, int v_col0, int n_sets, int L, int H,
                                  float k_scale, void* k_tiles, void* v_tiles, void* stream_) {
    if (dtype != GVF_DT_BF16 && dtype != GVF_DT_F16) return GVF_EINVAL;
    if (n_sets < 0 || L <= 0 || H <= 0 || ld <= 0 || k_col0 < 0 || v_col0 < 0) return GVF_EINVAL;
    if (n_sets == 0) return GVF_OK;
    if (!kv || !k_tiles || !v_tiles) return GVF_EINVAL;
    if ((((uintptr_t)k_tiles) & 15) || (((uintptr_t)v_tiles) & 15)) return GVF_EINVAL;
    const int n_tiles = (L + X64_KT - 1) / X64_KT;
    const long long blocks = (long long)n_sets * H * n_tiles;
    if (blocks > 0x7fffffffLL) return GVF_EINVAL;
    const int al = kv_is_f32 ? 4 : 8;
    if ((ld % al) || (k_col0 % al) || (v_col0 % al) || (((uintptr_t)kv) & 15)) return GVF_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    (void)x64_set_lds_limit();               // eager, outside any capture; the attention launch insists
    (void)hipGetLastError();
    GVF_LP_DISPATCH(dtype,
        if (kv_is_f32)
            attn_pack_kv64_kernel<float, DT><<<dim3((unsigned)blocks), dim3(256), 0, stream>>>((const float*)kv, ld, k_col0, v_col0, L, H, n_tiles, k_scale,
                                                                                               (uint4*)k_tiles, (uint4*)v_tiles);
        else
            attn_pack_kv64_kernel<unsigned short, DT><<<dim3((unsigned)blocks), dim3(256), 0, stream>>>((const unsigned short*)kv, ld, k_col0, v_col0, L, H,
                                                                                                        n_tiles, k_scale, (uint4*)k_tiles, (uint4*)v_tiles));
    GVF_CHECK_LAUNCH();
    return GVF_OK;
}

// The resident key set needs more dynamic LDS than the default limit.  All four instantiations are configured together, once per device
// (gvf_common.h: the attribute is per device and may not be set during a capture): eagerly from the two pack entry points every caller runs
// before its first attention launch -- gvf_attn_pack_kv64, gvf_attn_fold_pack -- and again, as a no-op, from the launch itself.
static bool x64_set_lds_limit() {
    static GvfPerDeviceOnce once;
    return gvf_once_per_device(once, [] {
        const int bytes = (X64_MAX_TILES * X64_TILE + 4 * 512) * 16;
        const void* fns[4] = {reinterpret_cast<const void*>(&attn_xt64_kernel<GVF_DT_BF16, false>), reinterpret_cast<const void*>(&attn_xt64_kernel<GVF_DT_BF16, true>),
                              reinterpret_cast<const void*>(&attn_xt64_kernel<GVF_DT_F16, false>), reinterpret_cast<const void*>(&attn_xt64_kernel<GVF_DT_F16, true>)};
        for (const void* f : fns)
            if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
        return true;
    });
}

template <int DT, bool FOLD>
static int x64_launch(const X64Params& p, int force_safe, unsigned blocks, size_t lds, hipStream_t stream) {
    if (!x64_set_lds_limit()) return GVF_ELAUNCH;
    attn_xt64_kernel<DT, FOLD><<<dim3(blocks), dim3(X64_THREADS), lds, stream>>>(p, force_safe);
    return GVF_OK;
}

static int x64_run(int dtype, const void* q, const void* k_tiles, const void* v_tiles, void* out, const void* fold_frags, float* part, int n_outer,
                   int n_inner, int Lq, int Lk, int H, const int64_t* q_strides, const int64_t* o_strides, int64_t kv_set_stride_outer,
                   int64_t kv_set_stride_inner, int force_exact, int32_t* fallback_counter, void* stream_) {
    const bool fold = fold_frags != nullptr;
    if (dtype != GVF_DT_BF16 && dtype != GVF_DT_F16) return GVF_EINVAL;
    if (n_outer < 0 || n_inner <= 0 || Lq < 0 || Lk <= 0 || H <= 0) return GVF_EINVAL;
    if (Lk > X64_MAX_TILES * X64_KT) return GVF_EINVAL;               // the key set must fit in LDS
    if (n_outer == 0 || Lq == 0) return GVF_OK;
    if (!q || !k_tiles || !v_tiles || !q_strides || (fold ? !part : (!out || !o_strides))) return GVF_EINVAL;
    for (int i = 0; i < 4; ++i)
        if ((q_strides[i] % 8) || (!fold && (o_strides[i] % 8))) return GVF_EINVAL;
    if ((((uintptr_t)q) & 15) || (((uintptr_t)k_tiles) & 15) || (((uintptr_t)v_tiles) & 15)) return GVF_EINVAL;
    if (fold ? ((((uintptr_t)fold_frags) & 15) || (((uintptr_t)part) & 15)) : ((((uintptr_t)out) & 15) != 0)) return GVF_EINVAL;
    X64Params p;
    p.q = (const unsigned short*)q; p.out = (unsigned short*)out;
    p.kt = (const uint4*)k_tiles; p.vt = (const uint4*)v_tiles;
    p.n_outer = n_outer; p.n_inner = n_inner; p.Lq = Lq; p.Lk = Lk; p.H = H;
    p.q_blocks = (Lq + X64_PASSES * X64_THREADS - 1) / (X64_PASSES * X64_THREADS);
    p.n_tiles = (Lk + X64_KT - 1) / X64_KT;
    p.q_so = q_strides[0]; p.q_si = q_strides[1]; p.q_sl = q_strides[2]; p.q_sh = q_strides[3];
    p.o_so = p.o_si = p.o_sl = p.o_sh = 0;
    if (!fold) { p.o_so = o_strides[0]; p.o_si = o_strides[1]; p.o_sl = o_strides[2]; p.o_sh = o_strides[3]; }
    p.kv_so = kv_set_stride_outer; p.kv_si = kv_set_stride_inner;
    p.fallbacks = fallback_counter;
    p.fold_w = (const uint4*)fold_frags; p.part = part;
#ifdef X64_TIMING
    p.dbg = g_x64_dbg;
#endif
    const long long blocks = (long long)p.q_blocks * H * n_inner * n_outer;
    if (blocks > 0x7fffffffLL) return GVF_EINVAL;
    (void)hipGetLastError();
    int rc = GVF_OK;
    const size_t lds = ((size_t)p.n_tiles * X64_TILE + 4 * 512) * 16;
    const int fs = force_exact & GVF_ATTN_FORCE_EXACT;
    GVF_LP_DISPATCH(dtype, rc = fold ? x64_launch<DT, true>(p, fs, (unsigned)blocks, lds, (hipStream_t)stream_)
                                     : x64_launch<DT, false>(p, fs, (unsigned)blocks, lds, (hipStream_t)stream_));
    if (rc != GVF_OK) return rc;
    GVF_CHECK_LAUNCH();
    return GVF_OK;
}

extern "C" int gvf_attn_tiled64_fwd(int dtype, const void* q, const void* k_tiles, const void* v_tiles, void* out, int n_outer, int n_inner,
                                    int Lq, int Lk, int H, const int64_t* q_strides, const int64_t* o_strides,
                                    int64_t kv_set_stride_outer, int64_t kv_set_stride_inner, int force_exact, int32_t* fallback_counter,
                                    void* stream_) {
    return x64_run(dtype, q, k_tiles, v_tiles, out, nullptr, nullptr, n_outer, n_inner, Lq, Lk, H, q_strides, o_strides, kv_set_stride_outer,
                   kv_set_stride_inner, force_exact, fallback_counter, stream_);
}

extern "C" int gvf_attn_tiled64_fold_fwd(int dtype, const void* q, const void* k_tiles, const void* v_tiles, const void* fold_frags, float* part,
                                         int n_outer, int n_inner, int Lq, int Lk, int H, const int64_t* q_strides,
                                         int64_t kv_set_stride_outer, int64_t kv_set_stride_inner, int force_exact, int32_t* fallback_counter,
                                         void* stream_) {
    if (!fold_frags) return GVF_EINVAL;
    return x64_run(dtype, q, k_tiles, v_tiles, nullptr, fold_frags, part, n_outer, n_inner, Lq, Lk, H, q_strides, nullptr, kv_set_stride_outer,
                   kv_set_stride_inner, force_exact, fallback_counter, stream_);
}

// W (n_out <= 16 rows x H * 64 columns, row-major 16-bit with leading dimension ld) -> the fragment image attn_xt64_kernel<.., FOLD> reads
namespace {
__global__ void x64_pack_fold_kernel(const unsigned short* __restrict__ w, int ld, int n_out, int H, uint4* __restrict__ frags) {
    const int lane = threadIdx.x & 63, i = threadIdx.x >> 6, head = blockIdx.x;       // 256 threads: fragment i = 2 dt + j
    const int m = lane & 31, half = lane >> 5, dt = i >> 1, j = i & 1;
    unsigned short v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int d = 32 * dt + 8 * (2 * j + (e >> 2)) + 4 * half + (e & 3);
        v[e] = m < n_out ? w[(long long)m * ld + head * 64 + d] : (unsigned short)0;
    }
    frags[(head * 4 + i) * 64 + lane] = make_uint4(v[0] | ((unsigned)v[1] << 16), v[2] | ((unsigned)v[3] << 16), v[4] | ((unsigned)v[5] << 16), v[6] | ((unsigned)v[7] << 16));
}

// out[set][q][0 .. n_out) = bias + sum over heads of part[set][head][q][0 .. n_out)   (fixed order: deterministic)
__global__ __launch_bounds__(256) void x64_fold_reduce_kernel(const float* __restrict__ part, const float* __restrict__ bias, float* __restrict__ out,
                                                              int H, int Lq, int n_out, long long out_set_stride, long long out_row_stride) {
    const int q = blockIdx.x * 256 + threadIdx.x, set = blockIdx.y;
    if (q >= Lq) return;
    float acc[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) acc[m] = (bias != nullptr && m < n_out) ? bias[m] : 0.f;
    const float4* src = reinterpret_cast<const float4*>(part + (((long long)set * H) * Lq + q) * 16);
    for (int h = 0; h < H; ++h) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float4 v = src[(long long)h * Lq * 4 + u];
            acc[4 * u] += v.x; acc[4 * u + 1] += v.y; acc[4 * u + 2] += v.z; acc[4 * u + 3] += v.w;
        }
    }
    float* dst = out + set * out_set_stride + q * out_row_stride;
#pragma unroll
    for (int m = 0; m < 16; ++m)
        if (m < n_out) dst[m] = acc[m];
}
}  // namespace

extern "C" int gvf_attn_fold_pack(int dtype, const void* w, int ld, int n_out, int H, void* fold_frags, void* stream_) {
    if (dtype != GVF_DT_BF16 && dtype != GVF_DT_F16) return GVF_EINVAL;
    if (!w || !fold_frags || n_out <= 0 || n_out > 16 || H <= 0 || ld < H * 64 || (((uintptr_t)fold_frags) & 15)) return GVF_EINVAL;
    (void)x64_set_lds_limit();               // eager, outside any capture (see x64_set_lds_limit)
    (void)hipGetLastError();
    x64_pack_fold_kernel<<<dim3((unsigned)H), dim3(256), 0, (hipStream_t)stream_>>>((const unsigned short*)w, ld, n_out, H, (uint4*)fold_frags);
    GVF_CHECK_LAUNCH();
    return GVF_OK;
}

extern "C" int gvf_attn_fold_reduce(const float* part, const float* bias, float* out, int n_sets, int H, int Lq, int n_out,
                                    int64_t out_set_stride, int64_t out_row_stride, void* stream_) {
    if (n_sets < 0 || H <= 0 || Lq < 0 || n_out <= 0 || n_out > 16) return GVF_EINVAL;
    if (n_sets == 0 || Lq == 0) return GVF_OK;
    if (!part || !out || (((uintptr_t)part) & 15) || n_sets > 65535) return GVF_EINVAL;
    (void)hipGetLastError();
    x64_fold_reduce_kernel<<<dim3((unsigned)((Lq + 255) / 256), (unsigned)n_sets), dim3(256), 0, (hipStream_t)stream_>>>(part, bias, out, H, Lq, n_out,
                                                                                                                      out_set_stride, out_row_stride);
    GVF_CHECK_LAUNCH();
    return GVF_OK;
}
