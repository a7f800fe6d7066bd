// rowblock.hip -- one launch per sub-layer boundary of the DiT block, for gfx950 (MI355X), model_channels = 512:
//
//     x   <- x + gate * (A W1^T + b1)                      (the projection that closes a sub-layer: to_out / mlp.2 / input_layer)
//     hb  <- bf16( LayerNorm(x) * mul + add )              (the LayerNorm that opens the next one: adaLN modulate or affine)
//     out <- epi( hb W2^T + b2 )                           (its first projection: to_qkv / to_q / mlp.0 [+GELU])
//
// i.e. model/dit.py:236-277 of the reference (x = x + gate * attn(...); h = norm(x); h = h * (1 + scale) + shift; attn.to_q(h) ...)
// cut at the attention calls instead of at the nn.Module boundaries.  Unfused this is three launches (gemm.hip RESID epilogue,
// elem.hip ln_mod, gemm.hip) that move the fp32 stream x through HBM three times and the normalised copy twice.
//
// Shape of the work: M = B*T*N rows (12288 for one sample) by exactly 512 columns.  A workgroup owns 48 FULL rows -- 256
// workgroups for one sample, one per CU -- so that the row statistics of LayerNorm are available in the epilogue of the
// first GEMM, and the normalised rows never leave the CU: they are written to LDS already in MFMA-fragment order and are
// the activation operand of the second GEMM.
//
// Data flow per workgroup (4 waves; wave w owns output columns [128w, 128w + 128) of every 512-column pass):
//   * weights: pre-packed once (gvf_rowblock_pack_weight) in fragment order [pass][k-step][wave][column tile][lane][8 bf16], so a
//     wave's operand for one k-step (32 deep) is 8 fully coalesced 1 KiB loads, global -> VGPR, no LDS: every wave reads columns
//     nobody else in the workgroup needs.  Two k-steps are kept in flight (refilled in place behind the MFMAs that consume a
//     fragment); the stream runs on across the phase-1 / phase-2 boundary and across the passes of phase 2.  With 48 rows per
//     CU this stream (0.5 MiB of L2 reads per 512x512 weight matrix per CU) is what bounds the kernel: ~56 B/clk/CU of L2
//     bandwidth against 16 cycles per MFMA.
//   * activations of phase 1: 48 x 512 bf16 per chunk, LDS-DMA (global_load_lds_dwordx4) straight into fragment order
//     [k-step][row tile][lane][16 B] -- the gather is on the source side, the LDS side is linear -- two chunk buffers for K > 512.
//   * MFMA v_mfma_f32_16x16x32_bf16 with the WEIGHT fragment as the A operand: D[n][m], so a lane ends up with 4 consecutive
//     columns of one row: 16-byte accesses to the fp32 stream, 8-byte bf16 stores, and the LayerNorm row reduction is
//     4 lanes + 4 waves wide.
//   * the residual tile of x (96 VGPRs) is fetched before the k-loop; the per-column vectors (bias, gate, LayerNorm gain / shift)
//     sit in LDS.
#include <cstddef>
#include <cstdlib>
#include "gvf_common.h"
#include "gvf_lp.h"
#include "../../include/gvf_rast.h"
#include "../../include/gvf_dit.h"

namespace {

typedef gvf_f32x4 f32x4;

#ifndef RB_DEPTH
#define RB_DEPTH 4                               // k-steps of weight fragments in flight per wave (16 VGPRs each)
#endif
#ifndef RB_DEPTH_MLP
#define RB_DEPTH_MLP 4
#endif
constexpr int RB_C = 512;                        // columns of the stream (model_channels)
constexpr int RB_BM = 48;                        // rows per workgroup: 3 MFMA row tiles
constexpr int RB_THREADS = 512;                  // 8 waves, two per SIMD: wave w owns output columns [64w, 64w + 64) of every 512-column pass
constexpr int RB_NW = 8, RB_CT = 4;              // waves, 16-column tiles per wave
constexpr int RB_KS = 16;                        // k-steps (32 deep) of a 512-deep GEMM
constexpr int RB_BUF = RB_KS * 3 * 64;           // uint4 per 48 x 512 bf16 activation block in fragment order (48 KiB)
constexpr int RB_MAX_HIDDEN = 2048;
constexpr int RB_PAR = 2 * RB_BUF;               // floats: [b1 | gate1 | mul1 | add1 | b_fc2 | gate_m | mul2 | add2] x 512, b_fc1 x 2048, b3 x 1536, gamma_k x 512
constexpr int RB_MAX_N3 = 1536;
constexpr int RB_PAR_FLOATS = 8 * RB_C + RB_MAX_HIDDEN + RB_MAX_N3 + RB_C;     // ... and gamma_k x 512
constexpr int RB_RED = RB_PAR + RB_PAR_FLOATS / 4;       // [2][8 waves][48 rows] floats
constexpr int RB_SMEM = RB_RED + 2 * RB_NW * RB_BM / 4;
constexpr int RB_KPAD = 128;                      // K of a packed weight is padded to this (k-steps come in groups of RB_DEPTH)
constexpr int RB_STEP = 4 * 8 * 64;              // uint4 per k-step of a packed weight stream (32 KiB)

struct RbLn {                                    // LayerNorm(x) * (1 + scale) + shift and / or affine
    const float* ln_w; const float* ln_b;        // [512] or null
    const float* shift; const float* scale;      // row g of leading dimension mod_ld, or null
};

struct RbParams {
    const unsigned short* A; int lda, K1;        // phase-1 activations, bf16 [M][lda], K1 <= 512
    const uint4* W; const float* b1;             // the packed weight stream: W1 [K1/32 steps] | MLP [2 * 16 * hidden/512] | W3 [16 * N3/512]; bias [512] or null
    float* x; int M;                             // fp32 stream [M][512], updated in place
    const float* x_in; int x_in_period;          // optional: the residual is read from x_in[(row / rpg) * period + (row % rpg) % period] instead
    const float* in_x; const float* in_wt; const float* in_b; int in_cin;     // optional (K1 == 0): + in_x[row] W_in^T + b in fp32 (input_layer)
    const float* gate1;                          // row g of leading dimension mod_ld, or null (-> 1)
    RbLn ln1;
    int mod_ld, rpg; float eps;
    // optional MLP section: x += gate_m * (gelu(hb Wfc1^T + b_fc1) Wfc2^T + b_fc2), then LayerNorm ln2
    const float* b_fc1; const float* b_fc2; int hidden;      // stream: per 512 hidden units, 16 steps of mlp.0 then 16 of mlp.2
    const float* gate_m;
    RbLn ln2;
    // last projection, fed by the last LayerNorm
    const float* b3;                             // bias [N3] or null
    unsigned short* out3; int N3;          // bf16 [M][N3], or null: no last projection
    unsigned short* hb_out;                      // optional: the rows of the last LayerNorm, bf16 [M][512]
    // optional (N3 = 1536 = to_qkv of the spatial self attention): pass 0 (q) goes to out3 as [M][512]; pass 1 (k) and pass 2 (v) go
    // straight into the tiled K / V^T images of csrc/attn_xt.hip (what gvf_attn_pack_kv_bf16 would build from the row-major copy)
    uint4* kt; uint4* vt; int kv_L, kv_tiles; float k_scale; const float* gamma_k;
    int kv_group_rows;                  // > 0: only the first kv_group_rows rows of every rows_per_group group hold keys (padded groups)
    // temporal section (TEMPORAL kernels): between ln1 and the last projection the launch also runs
    //     [q | k | v] = hb Wqkv^T + b;  o = softmax over a token's t_T frames (per head of 32);  x += t_gate * (o Wout^T + t_bout);  hb = LN(x) * t_ln
    // on blocks of 48 / t_T TOKENS x t_T frames: local row r = tok * t_T + frame  <->  stream row  group * rpg + frame * t_N + (block * 48 / t_T + tok)
    int t_T, t_N;
    const float* t_bqkv; const float* t_gq; const float* t_gk; float t_kscale;
    const float* t_bout; const float* t_gate; RbLn t_ln;
    long long* dbg;                              // RB_TIMING builds only: [workgroup][16] s_memtime stamps
};

// GELU (tanh form) = x * sigmoid(2u), u = sqrt(2/pi) (x + 0.044715 x^3), as gemm.hip's epilogue -- with the hardware reciprocal (1 ulp)
// in place of the IEEE division (~10 instructions per element; 48 elements per lane per 512 hidden units, on the critical path
// between the two GEMMs of a slice)
__device__ __forceinline__ float rb_gelu_tanh(float x) {
    const float c0 = -2.0f * 0.7978845608028654f * 1.4426950408889634f, c1 = 0.044715f;
    const float x2 = x * x;
    const float e = __builtin_amdgcn_exp2f(c0 * (x + c1 * x2 * x));
    return x * __builtin_amdgcn_rcpf(1.0f + e);
}

__device__ __forceinline__ void rb_dma16(const unsigned short* g, uint4* l) {
    __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

template <int DT>
__device__ __forceinline__ typename GvfLp<DT>::x8 rb_ldw(const uint4* p) { return __builtin_bit_cast(typename GvfLp<DT>::x8, *p); }

// The weight stream of one wave: the k-steps of W1, then of the MLP section, then of W3, packed back to back (32 KiB per step).
// Past the end it repeats its last step (the refills behind the last MFMAs are redundant, never out of bounds).
struct RbStream {
    const uint4* w; int Gt;
#if defined(RB_ABL_WSMALL)        // timing experiment: every step reads the same 32 KiB (L1 / L2 hits only)
    __device__ __forceinline__ const uint4* at(int s) const { return w + (long long)(s & 1) * RB_STEP; }
#else
    __device__ __forceinline__ const uint4* at(int s) const { return w + (long long)(s < Gt ? s : Gt - 1) * RB_STEP; }
#endif
};

// `steps` k-steps (a multiple of D) of acc += act * W: 3 activation fragments per step from LDS (act = block base + lane), 24 MFMAs,
// the weight fragments refilled in place D steps ahead.  g = index of the first step in the stream.  D: the stream is new to the L2 in
// every launch (the DiT's weights are 100+ MB), so a refill is a miss to the Infinity Cache / HBM for the first workgroup of an XCD that
// asks and a wait on that miss for the others: ~1 us, i.e. 4+ k-steps of MFMA work.
// SWAP: the activation fragment is the A operand -> D[m][n]: a lane ends up with 4 consecutive ROWS of one column (the V^T layout of the
// temporal section); same fragments, same stream.
template <int D, int DT, bool SWAP = false>
__device__ __forceinline__ void rb_gemm(f32x4 (&acc)[3][RB_CT], typename GvfLp<DT>::x8 (&wf)[D][RB_CT], const uint4* act, int steps, int& g, const RbStream& st) {
    typedef typename GvfLp<DT>::x8 x8;
    static_assert(D % 2 == 0, "the activation fragments are double-buffered by step parity");
    // the activation fragments of step s + 1 are read from LDS while the MFMAs of step s run (the read past the last step stays
    // inside the LDS block and is never used)
    x8 af[2][3];
#pragma unroll
    for (int rt = 0; rt < 3; ++rt) af[0][rt] = __builtin_bit_cast(x8, act[rt * 64]);
    // (not unrolled: in straight-line code the scheduler sinks every refill to just before its use and the prefetch is gone)
#pragma clang loop unroll(disable)
    for (int ksl = 0; ksl < steps; ksl += D, g += D) {
#pragma unroll
        for (int b = 0; b < D; ++b) {
#pragma unroll
            for (int rt = 0; rt < 3; ++rt) {
#ifdef RB_ABL_NOLDS                // timing experiment: activation fragments as opaque register values
                asm volatile("" : "+v"(af[(b + 1) & 1][rt]));
#else
                af[(b + 1) & 1][rt] = __builtin_bit_cast(x8, act[((ksl + b + 1) * 3 + rt) * 64]);
#endif
            }
            __builtin_amdgcn_sched_barrier(0);   // ... issued BEFORE this step's MFMAs (the scheduler would sink them behind)
            const uint4* sn = st.at(g + b + D);
#pragma unroll
            for (int ct = 0; ct < RB_CT; ++ct) {
#pragma unroll
                for (int rt = 0; rt < 3; ++rt)
                    acc[rt][ct] = SWAP ? GvfLp<DT>::mfma16(af[b & 1][rt], wf[b][ct], acc[rt][ct]) : GvfLp<DT>::mfma16(wf[b][ct], af[b & 1][rt], acc[rt][ct]);
#if defined(RB_ABL_WFRAC)          // timing experiment: only RB_ABL_WFRAC of the 4 column tiles are refilled (what a column split across CUs would leave of the fill traffic)
                if (ct < RB_ABL_WFRAC) wf[b][ct] = rb_ldw<DT>(sn + ct * 64);
#elif !defined(RB_ABL_NOW)         // timing experiment: no weight refills at all
                wf[b][ct] = rb_ldw<DT>(sn + ct * 64);
#endif
            }
            __builtin_amdgcn_sched_barrier(0);   // the refills stay in the step that frees their registers
        }
    }
}

// Workgroup barrier that orders LDS traffic only: s_waitcnt lgkmcnt(0) + s_barrier.  __syncthreads() also drains vmcnt, i.e. waits for
// the whole weight prefetch in flight (a memory latency per barrier, ~20 barriers per launch).
__device__ __forceinline__ void rb_lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__device__ __forceinline__ void rb_zero(f32x4 (&acc)[3][RB_CT]) {
#pragma unroll
    for (int rt = 0; rt < 3; ++rt)
#pragma unroll
        for (int ct = 0; ct < RB_CT; ++ct) acc[rt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
}

// Column n = 64 wave + 16 ct + 4 lq + i of a 48 x 512 block is k index n of the GEMM that consumes it: k-step 2 wave + ct / 2,
// fragment lane (2 (ct & 1) + lq / 2) * 16 + l15, elements 4 (lq & 1) .. + 4 -> one 8-byte LDS write per (row tile, column tile).
__device__ __forceinline__ uint2* rb_frag_base(uint4* block, int wave, int lq, int l15) {      // the lane's part of the address
    return reinterpret_cast<uint2*>(block) + (wave * 6 * 64 + (lq >> 1) * 16 + l15) * 2 + (lq & 1);
}
__device__ __forceinline__ void rb_put_frag(uint2* base, int ct, int rt, uint2 o) {                // + a compile-time offset
    base[(((ct >> 1) * 3 + rt) * 64 + 2 * (ct & 1) * 16) * 2] = o;
}

// Row reductions over the 4 lanes (l15, lq = 0..3) that share a row, without index registers: lane ^ 16 by ds_swizzle (bit mode, inside each
// 32-lane half), lane ^ 32 by v_permlane32_swap (both results of the swap are the two halves' values: their sum / maximum IS the reduction).
// __shfl_xor keeps its bounds-checked ds_bpermute index alive across the whole kernel (one of them was the MLP variant's last spilled register).
__device__ __forceinline__ float rb_x16(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));
}
// (Pitfall, hipcc 7.2: __builtin_bit_cast(float, r[1]) on an ELEMENT of the builtin's vector result reads element 0 -- both "halves" came out
// as r[0], i.e. the other half of the wave silently dropped out of every sum.  The elements are copied into scalars first.)
__device__ __forceinline__ float rb_rowsum4(float v) {
    v += rb_x16(v);
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    const unsigned r0 = r[0], r1 = r[1];
    return __uint_as_float(r0) + __uint_as_float(r1);
}
__device__ __forceinline__ float rb_rowmax4(float v) {
    v = fmaxf(v, rb_x16(v));
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    const unsigned r0 = r[0], r1 = r[1];
    return fmaxf(__uint_as_float(r0), __uint_as_float(r1));
}

// LayerNorm of the 48 rows held in v (row 16 rt + l15, columns colw + 16 ct + i), times mul plus add, as bf16 fragments into
// `block` (and to hb_out).  v is left centred.  Two barriers; the caller adds the one that publishes `block`.
template <int DT>
__device__ __forceinline__ void rb_layernorm(f32x4 (&v)[3][RB_CT], float* sRed, const float* mul, const float* add, float eps, uint4* block,
                                             unsigned short* hb_out_row0, int wave, int lane, int lq, int l15, int colw) {
    float rsum[3];
#pragma unroll
    for (int rt = 0; rt < 3; ++rt) {
        float s = 0.f;
#pragma unroll
        for (int ct = 0; ct < RB_CT; ++ct) s += (v[rt][ct][0] + v[rt][ct][1]) + (v[rt][ct][2] + v[rt][ct][3]);
        s = rb_rowsum4(s);
        rsum[rt] = s;
        if (lane < 16) sRed[wave * RB_BM + 16 * rt + lane] = s;
    }
    rb_lds_barrier();                             // also: every wave is past the GEMM that read `block`'s previous contents
    float rstd[3];
#pragma unroll
    for (int rt = 0; rt < 3; ++rt) {
        const int r = 16 * rt + l15;
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < RB_NW; ++w) tot += sRed[w * RB_BM + r];
        const float mean = tot * (1.0f / RB_C);
        float q = 0.f;
#pragma unroll
        for (int ct = 0; ct < RB_CT; ++ct)
#pragma unroll
            for (int i = 0; i < 4; ++i) { const float d = v[rt][ct][i] - mean; v[rt][ct][i] = d; q += d * d; }
        q = rb_rowsum4(q);
        if (lane < 16) sRed[RB_NW * RB_BM + wave * RB_BM + 16 * rt + lane] = q;
    }
    rb_lds_barrier();
#pragma unroll
    for (int rt = 0; rt < 3; ++rt) {
        const int r = RB_NW * RB_BM + 16 * rt + l15;
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < RB_NW; ++w) tot += sRed[w * RB_BM + r];
        rstd[rt] = rsqrtf(tot * (1.0f / RB_C) + eps);
    }
    uint2* fb = rb_frag_base(block, wave, lq, l15);
    unsigned short* hbr[3];
#pragma unroll
    for (int rt = 0; rt < 3; ++rt) hbr[rt] = hb_out_row0 + (16 * rt + l15) * RB_C + colw;
#pragma unroll
    for (int ct = 0; ct < RB_CT; ++ct) {
        const f32x4 mu = *reinterpret_cast<const f32x4*>(mul + colw + 16 * ct);
        const f32x4 ad = *reinterpret_cast<const f32x4*>(add + colw + 16 * ct);
#pragma unroll
        for (int rt = 0; rt < 3; ++rt) {
            float y[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) y[i] = (v[rt][ct][i] * rstd[rt]) * mu[i] + ad[i];
            uint2 o;
            o.x = GvfLp<DT>::pack(y[0], y[1]);
            o.y = GvfLp<DT>::pack(y[2], y[3]);
            rb_put_frag(fb, ct, rt, o);
            if (hb_out_row0 != nullptr) *reinterpret_cast<uint2*>(hbr[rt] + 16 * ct) = o;
        }
    }
}

template <bool MLP, int D, int DT, bool TEMPORAL = false>
__global__ __launch_bounds__(RB_THREADS, 1) void rowblock_kernel(RbParams p) {
    typedef GvfLp<DT> LP;
    typedef typename LP::x8 x8;
    static_assert(!(MLP && TEMPORAL), "the temporal section belongs to the launch between the spatial and the image attention");
    __shared__ uint4 smem[RB_SMEM];              // the ONE LDS object (a second one makes hipcc drain vmcnt before every ds_read)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, lq = lane >> 4;
    const int m0 = blockIdx.x * RB_BM;           // M % 48 == 0: no row guards anywhere
#ifdef RB_TIMING
    int stamp_i = 0;
#define RB_STAMP() do { if (tid == 0 && p.dbg) p.dbg[blockIdx.x * 16 + stamp_i] = (long long)__builtin_amdgcn_s_memrealtime(); ++stamp_i; } while (0)
#else
#define RB_STAMP() do { } while (0)
#endif
    RB_STAMP();
    // the launcher refuses these options for a TEMPORAL launch: compile their code (and the registers it holds) out of that variant
    // (x_in / in_x belong to the first launch of a forward, which has neither an MLP nor a temporal section)
    const bool has_x_in = !TEMPORAL && !MLP && p.x_in != nullptr, has_in_x = !TEMPORAL && !MLP && p.in_x != nullptr, has_kt = !TEMPORAL && p.kt != nullptr;
    uint4* R0 = &smem[0];                        // normalised rows (operand of mlp.0 / of the last projection)
    uint4* R1 = &smem[RB_BUF];                   // phase-1 activations, then the GELU'd hidden units of one 512-wide slice
    float* sPar = reinterpret_cast<float*>(&smem[RB_PAR]);
    float* sRed = reinterpret_cast<float*>(&smem[RB_RED]);

    const int G1 = p.K1 >> 5;
    const int Pm = MLP ? p.hidden / RB_C : 0;
    const int P3 = p.out3 != nullptr ? p.N3 / RB_C : 0;
    RbStream st;
    st.w = p.W + wave * (RB_CT * 64) + lane;
    st.Gt = G1 + 2 * RB_KS * Pm + (TEMPORAL ? 4 * RB_KS : 0) + RB_KS * P3;
    // local row r of this block <-> row of the stream.  TEMPORAL: the block owns 48 / T tokens x all T frames of its group (sample), local
    // row = token * T + frame, so that a token's frames are neighbours (the keys of its attention) while the stream stays frame-major
    // A group whose token count is not a multiple of 48 / T ends with phantom tokens: their T rows each are the group's padding rows
    // (behind its T * t_N token rows), computed like any other row and read by nobody.
    const int t_T = TEMPORAL ? p.t_T : 1;
    long long t_grp0 = 0;
    int t_tok0 = 0;
    if (TEMPORAL) {
        const int bpg = p.rpg / RB_BM, grp = blockIdx.x / bpg;
        t_grp0 = (long long)grp * p.rpg;
        t_tok0 = (blockIdx.x - grp * bpg) * (RB_BM / t_T);
    }
    auto srow = [&](int r) -> long long {
        if (!TEMPORAL) return m0 + r;
        const int tl = r / t_T, f = r - tl * t_T, tok = t_tok0 + tl;
        return t_grp0 + (tok < p.t_N ? (long long)f * p.t_N + tok : (long long)t_T * p.t_N + (tok - p.t_N) * t_T + f);
    };

    // ---- prologue: everything the first k-steps and the epilogues need, issued back to back
    // phase-1 activations: LDS-DMA into fragment order; wave w stages k-steps w and w + 8 (3 row tiles each)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int ks = wave + RB_NW * j;
        if (ks < G1) {
#pragma unroll
            for (int rt = 0; rt < 3; ++rt)
                rb_dma16(p.A + srow(16 * rt + l15) * p.lda + ks * 32 + 8 * lq, &R1[(ks * 3 + rt) * 64]);
        }
    }
    x8 wf[D][RB_CT];
#pragma unroll
    for (int b = 0; b < D; ++b) {
        const uint4* s0 = st.at(b);
#pragma unroll
        for (int ct = 0; ct < RB_CT; ++ct) wf[b][ct] = rb_ldw<DT>(s0 + ct * 64);
    }
    const int colw = wave * 64 + 4 * lq;        // this lane's first column inside column tile 0
    // this lane's three rows of the stream, at its first column: 32-bit element offsets from p.x (M * 512 floats < 2^32; three registers
    // instead of three 64-bit pointers alive from the prologue to the last update)
    unsigned xo[3];
#pragma unroll
    for (int rt = 0; rt < 3; ++rt) xo[rt] = (unsigned)(srow(16 * rt + l15) * RB_C + colw);
#define xr(rt_) (p.x + xo[rt_])
    f32x4 rs[3][RB_CT];                              // residual tile of x (or of x_in: input_layer adds to the position embedding,
                                                     // [sample][N][512] broadcast over the frames -- no 25 MB copy to initialise the stream)
#pragma unroll
    for (int rt = 0; rt < 3; ++rt) {
        const float* src = xr(rt);
        if (has_x_in) {
            const int grp = p.rpg > 0 ? m0 / p.rpg : 0, in_grp = p.rpg > 0 ? m0 % p.rpg : m0;
            src = p.x_in + ((long long)grp * p.x_in_period + (in_grp + 16 * rt + l15) % p.x_in_period) * RB_C + colw;
        }
#pragma unroll
        for (int ct = 0; ct < RB_CT; ++ct) rs[rt][ct] = *reinterpret_cast<const f32x4*>(src + 16 * ct);
    }
    {
        // per-column vectors -> LDS.  Unconditional loads (a null vector reads the stream instead and is discarded): no branch + wait
        // per vector
        const int grp = p.rpg > 0 ? m0 / p.rpg : 0;
        {
            const int c = tid;
            const long long mo = (long long)grp * p.mod_ld + c;
            const float* dflt = p.x + c;
            const float vb = *(p.b1 ? p.b1 + c : dflt), vg = *(p.gate1 ? p.gate1 + mo : dflt), vw = *(p.ln1.ln_w ? p.ln1.ln_w + c : dflt);
            const float vlb = *(p.ln1.ln_b ? p.ln1.ln_b + c : dflt), vsc = *(p.ln1.scale ? p.ln1.scale + mo : dflt);
            const float vsh = *(p.ln1.shift ? p.ln1.shift + mo : dflt);
            const float sc = p.ln1.scale ? 1.0f + vsc : 1.0f;
            sPar[c] = p.b1 ? vb : 0.f;
            sPar[RB_C + c] = p.gate1 ? vg : 1.0f;
            sPar[2 * RB_C + c] = (p.ln1.ln_w ? vw : 1.0f) * sc;
            sPar[3 * RB_C + c] = (p.ln1.ln_b ? vlb * sc : 0.f) + (p.ln1.scale ? vsh : 0.f);
            if (MLP || TEMPORAL) {             // the second update of the stream: the MLP's, or the temporal attention's to_out
                const float* sb = TEMPORAL ? p.t_bout : p.b_fc2; const float* sg = TEMPORAL ? p.t_gate : p.gate_m;
                const RbLn& ln = TEMPORAL ? p.t_ln : p.ln2;
                const float ub = *(sb ? sb + c : dflt), ug = *(sg ? sg + mo : dflt), uw = *(ln.ln_w ? ln.ln_w + c : dflt);
                const float ulb = *(ln.ln_b ? ln.ln_b + c : dflt), usc = *(ln.scale ? ln.scale + mo : dflt);
                const float ush = *(ln.shift ? ln.shift + mo : dflt);
                const float sc2 = ln.scale ? 1.0f + usc : 1.0f;
                sPar[4 * RB_C + c] = sb ? ub : 0.f;
                sPar[5 * RB_C + c] = sg ? ug : 1.0f;
                sPar[6 * RB_C + c] = (ln.ln_w ? uw : 1.0f) * sc2;
                sPar[7 * RB_C + c] = (ln.ln_b ? ulb * sc2 : 0.f) + (ln.scale ? ush : 0.f);
            }
        }
        // (a global load in an epilogue would be waited for IN ORDER behind the whole weight prefetch: every bias sits in LDS)
        float vf1[RB_MAX_HIDDEN / RB_THREADS], vb3[RB_MAX_N3 / RB_THREADS];
#pragma unroll
        for (int j = 0; j < RB_MAX_HIDDEN / RB_THREADS; ++j) {
            const int c = tid + RB_THREADS * j;
            if (TEMPORAL) vf1[j] = *(j < 3 ? (p.t_bqkv ? p.t_bqkv + c : p.x + tid) : (p.t_gq ? p.t_gq + tid : p.x + tid));      // [b_q | b_k | b_v | gamma_q]
            else vf1[j] = *((MLP && p.b_fc1 && c < p.hidden) ? p.b_fc1 + c : p.x + tid);
        }
#pragma unroll
        for (int j = 0; j < RB_MAX_N3 / RB_THREADS; ++j) {
            const int c = tid + RB_THREADS * j;
            vb3[j] = *((p.b3 && c < P3 * RB_C) ? p.b3 + c : p.x + tid);
        }
#pragma unroll
        for (int j = 0; j < RB_MAX_HIDDEN / RB_THREADS; ++j)
            if (TEMPORAL) sPar[8 * RB_C + tid + RB_THREADS * j] = j < 3 ? (p.t_bqkv ? vf1[j] : 0.f) : (p.t_gq ? vf1[j] : 1.0f);
            else if (MLP) sPar[8 * RB_C + tid + RB_THREADS * j] = (p.b_fc1 && tid + RB_THREADS * j < p.hidden) ? vf1[j] : 0.f;
#pragma unroll
        for (int j = 0; j < RB_MAX_N3 / RB_THREADS; ++j)
            sPar[8 * RB_C + RB_MAX_HIDDEN + tid + RB_THREADS * j] = (p.b3 && tid + RB_THREADS * j < P3 * RB_C) ? vb3[j] : 0.f;
        if (TEMPORAL) sPar[8 * RB_C + RB_MAX_HIDDEN + RB_MAX_N3 + tid] = p.t_gk ? p.t_gk[tid] : 1.0f;
        else if (has_kt) sPar[8 * RB_C + RB_MAX_HIDDEN + RB_MAX_N3 + tid] = p.gamma_k ? p.gamma_k[tid] : 1.0f;
    }
    if (has_in_x) {
        // input_layer in fp32 on the plain ALUs (16 input channels: 0.4 MFLOP per workgroup): W_in^T [Cin][512] and the block's 48 input rows
        // go to R1 (free: there is no phase-1 operand)
        float* sWt = reinterpret_cast<float*>(R1);
        float* sXin = sWt + 16 * RB_C;
        for (int i = tid; i < p.in_cin * RB_C / 4; i += RB_THREADS)
            reinterpret_cast<f32x4*>(sWt)[i] = reinterpret_cast<const f32x4*>(p.in_wt)[i];
        for (int i = tid; i < RB_BM * 16; i += RB_THREADS) {
            const int r = i >> 4, k = i & 15;
            sXin[i] = k < p.in_cin ? p.in_x[(long long)(m0 + r) * p.in_cin + k] : 0.f;
        }
    }
    f32x4 acc[3][RB_CT];
    rb_zero(acc);
    // pin the residual tile here: left alone, the compiler sinks these loads to their first use -- the epilogue, after the k-loop.
    // (Requesting it later, piece by piece under the k-loop, measured 0.7 us better per launch and cost 9 spilled registers: VMEM results
    // are counted back in order, so the pieces have to ride in front of refills that are waited for 4 steps later.)
#pragma unroll
    for (int rt = 0; rt < 3; ++rt)
#pragma unroll
        for (int ct = 0; ct < RB_CT; ++ct) asm volatile("" : "+v"(rs[rt][ct]));
    __syncthreads();                             // activations landed (own DMA drained before the barrier), parameters visible
    RB_STAMP();

    if (has_in_x) {                     // rs (= position embedding or x) += b_in + in_x W_in^T, column by column in fp32
        const float* sWt = reinterpret_cast<const float*>(R1);
        const float* sXin = sWt + 16 * RB_C;
#pragma unroll
        for (int ct = 0; ct < RB_CT; ++ct) {
            if (p.in_b != nullptr) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.in_b + colw + 16 * ct);
#pragma unroll
                for (int rt = 0; rt < 3; ++rt) rs[rt][ct] += b4;
            }
        }
        for (int k4 = 0; k4 < p.in_cin; k4 += 4) {
            f32x4 xv[3];
#pragma unroll
            for (int rt = 0; rt < 3; ++rt) xv[rt] = *reinterpret_cast<const f32x4*>(sXin + (16 * rt + l15) * 16 + k4);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                for (int ct = 0; ct < RB_CT; ++ct) {
                    const f32x4 w4 = *reinterpret_cast<const f32x4*>(sWt + (k4 + kk) * RB_C + colw + 16 * ct);
#pragma unroll
                    for (int rt = 0; rt < 3; ++rt) rs[rt][ct] += xv[rt][kk] * w4;
                }
            }
        }
    }
    // ---- phase 1: x = x + gate1 * (A W1^T + b1), LayerNorm ln1 -> R0
    int g = 0;
    rb_gemm<D, DT>(acc, wf, R1 + lane, G1, g, st);
    RB_STAMP();
#pragma unroll
    for (int ct = 0; ct < RB_CT; ++ct) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(sPar + colw + 16 * ct);
        const f32x4 g4 = *reinterpret_cast<const f32x4*>(sPar + RB_C + colw + 16 * ct);
#pragma unroll
        for (int rt = 0; rt < 3; ++rt) {
            f32x4 v;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = rs[rt][ct][i] + g4[i] * (acc[rt][ct][i] + b4[i]);
            acc[rt][ct] = v;                     // LayerNorm works on (and overwrites) this copy;
            rs[rt][ct] = v;                      // the store reads this one: overwriting a register a store in flight still has to read
            if (!MLP) *reinterpret_cast<f32x4*>(xr(rt) + 16 * ct) = rs[rt][ct];      // means waiting for it, in order, behind the weight prefetch.
                                                 // (MLP: the stream is written once, after the second update.  TEMPORAL: written here and read
                                                 // back for the second update -- 48 registers that the q / k / v passes and the attention need)
        }
    }
    RB_STAMP();
    unsigned short* hb_rows = (!TEMPORAL && p.hb_out != nullptr) ? p.hb_out + (long long)m0 * RB_C : nullptr;
    rb_layernorm<DT>(acc, sRed, sPar + 2 * RB_C, sPar + 3 * RB_C, p.eps, R0, MLP ? nullptr : hb_rows, wave, lane, lq, l15, colw);

    RB_STAMP();
    if (TEMPORAL) {
        // ---- temporal self attention of the block's tokens, in registers.  hb (R0) -> q, k (normal accumulator layout: lane (l15, lq)
        // holds row 16 rt + l15, columns 4 lq .. + 4 of a column tile) and v (SWAPPED MFMA operands: lane holds rows 16 rt + 4 lq .. + 4
        // of column l15, i.e. the V^T operand of P V).  Wave w owns columns [64 w, 64 w + 64) = heads 2 w and 2 w + 1, so scores,
        // softmax and P V need nothing from another wave; only the output goes through LDS (R1), as the A operand of to_out.
        // Same rounding points as the unfused chain (to_qkv -> 16-bit, csrc/attn.hip's attn_small_kernel): q, k, v rounded to 16 bit,
        // MultiHeadRMSNorm in fp32 on the rounded values, scores and the row sum in fp32, P rounded for the MFMA.
        const float* tb = sPar + 8 * RB_C;                    // [b_q | b_k | b_v | gamma_q]
        const float* tgk = sPar + 8 * RB_C + RB_MAX_HIDDEN + RB_MAX_N3;
        const bool rms = p.t_gq != nullptr;
        uint4 qop[3][2], kop[3][2];                           // [row tile][head of the wave]: the 16 x 16 x 32 operand of (rows, head dims)
        auto qk_pass = [&](uint4 (&op)[3][2], const float* bias, const float* gam) {
            rb_zero(acc);
            rb_gemm<D, DT>(acc, wf, R0 + lane, RB_KS, g, st);
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                f32x4 b4[2], g4[2];
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2) {
                    b4[c2] = *reinterpret_cast<const f32x4*>(bias + colw + 16 * (2 * hh + c2));
                    g4[c2] = *reinterpret_cast<const f32x4*>(gam + colw + 16 * (2 * hh + c2));
                }
#pragma unroll
                for (int rt = 0; rt < 3; ++rt) {
                    // contraction slot e of the lane <-> head dim 16 (e >> 2) + 4 lq + (e & 3): any bijection, the same for q and k
                    float f[8], ss = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        f[e] = LP::from16(LP::to16(acc[rt][2 * hh + (e >> 2)][e & 3] + b4[e >> 2][e & 3]));
                        ss += f[e] * f[e];
                    }
                    if (rms) {
                        ss = rb_rowsum4(ss);
                        const float inv = 5.656854249492381f * __builtin_amdgcn_rsqf(fmaxf(ss, 1e-24f));      // sqrt(32) / max(|x|, 1e-12), hardware rsq (1 ulp)
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] = f[e] * inv * g4[e >> 2][e & 3];
                    }
                    op[rt][hh] = make_uint4(LP::pack(f[0], f[1]), LP::pack(f[2], f[3]), LP::pack(f[4], f[5]), LP::pack(f[6], f[7]));
                }
            }
        };
        rb_lds_barrier();                                     // the normalised rows are complete in R0
        // v FIRST (its weights lead the to_qkv segment of the stream: gvf_rowblock args / dit_ops.rowblock_pack_stream order the passes v | q | k):
        // v^T, transposed by the MFMA itself (swapped operands), waits as 24 packed registers for the probabilities -- the other way round the
        // 36 + 6 registers of probabilities and row sums would sit on top of the v pass's accumulators, weight fragments and the residual tile
        // (round 3: 39 registers spilled, one reload inside the k-step loop)
        // ... parked in the wave's own 6 KiB of R1 (free until the attention output goes there: head hh's output fragments are the 3 KiB behind
        // wave * 6 KiB + hh * 3 KiB, and that is where head hh's v^T waits -- it is read back before the head's output is written; LDS
        // operations of one wave execute in order).  24 registers less across the q and k passes.
        uint2* vst = reinterpret_cast<uint2*>(R1) + wave * (6 * 64 * 2) + lane;
        rb_zero(acc);
        rb_gemm<D, DT, true>(acc, wf, R0 + lane, RB_KS, g, st);
#pragma unroll
        for (int ct = 0; ct < RB_CT; ++ct) {
            const float bv = tb[2 * RB_C + wave * 64 + 16 * ct + l15];
#pragma unroll
            for (int rt = 0; rt < 3; ++rt)
                vst[(ct * 3 + rt) * 64] = make_uint2(LP::pack(acc[rt][ct][0] + bv, acc[rt][ct][1] + bv), LP::pack(acc[rt][ct][2] + bv, acc[rt][ct][3] + bv));
        }
        qk_pass(qop, tb, tb + 3 * RB_C);
        qk_pass(kop, tb + RB_C, tgk);
        // S^T tile (rk, rq) = K_rk Q_rq^T: lane (l15, lq) holds query row 16 rq + l15 against key rows 16 rk + 4 lq .. + 4.  A key counts
        // iff it is a frame of the query's token.
        unsigned same_tok[3];                                 // bit 4 rk + i
        {
            int tk[3][4];
#pragma unroll
            for (int rk = 0; rk < 3; ++rk)
#pragma unroll
                for (int i = 0; i < 4; ++i) tk[rk][i] = (16 * rk + 4 * lq + i) / t_T;
#pragma unroll
            for (int rq = 0; rq < 3; ++rq) {
                const int tq = (16 * rq + l15) / t_T;
                unsigned mm = 0;
#pragma unroll
                for (int rk = 0; rk < 3; ++rk)
#pragma unroll
                    for (int i = 0; i < 4; ++i) mm |= (tk[rk][i] == tq ? 1u : 0u) << (4 * rk + i);
                same_tok[rq] = mm;
            }
        }
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        uint2* fb1 = rb_frag_base(R1, wave, lq, l15);
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            uint2 ppk[3][3];                                  // [rq][rk]: this head's probabilities, 16 bit
            float linv[3];
#pragma unroll
            for (int rq = 0; rq < 3; ++rq) {
                f32x4 sc[3];
                float m = -INFINITY;
#pragma unroll
                for (int rk = 0; rk < 3; ++rk) {
                    sc[rk] = LP::mfma16(__builtin_bit_cast(x8, kop[rk][hh]), __builtin_bit_cast(x8, qop[rq][hh]), zero4);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (!((same_tok[rq] >> (4 * rk + i)) & 1u)) sc[rk][i] = -INFINITY;
                        m = fmaxf(m, sc[rk][i]);
                    }
                }
                m = rb_rowmax4(m);            // finite: the query's own row is one of its keys
                const float ms = m * p.t_kscale;
                float l = 0.f;
#pragma unroll
                for (int rk = 0; rk < 3; ++rk) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) { sc[rk][i] = __builtin_amdgcn_exp2f(sc[rk][i] * p.t_kscale - ms); l += sc[rk][i]; }
                    ppk[rq][rk] = make_uint2(LP::pack(sc[rk][0], sc[rk][1]), LP::pack(sc[rk][2], sc[rk][3]));
                }
                l = rb_rowsum4(l);
                linv[rq] = __builtin_amdgcn_rcpf(l);
            }
            // O^T = V^T P^T: contraction slots 0..3 <-> keys 4 lq .. + 4 of one 16-row tile, 4..7 of another (the third pairs with zeros)
            uint2 vpk[2][3];
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                for (int rt = 0; rt < 3; ++rt) vpk[c2][rt] = vst[((2 * hh + c2) * 3 + rt) * 64];
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) {
                const int ct = 2 * hh + c2;
                const x8 v01 = __builtin_bit_cast(x8, make_uint4(vpk[c2][0].x, vpk[c2][0].y, vpk[c2][1].x, vpk[c2][1].y));
                const x8 v2z = __builtin_bit_cast(x8, make_uint4(vpk[c2][2].x, vpk[c2][2].y, 0u, 0u));
#pragma unroll
                for (int rq = 0; rq < 3; ++rq) {
                    const uint2 (&pp)[3] = ppk[rq];
                    f32x4 o = LP::mfma16(v01, __builtin_bit_cast(x8, make_uint4(pp[0].x, pp[0].y, pp[1].x, pp[1].y)), zero4);
                    o = LP::mfma16(v2z, __builtin_bit_cast(x8, make_uint4(pp[2].x, pp[2].y, 0u, 0u)), o);
                    const float li = linv[rq];
                    rb_put_frag(fb1, ct, rq, make_uint2(LP::pack(o[0] * li, o[1] * li), LP::pack(o[2] * li, o[3] * li)));
                }
            }
            if (hh == 0) {
                // the residual tile comes back now (this lane's own stores of the phase-1 epilogue: same wave, same addresses, program
                // order), under the second head's softmax arithmetic (the first head's operands are dead); consumed behind the to_out GEMM
#pragma unroll
                for (int rt = 0; rt < 3; ++rt)
#pragma unroll
                    for (int ct = 0; ct < RB_CT; ++ct) rs[rt][ct] = *reinterpret_cast<const volatile f32x4*>(xr(rt) + 16 * ct);
            }
        }
        rb_lds_barrier();                                     // the attention output is complete in R1
        // ---- x += t_gate * (o Wout^T + t_bout), LayerNorm t_ln -> R0
        rb_zero(acc);
        rb_gemm<D, DT>(acc, wf, R1 + lane, RB_KS, g, st);
#pragma unroll
        for (int ct = 0; ct < RB_CT; ++ct) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(sPar + 4 * RB_C + colw + 16 * ct);
            const f32x4 g4 = *reinterpret_cast<const f32x4*>(sPar + 5 * RB_C + colw + 16 * ct);
#pragma unroll
            for (int rt = 0; rt < 3; ++rt) {
                f32x4 v;
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = rs[rt][ct][i] + g4[i] * (acc[rt][ct][i] + b4[i]);
                acc[rt][ct] = v;
                rs[rt][ct] = v;
                *reinterpret_cast<f32x4*>(xr(rt) + 16 * ct) = rs[rt][ct];
            }
        }
        rb_layernorm<DT>(acc, sRed, sPar + 6 * RB_C, sPar + 7 * RB_C, p.eps, R0, nullptr, wave, lane, lq, l15, colw);
    }
    if (MLP) {
        // ---- MLP: per 512 hidden units  h = gelu(R0 Wfc1[slice]^T + b) -> R1 (bf16 fragments);  acc2 += R1 Wfc2[:, slice]^T
        f32x4 acc2[3][RB_CT];
        rb_zero(acc2);
        uint2* fb1 = rb_frag_base(R1, wave, lq, l15);
        for (int s = 0; s < Pm; ++s) {
            rb_lds_barrier();                     // R0 complete (s = 0) / every wave done reading R1 (s > 0)
            rb_zero(acc);
            rb_gemm<D, DT>(acc, wf, R0 + lane, RB_KS, g, st);
#pragma unroll
            for (int ct = 0; ct < RB_CT; ++ct) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(sPar + 8 * RB_C + s * RB_C + colw + 16 * ct);
#pragma unroll
                for (int rt = 0; rt < 3; ++rt) {
                    float y[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) y[i] = rb_gelu_tanh(acc[rt][ct][i] + b4[i]);
                    uint2 o;
                    o.x = LP::pack(y[0], y[1]);
                    o.y = LP::pack(y[2], y[3]);
                    rb_put_frag(fb1, ct, rt, o);
                }
            }
            rb_lds_barrier();                     // the slice is complete in R1
            rb_gemm<D, DT>(acc2, wf, R1 + lane, RB_KS, g, st);
        }
#pragma unroll
        for (int ct = 0; ct < RB_CT; ++ct) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(sPar + 4 * RB_C + colw + 16 * ct);
            const f32x4 g4 = *reinterpret_cast<const f32x4*>(sPar + 5 * RB_C + colw + 16 * ct);
#pragma unroll
            for (int rt = 0; rt < 3; ++rt) {
                f32x4 v;
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = rs[rt][ct][i] + g4[i] * (acc2[rt][ct][i] + b4[i]);
                acc2[rt][ct] = v;
                rs[rt][ct] = v;
                *reinterpret_cast<f32x4*>(xr(rt) + 16 * ct) = rs[rt][ct];
            }
        }
        RB_STAMP();
        if (P3 == 0 && hb_rows == nullptr) return;       // last block: final_layer reads the stream itself (gvf_dit_final_layer_f32)
        rb_layernorm<DT>(acc2, sRed, sPar + 6 * RB_C, sPar + 7 * RB_C, p.eps, R0, hb_rows, wave, lane, lq, l15, colw);
        RB_STAMP();
    }
    if (P3 == 0) return;
    rb_lds_barrier();                             // the normalised rows are complete in R0
    RB_STAMP();

    // ---- last projection: out[:, 512 pass .. + 512] = R0 W3[pass]^T + b3.  A lane holds 4 columns of a row: written straight out that is
    // 8-byte pieces, 32 contiguous bytes per row per instruction (measured: 20 us per 12.6 MB pass).  So the tile goes through R1 (free by
    // now) as rows of 64 16-byte chunks, chunk c of row r at slot c ^ (r & 15) (conflict-free both ways), and leaves as whole 1 KiB rows.
    const float* b3p = sPar + 8 * RB_C + RB_MAX_HIDDEN;
    uint2* stg_w = reinterpret_cast<uint2*>(R1);
    for (int pass = 0; pass < P3; ++pass) {
        rb_zero(acc);
        rb_gemm<D, DT>(acc, wf, R0 + lane, RB_KS, g, st);
        const int col0 = pass * RB_C + colw;
        if (pass > 0) rb_lds_barrier();          // every wave has drained the previous tile
#pragma unroll
        for (int ct = 0; ct < RB_CT; ++ct) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(b3p + col0 + 16 * ct);
            const int chunk = 8 * wave + 2 * ct + (lq >> 1);
#pragma unroll
            for (int rt = 0; rt < 3; ++rt) {
                uint2 o;
                o.x = LP::pack(acc[rt][ct][0] + b4[0], acc[rt][ct][1] + b4[1]);
                o.y = LP::pack(acc[rt][ct][2] + b4[2], acc[rt][ct][3] + b4[3]);
                stg_w[((16 * rt + l15) * 64 + (chunk ^ l15)) * 2 + (lq & 1)] = o;
            }
        }
        rb_lds_barrier();
        if (has_kt && pass == 1) {
            // K rows -> tile images: lane = 16-byte chunk of the row = (head lane / 4, dims 8 (lane & 3) ..): MultiHeadRMSNorm over the 4
            // lanes of a head in fp32, gain, softmax scale * log2 e, ONE rounding; chunk c of key slot s at s * 4 + (c ^ ((s >> 2) & 3)).
            // Exactly gvf_attn_pack_kv_bf16's arithmetic on the same bf16 values: the tiles are bit-identical.
            const float* gk = sPar + 8 * RB_C + RB_MAX_HIDDEN + RB_MAX_N3 + 8 * lane;
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(gk), g1 = *reinterpret_cast<const f32x4*>(gk + 4);
            const float gg[8] = {g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2], g1[3]};
            const int h = lane >> 2, c = lane & 3;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int r = wave * 6 + j;
                const uint4 v = R1[r * 64 + (lane ^ (r & 15))];
                const unsigned w[4] = {v.x, v.y, v.z, v.w};
                float k8[8], ss = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    k8[2 * i] = LP::lo(w[i]); k8[2 * i + 1] = LP::hi(w[i]);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) ss += k8[e] * k8[e];
                ss += __shfl_xor(ss, 1, 64);
                ss += __shfl_xor(ss, 2, 64);
                const float mul = p.gamma_k != nullptr ? p.k_scale * 5.656854249492381f / fmaxf(sqrtf(ss), 1e-12f) : p.k_scale;
                uint4 o;
                o.x = LP::pack(k8[0] * mul * gg[0], k8[1] * mul * gg[1]); o.y = LP::pack(k8[2] * mul * gg[2], k8[3] * mul * gg[3]);
                o.z = LP::pack(k8[4] * mul * gg[4], k8[5] * mul * gg[5]); o.w = LP::pack(k8[6] * mul * gg[6], k8[7] * mul * gg[7]);
                int row = m0 + r;
                if (p.kv_group_rows > 0) {               // padded groups: rows behind a group's keys write nothing, key sets count keys only
                    const int grp = m0 / p.rpg, local = row - grp * p.rpg;
                    if (local >= p.kv_group_rows) continue;
                    row = grp * p.kv_group_rows + local;
                }
                const int set = row / p.kv_L, key = row - set * p.kv_L, key_l = key & 63;
                p.kt[((long long)(set * (RB_C / 32) + h) * p.kv_tiles + (key >> 6)) * 256 + key_l * 4 + (c ^ ((key_l >> 2) & 3))] = o;
            }
        } else if (has_kt && pass == 2) {
            // V rows -> V^T tile images: thread = one (head, d) column; per 16-row group g' and half hf one 16-byte chunk = the 8 keys
            // 16 g' + 4 hf + {0,1,2,3,8,9,10,11} (the order the score accumulator of attn_xt holds them), chunk j = 2 g + hf of row d at
            // d * 8 + (j ^ ((d >> 1) & 7)).  A 16-row group never straddles a frame or a 64-key tile (kv_L % 64 == 0, m0 % 16 == 0).
            const int h = tid >> 5, d = tid & 31, cch = tid >> 3, ce = tid & 7;
            const unsigned short* stg = reinterpret_cast<const unsigned short*>(R1);
#pragma unroll
            for (int gq = 0; gq < 3; ++gq) {
                int row0 = m0 + 16 * gq;
                if (p.kv_group_rows > 0) {               // (a 16-row group is all keys or all padding: kv_group_rows % 64 == 0)
                    const int grp = m0 / p.rpg, local = row0 - grp * p.rpg;
                    if (local >= p.kv_group_rows) continue;
                    row0 = grp * p.kv_group_rows + local;
                }
                const int set = row0 / p.kv_L, key0 = row0 - set * p.kv_L;
                uint4* dst = p.vt + ((long long)(set * (RB_C / 32) + h) * p.kv_tiles + (key0 >> 6)) * 256 + d * 8;
                const int g = (key0 & 63) >> 4;
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    unsigned vw[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int e0 = 2 * i, e1 = 2 * i + 1;
                        const int r0 = 16 * gq + 4 * hf + (e0 & 3) + 8 * (e0 >> 2), r1 = 16 * gq + 4 * hf + (e1 & 3) + 8 * (e1 >> 2);
                        const unsigned lo = stg[(r0 * 64 + (cch ^ (r0 & 15))) * 8 + ce], hi = stg[(r1 * 64 + (cch ^ (r1 & 15))) * 8 + ce];
                        vw[i] = lo | (hi << 16);
                    }
                    dst[(2 * g + hf) ^ ((d >> 1) & 7)] = make_uint4(vw[0], vw[1], vw[2], vw[3]);
                }
            }
        } else {
            const int ldo = has_kt ? RB_C : p.N3;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int r = wave * 6 + j;
                const uint4 v = R1[r * 64 + (lane ^ (r & 15))];
#ifdef RB_ABL_NOSTORE3
                if (p.M < 0)
#endif
                *reinterpret_cast<uint4*>(p.out3 + srow(r) * ldo + pass * RB_C + 8 * lane) = v;
            }
        }
        RB_STAMP();
    }
#undef RB_STAMP
#undef xr
}

// W bf16 [N][ldw] (nn.Linear layout) -> fragment order.  One 16-byte chunk per thread.
__global__ __launch_bounds__(256) void rowblock_pack_kernel(const unsigned short* __restrict__ W, int ldw, int N, int K, int Kp,
                                                            uint4* __restrict__ out, long long total) {
    const long long i = blockIdx.x * 256LL + threadIdx.x;
    if (i >= total) return;
    const int lane = (int)(i & 63), ct = (int)((i >> 6) & 7), wv = (int)((i >> 9) & 3);
    const long long s = i >> 11;                 // pass * (Kp / 32) + k-step
    const int G = Kp >> 5;
    const int pass = (int)(s / G), ks = (int)(s % G);
    const int n = pass * RB_C + wv * 128 + ct * 16 + (lane & 15), k = ks * 32 + 8 * (lane >> 4);
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (n < N) {
        const unsigned short* src = W + (long long)n * ldw + k;
        if (k + 8 <= K && (ldw % 8) == 0) {
            v = *reinterpret_cast<const uint4*>(src);
        } else {
            unsigned short e[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) e[j] = k + j < K ? src[j] : (unsigned short)0;
            v.x = e[0] | ((unsigned)e[1] << 16); v.y = e[2] | ((unsigned)e[3] << 16);
            v.z = e[4] | ((unsigned)e[5] << 16); v.w = e[6] | ((unsigned)e[7] << 16);
        }
    }
    out[i] = v;
}

// mlp.0 weight [hidden][512] and mlp.2 weight [512][hidden] -> the interleaved stream of the MLP section: per 512 hidden units, the 16
// k-steps of that slice of mlp.0 (a 512 x 512 pass), then the 16 k-steps of mlp.2 that consume it.
__global__ __launch_bounds__(256) void rowblock_pack_mlp_kernel(const unsigned short* __restrict__ W0, const unsigned short* __restrict__ W2,
                                                                int hidden, uint4* __restrict__ out, long long total) {
    const long long i = blockIdx.x * 256LL + threadIdx.x;
    if (i >= total) return;
    const int lane = (int)(i & 63), ct = (int)((i >> 6) & 7), wv = (int)((i >> 9) & 3);
    const int s = (int)(i >> 11);                // 32 steps per slice
    const int slice = s >> 5, ks = s & 15;
    const int nl = wv * 128 + ct * 16 + (lane & 15), kl = ks * 32 + 8 * (lane >> 4);
    const unsigned short* src = (s & 16) ? W2 + (long long)nl * hidden + slice * RB_C + kl           // mlp.2: output nl, hidden units of the slice
                                         : W0 + (long long)(slice * RB_C + nl) * RB_C + kl;           // mlp.0: hidden unit of the slice, input kl
    out[i] = *reinterpret_cast<const uint4*>(src);
}

}  // namespace

static long long* g_rb_dbg = nullptr;        // RB_TIMING builds of scripts/ubench/rowblock_bench.hip set it

// sizeof / field offsets of the argument struct, for bindings to check their own layout against (tests/test_capi_symbols.py)
extern "C" int gvf_rowblock_args_layout(int32_t* out, int n) {
    const int v[] = {(int)sizeof(gvf_rowblock_args), (int)offsetof(gvf_rowblock_args, x), (int)offsetof(gvf_rowblock_args, in_x), (int)offsetof(gvf_rowblock_args, gate1),
                     (int)offsetof(gvf_rowblock_args, mod_ld), (int)offsetof(gvf_rowblock_args, b_fc1), (int)offsetof(gvf_rowblock_args, ln2),
                     (int)offsetof(gvf_rowblock_args, b3), (int)offsetof(gvf_rowblock_args, hb_out), (int)offsetof(gvf_rowblock_args, k_tiles),
                     (int)offsetof(gvf_rowblock_args, gamma_k), (int)offsetof(gvf_rowblock_args, kv_group_rows), (int)offsetof(gvf_rowblock_args, dtype),
                     (int)offsetof(gvf_rowblock_args, t_frames), (int)offsetof(gvf_rowblock_args, t_b_qkv), (int)offsetof(gvf_rowblock_args, t_scale),
                     (int)offsetof(gvf_rowblock_args, t_ln)};
    const int m = (int)(sizeof(v) / sizeof(v[0]));
    for (int i = 0; i < n && i < m; ++i) out[i] = v[i];
    return m;
}

extern "C" int64_t gvf_rowblock_packed_bytes(int N, int K) {
    if (N <= 0 || K <= 0 || N % RB_C != 0) return GVF_EINVAL;
    const int Kp = (K + RB_KPAD - 1) / RB_KPAD * RB_KPAD;
    return (int64_t)N * Kp * 2;
}

extern "C" int gvf_rowblock_pack_weight(const void* w_bf16, int ldw, int N, int K, void* packed, void* stream_) {
    if (!w_bf16 || !packed || N <= 0 || K <= 0 || N % RB_C != 0 || ldw < K) return GVF_EINVAL;
    if ((((uintptr_t)w_bf16) & 15) || (((uintptr_t)packed) & 15)) return GVF_EINVAL;
    const int Kp = (K + RB_KPAD - 1) / RB_KPAD * RB_KPAD;
    const long long total = (long long)N * Kp / 8;
    (void)hipGetLastError();
    rowblock_pack_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream_>>>(
        (const unsigned short*)w_bf16, ldw, N, K, Kp, (uint4*)packed, total);
    GVF_CHECK_LAUNCH();
    return GVF_OK;
}

extern "C" int gvf_rowblock_pack_mlp(const void* w_fc1_bf16, const void* w_fc2_bf16, int hidden, void* packed, void* stream_) {
    if (!w_fc1_bf16 || !w_fc2_bf16 || !packed || hidden <= 0 || hidden % RB_C != 0 || hidden > RB_MAX_HIDDEN) return GVF_EINVAL;
    if ((((uintptr_t)w_fc1_bf16) & 15) || (((uintptr_t)w_fc2_bf16) & 15) || (((uintptr_t)packed) & 15)) return GVF_EINVAL;
    const long long total = 2LL * hidden * RB_C / 8;
    (void)hipGetLastError();
    rowblock_pack_mlp_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream_>>>(
        (const unsigned short*)w_fc1_bf16, (const unsigned short*)w_fc2_bf16, hidden, (uint4*)packed, total);
    GVF_CHECK_LAUNCH();
    return GVF_OK;
}

static int rowblock_launch(const gvf_rowblock_args* a, int dtype, void* stream_) {
    if (!a || (dtype != GVF_DT_BF16 && dtype != GVF_DT_F16)) return GVF_EINVAL;
    // K1 == 0: no closing projection (the stream already holds the sub-layer's result, e.g. input_layer in fp32): LayerNorm + projection only
    if (a->M < 0 || a->M % RB_BM != 0 || a->C != RB_C || a->K1 < 0 || a->K1 % (32 * RB_DEPTH) != 0 || a->K1 > 512) return GVF_EINVAL;
    if (a->K1 > 0 && (a->lda < a->K1 || a->lda % 8 != 0)) return GVF_EINVAL;
    if (a->M == 0) return GVF_OK;
    if ((a->K1 > 0 && !a->a) || !a->w || !a->x) return GVF_EINVAL;
    if (a->K1 == 0 && (a->gate1 != nullptr || a->b1 != nullptr)) return GVF_EINVAL;
    if (a->in_x != nullptr && (a->K1 != 0 || !a->in_wt || a->in_cin <= 0 || a->in_cin > 16 || (a->in_cin & 3) || (((uintptr_t)a->in_wt) & 15) ||
                               (((uintptr_t)a->in_b) & 15)))
        return GVF_EINVAL;
    if (a->x_in != nullptr && (a->x_in_period <= 0 || a->rows_per_group <= 0 || a->rows_per_group % RB_BM != 0 || (((uintptr_t)a->x_in) & 15)))
        return GVF_EINVAL;
    const gvf_rowblock_ln* lns[2] = {&a->ln1, &a->ln2};
    bool grouped = a->gate1 != nullptr || a->gate_m != nullptr;
    for (int i = 0; i < 2; ++i) {
        if (((lns[i]->ln_w == nullptr) != (lns[i]->ln_b == nullptr)) || ((lns[i]->shift == nullptr) != (lns[i]->scale == nullptr))) return GVF_EINVAL;
        grouped = grouped || lns[i]->scale != nullptr;
    }
    if (grouped && (a->rows_per_group <= 0 || a->rows_per_group % RB_BM != 0 || a->mod_ld < RB_C)) return GVF_EINVAL;
    const bool mlp = a->hidden != 0;
    if (mlp && (a->in_x != nullptr || a->x_in != nullptr)) return GVF_EINVAL;      // (compiled out of the MLP variant)
    if ((long long)a->M * RB_C >= (1LL << 32)) return GVF_EINVAL;                     // 32-bit element offsets into the stream
    if (mlp && (a->hidden < 0 || a->hidden % RB_C != 0 || a->hidden > RB_MAX_HIDDEN)) return GVF_EINVAL;
    if (a->N3 != 0 && (!a->out3 || a->N3 < 0 || a->N3 % RB_C != 0 || a->N3 > RB_MAX_N3 || a->epi3 != GVF_EPI_STORE_BF16))
        return GVF_EINVAL;
    if (a->N3 == 0 && a->hb_out == nullptr && a->hidden == 0) return GVF_EINVAL;       // nothing would consume the LayerNorm (with the MLP: the
                                                                                        // stream update alone is a result; LayerNorm ln2 is skipped)
    if ((a->k_tiles == nullptr) != (a->v_tiles == nullptr)) return GVF_EINVAL;
    if (a->k_tiles != nullptr && (a->N3 != 3 * RB_C || a->kv_L <= 0 || a->kv_L % 64 != 0 || !(a->k_scale > 0.f) ||
                                  (((uintptr_t)a->k_tiles) & 15) || (((uintptr_t)a->v_tiles) & 15)))
        return GVF_EINVAL;
    if (a->k_tiles != nullptr && a->kv_group_rows == 0 && a->M % a->kv_L != 0) return GVF_EINVAL;
    if (a->kv_group_rows != 0 && (a->k_tiles == nullptr || a->kv_group_rows < 0 || a->kv_group_rows % a->kv_L != 0 || a->rows_per_group <= 0 ||
                                  a->rows_per_group % RB_BM != 0 || a->kv_group_rows > a->rows_per_group || a->M % a->rows_per_group != 0))
        return GVF_EINVAL;
    if ((((uintptr_t)a->a) & 15) || (((uintptr_t)a->w) & 15) || (((uintptr_t)a->x) & 15) || (((uintptr_t)a->out3) & 7) ||
        (((uintptr_t)a->hb_out) & 7) || (((uintptr_t)a->b3) & 15))
        return GVF_EINVAL;
    const bool temporal = a->t_frames != 0;
    if (temporal) {
        // blocks of 48 / T tokens x T frames inside a group of T * t_stride rows; one stream: W1 | to_qkv | to_out | W3
        if (a->t_frames < 0 || RB_BM % a->t_frames != 0 || a->t_stride <= 0) return GVF_EINVAL;
        const int tpb = RB_BM / a->t_frames;                      // tokens per workgroup; a group = whole workgroups (its last one may hold phantom tokens)
        if (a->rows_per_group != (a->t_stride + tpb - 1) / tpb * RB_BM || a->M % a->rows_per_group != 0) return GVF_EINVAL;
        if (mlp || a->K1 == 0 || a->in_x != nullptr || a->x_in != nullptr || a->k_tiles != nullptr || a->hb_out != nullptr || a->N3 == 0) return GVF_EINVAL;
        if ((a->t_gamma_q == nullptr) != (a->t_gamma_k == nullptr) || !(a->t_scale > 0.f)) return GVF_EINVAL;
        if (((a->t_ln.ln_w == nullptr) != (a->t_ln.ln_b == nullptr)) || ((a->t_ln.shift == nullptr) != (a->t_ln.scale == nullptr))) return GVF_EINVAL;
        if ((a->t_gate != nullptr || a->t_ln.scale != nullptr) && a->mod_ld < RB_C) return GVF_EINVAL;
    }
    RbParams p;
    p.A = (const unsigned short*)a->a; p.lda = a->lda; p.K1 = a->K1;
    p.W = (const uint4*)a->w; p.b1 = a->b1;
    p.x = a->x; p.M = a->M;
    p.x_in = a->x_in; p.x_in_period = a->x_in_period;
    p.in_x = a->in_x; p.in_wt = a->in_wt; p.in_b = a->in_b; p.in_cin = a->in_cin;
    p.gate1 = a->gate1;
    p.ln1 = RbLn{a->ln1.ln_w, a->ln1.ln_b, a->ln1.shift, a->ln1.scale};
    p.mod_ld = a->mod_ld; p.rpg = (grouped || temporal || a->x_in != nullptr || a->kv_group_rows > 0) ? a->rows_per_group : 0; p.eps = a->eps;
    p.b_fc1 = a->b_fc1; p.b_fc2 = a->b_fc2; p.hidden = a->hidden; p.gate_m = a->gate_m;
    p.ln2 = RbLn{a->ln2.ln_w, a->ln2.ln_b, a->ln2.shift, a->ln2.scale};
    p.b3 = a->b3; p.out3 = a->N3 != 0 ? (unsigned short*)a->out3 : nullptr; p.N3 = a->N3;
    p.hb_out = (unsigned short*)a->hb_out;
    p.kt = (uint4*)a->k_tiles; p.vt = (uint4*)a->v_tiles; p.kv_L = a->kv_L; p.kv_tiles = a->kv_L / 64; p.k_scale = a->k_scale; p.gamma_k = a->gamma_k;
    p.kv_group_rows = a->kv_group_rows;
    p.t_T = a->t_frames; p.t_N = a->t_stride;
    p.t_bqkv = a->t_b_qkv; p.t_gq = a->t_gamma_q; p.t_gk = a->t_gamma_k; p.t_kscale = a->t_scale * 1.4426950408889634f;
    p.t_bout = a->t_b_out; p.t_gate = a->t_gate;
    p.t_ln = RbLn{a->t_ln.ln_w, a->t_ln.ln_b, a->t_ln.shift, a->t_ln.scale};
    p.dbg = g_rb_dbg;
    (void)hipGetLastError();
    const dim3 grid((unsigned)(a->M / RB_BM)), block(RB_THREADS);
    GVF_LP_DISPATCH(dtype,
        if (mlp) rowblock_kernel<true, RB_DEPTH_MLP, DT><<<grid, block, 0, (hipStream_t)stream_>>>(p);
        else if (temporal) rowblock_kernel<false, RB_DEPTH, DT, true><<<grid, block, 0, (hipStream_t)stream_>>>(p);
        else rowblock_kernel<false, RB_DEPTH, DT><<<grid, block, 0, (hipStream_t)stream_>>>(p));
    GVF_CHECK_LAUNCH();
    return GVF_OK;
}

extern "C" int gvf_rowblock_fused(const gvf_rowblock_args* a, void* stream_) { return rowblock_launch(a, a ? a->dtype : -1, stream_); }
extern "C" int gvf_rowblock_fused_bf16(const gvf_rowblock_args* a, void* stream_) { return rowblock_launch(a, GVF_DT_BF16, stream_); }
