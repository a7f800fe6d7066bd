// attn.hip -- bf16 MFMA flash attention forward, head_dim 32, for gfx950 (MI355X).
//
// Replaces the reference's attention operator seam, model/attention/full_attn.py:74-140
// (flash_attn_func / flash_attn_kvpacked_func / sdpa / naive behind scaled_dot_product_attention) for
// the four shapes of the DiT block (model/dit.py:246-270): spatial self (L 512), temporal self
// (L = frames), image cross (Lk 1370), static cross (Lk 4096), with the MultiHeadRMSNorm of q and k
// (model/attention/modules.py:8-15) fused into the operand loads.
//
// One workgroup = 128 query rows of one (batch, head); each of the 4 waves owns 32 rows.
// Swapped product: S^T = K Q^T with v_mfma_f32_32x32x16_bf16 (A = K rows, B = Q^T) so that a lane holds
// 16 scores of ONE query (lane & 31) -- the row max / row sum are in-lane reductions plus one exchange
// with lane ^ 32 -- and the bf16 probabilities it produces are already the B operand of the second
// product O^T = V^T P^T (A = V^T from a transposed LDS tile, contraction slots permuted to the
// accumulator's row order so P never moves between lanes).  K/V tiles of 64 keys are staged through
// LDS once per workgroup (K: 16-byte chunks XOR-swizzled over 4-row groups; V: written transposed).
// fp32 online softmax (running max / sum per query, exp2 with log2e folded into the scale).
#include "gvf_common.h"
#include "../../include/gvf_rast.h"
#include "../../include/gvf_dit.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int D = 32;            // head dim
constexpr int QB = 128;          // queries per workgroup
constexpr int KT = 64;           // keys per staged tile
constexpr int THREADS = 256;
constexpr int VT_LD = KT + 8;    // row stride (bf16 elements) of the transposed V tile

__device__ __forceinline__ unsigned short f2bf(float f) {
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }

struct AttnParams {
    const unsigned short *q, *k, *v;
    unsigned short* out;
    int n_inner, Lq, Lk, H, q_blocks;
    long long q_so, q_si, q_sl, k_so, k_si, k_sl, v_so, v_si, v_sl, o_so, o_si, o_sl;
    const float *gamma_q, *gamma_k;
    float scale_log2e;
};

// 8 bf16 (one 16-byte chunk of a head row) -> RMS-normalised * gamma * sqrt(32), given the row's sum of squares
__device__ __forceinline__ uint4 rms_apply(uint4 raw, float sumsq, const float* g8) {
    const float inv = 5.656854249492381f / fmaxf(sqrtf(sumsq), 1e-12f);   // sqrt(32) / max(||x||, eps)
    unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float lo = bf2f((unsigned short)(w[i] & 0xffffu)) * inv * g8[2 * i];
        float hi = bf2f((unsigned short)(w[i] >> 16)) * inv * g8[2 * i + 1];
        w[i] = (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16);
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}
__device__ __forceinline__ float sumsq8(uint4 raw) {
    unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float lo = bf2f((unsigned short)(w[i] & 0xffffu)), hi = bf2f((unsigned short)(w[i] >> 16));
        s += lo * lo + hi * hi;
    }
    return s;
}

__global__ __launch_bounds__(THREADS) void attn_fwd_kernel(AttnParams p) {
    __shared__ uint4 sK[KT * 4];                      // [key][4 chunks of 8 bf16], chunk ^= (key >> 2) & 3
    __shared__ unsigned short sVT[D * VT_LD];         // [d][key]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;

    int bid = blockIdx.x;
    const int qb = bid % p.q_blocks; bid /= p.q_blocks;
    const int head = bid % p.H; bid /= p.H;
    const int inner = bid % p.n_inner, outer = bid / p.n_inner;

    const unsigned short* qp = p.q + outer * p.q_so + inner * p.q_si + head * D;
    const unsigned short* kp = p.k + outer * p.k_so + inner * p.k_si + head * D;
    const unsigned short* vp = p.v + outer * p.v_so + inner * p.v_si + head * D;
    unsigned short* op = p.out + outer * p.o_so + inner * p.o_si + head * D;

    // ---- Q fragments: B operand of S^T = K Q^T.  Lane (q = lane&31, half): Q[q][16s + 8*half .. +7], s = 0,1
    const int qrow = qb * QB + wave * 32 + l31;
    const bool qvalid = qrow < p.Lq;
    uint4 qraw[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const uint4 v4 = *reinterpret_cast<const uint4*>(qp + (long long)(qvalid ? qrow : 0) * p.q_sl + 16 * s + 8 * half);
        const unsigned m = qvalid ? 0xffffffffu : 0u;
        qraw[s] = make_uint4(v4.x & m, v4.y & m, v4.z & m, v4.w & m);
    }
    if (p.gamma_q != nullptr) {
        float ss = sumsq8(qraw[0]) + sumsq8(qraw[1]);
        ss += __shfl_xor(ss, 32, 64);
#pragma unroll
        for (int s = 0; s < 2; ++s) qraw[s] = rms_apply(qraw[s], ss, p.gamma_q + head * D + 16 * s + 8 * half);
    }
    const bf16x8 qf0 = __builtin_bit_cast(bf16x8, qraw[0]);
    const bf16x8 qf1 = __builtin_bit_cast(bf16x8, qraw[1]);

    f32x16 o_acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) o_acc[r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // staging role of this thread: key row st_key, 16-byte chunk st_c of the 64-byte head row
    const int st_key = tid >> 2, st_c = tid & 3;
    float gk8[8];
    if (p.gamma_k != nullptr) {
#pragma unroll
        for (int e = 0; e < 8; ++e) gk8[e] = p.gamma_k[head * D + st_c * 8 + e];
    }

    const int n_tiles = (p.Lk + KT - 1) / KT;
    for (int kt = 0; kt < n_tiles; ++kt) {
        // ---- stage K (row-major, swizzled) and V (transposed) for keys [kt*64, kt*64+64)
        {
            const int key = kt * KT + st_key;
            const bool kvalid = key < p.Lk;
            const long long krow = kvalid ? key : 0;
            uint4 kraw = *reinterpret_cast<const uint4*>(kp + krow * p.k_sl + st_c * 8);
            uint4 vraw = *reinterpret_cast<const uint4*>(vp + krow * p.v_sl + st_c * 8);
            const unsigned m = kvalid ? 0xffffffffu : 0u;
            kraw = make_uint4(kraw.x & m, kraw.y & m, kraw.z & m, kraw.w & m);
            vraw = make_uint4(vraw.x & m, vraw.y & m, vraw.z & m, vraw.w & m);
            if (p.gamma_k != nullptr) {
                float ss = sumsq8(kraw);
                ss += __shfl_xor(ss, 1, 64);
                ss += __shfl_xor(ss, 2, 64);
                kraw = rms_apply(kraw, ss, gk8);
            }
            __syncthreads();   // previous tile fully consumed
            sK[st_key * 4 + (st_c ^ ((st_key >> 2) & 3))] = kraw;
            const unsigned w[4] = {vraw.x, vraw.y, vraw.z, vraw.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                sVT[(st_c * 8 + 2 * i) * VT_LD + st_key] = (unsigned short)(w[i] & 0xffffu);
                sVT[(st_c * 8 + 2 * i + 1) * VT_LD + st_key] = (unsigned short)(w[i] >> 16);
            }
            __syncthreads();
        }

#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int key0 = kt * KT + sub * 32;
            if (key0 >= p.Lk) break;
            // ---- S^T[key][q] = sum_d K[key][d] Q[q][d]
            f32x16 s_acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) s_acc[r] = 0.f;
            {
                const int krow_l = sub * 32 + l31;
                const int sw = (krow_l >> 2) & 3;
                const bf16x8 k0 = __builtin_bit_cast(bf16x8, sK[krow_l * 4 + ((0 + half) ^ sw)]);
                const bf16x8 k1 = __builtin_bit_cast(bf16x8, sK[krow_l * 4 + ((2 + half) ^ sw)]);
                s_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, qf0, s_acc, 0, 0, 0);
                s_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, qf1, s_acc, 0, 0, 0);
            }
            // accumulator row r of this lane is key  key0 + (r&3) + 8*(r>>2) + 4*half
            float mloc = -INFINITY;
            const bool partial = key0 + 32 > p.Lk;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float sv = s_acc[r] * p.scale_log2e;
                if (partial && (key0 + (r & 3) + 8 * (r >> 2) + 4 * half) >= p.Lk) sv = -INFINITY;
                s_acc[r] = sv;
                mloc = fmaxf(mloc, sv);
            }
            mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
            const float m_new = fmaxf(m_run, mloc);        // finite: every sub-tile holds >= 1 valid key
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            float psum = 0.f;
            unsigned pw[8];
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const float p0 = __builtin_amdgcn_exp2f(s_acc[r] - m_new);
                const float p1 = __builtin_amdgcn_exp2f(s_acc[r + 1] - m_new);
                psum += p0 + p1;
                pw[r >> 1] = (unsigned)f2bf(p0) | ((unsigned)f2bf(p1) << 16);
            }
            l_run = l_run * alpha + psum;
            m_run = m_new;
#pragma unroll
            for (int r = 0; r < 16; ++r) o_acc[r] *= alpha;
            // ---- O^T[d][q] += sum_slots V^T[d][key(slot)] P^T[key(slot)][q]; slot (u, half, e) = accumulator row 8u+e
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const bf16x8 pf = __builtin_bit_cast(bf16x8, make_uint4(pw[4 * u], pw[4 * u + 1], pw[4 * u + 2], pw[4 * u + 3]));
                const unsigned short* vrow = sVT + l31 * VT_LD + sub * 32 + 16 * u + 4 * half;
                const uint2 va = *reinterpret_cast<const uint2*>(vrow);        // keys +0..3
                const uint2 vb = *reinterpret_cast<const uint2*>(vrow + 8);    // keys +8..11
                const bf16x8 vf = __builtin_bit_cast(bf16x8, make_uint4(va.x, va.y, vb.x, vb.y));
                o_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o_acc, 0, 0, 0);
            }
        }
    }

    // ---- epilogue: O[q][d] / l, d = (r&3) + 8*(r>>2) + 4*half
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    if (qvalid) {
        const float inv = 1.0f / l_tot;
        unsigned short* orow = op + (long long)qrow * p.o_sl;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint2 w;
            w.x = (unsigned)f2bf(o_acc[4 * g] * inv) | ((unsigned)f2bf(o_acc[4 * g + 1] * inv) << 16);
            w.y = (unsigned)f2bf(o_acc[4 * g + 2] * inv) | ((unsigned)f2bf(o_acc[4 * g + 3] * inv) << 16);
            *reinterpret_cast<uint2*>(orow + 8 * g + 4 * half) = w;
        }
    }
}

}  // namespace

extern "C" int gvf_attn_fwd_bf16(const void* q, const void* k, const void* v, void* out, int n_outer, int n_inner,
                                 int Lq, int Lk, int H, int64_t q_so, int64_t q_si, int64_t q_sl, int64_t k_so,
                                 int64_t k_si, int64_t k_sl, int64_t v_so, int64_t v_si, int64_t v_sl, int64_t o_so,
                                 int64_t o_si, int64_t o_sl, const float* gamma_q, const float* gamma_k, float scale,
                                 void* stream_) {
    if (n_outer < 0 || n_inner <= 0 || Lq < 0 || Lk <= 0 || H <= 0) return GVF_EINVAL;
    if (n_outer == 0 || Lq == 0) return GVF_OK;
    if (!q || !k || !v || !out) return GVF_EINVAL;
    // 16-byte operand chunks: bases and every stride must keep 8-element alignment
    const int64_t strides[] = {q_so, q_si, q_sl, k_so, k_si, k_sl, v_so, v_si, v_sl};
    for (int64_t s : strides)
        if (s % 8 != 0) return GVF_EINVAL;
    if ((o_so % 4) || (o_si % 4) || (o_sl % 4)) return GVF_EINVAL;
    if ((((uintptr_t)q) & 15) || (((uintptr_t)k) & 15) || (((uintptr_t)v) & 15) || (((uintptr_t)out) & 7)) return GVF_EINVAL;
    AttnParams p;
    p.q = (const unsigned short*)q; p.k = (const unsigned short*)k; p.v = (const unsigned short*)v;
    p.out = (unsigned short*)out;
    p.n_inner = n_inner; p.Lq = Lq; p.Lk = Lk; p.H = H; p.q_blocks = (Lq + QB - 1) / QB;
    p.q_so = q_so; p.q_si = q_si; p.q_sl = q_sl; p.k_so = k_so; p.k_si = k_si; p.k_sl = k_sl;
    p.v_so = v_so; p.v_si = v_si; p.v_sl = v_sl; p.o_so = o_so; p.o_si = o_si; p.o_sl = o_sl;
    p.gamma_q = gamma_q; p.gamma_k = gamma_k;
    p.scale_log2e = scale * 1.4426950408889634f;
    const long long blocks = (long long)p.q_blocks * H * n_inner * n_outer;
    if (blocks > 0x7fffffffLL) return GVF_EINVAL;
    (void)hipGetLastError();
    hipLaunchKernelGGL(attn_fwd_kernel, dim3((unsigned)blocks), dim3(THREADS), 0, (hipStream_t)stream_, p);
    GVF_CHECK_LAUNCH();
    return GVF_OK;
}
