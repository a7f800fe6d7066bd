// elem.hip -- memory-bound helpers of the DiT block for gfx950: fused LayerNorm + (affine | adaLN
// modulate) writing the bf16 GEMM operand, and fp32 -> bf16 cast with zero padding / SiLU.
//
// Reference: model/dit.py:168-172,246-277 (norm1..5: LayerNorm(eps 1e-6), affine only on norm3/4;
// h * (1 + scale) + shift), model/dit.py:217-225,240-242 (SiLU in front of the adaLN Linear).
// One wave per row (C <= 1024): the row is read once with 16-byte loads and kept in registers for the
// mean / variance / normalise passes (two-pass variance, as torch's LayerNorm computes it).
#include <cmath>
#include "gvf_common.h"
#include "gvf_lp.h"
#include "../../include/gvf_rast.h"
#include "../../include/gvf_dit.h"

namespace {

__device__ __forceinline__ unsigned short f2bf(float f) {
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

// Sum over the 64 lanes, every lane gets it -- on the DPP / permlane data paths only (quad_perm, row mirrors, v_permlane16_swap,
// v_permlane32_swap), nothing through the LDS unit.  #ifdef GVF_WAVE_SUM_BPERMUTE: the round-1 form (six ds_bpermute butterflies).
__device__ __forceinline__ float wave_sum(float v) {
#ifdef GVF_WAVE_SUM_BPERMUTE
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
#else
#define GVF_DPP(x_, ctrl_) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x_), ctrl_, 0xF, 0xF, false))
    v += GVF_DPP(v, 0xB1);      // quad_perm [1,0,3,2]: lane ^ 1
    v += GVF_DPP(v, 0x4E);      // quad_perm [2,3,0,1]: lane ^ 2
    v += GVF_DPP(v, 0x141);     // row_half_mirror: lane i of 8 <-> 7 - i (the other quad of the half row)
    v += GVF_DPP(v, 0x140);     // row_mirror: lane i of 16 <-> 15 - i (the other half of the row)
#undef GVF_DPP
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto r16 = __builtin_amdgcn_permlane16_swap(u, u, false, false);      // {rows 0,0,2,2 | rows 1,1,3,3}: their sum = rows 0+1 | 2+3
    const unsigned a0 = r16[0], a1 = r16[1];
    const float w = __uint_as_float(a0) + __uint_as_float(a1);
    const unsigned uw = __builtin_bit_cast(unsigned, w);
    const auto r32 = __builtin_amdgcn_permlane32_swap(uw, uw, false, false);
    const unsigned b0 = r32[0], b1 = r32[1];
    return __uint_as_float(b0) + __uint_as_float(b1);
#endif
}

// VPL = float4 loads per lane: C = 64 * 4 * VPL
template <int VPL, int DT>
__global__ __launch_bounds__(256) void ln_mod_kernel(const float* __restrict__ x, unsigned short* __restrict__ out,
                                                     int rows, int C, float eps, const float* __restrict__ ln_w,
                                                     const float* __restrict__ ln_b, const float* __restrict__ shift,
                                                     const float* __restrict__ scale, int mod_ld, int rpg) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * C);
    float4 v[VPL];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        v[i] = xr[lane + 64 * i];
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
        q += (a * a + b * b) + (c * c + d * d);
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
    const int g = row / rpg;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c0 = (lane + 64 * i) * 4;
        float y[4] = {(v[i].x - mean) * rstd, (v[i].y - mean) * rstd, (v[i].z - mean) * rstd, (v[i].w - mean) * rstd};
        if (ln_w != nullptr) {
            const float4 w4 = *reinterpret_cast<const float4*>(ln_w + c0);
            const float4 b4 = *reinterpret_cast<const float4*>(ln_b + c0);
            y[0] = y[0] * w4.x + b4.x; y[1] = y[1] * w4.y + b4.y; y[2] = y[2] * w4.z + b4.z; y[3] = y[3] * w4.w + b4.w;
        }
        if (scale != nullptr) {
            const float4 sc = *reinterpret_cast<const float4*>(scale + (size_t)g * mod_ld + c0);
            const float4 sh = *reinterpret_cast<const float4*>(shift + (size_t)g * mod_ld + c0);
            y[0] = y[0] * (1.0f + sc.x) + sh.x; y[1] = y[1] * (1.0f + sc.y) + sh.y;
            y[2] = y[2] * (1.0f + sc.z) + sh.z; y[3] = y[3] * (1.0f + sc.w) + sh.w;
        }
        uint2 o;
        o.x = GvfLp<DT>::pack(y[0], y[1]);
        o.y = GvfLp<DT>::pack(y[2], y[3]);
        *reinterpret_cast<uint2*>(out + (size_t)row * C + c0) = o;
    }
}

// any C (multiple of 4): the row is re-read from cache for each pass
template <int DT>
__global__ __launch_bounds__(256) void ln_mod_generic_kernel(const float* __restrict__ x, unsigned short* __restrict__ out,
                                                             int rows, int C, float eps, const float* __restrict__ ln_w,
                                                             const float* __restrict__ ln_b, const float* __restrict__ shift,
                                                             const float* __restrict__ scale, int mod_ld, int rpg) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (size_t)row * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += xr[c];
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
    for (int c = lane; c < C; c += 64) { const float a = xr[c] - mean; q += a * a; }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
    const int g = row / rpg;
    for (int c = lane; c < C; c += 64) {
        float y = (xr[c] - mean) * rstd;
        if (ln_w != nullptr) y = y * ln_w[c] + ln_b[c];
        if (scale != nullptr) y = y * (1.0f + scale[(size_t)g * mod_ld + c]) + shift[(size_t)g * mod_ld + c];
        out[(size_t)row * C + c] = GvfLp<DT>::to16(y);
    }
}

template <int DT>
__global__ __launch_bounds__(256) void cast_pad_kernel(const float* __restrict__ src, int ld_src,
                                                       unsigned short* __restrict__ dst, int ld_dst, long long rows,
                                                       int cols, int act) {
    const long long total = rows * (long long)ld_dst;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / ld_dst;
        const int c = (int)(i - r * ld_dst);
        float v = 0.f;
        if (c < cols) {
            v = src[r * ld_src + c];
            if (act == 1) v = v / (1.0f + __expf(-v));
        }
        dst[i] = GvfLp<DT>::to16(v);
    }
}


// fp32 matrix -> its two-term bf16 expansion laid out along K for ONE plain bf16 GEMM (the hoisted condition projections, DiT.prepare_conditions):
//   x = hi + lo + O(2^-17 |x|),  hi = bf16(x),  lo = bf16(x - hi)        (bf16 keeps fp32's exponent range: no scaling, no subnormal cases)
//   activations (mode 0):  A' = [ hi | lo | hi ]   (rows x 3 Kp)
//   weights     (mode 1):  W' = [ hi | hi | lo ]   (rows x 3 Kp)
//   A' W'^T = A_hi W_hi^T + A_lo W_hi^T + A_hi W_lo^T = A W^T - A_lo W_lo^T   -- the fp32 product up to a relative 2^-16 per term, accumulated
// in fp32 by the MFMA in a fixed order per output row (no split-K), at 16-bit matrix-pipe rate: 3 products at 2.5 PFLOP/s peak against one at
// the 0.157 PFLOP/s of v_mfma_f32_32x32x2_f32.  Kp = K rounded up to 64 (zero padded), so every third of A' / W' is a whole number of k-steps.
__global__ __launch_bounds__(256) void split3_bf16_kernel(const float* __restrict__ src, long long ld_src, unsigned short* __restrict__ dst, long long rows,
                                                          int cols, int Kp, int mode) {
    const int per_row = Kp / 4;                                   // one thread = 4 consecutive columns
    const long long total = rows * (long long)per_row;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / per_row;
        const int c = (int)(i - r * per_row) * 4;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        const float* sp = src + r * ld_src + c;
        if (c + 3 < cols && ((((uintptr_t)sp) & 15) == 0)) {
            const float4 q = *reinterpret_cast<const float4*>(sp);
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = c + e < cols ? sp[e] : 0.f;
        }
        unsigned short hi[4], lo[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            hi[e] = f2bf(v[e]);
            lo[e] = f2bf(v[e] - __uint_as_float(((unsigned)hi[e]) << 16));       // exact difference (Sterbenz-like: |x - hi| <= ulp_bf16 / 2), one rounding
        }
        const uint2 h2 = make_uint2((unsigned)hi[0] | ((unsigned)hi[1] << 16), (unsigned)hi[2] | ((unsigned)hi[3] << 16));
        const uint2 l2 = make_uint2((unsigned)lo[0] | ((unsigned)lo[1] << 16), (unsigned)lo[2] | ((unsigned)lo[3] << 16));
        unsigned short* d = dst + r * (3LL * Kp) + c;
        *reinterpret_cast<uint2*>(d) = h2;
        *reinterpret_cast<uint2*>(d + Kp) = mode == 0 ? l2 : h2;
        *reinterpret_cast<uint2*>(d + 2 * Kp) = mode == 0 ? h2 : l2;
    }
}

// TimestepEmbedder + the SiLU in front of the adaLN projections in ONE launch (model/dit.py:59-100: sinusoid -> Linear -> SiLU ->
// Linear; model/dit.py:217-225: SiLU -> Linear): out[b] = bf16(silu(W2 bf16(silu(W0 bf16([cos | sin](t_b f)) + b0)) + b2)), the operand of
// the one GEMM that produces every block's modulation vectors.  One workgroup (16 waves) per sample; a wave computes 8 outputs at a time
// (the 64 lanes read a contiguous piece of the weight row, wave reduction): 0.4 MFLOP, launch-latency sized.  Rounding points as the
// launches it replaces (torch sin / cos, gvf_cast_pad_bf16, gvf_gemm_bf16 x 2): bf16 operands, fp32 accumulation.
__device__ __forceinline__ float bf2f_(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }

// dot products of 8 weight rows (o0 + 16 u) with the LDS vector v[0, K): all 8 rows' loads in flight together (one output at a time
// the kernel is a chain of ~2 us memory latencies: 0.5 ms)
__device__ __forceinline__ void te_dot8(const unsigned short* __restrict__ W, int ldw, int o0, int n_out, int K, const float* v, int lane, float (&acc)[8]) {
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = 0.f;
    for (int k = 4 * lane; k < K; k += 256) {
        uint2 w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int o = o0 + 16 * u;
            w[u] = o < n_out ? *reinterpret_cast<const uint2*>(W + (size_t)o * ldw + k) : make_uint2(0u, 0u);
        }
        const float4 x = *reinterpret_cast<const float4*>(v + k);
#pragma unroll
        for (int u = 0; u < 8; ++u)
            acc[u] += (__uint_as_float(w[u].x << 16) * x.x + __uint_as_float(w[u].x & 0xffff0000u) * x.y) +
                      (__uint_as_float(w[u].y << 16) * x.z + __uint_as_float(w[u].y & 0xffff0000u) * x.w);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = wave_sum(acc[u]);
}

__global__ __launch_bounds__(1024) void timestep_embed_kernel(const float* __restrict__ t, int F, float neg_log_period,
                                                              const unsigned short* __restrict__ W0, int ldw0, const float* __restrict__ b0,
                                                              const unsigned short* __restrict__ W2, int ldw2, const float* __restrict__ b2, int C,
                                                              unsigned short* __restrict__ out, int ld_out, float* __restrict__ t_emb) {
    __shared__ __attribute__((aligned(16))) float sA[1024], sB[1024];          // F <= 1024, C <= 1024 (zero padded to a multiple of 4)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
    const int half = F / 2;
    const float tv = t[b];
    for (int i = tid; i < 1024; i += 1024) {
        float v = 0.f;
        if (i < F) {
            const int j = i < half ? i : i - half;
            const float f = expf(neg_log_period * (float)j / (float)half);     // torch: exp(-log(max_period) * arange(half) / half), the scalar in fp32
            const float a = tv * f;
            v = bf2f_(f2bf(i < half ? cosf(a) : sinf(a)));
        }
        sA[i] = v;
        sB[i] = 0.f;
    }
    __syncthreads();
    float acc[8];
    for (int o0 = wave; o0 < C; o0 += 128) {                 // 16 waves x 8 rows per round
        te_dot8(W0, ldw0, o0, C, (F + 3) & ~3, sA, lane, acc);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int o = o0 + 16 * u;
            if (lane == 0 && o < C) { const float v = acc[u] + (b0 ? b0[o] : 0.f); sB[o] = bf2f_(f2bf(v / (1.0f + __expf(-v)))); }
        }
    }
    __syncthreads();
    // second Linear: this workgroup's slice of the outputs only (gridDim.y workgroups per sample; the first Linear is recomputed by each:
    // one CU pulls ~70 GB/s, so one workgroup alone would spend 13 us on the 0.75 MB of weights)
    const int per = (C + gridDim.y - 1) / gridDim.y, lo = blockIdx.y * per, hi = lo + per < C ? lo + per : C;
    for (int o0 = lo + wave; o0 < hi; o0 += 128) {
        te_dot8(W2, ldw2, o0, hi, (C + 3) & ~3, sB, lane, acc);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int o = o0 + 16 * u;
            if (lane == 0 && o < hi) {
                const float v = acc[u] + (b2 ? b2[o] : 0.f);
                if (t_emb != nullptr) t_emb[(size_t)b * C + o] = v;
                out[(size_t)b * ld_out + o] = f2bf(v / (1.0f + __expf(-v)));
            }
        }
    }
    if (blockIdx.y == 0)
        for (int o = C + tid; o < ld_out; o += 1024) out[(size_t)b * ld_out + o] = 0;
}


// ---- the small projections of the step in FP32 --------------------------------------------------------------------------------------------
// Where the denoiser's bf16 error comes from (measured by re-rounding the oracle site by site, full config: 4.97e-3 of the output with
// everything in bf16): not the big projections (all of them in fp16: 4.44e-3) but the small ones whose result multiplies or feeds
// everything else -- condition projections and to_kv (hoisted: 4.29e-3 with those in fp32), the timestep embedder, the adaLN modulation
// GEMV, final_layer (3.49e-3) and input_layer (2.82e-3).  They are 0.3 % of the FLOPs, so they run in fp32 on the plain ALUs.

// dot products of 8 fp32 weight rows (o0 + stride u) with the LDS vector v[0, K)
__device__ __forceinline__ void dot8_f32(const float* __restrict__ W, int ldw, int o0, int stride, int n_out, int K, const float* v, int lane, float (&acc)[8]) {
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = 0.f;
    for (int k = 4 * lane; k < K; k += 256) {
        float4 w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int o = o0 + stride * u;
            w[u] = o < n_out ? *reinterpret_cast<const float4*>(W + (size_t)o * ldw + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const float4 x = *reinterpret_cast<const float4*>(v + k);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] += (w[u].x * x.x + w[u].y * x.y) + (w[u].z * x.z + w[u].w * x.w);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = wave_sum(acc[u]);
}

// TimestepEmbedder in fp32: out[b] = silu(W2 silu(W0 [cos | sin](t_b f) + b0) + b2) (the input of every adaLN projection), t_emb optional
__global__ __launch_bounds__(1024) void timestep_embed_f32_kernel(const float* __restrict__ t, int F, float neg_log_period, const float* __restrict__ W0,
                                                                  const float* __restrict__ b0, const float* __restrict__ W2, const float* __restrict__ b2,
                                                                  int C, float* __restrict__ out, float* __restrict__ t_emb) {
    __shared__ __attribute__((aligned(16))) float sA[1024], sB[1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
    const int half = F / 2;
    const float tv = t[b];
    {
        float v = 0.f;
        if (tid < F) {
            const int j = tid < half ? tid : tid - half;
            const float a = tv * expf(neg_log_period * (float)j / (float)half);
            v = tid < half ? cosf(a) : sinf(a);
        }
        sA[tid] = v;
        sB[tid] = 0.f;
    }
    __syncthreads();
    float acc[8], acc2[8];
    for (int o0 = wave; o0 < C; o0 += 256) {               // two independent groups of 8 rows per trip: 16 rows' loads in flight per wave
        dot8_f32(W0, F, o0, 16, C, F, sA, lane, acc);
        dot8_f32(W0, F, o0 + 128, 16, C, F, sA, lane, acc2);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int o = o0 + 16 * u, o2 = o + 128;
            if (lane == 0 && o < C) { const float v = acc[u] + (b0 ? b0[o] : 0.f); sB[o] = v / (1.0f + __expf(-v)); }
            if (lane == 0 && o2 < C) { const float v = acc2[u] + (b0 ? b0[o2] : 0.f); sB[o2] = v / (1.0f + __expf(-v)); }
        }
    }
    __syncthreads();
    const int per = (C + gridDim.y - 1) / gridDim.y, lo = blockIdx.y * per, hi = lo + per < C ? lo + per : C;
    for (int o0 = lo + wave; o0 < hi; o0 += 128) {
        dot8_f32(W2, C, o0, 16, hi, C, sB, lane, acc);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int o = o0 + 16 * u;
            if (lane == 0 && o < hi) {
                const float v = acc[u] + (b2 ? b2[o] : 0.f);
                if (t_emb != nullptr) t_emb[(size_t)b * C + o] = v;
                out[(size_t)b * C + o] = v / (1.0f + __expf(-v));
            }
        }
    }
}

// every adaLN projection of the step as one fp32 GEMV per sample: out[b][n] = W[n] . s[b] + bias[n]; a wave takes 8 rows at a time
// (HBM-bound on the fp32 weights: 115 MB at the full config)
__global__ __launch_bounds__(256) void modulation_f32_kernel(const float* __restrict__ s, int B, int C, const float* __restrict__ W, const float* __restrict__ bias,
                                                             int N, float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float sS[1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rows_per_block = 32;                       // 4 waves x 8 rows
    for (int b = 0; b < B; ++b) {
        __syncthreads();
        for (int k = tid; k < 1024; k += 256) sS[k] = k < C ? s[(size_t)b * C + k] : 0.f;
        __syncthreads();
        for (int base = blockIdx.x * rows_per_block; base < N; base += gridDim.x * rows_per_block) {
            float acc[8];
            dot8_f32(W, C, base + wave, 4, N, C, sS, lane, acc);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int n = base + wave + 4 * u;
                if (lane == 0 && n < N) out[(size_t)b * N + n] = acc[u] + (bias ? bias[n] : 0.f);
            }
        }
    }
}

// input_layer in fp32 on top of the (broadcast) position embedding: out[row] = pos[(row / rpg) * period + (row % rpg) % period] + W x[row] + b.
// 16 rows per workgroup, thread = output columns tid and tid + 256 with their weights in registers (W^T [Cin][C], transposed once by the
// caller: coalesced loads); the input rows in LDS.  Cin <= 24, C <= 512.
constexpr int IN_ROWS = 16;
__global__ __launch_bounds__(256) void input_layer_f32_kernel(const float* __restrict__ x, int M, int Cin, const float* __restrict__ Wt, const float* __restrict__ bias,
                                                              const float* __restrict__ pos, int period, int rpg, int C, float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float sX[IN_ROWS * 24];
    const int tid = threadIdx.x;
    const int r0 = blockIdx.x * IN_ROWS;
    for (int i = tid; i < IN_ROWS * 24; i += 256) {
        const int r = i / 24, k = i - r * 24;
        sX[i] = (k < Cin && r0 + r < M) ? x[(size_t)(r0 + r) * Cin + k] : 0.f;
    }
    const int c0 = tid, c1 = tid + 256;
    const bool ok0 = c0 < C, ok1 = c1 < C;
    float w0[24], w1[24];                                  // this thread's two weight columns (coalesced across the threads: W^T rows)
#pragma unroll
    for (int k = 0; k < 24; ++k) {
        w0[k] = (k < Cin && ok0) ? Wt[k * C + c0] : 0.f;
        w1[k] = (k < Cin && ok1) ? Wt[k * C + c1] : 0.f;
    }
    const float b0 = (bias && ok0) ? bias[c0] : 0.f, b1 = (bias && ok1) ? bias[c1] : 0.f;
    __syncthreads();
#pragma unroll 4
    for (int r = 0; r < IN_ROWS; ++r) {
        const int row = r0 + r;
        if (row >= M) break;
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int k4 = 0; k4 < 6; ++k4) {
            const float4 xv = *reinterpret_cast<const float4*>(&sX[r * 24 + 4 * k4]);     // broadcast read
            a0 = fmaf(xv.x, w0[4 * k4], a0); a0 = fmaf(xv.y, w0[4 * k4 + 1], a0); a0 = fmaf(xv.z, w0[4 * k4 + 2], a0); a0 = fmaf(xv.w, w0[4 * k4 + 3], a0);
            a1 = fmaf(xv.x, w1[4 * k4], a1); a1 = fmaf(xv.y, w1[4 * k4 + 1], a1); a1 = fmaf(xv.z, w1[4 * k4 + 2], a1); a1 = fmaf(xv.w, w1[4 * k4 + 3], a1);
        }
        const size_t pr = pos != nullptr ? (size_t)(row / rpg) * period + (row % rpg) % period : 0;
        if (ok0) out[(size_t)row * C + c0] = (pos != nullptr ? pos[pr * C + c0] : 0.f) + (a0 + b0);
        if (ok1) out[(size_t)row * C + c1] = (pos != nullptr ? pos[pr * C + c1] : 0.f) + (a1 + b1);
    }
}

// FinalLayer in fp32 straight from the stream: out[row] = W (LayerNorm(x[row]) * (1 + scale[g]) + shift[g]) + b.  One wave per row (C <= 512, a
// multiple of 4: up to two float4 per lane), NP = 16 or 32 output columns (W rows beyond Cout are zero in LDS: no branches in the row loop).
template <int NP>
__global__ __launch_bounds__(256) void final_layer_f32_kernel(const float* __restrict__ x, int M, int C, float eps, const float* __restrict__ shift,
                                                              const float* __restrict__ scale, int mod_ld, int rpg, const float* __restrict__ W,
                                                              const float* __restrict__ bias, int Cout, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float sW[];        // NP * C floats (32 KiB at 16 x 512: 4-5 workgroups per CU)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {   // W -> LDS, all of a thread's loads in flight together (one element per trip is a chain of memory latencies)
        const int n4 = Cout * C / 4, n4p = NP * C / 4;
        const float4* W4 = reinterpret_cast<const float4*>(W);
        float4* s4 = reinterpret_cast<float4*>(sW);
        for (int base = 0; base < n4p; base += 256 * 8) {
            float4 tmp[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int i = base + u * 256 + tid; tmp[u] = i < n4 ? W4[i] : make_float4(0.f, 0.f, 0.f, 0.f); }
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int i = base + u * 256 + tid; if (i < n4p) s4[i] = tmp[u]; }
        }
    }
    __syncthreads();
    const bool ok0 = lane * 4 < C, ok1 = (64 + lane) * 4 < C;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const int stride = gridDim.x * 4;
    const float4* sW4 = reinterpret_cast<const float4*>(sW);
    const int i0 = ok0 ? lane : 0, i1 = ok1 ? 64 + lane : 0;           // LDS / global chunk indices (a masked chunk reads chunk 0 and is zeroed)
    float4 nxt0 = z4, nxt1 = z4;
    {
        const int row = blockIdx.x * 4 + wave;
        if (row < M) { const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * C); nxt0 = xr[i0]; nxt1 = xr[i1]; }
    }
    for (int row = blockIdx.x * 4 + wave; row < M; row += stride) {
        float4 v0 = ok0 ? nxt0 : z4, v1 = ok1 ? nxt1 : z4;
        if (row + stride < M) {                              // the next row of this wave is requested before this one is reduced
            const float4* xn = reinterpret_cast<const float4*>(x + (size_t)(row + stride) * C);
            nxt0 = xn[i0]; nxt1 = xn[i1];
        }
        const float mean = wave_sum((v0.x + v0.y) + (v0.z + v0.w) + (v1.x + v1.y) + (v1.z + v1.w)) / (float)C;
        if (ok0) { v0.x -= mean; v0.y -= mean; v0.z -= mean; v0.w -= mean; }
        if (ok1) { v1.x -= mean; v1.y -= mean; v1.z -= mean; v1.w -= mean; }
        const float rstd = rsqrtf(wave_sum((v0.x * v0.x + v0.y * v0.y) + (v0.z * v0.z + v0.w * v0.w) + (v1.x * v1.x + v1.y * v1.y) + (v1.z * v1.z + v1.w * v1.w)) / (float)C + eps);
        float4 sc0 = z4, sh0 = z4, sc1 = z4, sh1 = z4;
        if (scale != nullptr) {
            const size_t mo = (size_t)(rpg > 0 ? row / rpg : 0) * mod_ld;
            sc0 = *reinterpret_cast<const float4*>(scale + mo + i0 * 4); sh0 = *reinterpret_cast<const float4*>(shift + mo + i0 * 4);
            sc1 = *reinterpret_cast<const float4*>(scale + mo + i1 * 4); sh1 = *reinterpret_cast<const float4*>(shift + mo + i1 * 4);
        }
        float y[8] = {v0.x * rstd * (1.0f + sc0.x) + sh0.x, v0.y * rstd * (1.0f + sc0.y) + sh0.y, v0.z * rstd * (1.0f + sc0.z) + sh0.z, v0.w * rstd * (1.0f + sc0.w) + sh0.w,
                      v1.x * rstd * (1.0f + sc1.x) + sh1.x, v1.y * rstd * (1.0f + sc1.y) + sh1.y, v1.z * rstd * (1.0f + sc1.z) + sh1.z, v1.w * rstd * (1.0f + sc1.w) + sh1.w};
        if (!ok0) { y[0] = y[1] = y[2] = y[3] = 0.f; }
        if (!ok1) { y[4] = y[5] = y[6] = y[7] = 0.f; }
        // NP dot products: per-lane partials, then ONE butterfly over all of them: after step d a lane keeps the half of the remaining
        // outputs selected by its bit d, so the partials shrink NP -> NP/2 -> ... -> 1 while the lanes fold: lane l ends with the complete
        // sum of output (l & (NP - 1)) -- NP - 1 exchanges per lane instead of 6 per output
        float part[NP];
#pragma unroll
        for (int o = 0; o < NP; ++o) {
            const float4 w0 = sW4[o * (C / 4) + i0], w1 = sW4[o * (C / 4) + i1];
            part[o] = (y[0] * w0.x + y[1] * w0.y) + (y[2] * w0.z + y[3] * w0.w) + (y[4] * w1.x + y[5] * w1.y) + (y[6] * w1.z + y[7] * w1.w);
        }
        constexpr int STEPS = NP == 32 ? 5 : 4;
#pragma unroll
        for (int d = 0; d < STEPS; ++d) {
            const bool up = (lane >> d) & 1;
#pragma unroll
            for (int i = 0; i < (NP >> (d + 1)); ++i) {
                const float keep = up ? part[2 * i + 1] : part[2 * i], give = up ? part[2 * i] : part[2 * i + 1];
                part[i] = keep + __shfl_xor(give, 1 << d, 64);
            }
        }
        float tot = part[0];
#pragma unroll
        for (int d = STEPS; d < 6; ++d) tot += __shfl_xor(tot, 1 << d, 64);
        const int o = lane & (NP - 1);
        if (lane < NP && o < Cout) out[(size_t)row * Cout + o] = tot + (bias ? bias[o] : 0.f);
    }
}

}  // namespace

extern "C" int gvf_layernorm_modulate_bf16(const float* x, void* out_bf16, int rows, int C, float eps, const float* ln_w,
                                           const float* ln_b, const float* shift, const float* scale, int mod_ld,
                                           int rows_per_group, void* stream_) {
    return gvf_layernorm_modulate(GVF_DT_BF16, x, out_bf16, rows, C, eps, ln_w, ln_b, shift, scale, mod_ld, rows_per_group, stream_);
}

extern "C" int gvf_layernorm_modulate(int dtype, const float* x, void* out_bf16, int rows, int C, float eps, const float* ln_w,
                                      const float* ln_b, const float* shift, const float* scale, int mod_ld,
                                      int rows_per_group, void* stream_) {
    if (dtype != GVF_DT_BF16 && dtype != GVF_DT_F16) return GVF_EINVAL;
    if (rows < 0 || C <= 0) return GVF_EINVAL;
    if (rows == 0) return GVF_OK;
    if (!x || !out_bf16 || ((ln_w == nullptr) != (ln_b == nullptr)) || ((shift == nullptr) != (scale == nullptr)))
        return GVF_EINVAL;
    if (scale != nullptr && (rows_per_group <= 0 || ((C % 256) == 0 && (mod_ld % 4) != 0))) return GVF_EINVAL;
    if ((((uintptr_t)x) & 15) || (((uintptr_t)out_bf16) & 7)) return GVF_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    (void)hipGetLastError();
    const int rpg = rows_per_group > 0 ? rows_per_group : 1;
    const dim3 grid((rows + 3) / 4), block(256);
    unsigned short* o = (unsigned short*)out_bf16;
    GVF_LP_DISPATCH(dtype,
        if ((C % 256) != 0 || C > 1024) {
            hipLaunchKernelGGL(ln_mod_generic_kernel<DT>, grid, block, 0, stream, x, o, rows, C, eps, ln_w, ln_b, shift, scale, mod_ld, rpg);
        } else {
            switch (C / 256) {
                case 1: hipLaunchKernelGGL((ln_mod_kernel<1, DT>), grid, block, 0, stream, x, o, rows, C, eps, ln_w, ln_b, shift, scale, mod_ld, rpg); break;
                case 2: hipLaunchKernelGGL((ln_mod_kernel<2, DT>), grid, block, 0, stream, x, o, rows, C, eps, ln_w, ln_b, shift, scale, mod_ld, rpg); break;
                case 3: hipLaunchKernelGGL((ln_mod_kernel<3, DT>), grid, block, 0, stream, x, o, rows, C, eps, ln_w, ln_b, shift, scale, mod_ld, rpg); break;
                default: hipLaunchKernelGGL((ln_mod_kernel<4, DT>), grid, block, 0, stream, x, o, rows, C, eps, ln_w, ln_b, shift, scale, mod_ld, rpg); break;
            }
        });
    GVF_CHECK_LAUNCH();
    return GVF_OK;
}

extern "C" int gvf_cast_pad_bf16(const float* src, int ld_src, void* dst, int ld_dst, int64_t rows, int cols, int act,
                                 void* stream_) {
    return gvf_cast_pad(GVF_DT_BF16, src, ld_src, dst, ld_dst, rows, cols, act, stream_);
}

extern "C" int gvf_cast_pad(int dtype, const float* src, int ld_src, void* dst, int ld_dst, int64_t rows, int cols, int act,
                            void* stream_) {
    if (dtype != GVF_DT_BF16 && dtype != GVF_DT_F16) return GVF_EINVAL;
    if (rows < 0 || cols <= 0 || ld_src < cols || ld_dst < cols || (act != 0 && act != 1)) return GVF_EINVAL;
    if (rows == 0) return GVF_OK;
    if (!src || !dst) return GVF_EINVAL;
    (void)hipGetLastError();
    long long total = rows * (long long)ld_dst;
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    GVF_LP_DISPATCH(dtype, hipLaunchKernelGGL(cast_pad_kernel<DT>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_, src, ld_src,
                                              (unsigned short*)dst, ld_dst, (long long)rows, cols, act));
    GVF_CHECK_LAUNCH();
    return GVF_OK;
}

extern "C" int gvf_split3_bf16(const float* src, int64_t ld_src, void* dst, int64_t rows, int cols, int mode, void* stream_) {
    if (rows < 0 || cols <= 0 || ld_src < cols || (mode != 0 && mode != 1)) return GVF_EINVAL;
    if (rows == 0) return GVF_OK;
    if (!src || !dst || (((uintptr_t)dst) & 7)) return GVF_EINVAL;
    const int Kp = (cols + 63) / 64 * 64;
    (void)hipGetLastError();
    long long blocks = (rows * (long long)(Kp / 4) + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(split3_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_, src, (long long)ld_src, (unsigned short*)dst,
                       (long long)rows, cols, Kp, mode);
    GVF_CHECK_LAUNCH();
    return GVF_OK;
}

extern "C" int gvf_dit_timestep_embed_bf16(const float* t, int B, int freq_dim, float max_period, const void* w0_bf16, int ldw0, const float* b0,
                                           const void* w2_bf16, int ldw2, const float* b2, int C, void* out_bf16, int ld_out, float* t_emb,
                                           void* stream_) {
    if (B < 0 || freq_dim <= 0 || (freq_dim & 1) || freq_dim > 1024 || C <= 0 || C > 1024 || ldw0 < ((freq_dim + 3) & ~3) || ldw2 < ((C + 3) & ~3) || (ldw0 & 3) || (ldw2 & 3) || ld_out < C || !(max_period > 1.0f))
        return GVF_EINVAL;
    if (B == 0) return GVF_OK;
    if (!t || !w0_bf16 || !w2_bf16 || !out_bf16 || (((uintptr_t)w0_bf16) & 7) || (((uintptr_t)w2_bf16) & 7)) return GVF_EINVAL;
    (void)hipGetLastError();
    hipLaunchKernelGGL(timestep_embed_kernel, dim3(B, 16), dim3(1024), 0, (hipStream_t)stream_, t, freq_dim, (float)(-log((double)max_period)), (const unsigned short*)w0_bf16, ldw0,
                       b0, (const unsigned short*)w2_bf16, ldw2, b2, C, (unsigned short*)out_bf16, ld_out, t_emb);
    GVF_CHECK_LAUNCH();
    return GVF_OK;
}

extern "C" int gvf_dit_timestep_embed_f32(const float* t, int B, int freq_dim, float max_period, const float* w0, const float* b0, const float* w2,
                                          const float* b2, int C, float* out_silu, float* t_emb, void* stream_) {
    if (B < 0 || freq_dim <= 0 || (freq_dim & 3) || freq_dim > 1024 || C <= 0 || (C & 3) || C > 1024 || !(max_period > 1.0f)) return GVF_EINVAL;
    if (B == 0) return GVF_OK;
    if (!t || !w0 || !w2 || !out_silu || (((uintptr_t)w0) & 15) || (((uintptr_t)w2) & 15)) return GVF_EINVAL;
    (void)hipGetLastError();
    hipLaunchKernelGGL(timestep_embed_f32_kernel, dim3(B, 32), dim3(1024), 0, (hipStream_t)stream_, t, freq_dim, (float)(-log((double)max_period)), w0, b0, w2, b2,
                       C, out_silu, t_emb);
    GVF_CHECK_LAUNCH();
    return GVF_OK;
}

extern "C" int gvf_dit_modulation_f32(const float* s, int B, int C, const float* w, const float* bias, int N, float* out, void* stream_) {
    if (B < 0 || C <= 0 || (C & 3) || C > 1024 || N <= 0) return GVF_EINVAL;
    if (B == 0) return GVF_OK;
    if (!s || !w || !out || (((uintptr_t)w) & 15)) return GVF_EINVAL;
    (void)hipGetLastError();
    int blocks = (N + 31) / 32;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(modulation_f32_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, s, B, C, w, bias, N, out);
    GVF_CHECK_LAUNCH();
    return GVF_OK;
}

extern "C" int gvf_dit_input_layer_f32(const float* x, int M, int Cin, const float* w, const float* bias, const float* pos, int pos_period,
                                       int rows_per_group, int C, float* out, void* stream_) {
    if (M < 0 || Cin <= 0 || Cin > 24 || C <= 0 || C > 512) return GVF_EINVAL;
    if (M == 0) return GVF_OK;
    if (!x || !w || !out) return GVF_EINVAL;
    if (pos != nullptr && (pos_period <= 0 || rows_per_group <= 0 || rows_per_group % pos_period != 0)) return GVF_EINVAL;
    (void)hipGetLastError();
    hipLaunchKernelGGL(input_layer_f32_kernel, dim3((M + IN_ROWS - 1) / IN_ROWS), dim3(256), 0, (hipStream_t)stream_, x, M, Cin, w, bias, pos, pos_period,
                       rows_per_group > 0 ? rows_per_group : 1, C, out);
    GVF_CHECK_LAUNCH();
    return GVF_OK;
}

extern "C" int gvf_dit_final_layer_f32(const float* x, int M, int C, float eps, const float* shift, const float* scale, int mod_ld, int rows_per_group,
                                       const float* w, const float* bias, int Cout, float* out, void* stream_) {
    if (M < 0 || C <= 0 || C > 512 || (C & 3) || Cout <= 0 || Cout > 32) return GVF_EINVAL;
    if (M == 0) return GVF_OK;
    if (!x || !w || !out || ((shift == nullptr) != (scale == nullptr)) || (((uintptr_t)x) & 15) || (((uintptr_t)w) & 15)) return GVF_EINVAL;
    if (scale != nullptr && (rows_per_group <= 0 || (mod_ld & 3) || (((uintptr_t)scale) & 15) || (((uintptr_t)shift) & 15))) return GVF_EINVAL;
    (void)hipGetLastError();
    int blocks = (M + 3) / 4;
    if (blocks > 1024) blocks = 1024;                  // each workgroup copies W into LDS once: 4 resident workgroups per CU, 3 rows per wave
    if (Cout <= 16)
        hipLaunchKernelGGL(final_layer_f32_kernel<16>, dim3(blocks), dim3(256), (size_t)16 * C * sizeof(float), (hipStream_t)stream_, x, M, C, eps, shift, scale,
                           mod_ld, rows_per_group, w, bias, Cout, out);
    else
        hipLaunchKernelGGL(final_layer_f32_kernel<32>, dim3(blocks), dim3(256), (size_t)32 * C * sizeof(float), (hipStream_t)stream_, x, M, C, eps, shift, scale,
                           mod_ld, rows_per_group, w, bias, Cout, out);
    GVF_CHECK_LAUNCH();
    return GVF_OK;
}
