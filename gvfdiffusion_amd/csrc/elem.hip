// elem.hip -- memory-bound helpers of the DiT block for gfx950: fused LayerNorm + (affine | adaLN
// modulate) writing the bf16 GEMM operand, and fp32 -> bf16 cast with zero padding / SiLU.
//
// Reference: model/dit.py:168-172,246-277 (norm1..5: LayerNorm(eps 1e-6), affine only on norm3/4;
// h * (1 + scale) + shift), model/dit.py:217-225,240-242 (SiLU in front of the adaLN Linear).
// One wave per row (C <= 1024): the row is read once with 16-byte loads and kept in registers for the
// mean / variance / normalise passes (two-pass variance, as torch's LayerNorm computes it).
#include <cmath>
#include "gvf_common.h"
#include "../../include/gvf_rast.h"
#include "../../include/gvf_dit.h"

namespace {

__device__ __forceinline__ unsigned short f2bf(float f) {
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// VPL = float4 loads per lane: C = 64 * 4 * VPL
template <int VPL>
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
        o.x = (unsigned)f2bf(y[0]) | ((unsigned)f2bf(y[1]) << 16);
        o.y = (unsigned)f2bf(y[2]) | ((unsigned)f2bf(y[3]) << 16);
        *reinterpret_cast<uint2*>(out + (size_t)row * C + c0) = o;
    }
}

// any C (multiple of 4): the row is re-read from cache for each pass
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
        out[(size_t)row * C + c] = f2bf(y);
    }
}

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
        dst[i] = f2bf(v);
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
    float acc[8];
    for (int o0 = wave; o0 < C; o0 += 128) {
        dot8_f32(W0, F, o0, 16, C, F, sA, lane, acc);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int o = o0 + 16 * u;
            if (lane == 0 && o < C) { const float v = acc[u] + (b0 ? b0[o] : 0.f); sB[o] = v / (1.0f + __expf(-v)); }
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
// 16 rows per workgroup, thread = output columns tid and tid + 256; W^T and the 16 input rows in LDS.  Cin <= 24, C <= 512.
__global__ __launch_bounds__(256) void input_layer_f32_kernel(const float* __restrict__ x, int M, int Cin, const float* __restrict__ W, const float* __restrict__ bias,
                                                              const float* __restrict__ pos, int period, int rpg, int C, float* __restrict__ out) {
    __shared__ float sW[24 * 512];                        // [k][c]
    __shared__ float sX[16 * 24];
    const int tid = threadIdx.x;
    for (int i = tid; i < Cin * C; i += 256) { const int c = i / Cin, k = i - c * Cin; sW[k * C + c] = W[i]; }
    const int r0 = blockIdx.x * 16;
    for (int i = tid; i < 16 * Cin; i += 256) { const int r = i / Cin, k = i - r * Cin; sX[r * 24 + k] = r0 + r < M ? x[(size_t)(r0 + r) * Cin + k] : 0.f; }
    __syncthreads();
    const int c0 = tid, c1 = tid + 256;
    const bool ok0 = c0 < C, ok1 = c1 < C;
    float a0[16], a1[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; }
    for (int k = 0; k < Cin; ++k) {
        const float w0 = ok0 ? sW[k * C + c0] : 0.f, w1 = ok1 ? sW[k * C + c1] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float xv = sX[r * 24 + k]; a0[r] = fmaf(xv, w0, a0[r]); a1[r] = fmaf(xv, w1, a1[r]); }
    }
    const float b0 = (bias && ok0) ? bias[c0] : 0.f, b1 = (bias && ok1) ? bias[c1] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = r0 + r;
        if (row < M) {
            const size_t pr = pos != nullptr ? (size_t)(row / rpg) * period + (row % rpg) % period : 0;
            if (ok0) out[(size_t)row * C + c0] = (pos != nullptr ? pos[pr * C + c0] : 0.f) + (a0[r] + b0);
            if (ok1) out[(size_t)row * C + c1] = (pos != nullptr ? pos[pr * C + c1] : 0.f) + (a1[r] + b1);
        }
    }
}

// FinalLayer in fp32 straight from the stream: out[row] = W (LayerNorm(x[row]) * (1 + scale[g]) + shift[g]) + b.  One wave per row (C <= 512, a
// multiple of 4: up to two float4 per lane), Cout <= 32 output columns, W [Cout][C] in LDS.
__global__ __launch_bounds__(256) void final_layer_f32_kernel(const float* __restrict__ x, int M, int C, float eps, const float* __restrict__ shift,
                                                              const float* __restrict__ scale, int mod_ld, int rpg, const float* __restrict__ W,
                                                              const float* __restrict__ bias, int Cout, float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float sW[32 * 512];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < Cout * C; i += 256) sW[i] = W[i];
    __syncthreads();
    const bool ok[2] = {lane * 4 < C, (64 + lane) * 4 < C};
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
        const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * C);
        float4 v[2] = {ok[0] ? xr[lane] : z4, ok[1] ? xr[64 + lane] : z4};
        const float sum = (v[0].x + v[0].y) + (v[0].z + v[0].w) + (v[1].x + v[1].y) + (v[1].z + v[1].w);
        const float mean = wave_sum(sum) / (float)C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (ok[i]) {
                v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
                q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
            }
        }
        const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
        const int g = rpg > 0 ? row / rpg : 0;
        float y[8];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c0 = (lane + 64 * i) * 4;
            float4 sc = z4, sh = z4;
            if (scale != nullptr && ok[i]) {
                sc = *reinterpret_cast<const float4*>(scale + (size_t)g * mod_ld + c0);
                sh = *reinterpret_cast<const float4*>(shift + (size_t)g * mod_ld + c0);
            }
            y[4 * i + 0] = ok[i] ? v[i].x * rstd * (1.0f + sc.x) + sh.x : 0.f; y[4 * i + 1] = ok[i] ? v[i].y * rstd * (1.0f + sc.y) + sh.y : 0.f;
            y[4 * i + 2] = ok[i] ? v[i].z * rstd * (1.0f + sc.z) + sh.z : 0.f; y[4 * i + 3] = ok[i] ? v[i].w * rstd * (1.0f + sc.w) + sh.w : 0.f;
        }
        for (int o = 0; o < Cout; ++o) {
            const float4 w0 = ok[0] ? *reinterpret_cast<const float4*>(sW + o * C + lane * 4) : z4;
            const float4 w1 = ok[1] ? *reinterpret_cast<const float4*>(sW + o * C + (64 + lane) * 4) : z4;
            float acc = (y[0] * w0.x + y[1] * w0.y) + (y[2] * w0.z + y[3] * w0.w) + (y[4] * w1.x + y[5] * w1.y) + (y[6] * w1.z + y[7] * w1.w);
            acc = wave_sum(acc);
            if (lane == 0) out[(size_t)row * Cout + o] = acc + (bias ? bias[o] : 0.f);
        }
    }
}

}  // namespace

extern "C" int gvf_layernorm_modulate_bf16(const float* x, void* out_bf16, int rows, int C, float eps, const float* ln_w,
                                           const float* ln_b, const float* shift, const float* scale, int mod_ld,
                                           int rows_per_group, void* stream_) {
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
    if ((C % 256) != 0 || C > 1024) {
        hipLaunchKernelGGL(ln_mod_generic_kernel, grid, block, 0, stream, x, o, rows, C, eps, ln_w, ln_b, shift, scale, mod_ld, rpg);
        GVF_CHECK_LAUNCH();
        return GVF_OK;
    }
    switch (C / 256) {
        case 1: hipLaunchKernelGGL(ln_mod_kernel<1>, grid, block, 0, stream, x, o, rows, C, eps, ln_w, ln_b, shift, scale, mod_ld, rpg); break;
        case 2: hipLaunchKernelGGL(ln_mod_kernel<2>, grid, block, 0, stream, x, o, rows, C, eps, ln_w, ln_b, shift, scale, mod_ld, rpg); break;
        case 3: hipLaunchKernelGGL(ln_mod_kernel<3>, grid, block, 0, stream, x, o, rows, C, eps, ln_w, ln_b, shift, scale, mod_ld, rpg); break;
        default: hipLaunchKernelGGL(ln_mod_kernel<4>, grid, block, 0, stream, x, o, rows, C, eps, ln_w, ln_b, shift, scale, mod_ld, rpg); break;
    }
    GVF_CHECK_LAUNCH();
    return GVF_OK;
}

extern "C" int gvf_cast_pad_bf16(const float* src, int ld_src, void* dst, int ld_dst, int64_t rows, int cols, int act,
                                 void* stream_) {
    if (rows < 0 || cols <= 0 || ld_src < cols || ld_dst < cols || (act != 0 && act != 1)) return GVF_EINVAL;
    if (rows == 0) return GVF_OK;
    if (!src || !dst) return GVF_EINVAL;
    (void)hipGetLastError();
    long long total = rows * (long long)ld_dst;
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(cast_pad_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_, src, ld_src,
                       (unsigned short*)dst, ld_dst, (long long)rows, cols, act);
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
    hipLaunchKernelGGL(timestep_embed_f32_kernel, dim3(B, 16), dim3(1024), 0, (hipStream_t)stream_, t, freq_dim, (float)(-log((double)max_period)), w0, b0, w2, b2,
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
    hipLaunchKernelGGL(input_layer_f32_kernel, dim3((M + 15) / 16), dim3(256), 0, (hipStream_t)stream_, x, M, Cin, w, bias, pos, pos_period,
                       rows_per_group > 0 ? rows_per_group : 1, C, out);
    GVF_CHECK_LAUNCH();
    return GVF_OK;
}

extern "C" int gvf_dit_final_layer_f32(const float* x, int M, int C, float eps, const float* shift, const float* scale, int mod_ld, int rows_per_group,
                                       const float* w, const float* bias, int Cout, float* out, void* stream_) {
    if (M < 0 || C <= 0 || C > 512 || (C & 3) || Cout <= 0 || Cout > 32) return GVF_EINVAL;
    if (M == 0) return GVF_OK;
    if (!x || !w || !out || ((shift == nullptr) != (scale == nullptr)) || (((uintptr_t)x) & 15)) return GVF_EINVAL;
    if (scale != nullptr && (rows_per_group <= 0 || (mod_ld & 3) || (((uintptr_t)scale) & 15) || (((uintptr_t)shift) & 15))) return GVF_EINVAL;
    (void)hipGetLastError();
    int blocks = (M + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(final_layer_f32_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, x, M, C, eps, shift, scale, mod_ld, rows_per_group, w, bias, Cout,
                       out);
    GVF_CHECK_LAUNCH();
    return GVF_OK;
}
