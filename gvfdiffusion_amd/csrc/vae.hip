// vae.hip -- the two memory-bound kernels the motion-VAE decoder adds to the DiT kernel set (gfx950).
//
// Reference: model/autoencoder.py:90-93 (GEGLU), :250-301 (PointEmbed), :392-394 (gs_embedding /
// position_encoding), :561 (their sum) and :80-81 (the PreNorm LayerNorm in front of decoder_cross_attn).
// geglu: 16-byte loads of both halves, erf GELU in fp32, 16-byte store.
// query_embed: one wave per Gaussian; the 14 -> C Linear is done in fp32 from an LDS copy of W (C x qdim),
// each lane owning channels lane, lane+64, ...; three wave-level LayerNorms; the result is the bf16 operand
// of the to_q GEMM.  It depends on the static Gaussians only, so the host runs it once for all T frames.
#include "gvf_common.h"
#include "gvf_lp.h"
#include "../../include/gvf_rast.h"
#include "../../include/gvf_vae.h"
#include "../../include/gvf_dit.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

template <int DT>
__global__ __launch_bounds__(256) void geglu_kernel(const unsigned short* __restrict__ in, int ld_in,
                                                    unsigned short* __restrict__ out, int ld_out, long long rows, int F) {
    const int f8 = F >> 3;
    const long long total = rows * (long long)f8;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / f8;
        const int c = (int)(i - r * f8) << 3;
        const uint4 a = *reinterpret_cast<const uint4*>(in + r * ld_in + c);
        const uint4 g = *reinterpret_cast<const uint4*>(in + r * ld_in + F + c);
        const unsigned aw[4] = {a.x, a.y, a.z, a.w}, gw[4] = {g.x, g.y, g.z, g.w};
        unsigned ow[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float lo = GvfLp<DT>::lo(aw[k]) * gelu_erf(GvfLp<DT>::lo(gw[k]));
            const float hi = GvfLp<DT>::hi(aw[k]) * gelu_erf(GvfLp<DT>::hi(gw[k]));
            ow[k] = GvfLp<DT>::pack(lo, hi);
        }
        *reinterpret_cast<uint4*>(out + r * ld_out + c) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
    }
}

constexpr int QE_MAXI = 16;   // channels per lane: C <= 1024
constexpr int QE_MAXQ = 16;

__device__ __forceinline__ void wave_layernorm(float (&v)[QE_MAXI], int ni, int lane, int C, float eps) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < QE_MAXI; ++i)
        if (i < ni && lane + 64 * i < C) s += v[i];
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < QE_MAXI; ++i)
        if (i < ni && lane + 64 * i < C) { const float d = v[i] - mean; q += d * d; }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
    for (int i = 0; i < QE_MAXI; ++i) v[i] = (v[i] - mean) * rstd;
}

template <int DT>
__global__ __launch_bounds__(256) void query_embed_kernel(const float* __restrict__ queries, int qdim,
                                                          const float* __restrict__ W, const float* __restrict__ bias,
                                                          const float* __restrict__ omega, unsigned short* __restrict__ out,
                                                          float* __restrict__ out_embed, long long P, int C, float eps,
                                                          float eps_pre, int rows_per_block) {
    extern __shared__ float sW[];   // [C][qdim + 1] (odd-ish pitch keeps lanes on distinct banks), then bias[C], omega[C/6]
    const int pitch = qdim | 1;
    float* sB = sW + (size_t)C * pitch;
    float* sOm = sB + C;
    for (int i = threadIdx.x; i < C * qdim; i += blockDim.x) sW[(i / qdim) * pitch + (i % qdim)] = W[i];
    for (int i = threadIdx.x; i < C; i += blockDim.x) sB[i] = bias[i];
    const int E = C / 6;
    for (int i = threadIdx.x; i < E; i += blockDim.x) sOm[i] = omega[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ni = (C + 63) >> 6;
    const long long row0 = (long long)blockIdx.x * rows_per_block;
    for (int rr = wave; rr < rows_per_block; rr += 4) {
        const long long row = row0 + rr;
        if (row >= P) break;
        float qv[QE_MAXQ];
#pragma unroll
        for (int k = 0; k < QE_MAXQ; ++k) qv[k] = k < qdim ? queries[row * qdim + k] : 0.f;
        float e1[QE_MAXI], e2[QE_MAXI];
#pragma unroll
        for (int i = 0; i < QE_MAXI; ++i) {
            e1[i] = 0.f; e2[i] = 0.f;
            const int c = lane + 64 * i;
            if (i < ni && c < C) {
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < QE_MAXQ; ++k)
                    if (k < qdim) acc = fmaf(qv[k], sW[c * pitch + k], acc);
                e1[i] = acc + sB[c];
                const int axis = c / (2 * E), j = c - axis * 2 * E;
                const float p = axis == 0 ? qv[0] : (axis == 1 ? qv[1] : qv[2]);
                // PointEmbed's phases are |p omega| <= 0.5 rad (p in [-0.5, 0.5], omega <= 1): the hardware sine / cosine (v_sin_f32 / v_cos_f32 on
                // the phase in revolutions, ~1e-6 absolute) instead of libm's 25-instruction routines -- the value goes through a LayerNorm and a
                // 16-bit rounding (2^-9 relative) next
                e2[i] = j < E ? __sinf(p * sOm[j]) : __cosf(p * sOm[j - E]);
            }
        }
        wave_layernorm(e1, ni, lane, C, eps);
        wave_layernorm(e2, ni, lane, C, eps);
#pragma unroll
        for (int i = 0; i < QE_MAXI; ++i) e1[i] += e2[i];
        if (out_embed != nullptr) {                      // the embedding itself (the encoder's residual stream starts from it)
#pragma unroll
            for (int i = 0; i < QE_MAXI; ++i) {
                const int c = lane + 64 * i;
                if (i < ni && c < C) out_embed[row * C + c] = e1[i];
            }
        }
        wave_layernorm(e1, ni, lane, C, eps_pre);
#pragma unroll
        for (int i = 0; i < QE_MAXI; ++i) {
            const int c = lane + 64 * i;
            if (i < ni && c < C) out[row * C + c] = GvfLp<DT>::to16(e1[i]);
        }
    }
}

// The same embedding with the Linear's weights in REGISTERS: lane l owns channels l + 64 i, i < NI, and keeps their QD-wide weight rows (NI * QD
// registers: 168 for the released 768 x 14) across the Gaussians of its wave -- the LDS version pays a ds_read_b32 in front of every one of the
// NI * QD fused multiply-adds of a Gaussian (1.02 ms for 262 144 Gaussians; the arithmetic alone is a quarter of that).  C = 64 NI, qdim = QD only.
template <int DT, int NI, int QD>
__global__ __launch_bounds__(256) void query_embed_reg_kernel(const float* __restrict__ queries, const float* __restrict__ W, const float* __restrict__ bias,
                                                              const float* __restrict__ omega, unsigned short* __restrict__ out,
                                                              float* __restrict__ out_embed, long long P, float eps, float eps_pre, int rows_per_wave) {
    constexpr int C = 64 * NI, E = C / 6;
    const int lane = threadIdx.x & 63;
    const long long wave_id = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    float w[NI][QD], b[NI], om[NI];
    int axis[NI];
    bool is_sin[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int c = lane + 64 * i;
#pragma unroll
        for (int k = 0; k < QD; ++k) w[i][k] = W[c * QD + k];
        b[i] = bias[c];
        axis[i] = c / (2 * E);
        const int j = c - axis[i] * 2 * E;
        is_sin[i] = j < E;
        om[i] = omega[j < E ? j : j - E];
    }
    const long long row_end = (wave_id + 1) * rows_per_wave < P ? (wave_id + 1) * rows_per_wave : P;
    for (long long row = wave_id * rows_per_wave; row < row_end; ++row) {
        float qv[QD];
#pragma unroll
        for (int k = 0; k < QD; ++k) qv[k] = queries[row * QD + k];
        float e1[QE_MAXI], e2[QE_MAXI];
#pragma unroll
        for (int i = 0; i < QE_MAXI; ++i) {
            e1[i] = 0.f; e2[i] = 0.f;
            if (i < NI) {
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < QD; ++k) acc = fmaf(qv[k], w[i < NI ? i : 0][k], acc);
                e1[i] = acc + b[i < NI ? i : 0];
                const int ax = axis[i < NI ? i : 0];
                const float p = ax == 0 ? qv[0] : (ax == 1 ? qv[1] : qv[2]);
                const float ph = p * om[i < NI ? i : 0];
                e2[i] = is_sin[i < NI ? i : 0] ? __sinf(ph) : __cosf(ph);
            }
        }
        wave_layernorm(e1, NI, lane, C, eps);
        wave_layernorm(e2, NI, lane, C, eps);
#pragma unroll
        for (int i = 0; i < QE_MAXI; ++i) e1[i] += e2[i];
        if (out_embed != nullptr) {
#pragma unroll
            for (int i = 0; i < NI; ++i) out_embed[row * C + lane + 64 * i] = e1[i];
        }
        wave_layernorm(e1, NI, lane, C, eps_pre);
#pragma unroll
        for (int i = 0; i < NI; ++i) out[row * C + lane + 64 * i] = GvfLp<DT>::to16(e1[i]);
    }
}

}  // namespace

extern "C" int gvf_geglu_bf16(const void* in_bf16, int ld_in, void* out_bf16, int ld_out, int64_t rows, int F, void* stream_) {
    return gvf_geglu(GVF_DT_BF16, in_bf16, ld_in, out_bf16, ld_out, rows, F, stream_);
}

extern "C" int gvf_geglu(int dtype, const void* in_bf16, int ld_in, void* out_bf16, int ld_out, int64_t rows, int F, void* stream_) {
    if (dtype != GVF_DT_BF16 && dtype != GVF_DT_F16) return GVF_EINVAL;
    if (rows < 0 || F <= 0 || (F & 7) || (ld_in & 7) || (ld_out & 7) || ld_in < 2 * F || ld_out < F) return GVF_EINVAL;
    if (rows == 0) return GVF_OK;
    if (!in_bf16 || !out_bf16 || (((uintptr_t)in_bf16) & 15) || (((uintptr_t)out_bf16) & 15)) return GVF_EINVAL;
    (void)hipGetLastError();
    long long total = rows * (long long)(F >> 3);
    long long blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    GVF_LP_DISPATCH(dtype, hipLaunchKernelGGL(geglu_kernel<DT>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_,
                                              (const unsigned short*)in_bf16, ld_in, (unsigned short*)out_bf16, ld_out, (long long)rows, F));
    GVF_CHECK_LAUNCH();
    return GVF_OK;
}

extern "C" int gvf_vae_query_embed_bf16(const float* queries, int qdim, const float* W, const float* bias, const float* omega,
                                        void* out_bf16, int64_t P, int C, float eps_embed, float eps_prenorm, void* stream_) {
    return gvf_vae_embed(GVF_DT_BF16, queries, qdim, W, bias, omega, out_bf16, nullptr, P, C, eps_embed, eps_prenorm, stream_);
}

extern "C" int gvf_vae_embed_bf16_f32(const float* queries, int qdim, const float* W, const float* bias, const float* omega,
                                      void* out_bf16, float* out_embed_f32, int64_t P, int C, float eps_embed, float eps_prenorm,
                                      void* stream_) {
    return gvf_vae_embed(GVF_DT_BF16, queries, qdim, W, bias, omega, out_bf16, out_embed_f32, P, C, eps_embed, eps_prenorm, stream_);
}

extern "C" int gvf_vae_embed(int dtype, const float* queries, int qdim, const float* W, const float* bias, const float* omega,
                             void* out_bf16, float* out_embed_f32, int64_t P, int C, float eps_embed, float eps_prenorm,
                             void* stream_) {
    if (dtype != GVF_DT_BF16 && dtype != GVF_DT_F16) return GVF_EINVAL;
    if (P < 0 || qdim < 3 || qdim > QE_MAXQ || C <= 0 || C > 64 * QE_MAXI || (C % 6) != 0) return GVF_EINVAL;
    if (P == 0) return GVF_OK;
    if (!queries || !W || !bias || !omega || !out_bf16) return GVF_EINVAL;
    (void)hipGetLastError();
    static const int reg_mode = [] { const char* e = getenv("GVF_VAE_EMBED_REG"); return e == nullptr ? 1 : atoi(e); }();
    if (reg_mode != 0 && C == 768 && qdim == 14) {           // the released width: weights in registers (query_embed_reg_kernel)
        const int rows_per_wave = 64;
        const long long waves = (P + rows_per_wave - 1) / rows_per_wave, blocks_r = (waves + 3) / 4;
        GVF_LP_DISPATCH(dtype, hipLaunchKernelGGL((query_embed_reg_kernel<DT, 12, 14>), dim3((unsigned)blocks_r), dim3(256), 0, (hipStream_t)stream_, queries, W,
                                                  bias, omega, (unsigned short*)out_bf16, out_embed_f32, (long long)P, eps_embed, eps_prenorm, rows_per_wave));
        GVF_CHECK_LAUNCH();
        return GVF_OK;
    }
    const int rows_per_block = 64;   // amortises the W -> LDS copy (C*qdim floats) over 64 Gaussians
    const size_t smem = ((size_t)C * (qdim | 1) + C + C / 6) * sizeof(float);
    const long long blocks = (P + rows_per_block - 1) / rows_per_block;
    GVF_LP_DISPATCH(dtype, hipLaunchKernelGGL(query_embed_kernel<DT>, dim3((unsigned)blocks), dim3(256), smem, (hipStream_t)stream_, queries, qdim, W,
                                              bias, omega, (unsigned short*)out_bf16, out_embed_f32, (long long)P, C, eps_embed, eps_prenorm,
                                              rows_per_block));
    GVF_CHECK_LAUNCH();
    return GVF_OK;
}
