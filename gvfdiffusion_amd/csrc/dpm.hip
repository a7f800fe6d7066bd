// dpm.hip -- the tensor arithmetic of the DPM-Solver steps as single launches, for gfx950.
//
// Reference: model/dpmsolver.py -- data_prediction_fn (:450-461: x0 = (x - sigma_t noise) / alpha_t), dpm_solver_first_update (:564-609),
// singlestep_dpm_solver_second_update (:611-690), multistep_dpm_solver_second_update (:813-869), and the error norm of dpm_solver_adaptive
// (:1013-1019: delta = max(atol, rtol max(|x_lower|, |x_prev|)), E = max over samples of sqrt(mean(((x_higher - x_lower) / delta)^2))).
// Upstream writes these as chains of tensor operations (a step of the adaptive order-2 solver: ~30 launches of 4 us each on a 0.8 MB
// latent, and as many host dispatches -- 0.17-0.23 ms per model evaluation next to a 4.4 ms forward, profiles/r06_adaptive_fused_steps.txt);
// the schedule coefficients are host floats (gvfdiffusion_amd/model/dpmsolver.py keeps the times on the host), so every state update is one
// elementwise launch with scalar arguments.  Every product, sum and quotient is rounded to fp32 on its own, in the order the reference's
// expressions evaluate (no fused multiply-add: the file is compiled with -ffp-contract=off -- HIP's __fmul_rn / __fadd_rn are plain operators
// and contract like them --, the division is the correctly rounded one).
#include <cmath>
#include <cstdint>
#include "gvf_common.h"
#include "../../include/gvf_rast.h"
#include "../../include/gvf_dit.h"

namespace {

constexpr int DPM_THREADS = 256;
constexpr int DPM_PER_BLOCK = DPM_THREADS * 4;      // elements per workgroup

__device__ __forceinline__ float4 ld4(const float* p, long long i, long long n, bool vec) {
    if (vec) return *reinterpret_cast<const float4*>(p + i);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n) v.x = p[i];
    if (i + 1 < n) v.y = p[i + 1];
    if (i + 2 < n) v.z = p[i + 2];
    if (i + 3 < n) v.w = p[i + 3];
    return v;
}
__device__ __forceinline__ void st4(float* p, long long i, long long n, bool vec, float4 v) {
    if (vec) { *reinterpret_cast<float4*>(p + i) = v; return; }
    if (i < n) p[i] = v.x;
    if (i + 1 < n) p[i + 1] = v.y;
    if (i + 2 < n) p[i + 2] = v.z;
    if (i + 3 < n) p[i + 3] = v.w;
}
__device__ __forceinline__ float comp(const float4& v, int e) { return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; }

// x0 = (x - sigma * noise) / alpha
__global__ __launch_bounds__(DPM_THREADS) void dpm_x0_kernel(const float* __restrict__ x, const float* __restrict__ noise, float sigma, float alpha,
                                                             float* __restrict__ x0, long long n, int vec) {
    const long long i = ((long long)blockIdx.x * DPM_THREADS + threadIdx.x) * 4;
    if (i >= n) return;
    const bool v4 = vec && i + 3 < n;
    const float4 a = ld4(x, i, n, v4), b = ld4(noise, i, n, v4);
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = __fdiv_rn(__fsub_rn(comp(a, e), __fmul_rn(sigma, comp(b, e))), alpha);
    st4(x0, i, n, v4, make_float4(o[0], o[1], o[2], o[3]));
}

// out = ((a x) + (b m0)) + (c m1)      (m1 null: out = (a x) + (b m0))
__global__ __launch_bounds__(DPM_THREADS) void dpm_lincomb_kernel(const float* __restrict__ x, const float* __restrict__ m0, const float* __restrict__ m1,
                                                                  float a, float b, float c, float* __restrict__ out, long long n, int vec) {
    const long long i = ((long long)blockIdx.x * DPM_THREADS + threadIdx.x) * 4;
    if (i >= n) return;
    const bool v4 = vec && i + 3 < n;
    const float4 vx = ld4(x, i, n, v4), v0 = ld4(m0, i, n, v4);
    float o[4];
    if (m1 != nullptr) {
        const float4 v1 = ld4(m1, i, n, v4);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = __fadd_rn(__fadd_rn(__fmul_rn(a, comp(vx, e)), __fmul_rn(b, comp(v0, e))), __fmul_rn(c, comp(v1, e)));
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = __fadd_rn(__fmul_rn(a, comp(vx, e)), __fmul_rn(b, comp(v0, e)));
    }
    st4(out, i, n, v4, make_float4(o[0], o[1], o[2], o[3]));
}

// The closing launch of an adaptive order-2 step (dpmsolver++, solver_type "dpmsolver"):
//   x_lower  = (a x) - (b m)                                   (first-order update over the whole step)
//   x_higher = ((a x) - (b m)) - (c (m1 - m))                  (second-order update; m1 = the model at the intermediate time)
//   delta    = max(atol, rtol max(|x_lower|, |x_prev|));  v = (x_higher - x_lower) / delta
//   partial[sample][block] = sum of v^2 over the block's elements (fixed order: lanes, then waves)
__global__ __launch_bounds__(DPM_THREADS) void dpm_second_err_kernel(const float* __restrict__ x, const float* __restrict__ m, const float* __restrict__ m1,
                                                                     const float* __restrict__ x_prev, float a, float b, float c, float atol, float rtol,
                                                                     float* __restrict__ x_lower, float* __restrict__ x_higher,
                                                                     double* __restrict__ partial, long long n_per, int vec) {
    __shared__ double wsum[DPM_THREADS / 64];
    const long long sample = blockIdx.y, base = sample * n_per;
    const long long i = ((long long)blockIdx.x * DPM_THREADS + threadIdx.x) * 4;
    double acc = 0.0;
    if (i < n_per) {
        const bool v4 = vec && i + 3 < n_per;
        const float4 vx = ld4(x + base, i, n_per, v4), vm = ld4(m + base, i, n_per, v4), v1 = ld4(m1 + base, i, n_per, v4), vp = ld4(x_prev + base, i, n_per, v4);
        float lo[4], hi[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            lo[e] = __fsub_rn(__fmul_rn(a, comp(vx, e)), __fmul_rn(b, comp(vm, e)));
            hi[e] = __fsub_rn(lo[e], __fmul_rn(c, __fsub_rn(comp(v1, e), comp(vm, e))));
            const float delta = fmaxf(atol, __fmul_rn(rtol, fmaxf(fabsf(lo[e]), fabsf(comp(vp, e)))));
            const float v = __fdiv_rn(__fsub_rn(hi[e], lo[e]), delta);
            if (i + e < n_per) acc += (double)__fmul_rn(v, v);
        }
        st4(x_lower + base, i, n_per, v4, make_float4(lo[0], lo[1], lo[2], lo[3]));
        st4(x_higher + base, i, n_per, v4, make_float4(hi[0], hi[1], hi[2], hi[3]));
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < DPM_THREADS / 64; ++w) s += wsum[w];
        partial[sample * gridDim.x + blockIdx.x] = s;
    }
}

// E = max over samples of sqrt(mean(v^2)): one wave; a sample's block sums are added in a fixed order (lane l takes blocks l, l + 64, ...; then a
// butterfly over the lanes)
__global__ __launch_bounds__(64) void dpm_err_finish_kernel(const double* __restrict__ partial, int n_samples, int blocks, long long n_per, float* __restrict__ E) {
    float best = 0.f;
    bool nan = false;
    for (int s = 0; s < n_samples; ++s) {
        double t = 0.0;
        for (int k = (int)threadIdx.x; k < blocks; k += 64) t += partial[(long long)s * blocks + k];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) t += __shfl_xor(t, d, 64);
        const float e = sqrtf((float)(t / (double)n_per));
        nan = nan || (e != e);
        best = fmaxf(best, e);
    }
    if (threadIdx.x == 0) *E = nan ? __builtin_nanf("") : best;        // (torch's max propagates a NaN: the caller's accept test must see it)
}

inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

}  // namespace

extern "C" int gvf_dpm_x0(const float* x, const float* noise, float sigma, float alpha, float* x0, int64_t n, void* stream_) {
    if (n < 0) return GVF_EINVAL;
    if (n == 0) return GVF_OK;
    if (!x || !noise || !x0) return GVF_EINVAL;
    const int vec = aligned16(x) && aligned16(noise) && aligned16(x0);
    (void)hipGetLastError();
    dpm_x0_kernel<<<dim3((unsigned)((n + DPM_PER_BLOCK - 1) / DPM_PER_BLOCK)), dim3(DPM_THREADS), 0, (hipStream_t)stream_>>>(x, noise, sigma, alpha, x0, n, vec);
    GVF_CHECK_LAUNCH();
    return GVF_OK;
}

extern "C" int gvf_dpm_lincomb(const float* x, const float* m0, const float* m1, float a, float b, float c, float* out, int64_t n, void* stream_) {
    if (n < 0) return GVF_EINVAL;
    if (n == 0) return GVF_OK;
    if (!x || !m0 || !out) return GVF_EINVAL;
    const int vec = aligned16(x) && aligned16(m0) && aligned16(out) && (m1 == nullptr || aligned16(m1));
    (void)hipGetLastError();
    dpm_lincomb_kernel<<<dim3((unsigned)((n + DPM_PER_BLOCK - 1) / DPM_PER_BLOCK)), dim3(DPM_THREADS), 0, (hipStream_t)stream_>>>(x, m0, m1, a, b, c, out, n, vec);
    GVF_CHECK_LAUNCH();
    return GVF_OK;
}

extern "C" int64_t gvf_dpm_err_scratch_doubles(int n_samples, int64_t n_per_sample) {
    if (n_samples <= 0 || n_per_sample <= 0) return 0;
    return (int64_t)n_samples * ((n_per_sample + DPM_PER_BLOCK - 1) / DPM_PER_BLOCK);
}

extern "C" int gvf_dpm_second_err(const float* x, const float* m, const float* m1, const float* x_prev, float a, float b, float c, float atol, float rtol,
                                  float* x_lower, float* x_higher, int n_samples, int64_t n_per_sample, double* scratch, float* E, void* stream_) {
    if (n_samples < 0 || n_per_sample < 0 || n_samples > 65535) return GVF_EINVAL;
    if (n_samples == 0 || n_per_sample == 0) return GVF_EINVAL;          // (an error norm over nothing has no value)
    if (!x || !m || !m1 || !x_prev || !x_lower || !x_higher || !scratch || !E) return GVF_EINVAL;
    const long long blocks = (n_per_sample + DPM_PER_BLOCK - 1) / DPM_PER_BLOCK;
    if (blocks > 0x7fffffffLL) return GVF_EINVAL;
    const int vec = aligned16(x) && aligned16(m) && aligned16(m1) && aligned16(x_prev) && aligned16(x_lower) && aligned16(x_higher) && (n_per_sample % 4) == 0;
    hipStream_t stream = (hipStream_t)stream_;
    (void)hipGetLastError();
    dpm_second_err_kernel<<<dim3((unsigned)blocks, (unsigned)n_samples), dim3(DPM_THREADS), 0, stream>>>(x, m, m1, x_prev, a, b, c, atol, rtol, x_lower, x_higher,
                                                                                                        scratch, (long long)n_per_sample, vec);
    dpm_err_finish_kernel<<<dim3(1), dim3(64), 0, stream>>>(scratch, n_samples, (int)blocks, (long long)n_per_sample, E);
    GVF_CHECK_LAUNCH();
    return GVF_OK;
}
