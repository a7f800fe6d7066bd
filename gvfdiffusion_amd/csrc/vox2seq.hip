// vox2seq.hip -- Z-order (Morton) and Hilbert serialisation of 3 x 10-bit voxel coordinates, gfx950.
//
// Reference: model/sparse_voxel_diffusion/vox2seq/src/z_order.cu:35-66, src/hilbert.cu:35-133 (one thread per
// voxel, block 256: api.h:16).  Pure integer streaming work (12-16 B per voxel): HBM-bound, so the kernels are
// grid-stride with coalesced dword accesses and nothing else; bits are spread with the classic magic-mask
// sequence, the Hilbert transform is Skilling's transpose <-> axes algorithm unrolled for 3 axes x 10 bits.
#include "gvf_common.h"
#include "../../include/gvf_rast.h"
#include "../../include/gvf_sparse.h"

namespace {

constexpr int NBITS = 10;

// spread the low 10 bits of v so that bit k lands on bit 3k
__device__ __forceinline__ uint32_t spread3(uint32_t v) {
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}
__device__ __forceinline__ uint32_t compact3(uint32_t v) {
    v &= 0x09249249u;
    v = (v | (v >> 2)) & 0x030c30c3u;
    v = (v | (v >> 4)) & 0x0300f00fu;
    v = (v | (v >> 8)) & 0x030000ffu;
    v = (v | (v >> 16)) & 0x3ffu;
    return v;
}
__device__ __forceinline__ uint32_t interleave(uint32_t x, uint32_t y, uint32_t z) {
    return (spread3(x) << 2) | (spread3(y) << 1) | spread3(z);
}

__device__ __forceinline__ void axes_to_transpose(uint32_t (&X)[3]) {
    const uint32_t M = 1u << (NBITS - 1);
    for (uint32_t Q = M; Q > 1; Q >>= 1) {
        const uint32_t P = Q - 1;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (X[k] & Q) X[0] ^= P;
            else { const uint32_t t = (X[0] ^ X[k]) & P; X[0] ^= t; X[k] ^= t; }
        }
    }
    X[1] ^= X[0];
    X[2] ^= X[1];
    uint32_t t = 0;
    for (uint32_t Q = M; Q > 1; Q >>= 1)
        if (X[2] & Q) t ^= Q - 1;
    X[0] ^= t; X[1] ^= t; X[2] ^= t;
}
__device__ __forceinline__ void transpose_to_axes(uint32_t (&X)[3]) {
    const uint32_t N = 2u << (NBITS - 1);
    uint32_t t = X[2] >> 1;
    X[2] ^= X[1];
    X[1] ^= X[0];
    X[0] ^= t;
    for (uint32_t Q = 2; Q != N; Q <<= 1) {
        const uint32_t P = Q - 1;
#pragma unroll
        for (int k = 2; k >= 0; --k) {
            if (X[k] & Q) X[0] ^= P;
            else { t = (X[0] ^ X[k]) & P; X[0] ^= t; X[k] ^= t; }
        }
    }
}

template <bool HILBERT>
__global__ __launch_bounds__(256) void encode_kernel(const int32_t* __restrict__ x, const int32_t* __restrict__ y,
                                                     const int32_t* __restrict__ z, int32_t* __restrict__ code,
                                                     long long n) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        uint32_t X[3] = {(uint32_t)x[i], (uint32_t)y[i], (uint32_t)z[i]};
        if (HILBERT) axes_to_transpose(X);
        code[i] = (int32_t)interleave(X[0], X[1], X[2]);
    }
}
template <bool HILBERT>
__global__ __launch_bounds__(256) void decode_kernel(const int32_t* __restrict__ code, int32_t* __restrict__ x,
                                                     int32_t* __restrict__ y, int32_t* __restrict__ z, long long n) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const uint32_t c = (uint32_t)code[i];
        uint32_t X[3] = {compact3(c >> 2), compact3(c >> 1), compact3(c)};
        if (HILBERT) transpose_to_axes(X);
        x[i] = (int32_t)X[0]; y[i] = (int32_t)X[1]; z[i] = (int32_t)X[2];
    }
}

inline unsigned grid_for(long long n) {
    long long b = (n + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 8192 ? 8192 : b));   // 256 CUs x 8 WGs, grid-stride beyond
}

}  // namespace

#define GVF_VOX_ENTRY(NAME, KERNEL, ...)                                                              \
    if (n < 0) return GVF_EINVAL;                                                                     \
    if (n == 0) return GVF_OK;                                                                        \
    (void)hipGetLastError();                                                                          \
    hipLaunchKernelGGL(KERNEL, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__, (long long)n); \
    GVF_CHECK_LAUNCH();                                                                               \
    return GVF_OK;

extern "C" int gvf_z_order_encode(const int32_t* x, const int32_t* y, const int32_t* z, int32_t* code, int64_t n, void* stream) {
    if (n > 0 && (!x || !y || !z || !code)) return GVF_EINVAL;
    GVF_VOX_ENTRY(z_order_encode, encode_kernel<false>, x, y, z, code)
}
extern "C" int gvf_z_order_decode(const int32_t* code, int32_t* x, int32_t* y, int32_t* z, int64_t n, void* stream) {
    if (n > 0 && (!x || !y || !z || !code)) return GVF_EINVAL;
    GVF_VOX_ENTRY(z_order_decode, decode_kernel<false>, code, x, y, z)
}
extern "C" int gvf_hilbert_encode(const int32_t* x, const int32_t* y, const int32_t* z, int32_t* code, int64_t n, void* stream) {
    if (n > 0 && (!x || !y || !z || !code)) return GVF_EINVAL;
    GVF_VOX_ENTRY(hilbert_encode, encode_kernel<true>, x, y, z, code)
}
extern "C" int gvf_hilbert_decode(const int32_t* code, int32_t* x, int32_t* y, int32_t* z, int64_t n, void* stream) {
    if (n > 0 && (!x || !y || !z || !code)) return GVF_EINVAL;
    GVF_VOX_ENTRY(hilbert_decode, decode_kernel<true>, code, x, y, z)
}
