/*
 * gvf_sparse.h -- C ABI of the sparse-voxel helpers of the path (secondary rows SP1-SP3).
 *
 * Replaces the reference's in-tree CUDA extension `vox2seq`
 *   model/sparse_voxel_diffusion/vox2seq/src/api.cu:17,39,61,83 (launches), src/z_order.cu:35-66,
 *   src/hilbert.cu:35-133 (kernels), src/ext.cpp:5-9 (pybind: z_order_encode / z_order_decode /
 *   hilbert_encode / hilbert_decode), used by model/sparse_attention/serialized_attn.py:62-75.
 * Codes are the 30-bit interleave of three 10-bit coordinates (x at bit 3k+2, y at 3k+1, z at 3k);
 * bit-exact with the vox2seq/pytorch sources -- the equality vox2seq/test.py:5-24 asserts for the CUDA extension.
 * Conventions as in gvf_rast.h (device pointers, explicit stream, int status).
 */
#ifndef GVF_SPARSE_H
#define GVF_SPARSE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int gvf_z_order_encode(const int32_t* x, const int32_t* y, const int32_t* z, int32_t* code, int64_t n, void* stream);
int gvf_z_order_decode(const int32_t* code, int32_t* x, int32_t* y, int32_t* z, int64_t n, void* stream);
int gvf_hilbert_encode(const int32_t* x, const int32_t* y, const int32_t* z, int32_t* code, int64_t n, void* stream);
int gvf_hilbert_decode(const int32_t* code, int32_t* x, int32_t* y, int32_t* z, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GVF_SPARSE_H */
