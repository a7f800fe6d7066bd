/*
 * gvf_points.h -- C ABI of the point-set helper of the path's caller side: farthest point sampling.
 *
 * Replaces `torch_cluster.fps` (third-party CUDA wheel, absent from /root/reference; imported at
 * utils/inference_utils.py:8, model/autoencoder.py:13, encode_latent.py:18) for its uses on the inference path:
 *   utils/inference_utils.py:180-198  sample_gs(): 4096 / num_latents positions out of a sample's static Gaussians --
 *       the DiT's `static_latent` and `deformation_position_xyz` conditions (inference_dpm_latent.py:208-218);
 *   model/autoencoder.py (encode): the latent queries of the motion VAE.
 * Algorithm (the published one, restated in oracle/points_ref.py): per batch element, start from a given point, keep for
 * every point the squared distance to the nearest selected point ((dx*dx + dy*dy) + dz*dz in binary32, no contraction),
 * repeatedly select the point where it is largest (ties: lowest index) -- indices are returned in selection order, as
 * row numbers of `pos`.  Integer output: bit-exact against the oracle.
 * Conventions as in gvf_rast.h (device pointers unless *_host, explicit stream, int status, caller-owned scratch).
 */
#ifndef GVF_POINTS_H
#define GVF_POINTS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GVF_FPS_MAX_BATCH   16          /* batch elements per call */
#define GVF_FPS_MAX_POINTS  1048576     /* points per batch element (256 workgroups x 4096 register-resident points) */

/* scratch for n_batches elements of at most max_k samples each (per-iteration hand-off slots; zeroed by gvf_fps) */
int gvf_fps_scratch_bytes(int n_batches, int max_k, size_t* bytes);

/* pos[N][3] fp32; batch b owns rows [ptr_host[b], ptr_host[b+1]); k_host[b] samples are drawn from it, the first one
 * being row ptr_host[b] + start_host[b]; out_idx (int64, device) receives sum(k) row numbers, batch after batch.
 * status_out (device int32, may be null) is set to 1 if the in-kernel hand-off timed out (output then invalid). */
int gvf_fps(const float* pos, const int32_t* ptr_host, int n_batches, const int32_t* k_host, const int32_t* start_host,
            int64_t* out_idx, void* scratch, size_t scratch_bytes, int32_t* status_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GVF_POINTS_H */
