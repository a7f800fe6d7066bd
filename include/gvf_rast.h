/*
 * gvf_rast.h -- C ABI of the MI355X (gfx950) tile-based 3D-Gaussian-splatting rasteriser.
 *
 * This is the drop-in boundary for the reference's rasteriser operator seam:
 *   renderers/gaussian_render.py:110-143  GaussianRasterizationSettings(...)   -> GvfRastSettings / GvfRastFrame
 *   renderers/gaussian_render.py:198-220  GaussianRasterizer(...)(means3D, ...) -> gvf_rast_forward()
 *   renderers/gaussian_render.py:154-160 + representations/gaussian/gaussian_model.py:84-114
 *        (get_*_with_delta, fused in front of the rasteriser)                  -> gvf_rast_forward_batched()
 * The external CUDA packages behind that seam (diff_gaussian_rasterization [mip-splatting fork]
 * and diff_gauss [slothfulxtx fork]) are NOT in /root/reference (setup.sh:111,220-227); the two
 * `mode`s below select their published behaviours.
 *
 * Conventions (all entry points):
 *   - extern "C", plain pointers and sizes, no torch types.
 *   - every data pointer is a DEVICE pointer unless the name ends in _host.
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*); nothing synchronises,
 *     nothing allocates: the caller owns every buffer, including the workspace.
 *   - return value: 0 on success, negative GVF_E* on a host-detectable error. Kernel launch
 *     failures are returned as GVF_ELAUNCH (hipGetLastError is consulted after each launch).
 *   - re-entrant per (device, stream); no global mutable state.
 */
#ifndef GVF_RAST_H
#define GVF_RAST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GVF_OK        0
#define GVF_EINVAL   (-1)   /* bad argument (null pointer, negative size, unsupported SH degree ...) */
#define GVF_ENOSPC   (-2)   /* workspace too small for the request */
#define GVF_ELAUNCH  (-3)   /* a kernel launch or HIP runtime call failed */

#define GVF_RAST_MODE_MIP      0   /* diff_gaussian_rasterization (mip-splatting): 2D mip filter, opacity*=coef; returns color,radii */
#define GVF_RAST_MODE_DILATE   1   /* diff_gauss (slothfulxtx): +0.3 px^2 dilation; also returns depth, alpha */

#define GVF_TILE 16                /* tile edge in pixels (BLOCK_X = BLOCK_Y = 16 upstream) */

/* Per-frame camera block == the camera fields of GaussianRasterizationSettings
 * (gaussian_render.py:110-125).  Matrices are the 16 floats of the tensors the reference passes:
 * viewmatrix = world_view_transform = V^T, projmatrix = full_proj_transform = (P V)^T, both
 * row-major, i.e. element [i] here is column-major V / PV exactly as the upstream kernels index. */
typedef struct GvfRastFrame {
    float viewmatrix[16];
    float projmatrix[16];
    float campos[3];
    float tanfovx;
    float tanfovy;
    int32_t delta_index;      /* batched path: which (P,14) slice of `delta` this frame uses; -1 = none */
    int32_t reserved[2];
} GvfRastFrame;

/* Frame-invariant settings == the remaining GaussianRasterizationSettings fields. */
typedef struct GvfRastSettings {
    int32_t image_height;
    int32_t image_width;
    int32_t sh_degree;        /* active degree D (0..3); colours come from shs[P][M][3] */
    int32_t mode;             /* GVF_RAST_MODE_* */
    float   kernel_size;      /* mip 2D filter variance in px^2 (pipe.kernel_size, 0.1 at inference) */
    float   scale_modifier;
    float   bg[3];
    int32_t prefiltered;      /* accepted for API parity; unused (as upstream when False) */
    int32_t debug;            /* accepted for API parity */
    int32_t upstream_binning; /* 0 (default): a (Gaussian, tile) pair is binned only if alpha = opacity * exp(power) can
                                 reach 1/255 somewhere in the tile (conservative box test) -- the blend skips every other
                                 pair at each pixel, so images are identical and num_rendered is smaller;
                                 1: bin the whole 3-sigma tile rect, num_rendered equals upstream's count */
    int32_t bin_algo;         /* GVF_RAST_BIN_*: how instances reach their tile segment (same image either way) */
} GvfRastSettings;

/* BUCKET (default): Gaussians are put in Morton order once per call, per-tile instance counts are taken in the
 * preprocess kernel (LDS histogram per block, one global atomic per touched tile), their scan gives the tile
 * ranges and a scatter pass writes (depth, id) into the segments; RADIX: upstream's scheme restricted to the
 * (frame, tile) key bits -- duplicate, two global 8-bit LSD passes, ranges.  Both finish with the per-tile
 * on-chip sort by (depth, id). */
#define GVF_RAST_BIN_AUTO   0
#define GVF_RAST_BIN_RADIX  1
#define GVF_RAST_BIN_BUCKET 2

/* Activation constants of GaussianModel (representations/gaussian/gaussian_model.py:24-41,84-114)
 * for the fused delta path.  scaling_activation: 0 = exp, 1 = softplus. */
typedef struct GvfGaussianActivation {
    float   aabb[6];              /* xyz = _xyz * aabb[3:6] + aabb[0:3] */
    float   scale_bias;           /* inverse_activation(scaling_bias), added before the activation */
    float   opacity_bias;         /* logit(opacity_bias) */
    float   min_kernel_size;      /* 3D filter: scale = sqrt(act(.)^2 + k^2) */
    int32_t scaling_activation;
} GvfGaussianActivation;

/* Bytes of workspace gvf_rast_forward*() needs for P Gaussians, F frames in one call, an H x W
 * image and room for at most `max_rendered` (splat,tile) instances summed over the F frames. */
int gvf_rast_workspace_bytes(int P, int F, int H, int W, int64_t max_rendered, size_t* bytes);

/* One frame, activated inputs: the GaussianRasterizer.__call__ operator.
 *   means3D[P][3], opacities[P], scales[P][3], rotations[P][4] (r,x,y,z; used un-normalised),
 *   exactly one of shs[P][M][3] / colors_precomp[P][3] non-null,
 *   cov3D_precomp[P][6] may replace scales+rotations (pass those null then),
 *   subpixel_offset[H][W][2] may be null (= zeros).
 * Outputs (caller-allocated): out_color[3][H][W]; out_radii[P] int32;
 *   out_alpha[H][W], out_depth[H][W] may be null;
 *   out_num_rendered: device uint32 receiving the instance count D. If D > max_rendered the
 *   frame is NOT rendered correctly: the caller must check it and retry with a larger workspace. */
int gvf_rast_forward(const GvfRastSettings* settings_host, const GvfRastFrame* frame_host,
                     int P, int M,
                     const float* means3D, const float* shs, const float* colors_precomp,
                     const float* opacities, const float* scales, const float* rotations,
                     const float* cov3D_precomp, const float* subpixel_offset,
                     void* workspace, size_t workspace_bytes, int64_t max_rendered,
                     float* out_color, float* out_alpha, float* out_depth,
                     int32_t* out_radii, uint32_t* out_num_rendered,
                     void* stream);

/* Backward of gvf_rast_forward() (the operator is differentiable: renderers/gaussian_render.py:198-220 is called
 * under autograd, train_vae.py:321-352 back-propagates a render loss through it; upstream RasterizeGaussiansBackward).
 * `workspace` must be the workspace of the forward call for the SAME inputs, untouched since (it holds the splat
 * records, the per-tile sorted lists and the tile ranges), with the same workspace_bytes / max_rendered.
 * dL_dcolor[3][H][W] is required; dL_dalpha[H][W], dL_ddepth[H][W] may be null (mode DILATE outputs).
 * scratch: >= gvf_rast_backward_scratch_bytes(P) bytes, 16-byte aligned (per-Gaussian accumulators, zeroed here).
 * Outputs (caller-allocated, fully written): dL_dmeans3D[P][3]; dL_dmeans2D[P][2] or null -- the screen-space
 * gradient in NDC units, as upstream reports it in screenspace_points.grad; dL_dshs[P][M][3] xor dL_dcolors[P][3];
 * dL_dopacities[P]; dL_dscales[P][3] + dL_drotations[P][4], or dL_dcov3D[P][6] (also written when non-null with
 * scales/rotations given).  Conventions taken over from upstream: the gradient passes through the alpha <= 0.99 clamp;
 * a view-space coordinate clamped to 1.3 tan(fov) gets no gradient. */
int gvf_rast_backward_scratch_bytes(int P, size_t* bytes);
int gvf_rast_backward(const GvfRastSettings* settings_host, const GvfRastFrame* frame_host, int P, int M,
                      const float* means3D, const float* shs, const float* colors_precomp,
                      const float* opacities, const float* scales, const float* rotations,
                      const float* cov3D_precomp, const float* subpixel_offset,
                      const void* workspace, size_t workspace_bytes, int64_t max_rendered,
                      const float* dL_dcolor, const float* dL_dalpha, const float* dL_ddepth,
                      void* scratch, size_t scratch_bytes,
                      float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dshs, float* dL_dcolors,
                      float* dL_dopacities, float* dL_dscales, float* dL_drotations, float* dL_dcov3D,
                      void* stream);

/* F frames in one call, GaussianModel parameters + per-frame deltas activated in-kernel
 * (render() with delta_pc, gaussian_render.py:154-160).  Raw parameters:
 *   xyz_raw[P][3], features_dc[P][M][3], scaling_raw[P][3], rotation_raw[P][4], opacity_raw[P];
 *   delta[n_delta][P][14] laid out [xyz3|scale3|rot4|rgb3|op1] (may be null: static render);
 *   frames_host[F].
 * Outputs: out_color[F][3][H][W]; out_alpha/out_depth [F][H][W] or null; out_radii[F][P] or null;
 *   out_num_rendered[F] device uint32 (per-frame instance counts; their sum must be
 *   <= max_rendered, else frames past the overflow point are not rendered correctly).
 * The call only enqueues kernels on `stream` (camera blocks travel as kernel arguments, per-call tables are cleared by the first
 * launch, nothing is read back): it can be captured in a hipGraph and replayed.  One function attribute of the large-segment sort
 * kernel has to be set outside any capture: gvf_rast_workspace_bytes -- which every caller runs, on the device's thread, to size
 * the workspace before its first forward -- does that once per process (under a lock; the entry points are re-entrant per
 * (device, stream) and may be called from several host threads). */
int gvf_rast_forward_batched(const GvfRastSettings* settings_host, const GvfRastFrame* frames_host, int F,
                             const GvfGaussianActivation* act_host,
                             int P, int M,
                             const float* xyz_raw, const float* features_dc, const float* scaling_raw,
                             const float* rotation_raw, const float* opacity_raw,
                             const float* delta, int n_delta,
                             void* workspace, size_t workspace_bytes, int64_t max_rendered,
                             float* out_color, float* out_alpha, float* out_depth,
                             int32_t* out_radii, uint32_t* out_num_rendered,
                             void* stream);

/* The same call with the frames leaving as uint8: out_rgb_u8[F][3][H][W] = (uint8) (clamp(rgb, 0, 1) * 255), the post-process the
 * reference applies to every rendered frame on the host (utils/inference_utils.py:280-286: np.clip(...) * 255 -> astype(np.uint8)),
 * computed in the compositing kernel's epilogue on the same fp32 value gvf_rast_forward_batched would have stored -- bit-identical to
 * gvf_rast_forward_batched + gvf_rgb_to_u8, without the fp32 frame's round trip through HBM (12 + 12 + 3 bytes per pixel -> 3).
 * No alpha / depth / radii outputs (the reference's render job reads none of them). */
int gvf_rast_forward_batched_u8(const GvfRastSettings* settings_host, const GvfRastFrame* frames_host, int F,
                                const GvfGaussianActivation* act_host,
                                int P, int M,
                                const float* xyz_raw, const float* features_dc, const float* scaling_raw,
                                const float* rotation_raw, const float* opacity_raw,
                                const float* delta, int n_delta,
                                void* workspace, size_t workspace_bytes, int64_t max_rendered,
                                uint8_t* out_rgb_u8, uint32_t* out_num_rendered,
                                void* stream);

/* GaussianModel activations alone (get_*_with_delta), for callers that want the activated
 * tensors (and for parity tests of the fused path): writes means3D[P][3], scales[P][3],
 * rotations[P][4], shs[P][M][3], opacities[P].  delta may be null. */
int gvf_gaussian_activate(const GvfGaussianActivation* act_host, int P, int M,
                          const float* xyz_raw, const float* features_dc, const float* scaling_raw,
                          const float* rotation_raw, const float* opacity_raw, const float* delta,
                          float* means3D, float* scales, float* rotations, float* shs, float* opacities,
                          void* stream);

/* Stable LSD radix sort of n (key,value) pairs on bits [0,end_bit) of 64-bit keys: the
 * rasteriser's R4 stage, exported for testing.  keys_alt/values_alt are scratch of the same
 * size; the sorted result is left in keys/values. tmp: >= gvf_sort_tmp_bytes(n) bytes. */
size_t gvf_sort_tmp_bytes(int64_t n);
int gvf_sort_pairs_u64(uint64_t* keys, uint64_t* keys_alt, uint32_t* values, uint32_t* values_alt,
                       int64_t n, int end_bit, void* tmp, size_t tmp_bytes, void* stream);

/* The second half of R4, exported for testing: every segment [ranges[2 s], ranges[2 s + 1]) of keys -- (depth bits << 32 | id),
 * ids unique inside a segment, as the binning stage leaves them per (frame, tile) -- is sorted ascending and the low words (the
 * ids) are written to the same positions of `ids`: upstream's stable (tile, depth) order.  Segments up to 2048 keys take a
 * distribution sort on depth (sorting network when one depth bucket gets crowded), up to 16384 a network in LDS, larger ones a
 * network in global memory (keys of those segments are permuted in place).  scratch: 2 + 2 nseg uint32. */
int gvf_tile_sort_u64(uint64_t* keys, const uint32_t* ranges, int nseg, uint32_t* ids, uint32_t* scratch, void* stream);

/* Frame post-process of the render loop (utils/inference_utils.py:280-286: image.clamp(0,1) -> * 255 ->
 * astype('uint8')) on the device: out[i] = (uint8)(clamp(rgb[i], 0, 1) * 255), n elements. */
int gvf_rgb_to_u8(const float* rgb, uint8_t* out, int64_t n, void* stream);

/* Opt-in per-stage GPU timing of gvf_rast_forward*(): HIP events are recorded on the caller's
 * stream at the stage boundaries of the next (at most 256) calls.  Stages, in order:
 * order (Morton order + input gather), preprocess, scan, bin (scatter | duplicate), sort (radix passes + ranges;
 * empty for bucket binning), tile_sort, blend.  gvf_rast_profile_read() synchronises on the
 * recorded events, writes the summed milliseconds per stage over `*calls` calls, and resets.
 * This is the library's only process-global state (off by default; not thread-safe). */
#define GVF_RAST_NSTAGES 7
int gvf_rast_profile_enable(int on);
int gvf_rast_profile_read(float* ms_sum /*[GVF_RAST_NSTAGES]*/, int* calls);

/* How many gvf_rast_forward_batched() calls of this process took the shared-activation path: when the F frames of a call select few
 * distinct delta slices (the reference's render loop, utils/inference_utils.py:256-269: 128 cameras per timestep), the GaussianModel
 * activations (gaussian_model.py:84-114) and the 3-D covariances are computed once per (slice, Gaussian) instead of once per frame;
 * the outputs are the same bits either way (GVF_RAST_SHARED_ACT=0 in the environment forces the per-frame form).  A test aid. */
int64_t gvf_rast_shared_activation_calls(void);

/* Diagnostic of the per-tile sort (R4): how many (frame, tile) segments of the LAST gvf_rast_forward*() call on this workspace fell into the
 * size classes above the one-workgroup register sort -- counts[0]: 1537 .. 16384 keys (the two LDS launches; 2049 .. up to round 6), counts[1]: more than 16384
 * (sorted in place in HBM).  Takes the arguments the forward call carved its workspace with; waits for `stream` and copies two words to the
 * host (tests use it to assert that a scene really entered those launches; not on any hot path). */
int gvf_rast_sort_class_counts(const void* workspace, size_t workspace_bytes, int P, int F, int H, int W, int64_t max_rendered,
                               uint32_t* counts_host /*[2]*/, void* stream);

/* Library identification: returns a static string "gvf_hip <version> gfx950". */
const char* gvf_version(void);

#ifdef __cplusplus
}
#endif
#endif /* GVF_RAST_H */
