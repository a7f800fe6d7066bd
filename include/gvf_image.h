/*
 * gvf_image.h -- C ABI of the frame post-process of the path's caller side: 8-bit separable resampling with placement
 * into a fixed canvas.
 *
 * Replaces, per rendered frame, the host round trip of utils/inference_utils.py:276-297 (render_and_save_images):
 *     rgb.cpu() -> uint8 -> PIL Image.resize((target, target), LANCZOS) -> paste into a white 512x512 canvas or
 *     centre-crop to 512x512 -> PNG
 * with device-side passes over the uint8 frames the rasteriser driver already produces (gvf_rgb_to_u8, gvf_rast.h), so that
 * only finished 512x512 frames cross PCIe.  The arithmetic is Pillow's 8-bit-per-channel resampler (third-party, not under
 * /root/reference; src/libImaging/Resample.c: precompute_coeffs, normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc /
 * Vertical_8bpc): per output sample an int32 accumulation  (1 << 21) + sum_k in[first + k] * coef[k]  followed by
 * >> 22 and a clamp to 0..255; horizontal pass first, its uint8 result feeding the vertical pass.  The coefficient tables
 * (22-bit fixed point, any filter) are computed by the host in double precision and handed over; restated in
 * oracle/resize_ref.py and checked bit-exactly against Pillow itself in the tests.
 * Conventions as in gvf_rast.h (device pointers, explicit stream, int status, caller-owned buffers).
 */
#ifndef GVF_IMAGE_H
#define GVF_IMAGE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GVF_RESAMPLE_PRECISION_BITS 22   /* 32 - 8 - 2, Pillow's PRECISION_BITS */
#define GVF_RESAMPLE_MAX_TAPS       256
#define GVF_RESAMPLE_MAX_WIDTH      16384

/* One separable pass table for `n_out` output samples: first[n_out] = index of the first input sample, count[n_out] = taps
 * used (<= ksize), coef[ksize][n_out] int32 fixed point (tap-major, so that neighbouring outputs read neighbouring words);
 * taps beyond count are zero. */
typedef struct GvfResampleTable {
    const int32_t* first;
    const int32_t* count;
    const int32_t* coef;
    int32_t ksize;
    int32_t n_out;
} GvfResampleTable;

/* src: planes x in_h x in_w uint8 (planes = frames x channels).  Horizontal pass with `tab_h` (n_out = mid_w) into
 * tmp (planes x in_h rows of pitch (mid_w + 3) & ~3 bytes; may be null when tab_h is null = no horizontal resampling,
 * mid_w = in_w), vertical pass with
 * `tab_v` (n_out = mid_h; null = none, mid_h = in_h), written into dst (planes x dst_h x dst_w) with the resampled image's
 * top-left corner at (off_y, off_x) -- negative offsets crop, uncovered canvas pixels take pad_value. */
int gvf_resample_place_u8(const uint8_t* src, int64_t planes, int in_h, int in_w, const GvfResampleTable* tab_h,
                          const GvfResampleTable* tab_v, uint8_t* tmp, uint8_t* dst, int dst_h, int dst_w, int off_y, int off_x,
                          int pad_value, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GVF_IMAGE_H */
