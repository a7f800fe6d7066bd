// resize.hip -- 8-bit separable resampling (Pillow's fixed-point arithmetic) + placement into a canvas, gfx950.
//
// Replaces the per-frame host work of utils/inference_utils.py:276-297 (see include/gvf_image.h).  Byte streaming work,
// HBM-bound: 1 B read per input sample (horizontal pass), ~ksize cached re-reads per output, 1 B written per output.
//   horizontal: one workgroup per 8 image rows staged in LDS (16-byte loads); thread x produces output column x of all 8, so
//               a coefficient is fetched once per 8 rows; the tap-major table makes those fetches coalesced.
//   vertical + place: one thread per four neighbouring canvas pixels; inside the image a tap is one 32-bit load of the
//               intermediate image (row pitch padded to 4 bytes), the result one 32-bit store.
#include "gvf_common.h"
#include "../../include/gvf_rast.h"
#include "../../include/gvf_image.h"

namespace {

constexpr int RB = 256;
constexpr int PREC = GVF_RESAMPLE_PRECISION_BITS;

__device__ __forceinline__ uint8_t clip8(int32_t acc) {
    const int32_t v = acc >> PREC;                       // arithmetic shift, as Pillow's clip8 lookup index
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

constexpr int HR = 8;     // image rows per workgroup of the horizontal pass

// One workgroup = HR consecutive rows (of the planes x in_h row list), staged in LDS; thread x owns output column x of all
// of them, so every coefficient is fetched once per HR rows.  dst rows have pitch `out_pitch` (a multiple of 4 bytes).
__global__ __launch_bounds__(RB) void resample_h_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int64_t n_rows, int in_w,
                                                        int out_w, int out_pitch, const int32_t* __restrict__ first,
                                                        const int32_t* __restrict__ count, const int32_t* __restrict__ coef) {
    extern __shared__ uint8_t rows[];                     // HR x in_pitch
    const int in_pitch = (in_w + 15) & ~15;
    const int64_t r0 = (int64_t)blockIdx.x * HR;
    const int nr = (int)min((int64_t)HR, n_rows - r0);
    const uint8_t* in = src + r0 * in_w;
    const int64_t total = (int64_t)nr * in_w;
    if ((((uintptr_t)in) & 15) == 0 && (in_w & 15) == 0) {
        for (int i = threadIdx.x; i < (int)(total >> 4); i += RB) ((uint4*)rows)[i] = ((const uint4*)in)[i];   // in_pitch == in_w
    } else {
        for (int i = threadIdx.x; i < (int)total; i += RB) {
            const int r = i / in_w;
            rows[r * in_pitch + (i - r * in_w)] = in[i];
        }
    }
    __syncthreads();
    for (int x = threadIdx.x; x < out_w; x += RB) {
        const int f = first[x], n = count[x];
        int32_t acc[HR];
#pragma unroll
        for (int r = 0; r < HR; ++r) acc[r] = 1 << (PREC - 1);
        for (int k = 0; k < n; ++k) {
            const int32_t c = coef[(int64_t)k * out_w + x];
#pragma unroll
            for (int r = 0; r < HR; ++r) acc[r] += (int32_t)rows[r * in_pitch + f + k] * c;
        }
#pragma unroll
        for (int r = 0; r < HR; ++r)
            if (r < nr) dst[(r0 + r) * out_pitch + x] = clip8(acc[r]);
    }
}

struct __attribute__((packed, aligned(1))) U32u { uint32_t v; };

// mid: planes x mid_rows x mid_pitch bytes (mid_w valid per row).  One thread = four neighbouring canvas pixels of one row
// (one aligned 32-bit store); inside the image their taps are one 32-bit load per input row.
template <bool V>
__global__ __launch_bounds__(RB) void resample_v_place_kernel(const uint8_t* __restrict__ mid, uint8_t* __restrict__ dst, int mid_rows,
                                                              int mid_w, int mid_pitch, int out_h, int dst_h, int dst_w, int off_y, int off_x,
                                                              int pad, const int32_t* __restrict__ first, const int32_t* __restrict__ count,
                                                              const int32_t* __restrict__ coef) {
    const int x0 = (blockIdx.x * RB + threadIdx.x) * 4;
    const int y = blockIdx.y;
    const int64_t p = blockIdx.z;
    if (x0 >= dst_w) return;
    const int sy = y - off_y, sx0 = x0 - off_x;
    uint8_t v[4] = {(uint8_t)pad, (uint8_t)pad, (uint8_t)pad, (uint8_t)pad};
    const uint8_t* plane = mid + p * (int64_t)mid_rows * mid_pitch;
    if (sy >= 0 && sy < out_h && sx0 + 3 >= 0 && sx0 < mid_w) {
        const int f = V ? first[sy] : sy, n = V ? count[sy] : 1;
        if (sx0 >= 0 && sx0 + 3 < mid_w) {
            int32_t a0 = 1 << (PREC - 1), a1 = a0, a2 = a0, a3 = a0;
            if (V) {
                for (int k = 0; k < n; ++k) {
                    const uint32_t w = ((const U32u*)(plane + (int64_t)(f + k) * mid_pitch + sx0))->v;
                    const int32_t c = coef[(int64_t)k * out_h + sy];
                    a0 += (int32_t)(w & 255u) * c;
                    a1 += (int32_t)((w >> 8) & 255u) * c;
                    a2 += (int32_t)((w >> 16) & 255u) * c;
                    a3 += (int32_t)(w >> 24) * c;
                }
                v[0] = clip8(a0); v[1] = clip8(a1); v[2] = clip8(a2); v[3] = clip8(a3);
            } else {
                const uint32_t w = ((const U32u*)(plane + (int64_t)f * mid_pitch + sx0))->v;
                v[0] = w & 255u; v[1] = (w >> 8) & 255u; v[2] = (w >> 16) & 255u; v[3] = w >> 24;
            }
        } else {
            for (int i = 0; i < 4; ++i) {
                const int sx = sx0 + i;
                if (sx < 0 || sx >= mid_w) continue;
                if (V) {
                    int32_t acc = 1 << (PREC - 1);
                    for (int k = 0; k < n; ++k) acc += (int32_t)plane[(int64_t)(f + k) * mid_pitch + sx] * coef[(int64_t)k * out_h + sy];
                    v[i] = clip8(acc);
                } else {
                    v[i] = plane[(int64_t)f * mid_pitch + sx];
                }
            }
        }
    }
    uint8_t* o = dst + (p * dst_h + y) * (int64_t)dst_w + x0;
    if (x0 + 3 < dst_w && (dst_w & 3) == 0) {
        *(uint32_t*)o = (uint32_t)v[0] | ((uint32_t)v[1] << 8) | ((uint32_t)v[2] << 16) | ((uint32_t)v[3] << 24);
    } else {
        for (int i = 0; i < 4 && x0 + i < dst_w; ++i) o[i] = v[i];
    }
}

bool table_ok(const GvfResampleTable* t, int n_in) {
    return t->first && t->count && t->coef && t->ksize >= 1 && t->ksize <= GVF_RESAMPLE_MAX_TAPS && t->n_out >= 1 &&
           t->n_out <= GVF_RESAMPLE_MAX_WIDTH && n_in >= 1;
}

}  // namespace

extern "C" int gvf_resample_place_u8(const uint8_t* src, int64_t planes, int in_h, int in_w, const GvfResampleTable* tab_h,
                                     const GvfResampleTable* tab_v, uint8_t* tmp, uint8_t* dst, int dst_h, int dst_w, int off_y,
                                     int off_x, int pad_value, void* stream) {
    if (!src || !dst || planes < 0 || in_h < 1 || in_w < 1 || in_w > GVF_RESAMPLE_MAX_WIDTH || dst_h < 1 || dst_w < 1 ||
        dst_h > 65535 || planes > 0x7fffffff / (int64_t)in_h || pad_value < 0 || pad_value > 255)
        return GVF_EINVAL;
    if (tab_h && (!table_ok(tab_h, in_w) || !tmp)) return GVF_EINVAL;
    if (tab_v && !table_ok(tab_v, in_h)) return GVF_EINVAL;
    if (planes == 0) return GVF_OK;
    hipStream_t s = (hipStream_t)stream;
    const uint8_t* mid = src;
    int mid_w = in_w, mid_pitch = in_w;
    if (tab_h) {
        mid_w = tab_h->n_out;
        mid_pitch = (mid_w + 3) & ~3;                       // tmp must hold planes * in_h * mid_pitch bytes
        const int64_t n_rows = planes * in_h;
        const size_t lds = (size_t)HR * ((in_w + 15) & ~15);
        resample_h_kernel<<<dim3((unsigned)((n_rows + HR - 1) / HR)), RB, lds, s>>>(src, tmp, n_rows, in_w, mid_w, mid_pitch, tab_h->first,
                                                                                   tab_h->count, tab_h->coef);
        GVF_CHECK_LAUNCH();
        mid = tmp;
    }
    const int out_h = tab_v ? tab_v->n_out : in_h;
    if (planes > 65535) return GVF_EINVAL;
    dim3 grid((unsigned)(((dst_w + 3) / 4 + RB - 1) / RB), (unsigned)dst_h, (unsigned)planes);
    if (tab_v)
        resample_v_place_kernel<true><<<grid, RB, 0, s>>>(mid, dst, in_h, mid_w, mid_pitch, out_h, dst_h, dst_w, off_y, off_x, pad_value,
                                                          tab_v->first, tab_v->count, tab_v->coef);
    else
        resample_v_place_kernel<false><<<grid, RB, 0, s>>>(mid, dst, in_h, mid_w, mid_pitch, out_h, dst_h, dst_w, off_y, off_x, pad_value,
                                                           nullptr, nullptr, nullptr);
    GVF_CHECK_LAUNCH();
    return GVF_OK;
}
