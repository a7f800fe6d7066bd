// rast.hip -- tile-based 3D-Gaussian-splatting forward rasteriser for gfx950 (MI355X).
//
// Implements the C ABI of include/gvf_rast.h, i.e. the operator behind the reference's
// GaussianRasterizer seam (renderers/gaussian_render.py:110-143,198-220) with the GaussianModel
// delta activations (representations/gaussian/gaussian_model.py:84-114) optionally fused in front.
// Stages follow SURVEY.md section 8a rows R1..R6 + G1; the arithmetic (operation order, fmaf
// placement, correctly rounded div/sqrt, no contraction: this file is compiled with
// -ffp-contract=off) is the floating-point contract stated in oracle/rast_oracle.c, so that all
// discrete decisions (cull, radius, tile rect, sort order) match the oracle bit for bit.
//
// Launch structure (F frames per call, everything stream-ordered, no host sync; D is read from device memory).
// Default = BUCKET binning (GvfRastSettings.bin_algo):
//   bbox, morton_count/scan/scatter      once per call: 15-bit Morton order of the Gaussians (locality for the bin passes)
//   preprocess   grid (ceil(P/256), F/4) activations (+deltas), EWA covariance, 2D filter, radius, tile rect, alpha-box
//                                        instance culling, SH -> RGB; writes one 64-byte splat record + a 16-byte bin record
//   bin<count>   grid (ceil(P/1024), F)  per-(frame, tile) instance counts: LDS histogram per block, one global atomic
//                                        per touched tile
//   seg_sums, seg_scan, frame_counts     exclusive scan of the counters = tile ranges + cursors, per-frame D, overflow guard
//   bin<scatter> grid (ceil(P/1024), F)  (depth_bits << 32 | id) into the tile segments (order inside a segment arbitrary)
//   classify + tile_sort                 per-tile sort: registers+shuffles / static LDS for segments <= 1536 (SORT_SMALL_N), LDS <= 16384,
//                                        in-place global beyond; writes ordered ids (= upstream's stable (tile, depth) order)
//   blend        grid (tiles, F)         16x16 px per workgroup, 4 waves = the four 8x8 quadrants
// RADIX binning (kept for comparison): preprocess (+block sums) -> scan_sums -> duplicate ((frame*tiles + tile) << 32 |
// depth keys + ids) -> two stable 8-bit LSD passes over the (frame, tile) key bits (sort.hip) -> ranges -> tile_sort.
// Upstream sorts all 64-bit (tile, depth) keys with one global radix sort (>= 6 passes over 12 B per instance);
// here no global sort is left and the depth order is produced on chip.
//
// HBM layout (caller-owned workspace, carved below): per (frame, Gaussian) ONE 64-byte, 64-byte-aligned record
//   float4 {x, y, conic_a, conic_b} | float4 {conic_c, opacity, r, g} | float4 {b, depth, hx, hy} | 16 B unused
// so the blend's gather of a (splat, tile) instance touches exactly one cache line (three 40-B-total arrays cost
// three lines per instance: 5.7 GB of fabric reads per 24-frame step measured with FETCH_SIZE, vs 1.1 GB
// algorithmic); hx, hy = half extents of the region where alpha can reach 1/255 (instance and quadrant culling),
// computed once per visible Gaussian.  A 16-byte bin record {x0|y0<<16, x1|y1<<16, depth bits, slab}; per
// (frame, tile) a range, a counter and a cursor; per instance one u64 key and one u32 ordered id (radix path:
// u64 key + u32 id, double buffered).
#include <atomic>
#include <cstdlib>
#include <mutex>
#include <vector>
#include "gvf_common.h"
#include "gvf_sort.h"
#include "../../include/gvf_rast.h"

namespace {

constexpr int PRE_THREADS = 256;
// Depth slabs per tile (bucket binning): segment = (frame, tile, slab), each sorted independently, a tile's slabs are
// contiguous and ordered.  Measured at the bench shape with 8 slabs: the per-tile sort does 1.8x fewer key-rounds
// but becomes dispatch-bound (480 k mostly empty workgroups, 0.41 -> 0.63 ms) and the bin passes pay 2-4x more global
// atomics (+0.24 ms) -- a net loss, so the machinery is kept but compiled for ONE slab.
constexpr int NSLAB = 1;
#ifndef GVF_PRE_FB
#define GVF_PRE_FB 4
#endif
constexpr int PRE_FB = GVF_PRE_FB;   // frames per preprocess workgroup: the frame-invariant inputs (xyz, scale, rotation, opacity,
                            // SH: 164 of the 220 input bytes per Gaussian at degree 2) come from HBM once per PRE_FB frames
// Workgroup -> (Gaussian block, frame group) of preprocess_kernel.  1: XCD-aware (round 5 experiment): workgroup n of the 1-D grid runs on XCD n % 8
// (round-robin dispatch); the FY frame groups of one Gaussian block are consecutive workgroups OF ONE XCD, so the block's frame-invariant
// 164 B per Gaussian cross the fabric once and are L2 hits for the other FY - 1 groups.  0 (default): round 1-4's (blocks, frame groups) 2-D grid,
// in which a frame group walks all 43 MB of static inputs before the next one starts.  Measured: 0.363-0.366 ms with the XCD order against
// 0.358 ms without (profiles/r05_preprocess_variants_ab.txt) -- the launch does not wait for those bytes.
#ifndef GVF_PRE_XCD
#define GVF_PRE_XCD 0
#endif
// 1: preprocess_kernel<true> (the shared-activation launch) stores its splat records quad-transposed, whole 64-byte lines per instruction (see there);
// the fused launch keeps three 16-byte stores per lane (the transposed form costs it 13 registers; 0.377 -> 0.368 ms, inside the noise)
#ifndef GVF_PRE_REC_QUAD
#define GVF_PRE_REC_QUAD 1
#endif
// 1: the delta row of frame ff + 1 is requested before frame ff's arithmetic (experiment; no gain: same file)
#ifndef GVF_PRE_PREFETCH
#define GVF_PRE_PREFETCH 0
#endif
constexpr int TILE = GVF_TILE;
constexpr int BLEND_THREADS = TILE * TILE;
constexpr int MAX_SH_COEFFS = 16;
// The splat record holds the conic PRE-SCALED for the blend: (a, b, c) -> (CONIC_K1 a, CONIC_K2 b, CONIC_K1 c), so that
//   log2(e) * power = log2(e) * (-0.5 (a dx^2 + c dy^2) - b dx dy) = a' dx^2 + c' dy^2 + b' dx dy
// needs no scaling on its way into v_exp_f32 (upstream's form costs nine VALU instructions plus the exp's own log2(e) multiply; the
// compositing loop is VALU-bound).  Readers that need the conic itself un-scale it; the compositing kernels factor it (splat_cholesky below).
constexpr float CONIC_K1 = -0.7213475204444817f;   // -0.5 log2(e)
constexpr float CONIC_K2 = -1.4426950408889634f;   // -log2(e)
constexpr float CONIC_IK1 = -1.3862943611198906f;  // 1 / CONIC_K1 = -2 ln 2
constexpr float CONIC_IK2 = -0.6931471805599453f;  // 1 / CONIC_K2 = -ln 2
// The compositing kernels evaluate the exponent from the CHOLESKY factor of the (scaled, negated) conic in tile-relative coordinates:
//   -power_oct = m11 dx^2 + 2 m12 dx dy + m22 dy^2 = s1^2 + s2^2,   s1 = l11 dx + l12 dy,  s2 = l22 dy,   M = [[-a', -b'/2], [-b'/2, -c']]
// with dx = xr - px, dy = yr - py (splat centre and pixel relative to the tile origin):  s1 = c1 - l11 px - l12 py,  s2 = c2 - l22 py.
// Per (pixel, splat) that is 3 fma + 1 mul + 1 fma and the negation rides on v_exp_f32's source modifier -- against 2 subtractions + 5 for the
// conic form -- and the exponent cannot come out positive, so upstream's `power > 0` test (which only ever fires on rounding noise at the
// centre of a valid splat) has nothing to do: 3 of the ~21 vector instructions of a compositing step.  |c1|, |c2| stay small because a splat
// reaches a tile only within ~3 sigma (|c| <~ 3 + 16 / sigma), so the cancellation in s1 costs ~1e-5 of the exponent.  A conic that is not
// positive definite (NaN / overflowed covariances: upstream composites an indefinite form there) is dropped: its opacity is staged as 0.
struct SplatChol { float l11, l12, l22, c1, c2; bool ok; };
__device__ __forceinline__ SplatChol splat_cholesky(float x, float y, float ap, float bp, float cp, float tile_x0, float tile_y0) {
    SplatChol r;
    const float m11 = -ap, m12 = -0.5f * bp, m22 = -cp;
    const float il = __builtin_amdgcn_rsqf(m11);
    r.l11 = m11 * il;                                   // sqrt(m11)
    r.l12 = m12 * il;
    const float d = m22 - r.l12 * r.l12;
    r.l22 = __builtin_amdgcn_sqrtf(d);
    r.ok = m11 > 0.0f && d > 0.0f && m11 < __builtin_inff() && d < __builtin_inff();
    if (!r.ok) { r.l11 = 0.f; r.l12 = 0.f; r.l22 = 0.f; }
    const float xr = x - tile_x0, yr = y - tile_y0;
    r.c1 = r.ok ? __builtin_fmaf(r.l11, xr, r.l12 * yr) : 0.f;
    r.c2 = r.ok ? r.l22 * yr : 0.f;
    return r;
}
// s1^2 + s2^2 - lo = -(exponent + lo) (octaves) at tile-relative pixel (px, py).  lo = 0: minus the exponent itself; lo = log2(opacity): the
// compositing kernels' form -- alpha = exp2(log2(opacity) + exponent) costs no multiply by the opacity (the constant rides in the first square's
// fma), and an opacity of 0 (or a dropped splat) is lo = -inf -> alpha = 0.
__device__ __forceinline__ float splat_neg_exponent(float l11, float l12, float l22, float c1, float c2, float px, float py, float lo = 0.0f) {
    const float s1 = __builtin_fmaf(-l11, px, __builtin_fmaf(-l12, py, c1));
    const float s2 = __builtin_fmaf(-l22, py, c2);
    return __builtin_fmaf(s2, s2, __builtin_fmaf(s1, s1, -lo));
}

__constant__ float SH_C0 = 0.28209479177387814f;
__constant__ float SH_C1 = 0.4886025119029199f;
__constant__ float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
__constant__ float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                               0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                               -0.5900435899266435f};

struct PreParams {
    int P, M, deg, H, W, mode;
    int gx, gy;           // tile grid
    float kernel_size, scale_modifier;
    // fused activation (raw GaussianModel parameters) -- used when fused != 0
    int fused;
    GvfGaussianActivation act;
    int n_delta;
    int upstream_binning;   // 1: bin the whole 3-sigma tile rect as upstream does
    int F;                  // frames of the call (grid.y covers them PRE_FB at a time)
};

// ---------------------------------------------------------------------------------------------
// G1: GaussianModel activations (gaussian_model.py:84-114); delta layout [xyz3|scale3|rot4|rgb3|op1]
// ---------------------------------------------------------------------------------------------
// exp / log1p of the activations: the SAME sequence of correctly rounded operations as oracle/rast_oracle.c::act_expf / act_log1pf (fma where
// written, + - * /, float <-> int conversions, bit operations; this file is compiled with -ffp-contract=off), so that scales and opacities --
// and with them every radius, tile rect, instance count and sort key of the fused-activation path -- are bit-identical to the oracle's
// (round 6; up to round 5 the device's math library and the oracle's libm differed by an ulp or two and a few of 6.3 M radii flipped).
// Each is within 1 ulp of the true value (tests/test_oracle_rast.py::test_shared_activation_arithmetic_stays_within_2ulp_of_libm).
__device__ __forceinline__ float act_expf(float x) {
    if (x != x) return x;
    if (x > 88.72283f) return __builtin_inff();
    if (x < -103.97208f) return 0.0f;
    const float kf = x * 1.44269502f + (x < 0.0f ? -0.5f : 0.5f);
    const int k = (int)kf;                                   // truncation toward zero = round half away of x log2 e
    const float t = (float)k;
    float r = __builtin_fmaf(t, -0.693145751953125f, x);     // ln 2 = 0.693145751953125 (16 bits: t * it is exact) + 1.42860677e-6
    r = __builtin_fmaf(t, -1.42860677e-6f, r);
    float p = 1.98412698e-4f;                                // e^r, |r| <= 0.347: degree-7 Taylor polynomial, Horner
    p = __builtin_fmaf(p, r, 1.38888889e-3f);
    p = __builtin_fmaf(p, r, 8.33333377e-3f);
    p = __builtin_fmaf(p, r, 4.16666679e-2f);
    p = __builtin_fmaf(p, r, 1.66666672e-1f);
    p = __builtin_fmaf(p, r, 0.5f);
    p = __builtin_fmaf(p, r, 1.0f);
    p = __builtin_fmaf(p, r, 1.0f);
    const int k1 = k / 2, k2 = k - k1;                       // k in [-150, 128]: both factors are normal powers of two
    return (p * __uint_as_float((uint32_t)(k1 + 127) << 23)) * __uint_as_float((uint32_t)(k2 + 127) << 23);
}
__device__ __forceinline__ float act_log1pf(float y) {      // y >= 0 (or NaN)
    if (!(y >= 5.9604645e-8f)) return y;                     // < 2^-24: log1p(y) = y to the last bit (and NaN)
    if (y > 3.4028235e38f) return y;                         // +inf
    int k = 0;
    float c = 0.0f, f = y;
    if (y >= 0.41421354f) {                                  // 1 + y >= sqrt 2: split off the exponent
        const float u = 1.0f + y;
        uint32_t iu = __float_as_uint(u) + (0x3f800000u - 0x3f3504f3u);
        k = (int)(iu >> 23) - 127;
        if (k < 25) c = (k >= 2 ? 1.0f - (u - y) : y - (u - 1.0f)) / u;
        iu = (iu & 0x007fffffu) + 0x3f3504f3u;
        f = __uint_as_float(iu) - 1.0f;
    }
    const float s = f / (2.0f + f);
    const float z = s * s, w = z * z;
    const float t1 = w * (0.40000972152f + w * 0.24279078841f);
    const float t2 = z * (0.66666662693f + w * 0.28498786688f);
    const float R = t2 + t1;
    const float hfsq = (0.5f * f) * f;
    const float dk = (float)k;
    float acc = s * (hfsq + R);
    acc = acc + (dk * 9.0580006145e-6f + c);
    acc = acc - hfsq;
    acc = acc + f;
    return acc + dk * 6.9313812256e-1f;
}
#ifdef ACT_ABL_LIBM     // timing experiment only (variant build): the device math library's expf / log1pf, as up to round 5 (NOT bit-shared with the oracle)
#define act_expf expf
#define act_log1pf log1pf
#endif
__device__ __forceinline__ float act_scale(float x, const GvfGaussianActivation& a) {
    float s = a.scaling_activation == 0 ? act_expf(x) : (x > 20.0f ? x : act_log1pf(act_expf(x)));
    return sqrtf(s * s + a.min_kernel_size * a.min_kernel_size);
}

struct ActGaussian {
    float p[3], s[3], q[4], op, drgb[3];
};

// dl: the delta row (zeros when d is false -- they are not added then, as the reference's get_* accessors do without a delta)
__device__ __forceinline__ ActGaussian activate_vals(int i, const GvfGaussianActivation& a,
                                                     const float* __restrict__ xyz_raw,
                                                     const float* __restrict__ scaling_raw,
                                                     const float* __restrict__ rotation_raw,
                                                     const float* __restrict__ opacity_raw,
                                                     const float (&dl)[14], bool d) {
    ActGaussian g;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float v = xyz_raw[3 * (size_t)i + k] * a.aabb[3 + k] + a.aabb[k];
        g.p[k] = d ? v + dl[k] : v;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float x = scaling_raw[3 * (size_t)i + k] + a.scale_bias;
        if (d) x = x + dl[3 + k];
        g.s[k] = act_scale(x, a);
    }
    float q[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        q[k] = rotation_raw[4 * (size_t)i + k] + (k == 0 ? 1.0f : 0.0f);
        if (d) q[k] = q[k] + dl[6 + k];
    }
    float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    n = fmaxf(n, 1e-12f);
#pragma unroll
    for (int k = 0; k < 4; ++k) g.q[k] = q[k] / n;
    float x = opacity_raw[i] + a.opacity_bias;
    if (d) x = x + dl[13];
    g.op = 1.0f / (1.0f + act_expf(-x));
    g.drgb[0] = dl[10]; g.drgb[1] = dl[11]; g.drgb[2] = dl[12];
    return g;
}
__device__ __forceinline__ ActGaussian activate_one(int i, const GvfGaussianActivation& a,
                                                    const float* __restrict__ xyz_raw,
                                                    const float* __restrict__ scaling_raw,
                                                    const float* __restrict__ rotation_raw,
                                                    const float* __restrict__ opacity_raw,
                                                    const float* __restrict__ d /* delta row or null */) {
    float dl[14];
#pragma unroll
    for (int k = 0; k < 14; ++k) dl[k] = d ? d[k] : 0.0f;
    return activate_vals(i, a, xyz_raw, scaling_raw, rotation_raw, opacity_raw, dl, d != nullptr);
}

__global__ __launch_bounds__(256) void activate_kernel(GvfGaussianActivation a, int P, int M,
                                                       const float* __restrict__ xyz_raw,
                                                       const float* __restrict__ features_dc,
                                                       const float* __restrict__ scaling_raw,
                                                       const float* __restrict__ rotation_raw,
                                                       const float* __restrict__ opacity_raw,
                                                       const float* __restrict__ delta, float* __restrict__ means3D,
                                                       float* __restrict__ scales, float* __restrict__ rotations,
                                                       float* __restrict__ shs, float* __restrict__ opacities) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const float* d = delta ? delta + 14 * (size_t)i : nullptr;
    ActGaussian g = activate_one(i, a, xyz_raw, scaling_raw, rotation_raw, opacity_raw, d);
    for (int k = 0; k < 3; ++k) { means3D[3 * (size_t)i + k] = g.p[k]; scales[3 * (size_t)i + k] = g.s[k]; }
    for (int k = 0; k < 4; ++k) rotations[4 * (size_t)i + k] = g.q[k];
    opacities[i] = g.op;
    for (int m = 0; m < M; ++m)
        for (int c = 0; c < 3; ++c) {
            float v = features_dc[((size_t)i * M + m) * 3 + c];
            shs[((size_t)i * M + m) * 3 + c] = d ? v + g.drgb[c] : v;
        }
}

// ---------------------------------------------------------------------------------------------
// R1: preprocess
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void xform43(const float* m, const float* p, float* o) {
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}
__device__ __forceinline__ void xform44(const float* m, const float* p, float* o) {
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
    o[3] = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}

__device__ __forceinline__ void cov3d_from_scale_rot(const float* s, float mod, const float* q, float* c6) {
    float sx = mod * s[0], sy = mod * s[1], sz = mod * s[2];
    float r = q[0], x = q[1], y = q[2], z = q[3];
    float R00 = 1.f - 2.f * (y * y + z * z), R01 = 2.f * (x * y - r * z), R02 = 2.f * (x * z + r * y);
    float R10 = 2.f * (x * y + r * z), R11 = 1.f - 2.f * (x * x + z * z), R12 = 2.f * (y * z - r * x);
    float R20 = 2.f * (x * z - r * y), R21 = 2.f * (y * z + r * x), R22 = 1.f - 2.f * (x * x + y * y);
    float L00 = R00 * sx, L01 = R01 * sy, L02 = R02 * sz;
    float L10 = R10 * sx, L11 = R11 * sy, L12 = R12 * sz;
    float L20 = R20 * sx, L21 = R21 * sy, L22 = R22 * sz;
    c6[0] = L00 * L00 + L01 * L01 + L02 * L02;
    c6[1] = L00 * L10 + L01 * L11 + L02 * L12;
    c6[2] = L00 * L20 + L01 * L21 + L02 * L22;
    c6[3] = L10 * L10 + L11 * L11 + L12 * L12;
    c6[4] = L10 * L20 + L11 * L21 + L12 * L22;
    c6[5] = L20 * L20 + L21 * L21 + L22 * L22;
}

// sh: this Gaussian's coefficients in LDS, [M][3]; dadd: rgb delta added to every coefficient
__device__ __forceinline__ void sh_to_rgb(int deg, const float* sh, const float* dadd, const float* p,
                                          const float* cam, float* rgb) {
    float dx = p[0] - cam[0], dy = p[1] - cam[1], dz = p[2] - cam[2];
    float len = sqrtf(dx * dx + dy * dy + dz * dz);
    float x = dx / len, y = dy / len, z = dz / len;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float da = dadd[c];
        float res = SH_C0 * (sh[0 * 3 + c] + da);
        if (deg > 0) {
            res = res - SH_C1 * y * (sh[1 * 3 + c] + da) + SH_C1 * z * (sh[2 * 3 + c] + da) -
                  SH_C1 * x * (sh[3 * 3 + c] + da);
            if (deg > 1) {
                float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                res = res + SH_C2[0] * xy * (sh[4 * 3 + c] + da) + SH_C2[1] * yz * (sh[5 * 3 + c] + da) +
                      SH_C2[2] * (2.0f * zz - xx - yy) * (sh[6 * 3 + c] + da) +
                      SH_C2[3] * xz * (sh[7 * 3 + c] + da) + SH_C2[4] * (xx - yy) * (sh[8 * 3 + c] + da);
                if (deg > 2) {
                    res = res + SH_C3[0] * y * (3.0f * xx - yy) * (sh[9 * 3 + c] + da) +
                          SH_C3[1] * xy * z * (sh[10 * 3 + c] + da) +
                          SH_C3[2] * y * (4.0f * zz - xx - yy) * (sh[11 * 3 + c] + da) +
                          SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * (sh[12 * 3 + c] + da) +
                          SH_C3[4] * x * (4.0f * zz - xx - yy) * (sh[13 * 3 + c] + da) +
                          SH_C3[5] * z * (xx - yy) * (sh[14 * 3 + c] + da) +
                          SH_C3[6] * x * (xx - 3.0f * yy) * (sh[15 * 3 + c] + da);
                }
            }
        }
        res += 0.5f;
        rgb[c] = res < 0.f ? 0.f : res;
    }
}

struct TileRect { int x0, y0, x1, y1; };
__device__ __forceinline__ TileRect get_rect(float px, float py, float radius, int gx, int gy) {
    TileRect r;
    r.x0 = min(gx, max(0, (int)((px - radius) / (float)TILE)));
    r.y0 = min(gy, max(0, (int)((py - radius) / (float)TILE)));
    r.x1 = min(gx, max(0, (int)((px + radius + (float)(TILE - 1)) / (float)TILE)));
    r.y1 = min(gy, max(0, (int)((py + radius + (float)(TILE - 1)) / (float)TILE)));
    return r;
}

// Shared activation (round 5).  The reference renders one timestep from many cameras (utils/inference_utils.py:256-269: for t in 32, for cam in
// 128), i.e. consecutive frames of a batched call select the SAME delta slice: activations (gaussian_model.py:84-114) and the 3-D covariance do
// not depend on the camera.  When a call's F frames use few distinct slices, stage A computes them once per (slice, Gaussian) into a 64-byte
// record {x, y, z, opacity | S00, S01, S02, S11 | S12, S22, drgb0, drgb1 | drgb2, -, -, -} and preprocess_kernel<true> reads the record
// (one cache line, four 16-byte loads) instead of 112 B of strided raw inputs + the activation arithmetic per frame.  Same functions, same
// operation order, no contraction (-ffp-contract=off): every output bit equals the fused path's (tests/test_rast_gpu.py::test_shared_activation_*).
constexpr int ACT_MAX_SLICES = 16;
struct ActSlices { int n; int di[ACT_MAX_SLICES]; };
__global__ __launch_bounds__(256) void activate_cov_kernel(GvfGaussianActivation a, float scale_modifier, int P, ActSlices sl,
                                                           const float* __restrict__ xyz_raw, const float* __restrict__ scaling_raw,
                                                           const float* __restrict__ rotation_raw, const float* __restrict__ opacity_raw,
                                                           const float* __restrict__ delta, float4* __restrict__ rec3d,
                                                           const uint32_t* __restrict__ slot_of /* Gaussian -> Morton slot, or null */,
                                                           const float* __restrict__ sh, int sh_floats, float* __restrict__ sh_by_slot) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    // SLOT ORDER (slot_of != null): the record of Gaussian i goes to its Morton slot (a whole 64-byte line, scattered ONCE per slice) and slice 0's
    // workgroups also copy the SH rows into slot order; the per-frame launch then runs over slots: its record reads, its splat records and its bin
    // records (16 bytes each, scattered to the slot by the index-ordered form: 21 % of that launch) are all contiguous.
    const size_t dst = slot_of != nullptr ? (size_t)slot_of[i] : (size_t)i;
    if (sh_by_slot != nullptr && blockIdx.y == 0)
        for (int k = 0; k < sh_floats; ++k) sh_by_slot[dst * sh_floats + k] = sh[(size_t)i * sh_floats + k];
    const int di = sl.di[blockIdx.y];
    const float* d = (delta != nullptr && di >= 0) ? delta + ((size_t)di * P + i) * 14 : nullptr;
    const ActGaussian g = activate_one(i, a, xyz_raw, scaling_raw, rotation_raw, opacity_raw, d);
    float c6[6];
    cov3d_from_scale_rot(g.s, scale_modifier, g.q, c6);
    float4* r = rec3d + 4 * ((size_t)blockIdx.y * P + dst);
    r[0] = make_float4(g.p[0], g.p[1], g.p[2], g.op);
    r[1] = make_float4(c6[0], c6[1], c6[2], c6[3]);
    r[2] = make_float4(c6[4], c6[5], g.drgb[0], g.drgb[1]);
    r[3] = make_float4(g.drgb[2], 0.f, 0.f, 0.f);
}

// Inputs are either activated tensors (fused == 0: a0=means3D, a1=scales, a2=rotations, a3=opacities,
// sh=shs/colors) or raw GaussianModel parameters (fused != 0: a0=_xyz, a1=_scaling, a2=_rotation,
// a3=_opacity, sh=_features_dc, delta[n_delta][P][14]).
// Half extents (hx, hy) of the axis-aligned box around the splat centre outside which
// alpha = opacity * exp(power) < 1/255 for certain ( ca dx^2 + 2 cb dx dy + cc dy^2 <= 2 ln(255 opacity) ),
// inflated so that float noise can only keep extra splats, never drop one.  hx < 0: never visible;
// +inf: degenerate conic, keep everywhere.  Built from correctly rounded + * / sqrt and bit operations only
// (ln_upper: exponent + tangent envelope of log2, no libm), so oracle/rast_oracle.c reproduces them bit for
// bit and the culled instance counts can be compared exactly.  Used for (1) instance culling: a (Gaussian,
// tile) pair whose tile the box does not reach is never binned -- the blend would skip it at every pixel --
// and (2) the blend's per-quadrant culling.
__device__ __forceinline__ float ln_upper(float z) {
    const uint32_t b = __float_as_uint(z);
    const int e = (int)(b >> 23) - 127;
    const float m = __uint_as_float((b & 0x7fffffu) | 0x3f800000u);
    float L = (m - 1.0f) * 1.4426951f;
    L = fminf(L, 0.32192809f + (m - 1.25f) * 1.1541561f);
    L = fminf(L, 0.5849625f + (m - 1.5f) * 0.96179669f);
    L = fminf(L, 0.80735492f + (m - 1.75f) * 0.8243972f);
    L = fminf(L, 1.0f + (m - 2.0f) * 0.72134752f);
    return ((float)e + (L + 1e-5f)) * 0.69314724f;
}

__device__ __forceinline__ float2 cull_extent(float ca, float cb, float cc, float op) {
    if (op < 1.0f / 255.0f) return make_float2(-1.0f, -1.0f);
    const float det = ca * cc - cb * cb;
    const float inf = __builtin_inff();
    if (!(det > 0.0f) || !(ca > 0.0f) || !(cc > 0.0f)) return make_float2(inf, inf);
    const float tau = 2.0f * ln_upper(255.0f * op) * 1.001f + 1e-3f;
    const float inv = 1.0f / det;
    return make_float2(sqrtf(tau * cc * inv) * 1.001f + 0.01f, sqrtf(tau * ca * inv) * 1.001f + 0.01f);
}

// tiles t owning a pixel p in [16t, 16t+15] with |p - c| <= h, clipped to [lo, hi)
__device__ __forceinline__ void tight_range(float c, float h, int lo, int hi, int n, int& t0, int& t1) {
    if (h < 0.0f) { t0 = lo; t1 = lo; return; }
    const float a = fminf(fmaxf(ceilf((c - h - (float)(TILE - 1)) / (float)TILE), 0.0f), (float)n);
    const float b = fminf(fmaxf(floorf((c + h) / (float)TILE) + 1.0f, 0.0f), (float)n);
    t0 = max((int)a, lo);
    t1 = min((int)b, hi);
    if (t1 < t0) t1 = t0;
}

__device__ __forceinline__ TileRect tight_rect(TileRect r, float px, float py, float hx, float hy, int gx, int gy) {
    TileRect t;
    tight_range(px, hx, r.x0, r.x1, gx, t.x0, t.x1);
    tight_range(py, hy, r.y0, r.y1, gy, t.y0, t.y1);
    return t;
}

// SHARED: a0 = the stage-A records [slices][P][4 x float4] (see activate_cov_kernel), frames[f].reserved[0] = the frame's slice
template <bool SHARED>
__global__ __launch_bounds__(PRE_THREADS) void preprocess_kernel(
    PreParams pp, const GvfRastFrame* __restrict__ frames, const float* __restrict__ a0,
    const float* __restrict__ a1, const float* __restrict__ a2, const float* __restrict__ a3,
    const float* __restrict__ sh, const float* __restrict__ colors_precomp,
    const float* __restrict__ cov3D_precomp, const float* __restrict__ delta, float4* __restrict__ splats,
    uint32_t* __restrict__ tiles_touched,
    int32_t* __restrict__ radii, uint32_t* __restrict__ block_sums, uint4* __restrict__ binrec,
    const uint32_t* __restrict__ bin_slot /* Gaussian -> position of its bin record inside a frame; null = identity */,
    const float2* __restrict__ zrange /* per frame {z_lo, slabs per unit depth}; null = one slab */) {
    extern __shared__ __attribute__((aligned(16))) float sh_lds[];  // [PRE_THREADS][M*3] + 4 wave sums
    const int t = threadIdx.x;
    const int P = pp.P, M = pp.M;
    const int nbx = (P + PRE_THREADS - 1) / PRE_THREADS;
    int bx = blockIdx.x, by = blockIdx.y;
    if (GVF_PRE_XCD) {
        const int FY = (pp.F + PRE_FB - 1) / PRE_FB, k = (int)(blockIdx.x >> 3);
        bx = (k / FY) * 8 + (int)(blockIdx.x & 7u);
        by = k - (k / FY) * FY;
        if (bx >= nbx) return;                  // the grid is rounded up to whole groups of 8 blocks (workgroup-uniform)
    }
    const int i = bx * PRE_THREADS + t;

    // Stage this block's SH coefficients through LDS with coalesced 16-byte loads: 256 Gaussians x
    // M*3 floats are one contiguous span of the [P][M][3] tensor.
    const int sh_stride = M * 3;
    if (sh != nullptr) {
        const size_t span0 = (size_t)bx * PRE_THREADS * sh_stride;
        const int nvalid = min(PRE_THREADS, P - bx * PRE_THREADS);
        const int total = nvalid * sh_stride;
        const float4* src4 = reinterpret_cast<const float4*>(sh + span0);  // span0*4 B is 16-B aligned
        float4* dst4 = reinterpret_cast<float4*>(sh_lds);
        const int n4 = total >> 2;
        for (int k = t; k < n4; k += PRE_THREADS) dst4[k] = src4[k];
        for (int k = (n4 << 2) + t; k < total; k += PRE_THREADS) sh_lds[k] = sh[span0 + k];
    }
    __syncthreads();

  const uint32_t my_slot = (bin_slot != nullptr && i < P) ? bin_slot[i] : (uint32_t)i;
#if GVF_PRE_PREFETCH
  // the delta row of the NEXT frame of this workgroup is requested before the current frame's arithmetic
  float dnx[14];
  bool dnx_has = false;
  auto fetch_delta = [&](int f_) {
      dnx_has = false;
      if (pp.fused && delta != nullptr && i < P && f_ < pp.F) {
          const int di = frames[f_].delta_index;
          if (di >= 0) {
              const float* d = delta + ((size_t)di * P + i) * 14;
#pragma unroll
              for (int k = 0; k < 14; ++k) dnx[k] = d[k];
              dnx_has = true;
          }
      }
      if (!dnx_has) {
#pragma unroll
          for (int k = 0; k < 14; ++k) dnx[k] = 0.0f;
      }
  };
  fetch_delta(by * PRE_FB);
#endif
  for (int ff = 0; ff < PRE_FB; ++ff) {
    const int f = by * PRE_FB + ff;
    if (f >= pp.F) break;
    const GvfRastFrame* fr = frames + f;
#if GVF_PRE_PREFETCH
    float dcur[14];
#pragma unroll
    for (int k = 0; k < 14; ++k) dcur[k] = dnx[k];
    const bool dcur_has = dnx_has;
    if (ff + 1 < PRE_FB) fetch_delta(f + 1);
#endif
    uint32_t touched = 0;
    int radius_out = 0;
    float4 gA = make_float4(0.f, 0.f, 0.f, 0.f), gB = gA, gC = gA;
    TileRect rect = {0, 0, 0, 0};

    if (i < P) {
        float p[3], s[3], q[4], op, dadd[3] = {0.f, 0.f, 0.f};
        float c6s[6];
        if (SHARED) {
            const float4* r3 = reinterpret_cast<const float4*>(a0) + 4 * ((size_t)fr->reserved[0] * P + i);
            const float4 r0 = r3[0], r1 = r3[1], r2 = r3[2], r3v = r3[3];
            p[0] = r0.x; p[1] = r0.y; p[2] = r0.z; op = r0.w;
            c6s[0] = r1.x; c6s[1] = r1.y; c6s[2] = r1.z; c6s[3] = r1.w; c6s[4] = r2.x; c6s[5] = r2.y;
            dadd[0] = r2.z; dadd[1] = r2.w; dadd[2] = r3v.x;
        } else if (pp.fused) {
#if GVF_PRE_PREFETCH
            ActGaussian g = activate_vals(i, pp.act, a0, a1, a2, a3, dcur, dcur_has);
#else
            const int di = fr->delta_index;
            const float* d = (delta != nullptr && di >= 0) ? delta + ((size_t)di * P + i) * 14 : nullptr;
            ActGaussian g = activate_one(i, pp.act, a0, a1, a2, a3, d);
#endif
#pragma unroll
            for (int k = 0; k < 3; ++k) { p[k] = g.p[k]; s[k] = g.s[k]; dadd[k] = g.drgb[k]; }
#pragma unroll
            for (int k = 0; k < 4; ++k) q[k] = g.q[k];
            op = g.op;
        } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) p[k] = a0[3 * (size_t)i + k];
            if (cov3D_precomp == nullptr) {
#pragma unroll
                for (int k = 0; k < 3; ++k) s[k] = a1[3 * (size_t)i + k];
#pragma unroll
                for (int k = 0; k < 4; ++k) q[k] = a2[4 * (size_t)i + k];
            }
            op = a3[i];
        }

        float pv[3];
        xform43(fr->viewmatrix, p, pv);
        bool vis = pv[2] > 0.2f;
        if (vis) {
            float ph[4];
            xform44(fr->projmatrix, p, ph);
            float pw = 1.0f / (ph[3] + 0.0000001f);
            float projx = ph[0] * pw, projy = ph[1] * pw;

            float c6[6];
            if (SHARED) {
#pragma unroll
                for (int k = 0; k < 6; ++k) c6[k] = c6s[k];
            } else if (!pp.fused && cov3D_precomp != nullptr) {
#pragma unroll
                for (int k = 0; k < 6; ++k) c6[k] = cov3D_precomp[6 * (size_t)i + k];
            } else {
                cov3d_from_scale_rot(s, pp.scale_modifier, q, c6);
            }

            const float tanfovx = fr->tanfovx, tanfovy = fr->tanfovy;
            float focal_x = (float)pp.W / (2.0f * tanfovx);
            float focal_y = (float)pp.H / (2.0f * tanfovy);
            float limx = 1.3f * tanfovx, limy = 1.3f * tanfovy;
            float txtz = pv[0] / pv[2], tytz = pv[1] / pv[2];
            float tx = fminf(limx, fmaxf(-limx, txtz)) * pv[2];
            float ty = fminf(limy, fmaxf(-limy, tytz)) * pv[2];
            float tz = pv[2];
            float J00 = focal_x / tz, J02 = -(focal_x * tx) / (tz * tz);
            float J11 = focal_y / tz, J12 = -(focal_y * ty) / (tz * tz);
            float A0[3], A1[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float w0 = fr->viewmatrix[c * 4 + 0], w1 = fr->viewmatrix[c * 4 + 1], w2 = fr->viewmatrix[c * 4 + 2];
                A0[c] = J00 * w0 + J02 * w2;
                A1[c] = J11 * w1 + J12 * w2;
            }
            float S[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
            float B0[3], B1[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                B0[c] = A0[0] * S[0][c] + A0[1] * S[1][c] + A0[2] * S[2][c];
                B1[c] = A1[0] * S[0][c] + A1[1] * S[1][c] + A1[2] * S[2][c];
            }
            float cxx = B0[0] * A0[0] + B0[1] * A0[1] + B0[2] * A0[2];
            float cxy = B0[0] * A1[0] + B0[1] * A1[1] + B0[2] * A1[2];
            float cyy = B1[0] * A1[0] + B1[1] * A1[1] + B1[2] * A1[2];

            float coef = 1.0f;
            if (pp.mode == GVF_RAST_MODE_MIP) {
                float det0 = fmaxf(1e-6f, cxx * cyy - cxy * cxy);
                float det1 = fmaxf(1e-6f, (cxx + pp.kernel_size) * (cyy + pp.kernel_size) - cxy * cxy);
                coef = sqrtf(det0 / (det1 + 1e-6f) + 1e-6f);
                if (det0 <= 1e-6f || det1 <= 1e-6f) coef = 0.0f;
                cxx += pp.kernel_size;
                cyy += pp.kernel_size;
            } else {
                cxx += 0.3f;
                cyy += 0.3f;
            }
            float det = cxx * cyy - cxy * cxy;
            if (det != 0.0f) {
                float det_inv = 1.f / det;
                float ca = cyy * det_inv, cb = -cxy * det_inv, cc = cxx * det_inv;
                float mid = 0.5f * (cxx + cyy);
                float lam1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
                float lam2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
                float my_radius = ceilf(3.f * sqrtf(fmaxf(lam1, lam2)));
                float px = ((projx + 1.0f) * (float)pp.W - 1.0f) * 0.5f;
                float py = ((projy + 1.0f) * (float)pp.H - 1.0f) * 0.5f;
                TileRect r = get_rect(px, py, my_radius, pp.gx, pp.gy);
                uint32_t cnt = (uint32_t)((r.x1 - r.x0) * (r.y1 - r.y0));
                if (cnt != 0) {
                    float rgb[3];
                    if (colors_precomp != nullptr) {
                        rgb[0] = colors_precomp[3 * (size_t)i + 0];
                        rgb[1] = colors_precomp[3 * (size_t)i + 1];
                        rgb[2] = colors_precomp[3 * (size_t)i + 2];
                    } else {
                        sh_to_rgb(pp.deg, sh_lds + t * sh_stride, dadd, p, fr->campos, rgb);
                    }
                    radius_out = (int)my_radius;
                    const float2 ext = cull_extent(ca, cb, cc, op * coef);
                    if (!pp.upstream_binning) {
                        r = tight_rect(r, px, py, ext.x, ext.y, pp.gx, pp.gy);
                        cnt = (uint32_t)((r.x1 - r.x0) * (r.y1 - r.y0));
                    }
                    touched = cnt;
                    rect = r;
                    gA = make_float4(px, py, ca * CONIC_K1, cb * CONIC_K2);
                    gB = make_float4(cc * CONIC_K1, op * coef, rgb[0], rgb[1]);
                    gC = make_float4(rgb[2], pv[2], ext.x, ext.y);
                }
            }
        }
        const size_t o = (size_t)f * P + i;
#if defined(PRE_ABL_NOREC)      // timing experiment: no record stores (one dword keeps the arithmetic alive)
        if (touched != 0 && gA.x == 12345.678f) splats[4 * o] = gA;
#else
        if (!(GVF_PRE_REC_QUAD && SHARED) && touched != 0) {   // records of culled Gaussians are never read (no instance refers to them)
            float4* rec = splats + 4 * o;
            rec[0] = gA; rec[1] = gB; rec[2] = gC;
        }
#endif
        if (tiles_touched != nullptr) tiles_touched[o] = touched;   // radix binning only
        if (radii != nullptr) {
            // slot order (SHARED with a1 = the slot -> Gaussian table): thread i works on slot i, the radii stay indexed by Gaussian
            const uint32_t* gid_of = SHARED ? reinterpret_cast<const uint32_t*>(a1) : nullptr;
            radii[gid_of != nullptr ? (size_t)f * P + gid_of[i] : o] = radius_out;
        }
        // bucket binning: the final tile rect and the depth, 16 B that the count / scatter passes gather by id
        if (binrec != nullptr) {
            // depth slab: any monotone function of depth keeps the concatenation of the sorted slabs sorted
            uint32_t slab = 0u;
            if (zrange != nullptr) {
                const float2 zr = zrange[f];
                slab = (uint32_t)fminf(fmaxf((gC.y - zr.x) * zr.y, 0.0f), (float)(NSLAB - 1));
            }
#if defined(PRE_ABL_NOBIN)      // timing experiment: no bin-record stores
            if (gC.y == 12345.678f)
#elif defined(PRE_ABL_BINLINEAR) // timing experiment: bin records at the Gaussian's own index (coalesced) instead of its Morton slot
            binrec[(size_t)f * P + i] = make_uint4((uint32_t)rect.x0 | ((uint32_t)rect.y0 << 16), (uint32_t)rect.x1 | ((uint32_t)rect.y1 << 16), __float_as_uint(gC.y), slab);
            if (false)
#endif
            binrec[(size_t)f * P + my_slot] = make_uint4((uint32_t)rect.x0 | ((uint32_t)rect.y0 << 16),
                                                         (uint32_t)rect.x1 | ((uint32_t)rect.y1 << 16), __float_as_uint(gC.y), slab);
        }
    }

#if !defined(PRE_ABL_NOREC)
    if (GVF_PRE_REC_QUAD && SHARED) {
        // Quad-transposed record store (shared-activation launch): lane 4 g + j writes piece j (16 bytes; piece 3 = the padding) of the records of lanes
        // 4 g + k, k = 0 .. 3, so ONE instruction stores 16 whole 64-byte lines where the per-lane form stores 64 quarter lines three times (48 of a
        // line's 64 bytes, masked at the memory side).  The launch is bound by its stores (no record stores: -26 %, no bin-record stores: -21 %,
        // profiles/r05_preprocess_store_ablation.txt); this form: live job 146-148 -> 141-142 ms per sample.  All 64 lanes run it (a lane past P or
        // with a culled Gaussian has touched = 0 and zero pieces), lanes of a quad exchange through DPP quad broadcasts.
        const int lj = t & 3;
        float4* qbase = splats + 4 * ((size_t)f * P + (size_t)(i & ~3));
#define GVF_QB(v_, k_) __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, (v_)), (k_) * 0x55, 0xF, 0xF, true))
#define GVF_QSTORE(k_)                                                                                          \
        {                                                                                                       \
            const int tk = __builtin_amdgcn_mov_dpp((int)touched, (k_) * 0x55, 0xF, 0xF, true);                 \
            float4 o4;                                                                                          \
            { const float a = GVF_QB(gA.x, k_), b = GVF_QB(gB.x, k_), c = GVF_QB(gC.x, k_); o4.x = lj == 0 ? a : (lj == 1 ? b : (lj == 2 ? c : 0.f)); } \
            { const float a = GVF_QB(gA.y, k_), b = GVF_QB(gB.y, k_), c = GVF_QB(gC.y, k_); o4.y = lj == 0 ? a : (lj == 1 ? b : (lj == 2 ? c : 0.f)); } \
            { const float a = GVF_QB(gA.z, k_), b = GVF_QB(gB.z, k_), c = GVF_QB(gC.z, k_); o4.z = lj == 0 ? a : (lj == 1 ? b : (lj == 2 ? c : 0.f)); } \
            { const float a = GVF_QB(gA.w, k_), b = GVF_QB(gB.w, k_), c = GVF_QB(gC.w, k_); o4.w = lj == 0 ? a : (lj == 1 ? b : (lj == 2 ? c : 0.f)); } \
            if (tk != 0) qbase[4 * (k_) + lj] = o4;                                                             \
        }
        GVF_QSTORE(0) GVF_QSTORE(1) GVF_QSTORE(2) GVF_QSTORE(3)
#undef GVF_QSTORE
#undef GVF_QB
    }
#endif

    // block sum of tiles_touched (feeds the instance-offset scan, R2; radix binning only)
    if (block_sums != nullptr) {
        const unsigned lane = t & 63, w = t >> 6;
        uint32_t incl = gvf_wave_incl_scan(touched, lane);
        __shared__ uint32_t wsum[PRE_THREADS / GVF_WAVE];
        __syncthreads();
        if (lane == 63) wsum[w] = incl;
        __syncthreads();
        if (t == 0) block_sums[(size_t)f * nbx + bx] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// R2: exclusive scan of the F*nb block sums (single workgroup), per-frame counts and grand total
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void scan_sums_kernel(uint32_t* __restrict__ block_sums, int nb, int F,
                                                         uint32_t* __restrict__ frame_base /*[F+1]*/,
                                                         uint32_t* __restrict__ num_rendered /*[F]*/,
                                                         uint32_t* __restrict__ total_out, uint32_t max_rendered) {
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry_s;
    const int t = threadIdx.x;
    const unsigned lane = t & 63, w = t >> 6;
    const int n = nb * F;
    if (t == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        int j = base + t;
        uint32_t v = j < n ? block_sums[j] : 0u;
        uint32_t incl = gvf_wave_incl_scan(v, lane);
        if (lane == 63) wsum[w] = incl;
        __syncthreads();
        uint32_t wbase = 0, tot = 0;
        for (unsigned k = 0; k < 16; ++k) { if (k < w) wbase += wsum[k]; tot += wsum[k]; }
        uint32_t carry = carry_s;
        uint32_t excl = carry + wbase + incl - v;
        if (j < n) {
            block_sums[j] = excl;
            if (j % nb == 0) frame_base[j / nb] = excl;
        }
        __syncthreads();
        if (t == 0) carry_s = carry + tot;
        __syncthreads();
    }
    // Overflow (D > workspace capacity): render nothing (n = 0) but still report the true counts, so
    // the caller can detect it from num_rendered and retry with a larger workspace.
    if (t == 0) { frame_base[F] = carry_s; *total_out = carry_s > max_rendered ? 0u : carry_s; }
    __syncthreads();
    for (int f = t; f < F; f += 1024) num_rendered[f] = frame_base[f + 1] - frame_base[f];
}

// ---------------------------------------------------------------------------------------------
// R3: duplicate with keys
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(PRE_THREADS) void duplicate_kernel(
    int P, int gx, int gy, const float4* __restrict__ splats,
    const uint32_t* __restrict__ tiles_touched, const int32_t* __restrict__ radii_ws,
    const uint32_t* __restrict__ block_base, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals,
    uint32_t max_rendered, int upstream_binning) {
    __shared__ uint32_t wsum[PRE_THREADS / GVF_WAVE];
    const int t = threadIdx.x, f = blockIdx.y;
    const int i = blockIdx.x * PRE_THREADS + t;
    const unsigned lane = t & 63, w = t >> 6;
    const size_t o = (size_t)f * P + i;
    uint32_t touched = i < P ? tiles_touched[o] : 0u;
    uint32_t incl = gvf_wave_incl_scan(touched, lane);
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    uint32_t wbase = 0;
    for (unsigned k = 0; k < w; ++k) wbase += wsum[k];
    uint32_t off = block_base[(size_t)f * gridDim.x + blockIdx.x] + wbase + incl - touched;
    if (touched == 0) return;
    if ((uint64_t)off + touched > (uint64_t)max_rendered) return;  // overflow: caller checks num_rendered
    const float4 a = splats[4 * o];
    const float4 c = splats[4 * o + 2];
    const float depth = c.y;
    TileRect r = get_rect(a.x, a.y, (float)radii_ws[o], gx, gy);
    if (!upstream_binning) r = tight_rect(r, a.x, a.y, c.z, c.w, gx, gy);
    const uint64_t tile0 = (uint64_t)f * (uint32_t)(gx * gy);
    const uint32_t dbits = __float_as_uint(depth);
    for (int y = r.y0; y < r.y1; ++y)
        for (int x = r.x0; x < r.x1; ++x) {
            uint64_t key = ((tile0 + (uint32_t)(y * gx + x)) << 32) | dbits;
            keys[off] = key;
            vals[off] = (uint32_t)i;
            ++off;
        }
}

// ---------------------------------------------------------------------------------------------
// R2-R5, bucket form: per-tile counts (preprocess) -> exclusive scan = tile ranges -> scatter into the tile segments.
// No global sort: the order inside a segment is whatever the atomics produced, the per-tile sort below keys on
// (depth, id) and makes it deterministic.
// ---------------------------------------------------------------------------------------------
// order-preserving float <-> uint32 map (atomicMin / atomicMax on floats of either sign)
__device__ __forceinline__ uint32_t float_ordered(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float float_unordered(uint32_t o) {
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

// Exclusive scan of the per-segment counters (segment = (frame, tile, slab); F * tiles * NSLAB of them) in two
// launches: block sums of 4096 counters, then every block adds up the sums in front of it (at most a few hundred
// values) and rescans its own chunk.  Writes segment ranges + cursors, per-frame D, the grand total.
constexpr int SCAN_CHUNK = 4096;
// Size classes of the per-tile sort (R4, below).  SORT_SMALL_N: segments of up to this many keys are sorted in static LDS by tile_sort_kernel<0>, one
// workgroup of 256 threads each.  1536 since the end of round 6 (2048 before): 12 bytes of LDS per key = 18.4 KiB = EIGHT workgroups per CU instead
// of six -- the launch lives on how many segments are in flight (its waves are parked 75 % of the time) --; the few segments of 1537-2048 keys join
// the 512-thread LDS class.  Tile sort 0.137 -> 0.119 ms at the bench shape, the live render job -4 % (profiles/r06_tile_sort_classes.txt; sorting
// the segments of up to 256-512 keys four to a workgroup, one WAVE each, was built and measured on top of it: -3 % of the launch at best, not kept).
#ifndef GVF_SORT_SMALL_N
#define GVF_SORT_SMALL_N 1536
#endif
constexpr int SORT_SMALL_N = GVF_SORT_SMALL_N;
static_assert(SORT_SMALL_N == 1536 || SORT_SMALL_N == 2048, "register class of the per-tile sort: 6 or 8 keys per thread");
constexpr int SORT_LARGE_N = 16384;
constexpr int SORT_LARGE_BLOCKS = 256, SORT_HUGE_BLOCKS = 64;   // grid of the launch that walks the two rare classes
#ifndef SORT_LIST_BIT
#define SORT_LIST_BIT 1
#endif
constexpr int SORT_MEDIUM_N = 4096, SORT_MEDIUM_BLOCKS = 768;   // the LDS class's lower half has a launch of its own (tile_sort_kernel<1>)
__global__ __launch_bounds__(1024) void seg_sums_kernel(const uint32_t* __restrict__ cnt, int n, uint32_t* __restrict__ partial) {
    __shared__ uint32_t wsum[16];
    const int t = threadIdx.x, j0 = blockIdx.x * SCAN_CHUNK + 4 * t;
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) s += j0 + k < n ? cnt[j0 + k] : 0u;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    if ((t & 63) == 0) wsum[t >> 6] = s;
    __syncthreads();
    if (t == 0) { uint32_t tot = 0; for (int k = 0; k < 16; ++k) tot += wsum[k]; partial[blockIdx.x] = tot; }
}

__global__ __launch_bounds__(1024) void seg_scan_kernel(const uint32_t* __restrict__ cnt, int n, int per_frame, int F,
                                                        const uint32_t* __restrict__ partial, int nblocks,
                                                        uint2* __restrict__ ranges, uint32_t* __restrict__ cursor,
                                                        uint32_t* __restrict__ frame_base /*[F+1]*/,
                                                        uint32_t* __restrict__ num_rendered /*[F]*/,
                                                        uint32_t* __restrict__ total_out, uint32_t max_rendered,
                                                        uint32_t* __restrict__ cls /* sort size classes, as classify_kernel */) {
    __shared__ uint32_t wsum[16], wtot[16];
    __shared__ uint32_t s_base, s_total;
    const int t = threadIdx.x;
    const unsigned lane = t & 63, w = t >> 6;
    // offset of this chunk and the grand total from the block sums
    uint32_t before = 0, all = 0;
    for (int k = t; k < nblocks; k += 1024) { const uint32_t p = partial[k]; all += p; if (k < (int)blockIdx.x) before += p; }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { before += __shfl_xor(before, d, 64); all += __shfl_xor(all, d, 64); }
    if (lane == 0) { wsum[w] = before; wtot[w] = all; }
    __syncthreads();
    if (t == 0) {
        uint32_t b = 0, a = 0;
        for (int k = 0; k < 16; ++k) { b += wsum[k]; a += wtot[k]; }
        s_base = b; s_total = a;
    }
    __syncthreads();
    const uint32_t total = s_total;
    const bool overflow = total > max_rendered;
    const int j0 = blockIdx.x * SCAN_CHUNK + 4 * t;
    uint32_t c[4], s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { c[k] = j0 + k < n ? cnt[j0 + k] : 0u; s += c[k]; }
    const uint32_t incl = gvf_wave_incl_scan(s, lane);
    __syncthreads();
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    uint32_t run = s_base + incl - s;
    for (unsigned k = 0; k < w; ++k) run += wsum[k];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int j = j0 + k;
        if (j < n) {
            // Overflow (D > workspace capacity): render nothing, but the true counts let the caller retry.
            ranges[j] = overflow ? make_uint2(0u, 0u) : make_uint2(run, run + c[k]);
            cursor[j] = run;
            if (!overflow && c[k] > (uint32_t)SORT_SMALL_N) {           // rare: a segment for the LDS / global sort classes
                if (c[k] > (uint32_t)SORT_LARGE_N) cls[2 + n + atomicAdd(&cls[1], 1u)] = (uint32_t)j;
                else cls[2 + atomicAdd(&cls[0], 1u)] = (uint32_t)j | (c[k] > (uint32_t)SORT_MEDIUM_N ? 0x80000000u : 0u);   // top bit: the upper half of the LDS class
            }
            if (j % per_frame == 0) frame_base[j / per_frame] = run;
        }
        run += c[k];
    }
    if (blockIdx.x == 0 && t == 0) { frame_base[F] = total; *total_out = overflow ? 0u : total; }
}

// per-frame instance counts from the frame bases (separate tiny launch: needs every block of seg_scan_kernel done)
__global__ void frame_counts_kernel(const uint32_t* __restrict__ frame_base, int F, uint32_t* __restrict__ num_rendered) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f < F) num_rendered[f] = frame_base[f + 1] - frame_base[f];
}

// Depth range of the scene per frame: view-space z of the 8 corners of the Gaussians' bounding box (the Morton
// stage's min/max, mapped through the GaussianModel aabb for raw inputs), widened by 2 %.  Only used to balance the
// depth slabs: instances outside the range land in the first / last slab.
__global__ void frame_zrange_kernel(const GvfRastFrame* __restrict__ frames, int F, const uint32_t* __restrict__ mm,
                                    int fused, GvfGaussianActivation act, float2* __restrict__ zrange) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    float lo[3], hi[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        lo[k] = float_unordered(mm[k]); hi[k] = float_unordered(mm[3 + k]);
        if (fused) { lo[k] = lo[k] * act.aabb[3 + k] + act.aabb[k]; hi[k] = hi[k] * act.aabb[3 + k] + act.aabb[k]; }
        if (lo[k] > hi[k]) { const float tmp = lo[k]; lo[k] = hi[k]; hi[k] = tmp; }
    }
    const float* V = frames[f].viewmatrix;
    float zmin = 3.0e38f, zmax = -3.0e38f;
    for (int c = 0; c < 8; ++c) {
        const float p[3] = {(c & 1) ? hi[0] : lo[0], (c & 2) ? hi[1] : lo[1], (c & 4) ? hi[2] : lo[2]};
        const float z = V[2] * p[0] + V[6] * p[1] + V[10] * p[2] + V[14];
        zmin = fminf(zmin, z); zmax = fmaxf(zmax, z);
    }
    zmin = fmaxf(zmin, 0.2f);
    const float ext = fmaxf(zmax - zmin, 1e-6f);
    zrange[f] = make_float2(zmin - 0.02f * ext, (float)NSLAB / (1.04f * ext));
}

// Count / scatter passes over the compact bin records, BIN_SPT slots per thread, slots taken in Morton order
// (`order` = the Gaussian of each slot, may be null = identity; the records themselves are stored by slot): the rects of one block then fall into a small window of tiles, instances are
// counted in an LDS table and every touched tile costs ONE global atomic per block (count pass: += tile_count;
// scatter pass: cursor allocation, the base is left in the table and an LDS counter hands out the slots).  A block
// whose window exceeds WIN_MAX tiles (incoherent order) pays one global atomic per instance instead.
constexpr int WIN_MAX = 2048;
constexpr int BIN_SPT = 4;
constexpr int BIN_SLOTS = PRE_THREADS * BIN_SPT;

template <bool SCATTER>
__global__ __launch_bounds__(PRE_THREADS) void bin_kernel(int P, int gx, int gy, const uint4* __restrict__ binrec,
                                                          const uint32_t* __restrict__ order,
                                                          uint32_t* __restrict__ tile_count /* count pass */,
                                                          uint32_t* __restrict__ cursor /* scatter pass */,
                                                          const uint32_t* __restrict__ total,
                                                          uint64_t* __restrict__ payload, int nslab,
                                                          const uint32_t* __restrict__ frame_base, uint32_t* __restrict__ num_rendered) {
    __shared__ uint32_t s_tab[WIN_MAX];
    __shared__ uint32_t s_run[SCATTER ? WIN_MAX : 1];
    __shared__ int s_box[4];
    // scatter pass: the per-frame instance counts from the frame bases the scan left (saves a launch of its own)
    if (SCATTER && num_rendered != nullptr && blockIdx.x == 0 && threadIdx.x == 0)
        num_rendered[blockIdx.y] = frame_base[blockIdx.y + 1] - frame_base[blockIdx.y];
    if (SCATTER && *total == 0u) return;            // nothing visible, or capacity overflow (uniform)
    const int t = threadIdx.x, lane = t & 63, f = blockIdx.y;
    if (t == 0) { s_box[0] = 0x7fffffff; s_box[1] = 0x7fffffff; s_box[2] = 0; s_box[3] = 0; }
    int x0[BIN_SPT], y0[BIN_SPT], x1[BIN_SPT], y1[BIN_SPT], sl[BIN_SPT];
    uint64_t key[BIN_SPT];
    int bx0 = 0x7fffffff, by0 = 0x7fffffff, bx1 = 0, by1 = 0;
#pragma unroll
    for (int k = 0; k < BIN_SPT; ++k) {
        const int s = blockIdx.x * BIN_SLOTS + k * PRE_THREADS + t;
        x0[k] = y0[k] = x1[k] = y1[k] = 0; sl[k] = 0; key[k] = 0;
        if (s < P) {
            const uint32_t id = order != nullptr ? order[s] : (uint32_t)s;
            const uint4 br = binrec[(size_t)f * P + s];                // records sit at their slots (preprocess_kernel, bin_slot)
            x0[k] = (int)(br.x & 0xffffu); y0[k] = (int)(br.x >> 16);
            x1[k] = (int)(br.y & 0xffffu); y1[k] = (int)(br.y >> 16);
            sl[k] = (int)br.w;
            key[k] = ((uint64_t)br.z << 32) | id;                      // depth bits above the Gaussian id
            if (x1[k] > x0[k] && y1[k] > y0[k]) {
                bx0 = min(bx0, x0[k]); by0 = min(by0, y0[k]); bx1 = max(bx1, x1[k]); by1 = max(by1, y1[k]);
            } else {
                x1[k] = x0[k];                                          // empty: the loops below do nothing
            }
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        bx0 = min(bx0, __shfl_xor(bx0, d, 64)); by0 = min(by0, __shfl_xor(by0, d, 64));
        bx1 = max(bx1, __shfl_xor(bx1, d, 64)); by1 = max(by1, __shfl_xor(by1, d, 64));
    }
    __syncthreads();                                 // s_box initialised
    if (lane == 0 && bx1 > bx0) {
        atomicMin(&s_box[0], bx0); atomicMin(&s_box[1], by0); atomicMax(&s_box[2], bx1); atomicMax(&s_box[3], by1);
    }
    __syncthreads();
    const int wx0 = s_box[0], wy0 = s_box[1], ww = s_box[2] - s_box[0], wh = s_box[3] - s_box[1];
    if (ww <= 0 || wh <= 0) return;                  // no instance in this block (uniform)
    uint32_t* gtab = (SCATTER ? cursor : tile_count) + (size_t)f * gx * gy * nslab;   // [tile][slab]
    if (ww * wh * nslab <= WIN_MAX) {
        const int area = ww * wh * nslab;
        for (int e = t; e < area; e += PRE_THREADS) { s_tab[e] = 0u; if (SCATTER) s_run[e] = 0u; }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < BIN_SPT; ++k)
            for (int y = y0[k]; y < y1[k]; ++y)
                for (int x = x0[k]; x < x1[k]; ++x) atomicAdd(&s_tab[((y - wy0) * ww + (x - wx0)) * nslab + sl[k]], 1u);
        __syncthreads();
        for (int e = t; e < area; e += PRE_THREADS) {
            const uint32_t c = s_tab[e];
            if (c != 0u) {
                const int wt = e / nslab;
                const int seg = ((wy0 + wt / ww) * gx + wx0 + wt % ww) * nslab + (e - wt * nslab);
                if (SCATTER) s_tab[e] = atomicAdd(&gtab[seg], c);
                else atomicAdd(&gtab[seg], c);
            }
        }
        if (SCATTER) {
            __syncthreads();
#pragma unroll
            for (int k = 0; k < BIN_SPT; ++k)
                for (int y = y0[k]; y < y1[k]; ++y)
                    for (int x = x0[k]; x < x1[k]; ++x) {
                        const int e = ((y - wy0) * ww + (x - wx0)) * nslab + sl[k];
                        payload[s_tab[e] + atomicAdd(&s_run[e], 1u)] = key[k];
                    }
        }
    } else {
#pragma unroll
        for (int k = 0; k < BIN_SPT; ++k)
            for (int y = y0[k]; y < y1[k]; ++y)
                for (int x = x0[k]; x < x1[k]; ++x) {
                    const uint32_t pos = atomicAdd(&gtab[(y * gx + x) * nslab + sl[k]], 1u);
                    if (SCATTER) payload[pos] = key[k];
                }
    }
}

// ---------------------------------------------------------------------------------------------
// Spatial order of the Gaussians (once per call, shared by all frames): 15-bit Morton code of the position inside
// the bounding box, counting-sorted.  Purely a locality device for the bin passes: any order gives the same image.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bbox_kernel(int P, const float* __restrict__ xyz, uint32_t* __restrict__ mm) {
    __shared__ uint32_t s_mm[6];
    if (threadIdx.x < 6) s_mm[threadIdx.x] = threadIdx.x < 3 ? 0xffffffffu : 0u;
    __syncthreads();
    uint32_t lo[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, hi[3] = {0u, 0u, 0u};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P; i += gridDim.x * blockDim.x) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float v = xyz[3 * (size_t)i + k];
            if (v == v && fabsf(v) < 3.0e38f) {
                const uint32_t o = float_ordered(v);
                lo[k] = min(lo[k], o); hi[k] = max(hi[k], o);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            lo[k] = min(lo[k], (uint32_t)__shfl_xor((int)lo[k], d, 64));
            hi[k] = max(hi[k], (uint32_t)__shfl_xor((int)hi[k], d, 64));
        }
        if ((threadIdx.x & 63) == 0) { atomicMin(&s_mm[k], lo[k]); atomicMax(&s_mm[3 + k], hi[k]); }
    }
    __syncthreads();
    if (threadIdx.x < 3) atomicMin(&mm[threadIdx.x], s_mm[threadIdx.x]);
    else if (threadIdx.x < 6) atomicMax(&mm[threadIdx.x], s_mm[threadIdx.x]);
}

__device__ __forceinline__ uint32_t spread5(uint32_t v) {   // 5 bits -> every third bit
    return (v & 1u) | ((v & 2u) << 2) | ((v & 4u) << 4) | ((v & 8u) << 6) | ((v & 16u) << 8);
}

constexpr int MORTON_BINS = 1 << 15;

__device__ __forceinline__ uint32_t morton15(int i, const float* __restrict__ xyz, const uint32_t* __restrict__ mm) {
    uint32_t code = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float lo = float_unordered(mm[k]), hi = float_unordered(mm[3 + k]);
        const float v = xyz[3 * (size_t)i + k];
        float q = (v - lo) / fmaxf(hi - lo, 1e-30f) * 32.0f;
        q = (q == q) ? fminf(fmaxf(q, 0.0f), 31.0f) : 0.0f;
        code |= spread5((uint32_t)q) << k;
    }
    return code;
}

// counting sort by Morton cell: histogram -> exclusive scan -> scatter (order inside a cell is arbitrary)
__global__ __launch_bounds__(256) void morton_count_kernel(int P, const float* __restrict__ xyz,
                                                           const uint32_t* __restrict__ mm, uint32_t* __restrict__ codes,
                                                           uint32_t* __restrict__ hist) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const uint32_t c = morton15(i, xyz, mm);
    codes[i] = c;
    atomicAdd(&hist[c], 1u);
}

__global__ __launch_bounds__(1024) void morton_scan_kernel(uint32_t* __restrict__ hist) {
    __shared__ uint32_t wsum[16];
    const int t = threadIdx.x;
    const unsigned lane = t & 63, w = t >> 6;
    constexpr int PER = MORTON_BINS / 1024;
    uint32_t v[PER], s = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) { v[k] = hist[t * PER + k]; s += v[k]; }
    const uint32_t incl = gvf_wave_incl_scan(s, lane);
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    uint32_t run = incl - s;
    for (unsigned k = 0; k < w; ++k) run += wsum[k];
#pragma unroll
    for (int k = 0; k < PER; ++k) { hist[t * PER + k] = run; run += v[k]; }
}

// codes_rank: in = the Gaussian's Morton code, out = its slot in the order (the inverse permutation: preprocess writes the bin
// records at their slots, so the bin passes read them as one contiguous stream instead of gathering a 64-byte line per 16-byte record)
__global__ __launch_bounds__(256) void morton_scatter_kernel(int P, uint32_t* codes_rank, uint32_t* __restrict__ hist,
                                                             uint32_t* __restrict__ order) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const uint32_t slot = atomicAdd(&hist[codes_rank[i]], 1u);
    order[slot] = (uint32_t)i;
    codes_rank[i] = slot;
}

// ---------------------------------------------------------------------------------------------
// R5: tile ranges
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ranges_kernel(const uint64_t* __restrict__ keys,
                                                     const uint32_t* __restrict__ n_ptr, uint32_t n_cap,
                                                     uint2* __restrict__ ranges, uint32_t n_ranges) {
    uint32_t n = *n_ptr;
    n = n < n_cap ? n : n_cap;
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        uint32_t cur = (uint32_t)(keys[k] >> 32);
        if (cur >= n_ranges) continue;
        if (k == 0) ranges[cur].x = 0;
        else {
            uint32_t prev = (uint32_t)(keys[k - 1] >> 32);
            if (cur != prev) { if (prev < n_ranges) ranges[prev].y = k; ranges[cur].x = k; }
        }
        if (k == n - 1) ranges[cur].y = n;
    }
}

// ---------------------------------------------------------------------------------------------
// R4 (second half): per-tile sort.  The radix sort above only ordered the instances by (frame, tile) -- two 8-bit
// passes instead of six; one workgroup per (frame, tile) now sorts its segment by (depth, id) with a bitonic
// network on chip and writes the Gaussian ids in order (identical to upstream's stable (tile, depth) sort).  Three size classes share the code: segments up to
// SMALL_N keys in 16 KiB of static LDS (256 threads; the common case, ~460 keys per tile at the bench shape),
// up to LARGE_N keys in 128 KiB of dynamic LDS (1024 threads), anything larger in place in global memory
// (slow, correct: a whole scene projected onto one tile).  All three are launched over all tiles; a
// workgroup whose segment is not in its class exits at once.
// ---------------------------------------------------------------------------------------------
#ifndef SORT_BUCKETS
#define SORT_BUCKETS 1            // 0: every small segment through the sorting network (the round-1 path)
#endif

// Bitonic sorting network in its "all comparators ascending" form (the first step of every merge compares
// mirrored partners i <-> block_end - i, the remaining steps are the usual half-cleaners).  Because every
// compare-exchange puts the larger key at the higher index, virtual +inf padding above n never moves: pairs
// whose upper index is >= n are simply skipped, so n need not be a power of two and nothing is padded.
template <typename Ptr>
__device__ __forceinline__ void bitonic_sort_asc(Ptr keys, int n, int tid, int nthreads) {
    int npad = 2;
    while (npad < n) npad <<= 1;
    const int half = npad >> 1;
    for (int k = 2; k <= npad; k <<= 1) {
        for (int i = tid; i < half; i += nthreads) {
            const int blk = i / (k >> 1), off = i % (k >> 1);
            const int lo = blk * k + off, hi = blk * k + k - 1 - off;
            if (hi < n) {
                const uint64_t a = keys[lo], b = keys[hi];
                if (a > b) { keys[lo] = b; keys[hi] = a; }
            }
        }
        __syncthreads();
        for (int j = k >> 2; j > 0; j >>= 1) {
            for (int i = tid; i < half; i += nthreads) {
                const int lo = 2 * i - (i & (j - 1));
                const int hi = lo + j;
                if (hi < n) {
                    const uint64_t a = keys[lo], b = keys[hi];
                    if (a > b) { keys[lo] = b; keys[hi] = a; }
                }
            }
            __syncthreads();
        }
    }
}

// Small segments (<= SORT_SMALL_N keys, i.e. practically every tile): E = npad / 256 keys per thread live in REGISTERS
// (key index e = tid * E + r).  Same all-ascending network as above: every step pairs e with e ^ m (m = k - 1 for the
// mirrored first step of a merge, m = j for the half-cleaners), the lower index keeps the minimum.  Partners are in
// the same thread (m < E), the same wave (one 64-bit lane exchange, no LDS, no barrier) or another wave (LDS round
// trip).  The +inf padding above n never moves, so a wave that holds nothing but padding (wave 3 for n <= 1536, wave
// 2 for n <= 1024 at E = 8: the typical dense tile has ~1100 keys) skips everything except the barriers.
template <int E, int NP, bool OUT_LDS = false>
__device__ __forceinline__ void tile_sort_regs(const uint64_t* __restrict__ k, const uint32_t* __restrict__ v,
                                               uint32_t* __restrict__ ids, int n, uint64_t* __restrict__ lds) {
    // OUT_LDS: leave the sorted 64-bit keys in lds[0, n) (for the two-run merge below) instead of writing the ids
    // v == nullptr: k already holds (depth bits << 32 | id) (bucket binning); else k = (tile << 32 | depth), v = id
    // NP <= 256 * E keys take part (threads >= NP / E only ever hold padding and idle with their wave)
    const int tid = threadIdx.x;
    const bool live = (tid & ~63) * E < n;                  // this wave holds at least one real key (wave-uniform)
    uint64_t key[E];
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const int e = tid * E + r;
        key[r] = e < n ? (v != nullptr ? ((k[e] << 32) | v[e]) : k[e]) : ~0ull;    // depth bits above the Gaussian id
    }
#pragma unroll
    for (int k2 = 2; k2 <= NP; k2 <<= 1) {
#pragma unroll
        for (int step = 0, j = k2 >> 1; j > 0; ++step, j >>= 1) {
            const int m = step == 0 ? k2 - 1 : j;           // xor mask in key-index space
            const int mr = m & (E - 1), mt = m / E;         // ... on the register index / on the thread index
            if (mt == 0) {
                if (live) {
#pragma unroll
                    for (int r = 0; r < E; ++r) {
                        if (r < (r ^ mr)) {
                            const uint64_t a = key[r], b = key[r ^ mr];
                            if (a > b) { key[r] = b; key[r ^ mr] = a; }
                        }
                    }
                }
            } else if (mt < 64) {
                if (live) {
                    const int hb = step == 0 ? (k2 / E) >> 1 : mt;          // highest set bit of mt
                    const bool lower = (tid & hb) == 0;
                    uint64_t other[E];
#pragma unroll
                    for (int r = 0; r < E; ++r) {
                        const uint64_t src = key[r ^ mr];
                        const unsigned lo = __shfl_xor((unsigned)src, mt, 64);
                        const unsigned hi = __shfl_xor((unsigned)(src >> 32), mt, 64);
                        other[r] = ((uint64_t)hi << 32) | lo;
                    }
#pragma unroll
                    for (int r = 0; r < E; ++r)
                        key[r] = lower ? (other[r] < key[r] ? other[r] : key[r]) : (other[r] > key[r] ? other[r] : key[r]);
                }
            } else {
                __syncthreads();
                if (live) {
#pragma unroll
                    for (int r = 0; r < E; ++r) lds[tid * E + r] = key[r];
                }
                __syncthreads();
                if (live) {
#pragma unroll
                    for (int r = 0; r < E; ++r) {
                        const int e = tid * E + r, pe = e ^ m;
                        if (pe < n) {                       // partner above n is +inf: an upper partner changes nothing,
                            const uint64_t other = lds[pe]; // and e < n <= pe cannot be the upper side
                            key[r] = e < pe ? (other < key[r] ? other : key[r]) : (other > key[r] ? other : key[r]);
                        }
                    }
                }
            }
        }
    }
    if (OUT_LDS) {
        __syncthreads();                                    // the last exchange step may still be reading lds
#pragma unroll
        for (int r = 0; r < E; ++r) {
            const int e = tid * E + r;
            if (e < n) lds[e] = key[r];
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const int e = tid * E + r;
        if (e < n) ids[e] = (uint32_t)key[r];
    }
}

// Distribution sort of one small segment (the common path since round 2; the network above is the fallback).
// The keys of a (frame, tile) segment are (depth bits << 32 | id) with depths spread over [z_lo, z_hi] of the tile, so
//   bucket(key) = min(NB - 1, int(float(bits - bits_lo) * (NB / float(bits_hi - bits_lo))))      NB = 256 E >= n buckets,
// bits = the depth's bit pattern as an unsigned integer (the high word of the key),
// is a monotone function of the key (unsigned subtract, int -> float, multiply by a positive constant, truncate and clamp all are)
// for ANY key values -- no assumption on sign or finiteness of the depth --, i.e. every
// key of bucket b sorts before every key of bucket b + 1, and a bucket holds ~1 key on average: a histogram (one LDS
// atomic per key, which also hands out the key's slot inside its bucket), an exclusive scan of NB counters, a scatter into
// bucket order, and -- exactness -- each key's rank inside its own bucket by counting the smaller 64-bit keys there.
// ~60 instructions per key instead of the ~300 of the 55-round network at 1024 keys.  Keys are unique (they end in the id), so
// the ranks are a permutation.  A bucket longer than BKT_MAX_RUN (many splats at one depth: a wall facing the camera) makes
// the counting quadratic: the workgroup then returns false and its segment goes through the network (exact for any input).
constexpr int BKT_MAX_RUN = 40;
constexpr int BKT_AUX = 64;            // per wave: minimum, maximum, total, longest run (4 x up to 16 waves)
constexpr int BKT_LARGE_NB = 4096;     // buckets of the 512- / 1024-thread classes (SORT_SMALL_N + 1 .. 16384 keys: 0.4 .. 4 keys per bucket)

#ifdef SORT_STATS
__device__ unsigned long long g_sort_stats[16];
extern "C" int gvf_debug_sort_stats(unsigned long long* out16, int reset) {
    if (out16 != nullptr && hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_sort_stats), sizeof(g_sort_stats)) != hipSuccess) return 1;
    if (reset) { unsigned long long z[16] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_sort_stats), z, sizeof(z)) != hipSuccess) return 1; }
    return 0;
}
#endif
// E keys per thread, T threads, C counters per thread: n <= T E keys into NB = T C buckets
template <int E, int T, int C>
__device__ __forceinline__ bool tile_sort_buckets(const uint64_t* __restrict__ k, const uint32_t* __restrict__ v,
                                                  uint32_t* __restrict__ ids, int n, uint64_t* __restrict__ s_keys /*[T E]*/,
                                                  uint32_t* __restrict__ s_hist /*[NB + 1 + BKT_AUX]*/) {
    constexpr int NB = T * C, NW = T / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t* s_aux = s_hist + NB + 1;
    uint64_t key[E];
    uint32_t dmin = ~0u, dmax = 0u;
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const int e = tid + T * r;
        key[r] = e < n ? (v != nullptr ? ((k[e] << 32) | v[e]) : k[e]) : 0ull;
        if (e < n) {
            const uint32_t d = (uint32_t)(key[r] >> 32);
            dmin = min(dmin, d);
            dmax = max(dmax, d);
        }
    }
#pragma unroll
    for (int i = 0; i < C; ++i) s_hist[tid + T * i] = 0u;
    dmin = gvf_wave_umin(dmin);
    dmax = gvf_wave_umax(dmax);
    if (lane == 0) { s_aux[wave] = dmin; s_aux[NW + wave] = dmax; }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < NW; ++w) { dmin = min(dmin, s_aux[w]); dmax = max(dmax, s_aux[NW + w]); }
    // buckets are linear in the BIT PATTERN of the depth (as an unsigned integer, the way the key itself orders): monotone for any
    // key whatsoever, and for the positive depths of a frame (near cull 0.2) piecewise linear in the depth itself
    const float scale = dmax > dmin ? (float)NB / (float)(dmax - dmin) : 0.0f;     // one depth: everything in bucket 0
    uint32_t bkt[E], slot[E];
#pragma unroll
    for (int r = 0; r < E; ++r) {
        if (tid + T * r < n) {
            bkt[r] = (uint32_t)min(NB - 1, (int)((float)((uint32_t)(key[r] >> 32) - dmin) * scale));
            slot[r] = atomicAdd(&s_hist[bkt[r]], 1u);
        }
    }
    __syncthreads();
    // exclusive scan of the NB counters: thread t owns counters [t C, (t + 1) C)
    uint32_t cnt[C], tot = 0u, run = 0u;
#pragma unroll
    for (int i = 0; i < C; ++i) {
        cnt[i] = s_hist[tid * C + i];
        tot += cnt[i];
        run = max(run, cnt[i]);
    }
    const uint32_t incl = gvf_wave_incl_scan_dpp(tot);
    run = gvf_wave_umax(run);
    if (lane == 63) s_aux[2 * NW + wave] = incl;
    if (lane == 0) s_aux[3 * NW + wave] = run;
    __syncthreads();
    uint32_t base = incl - tot, longest = 0u;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        base += w < wave ? s_aux[2 * NW + w] : 0u;
        longest = max(longest, s_aux[3 * NW + w]);
    }
#ifdef SORT_STATS            // measurement builds only: [0] segments through the distribution sort, [1] of them sent to the network (crowded bucket),
                             // [2] keys of [0], [3] keys of [1], [4 + min(11, longest / 8)] histogram of the longest bucket
    if (tid == 0) {
        atomicAdd(&g_sort_stats[0], 1ull); atomicAdd(&g_sort_stats[2], (unsigned long long)n);
        if (longest > (uint32_t)BKT_MAX_RUN) { atomicAdd(&g_sort_stats[1], 1ull); atomicAdd(&g_sort_stats[3], (unsigned long long)n); }
        atomicAdd(&g_sort_stats[4 + min(11u, longest >> 3)], 1ull);
    }
#endif
    if (longest > (uint32_t)BKT_MAX_RUN) return false;                  // workgroup-uniform
#pragma unroll
    for (int i = 0; i < C; ++i) {
        s_hist[tid * C + i] = base;
        base += cnt[i];
    }
    if (tid == T - 1) s_hist[NB] = (uint32_t)n;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < E; ++r)
        if (tid + T * r < n) s_keys[s_hist[bkt[r]] + slot[r]] = key[r];
    __syncthreads();
    // position p of the bucket-ordered array: neighbouring lanes sit in the same or the next bucket (broadcast LDS reads, and ids
    // written next to each other)
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const int p = tid + T * r;
        if (p < n) {
            const uint64_t mine = s_keys[p];
            const int b = min(NB - 1, (int)((float)((uint32_t)(mine >> 32) - dmin) * scale));
            const uint32_t lo = s_hist[b], hi = s_hist[b + 1];
            uint32_t rank = lo;
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j)         // a bucket holds ~1 key: four independent reads (clamped into the array), then the rest
                rank += (lo + j < hi && s_keys[min(lo + j, (uint32_t)(T * E) - 1u)] < mine) ? 1u : 0u;
            for (uint32_t j = lo + 4; j < hi; ++j) rank += s_keys[j] < mine ? 1u : 0u;
            ids[rank] = (uint32_t)mine;
        }
    }
    return true;
}

// number of keys < x in the sorted run a[0, n)
__device__ __forceinline__ int lower_bound_u64(const uint64_t* a, int n, uint64_t x) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] < x) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// tile lists of the two rare size classes, filled by classify_kernel: [0] count large, [1] count huge, then indices
__global__ __launch_bounds__(256) void classify_kernel(const uint2* __restrict__ ranges, uint32_t nseg,
                                                       uint32_t* __restrict__ cls /*[2 + 2*nseg]*/) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nseg) return;
    const uint2 r = ranges[i];
    const uint32_t n = r.y - r.x;
    if (n > (uint32_t)SORT_LARGE_N) cls[2 + nseg + atomicAdd(&cls[1], 1u)] = i;
    else if (n > (uint32_t)SORT_SMALL_N) cls[2 + atomicAdd(&cls[0], 1u)] = i | (n > (uint32_t)SORT_MEDIUM_N ? 0x80000000u : 0u);
}

template <int MODE>   // 0: small (registers + shuffles), 1: large (dynamic LDS), 2: huge (global, in place), 3: the large class's lower half
__global__ __launch_bounds__(MODE == 3 ? 512 : 1024, MODE == 3 ? 2 : 1) void tile_sort_kernel(const uint2* __restrict__ ranges, uint64_t* __restrict__ keys,
                                 const uint32_t* __restrict__ vals, uint32_t* __restrict__ ids,
                                 const uint32_t* __restrict__ cls, uint32_t nseg, uint32_t split = 0u /*MODE 1: MODE 3 ran too*/) {
    __shared__ uint64_t s_small[MODE == 0 ? SORT_SMALL_N : 1];
    __shared__ uint32_t s_hist[MODE == 0 ? SORT_SMALL_N + 1 + BKT_AUX : (MODE == 1 || MODE == 3 ? BKT_LARGE_NB + 1 + BKT_AUX : 1)];
    extern __shared__ __attribute__((aligned(16))) uint64_t s_large[];
    if (MODE == 0) {
        const uint2 rng = ranges[blockIdx.x];
        const int n = (int)(rng.y - rng.x);
        if (n <= 0 || n > SORT_SMALL_N) return;
        const uint64_t* k = keys + rng.x;
        const uint32_t* v = vals != nullptr ? vals + rng.x : nullptr;
        uint32_t* o = ids + rng.x;
        if (SORT_BUCKETS && n > 128) {               // distribution sort; false = a long run of near-equal depths, take the network
            bool done;
            if (n <= 256) done = tile_sort_buckets<1, 256, 1>(k, v, o, n, s_small, s_hist);
            else if (n <= 512) done = tile_sort_buckets<2, 256, 2>(k, v, o, n, s_small, s_hist);
            else if (n <= 768) done = tile_sort_buckets<3, 256, 3>(k, v, o, n, s_small, s_hist);
            else if (n <= 1024) done = tile_sort_buckets<4, 256, 4>(k, v, o, n, s_small, s_hist);
            else if (n <= 1280) done = tile_sort_buckets<5, 256, 5>(k, v, o, n, s_small, s_hist);
            else if (n <= 1536 || SORT_SMALL_N == 1536) done = tile_sort_buckets<6, 256, 6>(k, v, o, n, s_small, s_hist);
            else done = tile_sort_buckets<SORT_SMALL_N / 256, 256, SORT_SMALL_N / 256>(k, v, o, n, s_small, s_hist);
            if (done) return;
            __syncthreads();
        }
        if (n <= 64) tile_sort_regs<1, 64>(k, v, o, n, s_small);
        else if (n <= 128) tile_sort_regs<1, 128>(k, v, o, n, s_small);
        else if (n <= 256) tile_sort_regs<1, 256>(k, v, o, n, s_small);
        else if (n <= 512) tile_sort_regs<2, 512>(k, v, o, n, s_small);
        else if (n <= 1024) tile_sort_regs<4, 1024>(k, v, o, n, s_small);
        else {
            // 1024 < n <= SORT_SMALL_N.  One 2048-key network would cost 66 rounds x 8 keys per thread even for 1025 keys (and
            // dense tiles sit just above 1024: 61 % of all keys at the bench shape are in segments of 1025-1280).
            // Instead: sort the first 1024 keys and the remaining n - 1024 as two runs (55 rounds x 4 keys + a small
            // network), then merge by rank -- keys are unique (they end in the Gaussian id), so an element's final
            // position is its index in its own run plus the number of smaller keys in the other run.
            const int nb = n - 1024;
            const uint32_t* vb = v != nullptr ? v + 1024 : nullptr;
            tile_sort_regs<4, 1024, true>(k, v, o, 1024, s_small);
            if (nb <= 64) tile_sort_regs<1, 64, true>(k + 1024, vb, o, nb, s_small + 1024);
            else if (nb <= 128) tile_sort_regs<1, 128, true>(k + 1024, vb, o, nb, s_small + 1024);
            else if (nb <= 256) tile_sort_regs<1, 256, true>(k + 1024, vb, o, nb, s_small + 1024);
            else if (nb <= 512 || SORT_SMALL_N == 1536) tile_sort_regs<2, 512, true>(k + 1024, vb, o, nb, s_small + 1024);
            else tile_sort_regs<4, 1024, true>(k + 1024, vb, o, nb, s_small + 1024);
            __syncthreads();
            for (int e = threadIdx.x; e < n; e += 256) {
                const uint64_t key = s_small[e];
                const int pos = e < 1024 ? e + lower_bound_u64(s_small + 1024, nb, key)
                                         : (e - 1024) + lower_bound_u64(s_small, 1024, key);
                o[pos] = (uint32_t)key;
            }
        }
        return;
    }
    // rare classes: a small fixed grid walks the lists built by classify_kernel / seg_scan_kernel.  MODE 1 is launched with
    // SORT_LARGE_BLOCKS + SORT_HUGE_BLOCKS blocks: the first take the LDS class (one workgroup per CU: 145 KB of LDS), the rest the
    // global class (one launch for both: at the bench shape they are empty)
    // The LDS class is walked by TWO launches over the same list: MODE 3 takes the segments of up to SORT_MEDIUM_N keys with workgroups of
    // 512 threads and 32 KiB of dynamic LDS, i.e. three per CU (a segment is a chain of ~6 barriers and LDS round trips: the other
    // workgroups fill one's waits), MODE 1 the rest with 1024 threads and the full 128 KiB.  At 512 x 512 with 262 144 Gaussians
    // about a fifth of the tiles are in this class (the reference's live render job), at 800 x 800 none.
    const bool huge = MODE == 2 || (MODE == 1 && blockIdx.x >= (uint32_t)SORT_LARGE_BLOCKS);
    const uint32_t bid = (MODE == 1 && huge) ? blockIdx.x - SORT_LARGE_BLOCKS : blockIdx.x;
    const uint32_t stride = MODE == 1 ? (huge ? (uint32_t)SORT_HUGE_BLOCKS : (uint32_t)SORT_LARGE_BLOCKS) : gridDim.x;
    const uint32_t count = cls[huge ? 1 : 0];
    const uint32_t* list = cls + 2 + (huge ? nseg : 0u);
    const int tid = threadIdx.x, nt = blockDim.x;
    for (uint32_t li = bid; li < count; li += stride) {
        // the LDS class's list says in its top bit which half an entry belongs to: the launch that does NOT own a segment skips it on the list word
        // alone (round 5: it used to read the segment's range first -- a dependent global load per skipped entry; the 1024-thread launch walked
        // ~80 entries per workgroup to find its few: 75 us per live chunk, mostly that)
        const uint32_t ent = list[li];
        const bool upper = !huge && (ent >> 31) != 0u;
#if SORT_LIST_BIT
        if (!huge && ((MODE == 3 && upper) || (MODE == 1 && split != 0u && !upper))) continue;       // the other launch's segment
#endif
        const uint2 rng = ranges[huge ? ent : (ent & 0x7fffffffu)];
        const int n = (int)(rng.y - rng.x);
#if !SORT_LIST_BIT                 // A/B switch: the round-4 form (decide on the range)
        if ((MODE == 3 && n > SORT_MEDIUM_N) || (MODE == 1 && !huge && split != 0u && n <= SORT_MEDIUM_N)) continue;
#endif
        uint64_t* k = keys + rng.x;
        const uint32_t* v = vals != nullptr ? vals + rng.x : nullptr;
        if (huge) {
            if (v != nullptr)
                for (int i = tid; i < n; i += nt) k[i] = (k[i] << 32) | v[i];   // in place in global memory
            __syncthreads();
            bitonic_sort_asc(k, n, tid, nt);
            for (int i = tid; i < n; i += nt) ids[rng.x + i] = (uint32_t)k[i];
        } else {
            if (MODE == 3 && SORT_BUCKETS) {             // the distribution sort on 512 threads
                const bool sorted = tile_sort_buckets<8, 512, BKT_LARGE_NB / 512>(k, v, ids + rng.x, n, s_large, s_hist);
                __syncthreads();
                if (sorted) continue;
            }
            if (MODE == 1 && SORT_BUCKETS) {             // the distribution sort on 1024 threads; false = crowded bucket, take the network
                bool sorted;
                if (n <= 4096) sorted = tile_sort_buckets<4, 1024, BKT_LARGE_NB / 1024>(k, v, ids + rng.x, n, s_large, s_hist);
                else if (n <= 8192) sorted = tile_sort_buckets<8, 1024, BKT_LARGE_NB / 1024>(k, v, ids + rng.x, n, s_large, s_hist);
                else sorted = tile_sort_buckets<16, 1024, BKT_LARGE_NB / 1024>(k, v, ids + rng.x, n, s_large, s_hist);
                __syncthreads();
                if (sorted) continue;
            }
            for (int i = tid; i < n; i += nt) s_large[i] = v != nullptr ? ((k[i] << 32) | v[i]) : k[i];
            __syncthreads();
            bitonic_sort_asc(s_large, n, tid, nt);
            for (int i = tid; i < n; i += nt) ids[rng.x + i] = (uint32_t)s_large[i];
        }
        __syncthreads();
    }
}

// classify + the three size classes of the per-tile sort over nseg segments (cls: 2 + 2 nseg words of scratch)
// The large-segment sort class needs more dynamic LDS than the default limit: raise it ONCE per (process, device), under a lock (the
// rasteriser is called from several host threads: utils/in_flight.py), and from gvf_rast_workspace_bytes too -- every caller sizes its
// workspace before its first forward, i.e. outside any hipGraph capture, where hipFuncSetAttribute would be illegal.
static int tile_sort_set_lds_limit() {
    static GvfPerDeviceOnce once;
    return gvf_once_per_device(once, [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&tile_sort_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, SORT_LARGE_N * 8) ==
               hipSuccess;
    }) ? GVF_OK : GVF_ELAUNCH;
}

// cls_state: 0 = cls holds nothing (clear + classify here), 1 = the two counters are cleared (classify here), 2 = classified
static int launch_tile_sort(hipStream_t stream, const uint2* ranges, uint64_t* keys, const uint32_t* vals, uint32_t* ids,
                            uint32_t* cls, uint32_t nseg, int cls_state) {
    if (cls_state == 0 && hipMemsetAsync(cls, 0, 2 * sizeof(uint32_t), stream) != hipSuccess) return GVF_ELAUNCH;
    if (cls_state < 2)
        hipLaunchKernelGGL(classify_kernel, dim3((nseg + 255) / 256), dim3(256), 0, stream, ranges, nseg, cls);
    hipLaunchKernelGGL(tile_sort_kernel<0>, dim3(nseg), dim3(256), 0, stream, ranges, keys, vals, ids, cls, nseg);
    if (tile_sort_set_lds_limit() != GVF_OK) return GVF_ELAUNCH;
    // the LDS class up to SORT_MEDIUM_N keys: three workgroups of 512 threads per CU (GVF_TILE_SORT_MEDIUM=0: measurement switch)
    static const bool medium = [] { const char* e = getenv("GVF_TILE_SORT_MEDIUM"); return !(e && e[0] == '0'); }();
    if (medium)
        hipLaunchKernelGGL(tile_sort_kernel<3>, dim3(SORT_MEDIUM_BLOCKS), dim3(512), SORT_MEDIUM_N * 8, stream, ranges, keys, vals, ids, cls,
                           nseg, 0u);
    // the rest of it and the global class in one launch
    hipLaunchKernelGGL(tile_sort_kernel<1>, dim3(SORT_LARGE_BLOCKS + SORT_HUGE_BLOCKS), dim3(1024), SORT_LARGE_N * 8, stream, ranges, keys,
                       vals, ids, cls, nseg, medium ? 1u : 0u);
    GVF_CHECK_LAUNCH();
    return GVF_OK;
}

// ---------------------------------------------------------------------------------------------
// R6: blend
// ---------------------------------------------------------------------------------------------
// One workgroup per 16x16 tile, as upstream's renderCUDA, but the 4 waves own the four 8x8 QUADRANTS of
// the tile and each wave only walks the splats that can reach its quadrant:
//   * the thread that stages splat j of a 256-splat batch into LDS also computes a 4-bit quadrant mask from
//     the axis-aligned bounding box of the region where alpha = opacity * exp(power) can reach 1/255
//     ( ca dx^2 + 2 cb dx dy + cc dy^2 <= 2 ln(255 opacity) ), inflated so that float noise can only keep
//     extra splats, never drop one: the list only decides WHICH (splat, quadrant) pairs are evaluated -- a culled pair
//     would have failed the alpha < 1/255 test at every pixel of the quadrant, so the image does not depend on it.
//     A kept splat is evaluated by splat_neg_exponent below, which is NOT upstream's expression since round 4:
//     upstream forms  power = -0.5 (A dx^2 + C dy^2) - B dx dy  from the pixel offsets and the conic, tests
//     power > 0, and takes  alpha = min(0.99, opacity * exp(power));  here the exponent comes from the Cholesky
//     factor of the scaled conic in tile-relative coordinates with log2(opacity) folded in,
//         alpha = min(0.99, exp2(lo - s1^2 - s2^2)),  s1 = c1 - l11 px - l12 py,  s2 = c2 - l22 py,
//     i.e. the same quadratic evaluated in another order (~1e-5 of the exponent apart: the cancellation in c - l p
//     over a tile's 16 pixels), with no `power > 0` case of its own (a sum of squares cannot come out positive;
//     upstream's test only fires on rounding noise at the centre of a valid splat) and with a conic that is not
//     positive definite dropped (opacity staged as 0) where upstream composites the indefinite form; the
//     transmittance update is  T - alpha T  (one fma) for upstream's  T (1 - alpha).  Every one of these stands
//     between this kernel and the published arithmetic only through oracle/rast_oracle.c, which keeps upstream's
//     forms: images agree to ~4e-7 except at pixels where a decision sits within that noise of its threshold
//     (flagged by the oracle);
//   * each wave compacts the batch into its own index list with wave64 ballots (order preserved = depth
//     order) and iterates over that list only.  A typical splat (3-sigma radius ~9 px) reaches ~40 % of the
//     quadrants of the tiles it was binned to, so ~60 % of upstream's (pixel, splat) evaluations disappear.
// Early termination is per wave (all 64 pixels saturated) and per workgroup (stop staging).
// Which of the tile's four 8x8 quadrants can a splat reach with alpha >= 1/255?  Exact (up to a safety margin) test of
// the ellipse  e(dx, dy) = a' dx^2 + b' dx dy + c' dy^2 >= -log2(255 opacity)  against each quadrant's rectangle: e is
// concave, so its maximum over a rectangle that does not contain the centre sits on one of the four edges, at the
// clamped vertex of a 1-D parabola.  (The axis-aligned box (hx, hy) that the binning uses keeps ~25 % more pairs: the
// corners of the box of a rotated, elongated ellipse.)  The test only decides which (splat, quadrant) pairs are
// evaluated; a kept splat is evaluated by the compositing step's own arithmetic (the Cholesky form described above) and
// a culled one would have failed alpha >= 1/255 at every pixel of the quadrant (margin: 0.02 octaves on the threshold
// against ~1e-5 of rounding), so images do not depend on the test.
// max over t in [lo, hi] of  qa fixed^2 + qb fixed t + qc t^2   (qc < 0), the vertex slope kv = -qb / (2 qc) handed in: one hardware
// reciprocal per splat and orientation instead of an IEEE division per edge (round 3: the staging loop spent 8 divisions = ~100 of its 265
// vector instructions per instance on them; blend 0.87 -> 0.82 ms).  An inexact vertex only LOWERS the value (any t of the interval is a
// lower bound of a concave function's maximum) by ~qc dt^2 ~ 1e-13 -- against the 0.02-octave margin of the test, i.e. never visibly; the
// forward and the backward kernel share this function, so they evaluate the same splats.
// (written with explicit fmas: this file is compiled with -ffp-contract=off for the arithmetic it shares with the oracle, and as separate multiplies
// and adds the four edges of the four quadrants were 110 of the staging pass's 204 vector instructions per instance; the test has a 0.02-octave
// margin and is shared by the forward and the backward kernel, so its rounding only has to be the same in both)
__device__ __forceinline__ float edge_max(float qa, float qb, float qc, float kv, float fixed, float lo, float hi) {
    const float t = fminf(fmaxf(kv * fixed, lo), hi);                                   // the vertex of the parabola along the edge, clamped
    return __builtin_fmaf(__builtin_fmaf(qc, t, qb * fixed), t, (qa * fixed) * fixed);  // qa fixed^2 + (qb fixed + qc t) t
}
__device__ __forceinline__ unsigned quadrant_mask(float x, float y, float ap, float bp, float cp, float op, float hx,
                                                  float tile_x0, float tile_y0, bool no_cull) {
    if (hx < 0.0f) return 0u;                            // opacity < 1/255: alpha < 1/255 at every pixel
    if (no_cull || !(hx < __builtin_inff())) return 0xFu; // sub-pixel offsets / degenerate conic: keep everywhere
    const float lim = -(__log2f(255.0f * op) + 0.02f);
    const float kx = -0.5f * bp * __builtin_amdgcn_rcpf(cp), ky = -0.5f * bp * __builtin_amdgcn_rcpf(ap);   // vertex slopes: dy* = kx dx, dx* = ky dy
    const float xr = x - tile_x0, yr = y - tile_y0;
    unsigned m = 0u;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float ox = (float)((q & 1) * 8), oy = (float)((q >> 1) * 8);
        const float dxl = xr - (ox + 7.0f), dxh = xr - ox, dyl = yr - (oy + 7.0f), dyh = yr - oy;   // offset ranges over the quadrant
        const bool inside = dxl <= 0.0f && dxh >= 0.0f && dyl <= 0.0f && dyh >= 0.0f;
        float e = edge_max(ap, bp, cp, kx, dxl, dyl, dyh);
        e = fmaxf(e, edge_max(ap, bp, cp, kx, dxh, dyl, dyh));
        e = fmaxf(e, edge_max(cp, bp, ap, ky, dyl, dxl, dxh));
        e = fmaxf(e, edge_max(cp, bp, ap, ky, dyh, dxl, dxh));
        m |= (inside || e >= lim) ? (1u << q) : 0u;
    }
    return m;
}

// Dispatch order of blend_kernel inside a frame: heaviest tiles first.  A workgroup's life is proportional to its tile's instance count
// (1 k .. 150 k cycles at the bench shape); in image order the heavy centre tiles of the LAST frame start late and the chip drains behind them:
// list scheduling of the measured workgroup durations on the 2048 resident slots puts the image order 5.7 % above sum / slots and
// heaviest-first 0.1-0.9 % above (scripts/blend_stamps.py, profiles/r05_blend_phase_stamps.txt).  One workgroup per frame: counting sort of the
// tiles by count class (64 classes of 32 instances, class 0 = 2016 and more; the order inside a class is whatever the atomics give -- tiles of a
// class are neighbours in the image anyway, so the splat records stay shared in L2).  Images do not depend on the order.
// Measured (profiles/r05_blend_order_ab.txt): the reference's live render job (512 x 512, tiles of up to 16 k instances beside empty ones) 167-176 ->
// 146-149 ms per sample; the bench shape (800 x 800, no tile above 2048) unchanged within noise -- there image order already mixes heavy (vector-pipe)
// and light (latency) workgroups on every CU.  So a frame is reordered only if it holds a tile of the top class.
__global__ __launch_bounds__(1024) void blend_order_kernel(const uint2* __restrict__ ranges, int ntiles, uint32_t* __restrict__ order) {
    __shared__ uint32_t hist[64];
    const int f = blockIdx.x, t = threadIdx.x;
    const uint2* r = ranges + (size_t)f * ntiles;
    if (t < 64) hist[t] = 0u;
    __syncthreads();
    for (int tile = t; tile < ntiles; tile += 1024) {
        const uint32_t n = r[tile].y - r[tile].x;
        atomicAdd(&hist[63u - min(63u, n >> 5)], 1u);
    }
    __syncthreads();
    const bool reorder = hist[0] != 0u;                // a tile in the top class (>= 2016 instances)?  Read by everyone BEFORE thread 0's scan overwrites it
    __syncthreads();
    if (!reorder) {                                    // image order
        for (int tile = t; tile < ntiles; tile += 1024) order[(size_t)f * ntiles + tile] = (uint32_t)tile;
        return;
    }
    if (t == 0) {
        uint32_t run = 0u;
        for (int c = 0; c < 64; ++c) { const uint32_t h = hist[c]; hist[c] = run; run += h; }
    }
    __syncthreads();
    for (int tile = t; tile < ntiles; tile += 1024) {
        const uint32_t n = r[tile].y - r[tile].x;
        order[(size_t)f * ntiles + atomicAdd(&hist[63u - min(63u, n >> 5)], 1u)] = (uint32_t)tile;
    }
}

#ifdef BLEND_TIMING
// timing builds only (scripts/blend_stamps.py, a variant library): wave-cycles per phase of blend_kernel, summed over every wave of a launch
//   [0] wait at the round's first barrier  [1] id load  [2] record gather  [3] Cholesky + quadrant mask + LDS writes  [4] wait at the second barrier
//   [5] list compaction  [6] compositing  [7] rounds  [8] list entries  [9] waves  [10] whole wave lifetime  [11] epilogue stores
__device__ unsigned long long* g_blend_buf;       // [waves of the launch][12], one row per wave (same-address atomics would serialise the launch)
__device__ unsigned long long g_blend_cap;
extern "C" int gvf_debug_blend_timing(unsigned long long* device_buf, unsigned long long rows) {
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_blend_buf), &device_buf, sizeof(device_buf)) != hipSuccess) return 1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_blend_cap), &rows, sizeof(rows)) != hipSuccess) return 1;
    return 0;
}
#define BT_DECL unsigned long long bt_acc[12] = {}; unsigned long long bt_last = __builtin_amdgcn_s_memtime(); const unsigned long long bt_first = bt_last;
#define BT(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); bt_acc[i] += n_ - bt_last; bt_last = n_; } while (0)
#define BT_VMWAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define BT_DECL
#define BT(i) do { } while (0)
#define BT_VMWAIT() do { } while (0)
#endif

#ifdef BLEND_CONSUMED
// counting builds only (scripts/blend_consumed.py, a variant library): per size class of a (frame, tile) segment -- <= 2048 keys, <= 4096, <= 16384,
// more -- [segments, keys sorted, keys the compositing had staged when every pixel of the tile was saturated (whole 256-key rounds)]: how much of
// the per-tile sort's work the blend ever looks at
__device__ unsigned long long g_blend_cons[4][5];   // + [staged keys whose quadrant mask is 0, sum of the masks' bit counts]
extern "C" int gvf_debug_blend_consumed(unsigned long long* out12, int reset) {
    if (out12 && hipMemcpyFromSymbol(out12, HIP_SYMBOL(g_blend_cons), sizeof(g_blend_cons)) != hipSuccess) return 1;
    if (reset) { unsigned long long z[20] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_blend_cons), z, sizeof(z)) != hipSuccess) return 1; }
    return 0;
}
#endif

// The pixel's "done" flag of blend_kernel (outside the image, or saturated) and the predicates of its compositing step.
// GVF_BLEND_LANE_MASKS = 1 (round 6, the product): the flag lives as the wave's 64-bit LANE MASK and the predicates are scalar operations on lane
// masks (ballot / inverse ballot).  As a per-lane bool (= 0, the form up to round 6, kept as the A/B variant) the compiler holds the same masks in
// scalar registers but pays one scalar instruction more per list entry (the complement of `done`) and rebuilds the flag in a vector register for
// every __all() (v_cndmask + v_cmp per trip); scalar instructions come out of the same wave's issue stream as the vector ones (a s_waitcnt does
// not: scripts/ubench/blend_step.hip MODE 4).  The loop alone: 41.4 -> 38.9 ticks per list entry per SIMD (MODE 2 there,
// profiles/r06_ubench_blend_step.txt).  Same arithmetic, same decisions: images are bit-identical.  The fence between the two pairs of a trip
// keeps the second pair's LDS reads behind the first pair's arithmetic: without it the scheduler requests all four entries' records at once --
// 68 vector registers, seven waves per SIMD instead of eight.
#ifndef GVF_BLEND_LANE_MASKS
#define GVF_BLEND_LANE_MASKS 1
#endif
#ifndef GVF_BLEND_FIRST_ROUND_REDUCE       // 1: the workgroup-wide "every pixel saturated?" reduction runs before the first round too (the form up to round 6; A/B variant)
#define GVF_BLEND_FIRST_ROUND_REDUCE 0
#endif
#if GVF_BLEND_LANE_MASKS
#define BL_DONE_INIT(init) unsigned long long done_m = __builtin_amdgcn_ballot_w64(init)      /* all 64 lanes are active: workgroups are whole */
#define BL_WAVE_DONE() (done_m == ~0ull)
#define BL_WORKGROUP_DONE() __syncthreads_and(done_m == ~0ull)
#define BL_STEP_PREDICATES(alpha, test_T)                                                                   \
            const unsigned long long ok_m = __builtin_amdgcn_ballot_w64(!((alpha) < 1.0f / 255.0f)) & ~done_m;   \
            const unsigned long long stop_m = ok_m & __builtin_amdgcn_ballot_w64((test_T) < 0.0001f);       \
            done_m |= stop_m;                                                                               \
            const bool acc = __builtin_amdgcn_inverse_ballot_w64(ok_m ^ stop_m);
#define BL_PAIR_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define BL_DONE_INIT(init) bool done = (init)
#define BL_WAVE_DONE() __all(done)
#define BL_WORKGROUP_DONE() (__syncthreads_count(done) == BLEND_THREADS)
#define BL_STEP_PREDICATES(alpha, test_T)                                                                   \
            const bool ok = !done && !((alpha) < 1.0f / 255.0f);                                            \
            const bool stop = ok && (test_T) < 0.0001f;                                                     \
            done = done || stop;                                                                            \
            const bool acc = ok != stop;           /* = ok && !stop (stop implies ok): a scalar xor of the two lane masks instead of a second compare */
#define BL_PAIR_FENCE() do { } while (0)
#endif

// DEPTH: accumulate the depth channel (diff_gauss outputs; one fma per evaluated splat that the mip path does not pay)
template <bool DEPTH>
__global__ __launch_bounds__(BLEND_THREADS) void blend_kernel(
    int P, int H, int W, int gx, int gy, float bg0, float bg1, float bg2, const uint2* __restrict__ ranges,
    const uint32_t* __restrict__ point_list, const float4* __restrict__ splats,
    const float* __restrict__ subpixel_offset, float* __restrict__ out_color,
    float* __restrict__ out_alpha, float* __restrict__ out_depth, int nslab,
    const uint32_t* __restrict__ tile_order /* [F][tiles]: workgroup blockIdx.x of frame f takes tile tile_order[f][blockIdx.x]; null = identity */,
    const uint32_t* __restrict__ rec_of /* Gaussian id -> index of its splat record inside a frame (slot order); null = the id itself */,
    unsigned char* __restrict__ out_u8 /* non-null: the frames leave as uint8 [F][3][H][W] = clamp(rgb, 0, 1) * 255 truncated (C1's post-process,
                                          rgb_to_u8_kernel's arithmetic on the same float) and out_color is not written */) {
    __shared__ float4 sA[BLEND_THREADS];
    __shared__ float4 sB[BLEND_THREADS];
    __shared__ float4 sC[BLEND_THREADS];                  // {b, depth, -, -}: 16-byte stride like sA / sB, one index shift per splat
    __shared__ unsigned char sMask[BLEND_THREADS];
    // per-wave compacted list of the batch: the BYTE OFFSET (16 x index) of each kept splat's records, one dword each -- four entries are one
    // ds_read_b128 whose registers ARE the addresses of the record reads (as byte indices they cost an extract-and-shift per compositing step)
    __shared__ unsigned sList[4][BLEND_THREADS];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int f = blockIdx.y;
    const int tile = tile_order != nullptr ? (int)tile_order[(size_t)f * gridDim.x + blockIdx.x] : (int)blockIdx.x;
    const int tx = tile % gx, ty = tile / gx;
    const int px = tx * TILE + (wave & 1) * 8 + (lane & 7), py = ty * TILE + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < W && py < H;
    const size_t pid = (size_t)py * W + px;
    float pxf = (float)px, pyf = (float)py;
    if (subpixel_offset != nullptr && inside) { pxf += subpixel_offset[2 * pid]; pyf += subpixel_offset[2 * pid + 1]; }

    // a tile's instances = its nslab consecutive depth-slab segments (contiguous in memory, each sorted, slabs ordered)
    const size_t seg0 = ((size_t)f * gx * gy + tile) * nslab;
    const uint2 rng = make_uint2(ranges[seg0].x, ranges[seg0 + nslab - 1].y);
    const size_t gbase = (size_t)f * P;
    const int rounds = (int)((rng.y - rng.x + BLEND_THREADS - 1) / BLEND_THREADS);
    int todo = (int)(rng.y - rng.x);
    const uint64_t lt_mask = (1ull << lane) - 1ull;

    const float pxr = pxf - (float)(tx * TILE), pyr = pyf - (float)(ty * TILE);      // tile-relative (exact: small integers + the sub-pixel offset)
    BL_DONE_INIT(!inside);
    float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, Dacc = 0.f;

    BT_DECL
#ifdef BLEND_CONSUMED
    int rounds_done = rounds;
#endif
    for (int r = 0; r < rounds; ++r, todo -= BLEND_THREADS) {
        BT(6);
        // (r > 0: before the first round no pixel inside the image is saturated and nothing in LDS has to be protected from a previous round;
        //  the reduction's LDS round trip and barriers are ~1 % of a wave's life at 1.17 rounds per tile)
#ifdef BLEND_CONSUMED
        if ((GVF_BLEND_FIRST_ROUND_REDUCE || r > 0) && BL_WORKGROUP_DONE()) { rounds_done = r; break; }
#else
        if ((GVF_BLEND_FIRST_ROUND_REDUCE || r > 0) && BL_WORKGROUP_DONE()) break;
#endif
        BT(0);
#ifdef BLEND_TIMING
        uint32_t id_ = 0;
        if (t < todo) id_ = point_list[rng.x + (uint32_t)r * BLEND_THREADS + t];
        if (t < todo && rec_of != nullptr) id_ = rec_of[id_];
        BT_VMWAIT(); BT(1);
        float4 a_ = {}, b_ = {}, c_ = {};
        if (t < todo) { const float4* rec = splats + 4 * (gbase + id_); a_ = rec[0]; c_ = rec[2]; b_ = rec[1]; }
        BT_VMWAIT(); BT(2);
        bt_acc[7] += 1;
#endif
        if (t < todo) {
#ifdef BLEND_TIMING
            const float4 a = a_, b = b_, c = c_;
#else
            uint32_t id = point_list[rng.x + (uint32_t)r * BLEND_THREADS + t];
            if (rec_of != nullptr) id = rec_of[id];          // (a 1 MB table, L2-resident; the sorted lists keep Gaussian ids: ties break by index)
            const float4* rec = splats + 4 * (gbase + id);
            const float4 a = rec[0];
            const float4 c = rec[2];
            const float4 b = rec[1];
#endif
            const SplatChol ch = splat_cholesky(a.x, a.y, a.z, a.w, b.x, (float)(tx * TILE), (float)(ty * TILE));
            sA[t] = make_float4(ch.l11, ch.l12, ch.l22, ch.c1);
            sB[t] = make_float4(ch.c2, ch.ok ? __builtin_amdgcn_logf(b.y) : -__builtin_inff(), b.z, b.w);       // log2(opacity)
            sC[t] = make_float4(c.x, c.y, 0.f, 0.f);
            sMask[t] = (unsigned char)quadrant_mask(a.x, a.y, a.z, a.w, b.x, b.y, c.z, (float)(tx * TILE), (float)(ty * TILE), subpixel_offset != nullptr);
#ifdef BLEND_CONSUMED
            {
                const unsigned n_ = rng.y - rng.x;
                const int cls_ = n_ <= 2048u ? 0 : n_ <= 4096u ? 1 : n_ <= 16384u ? 2 : 3;
                if (sMask[t] == 0) atomicAdd(&g_blend_cons[cls_][3], 1ull);
                atomicAdd(&g_blend_cons[cls_][4], (unsigned long long)__popc((unsigned)sMask[t]));
            }
#endif
        }
        BT(3);
        __syncthreads();
        BT(4);
        const int cnt = min(BLEND_THREADS, todo);
        if (BL_WAVE_DONE()) continue;                     // this quadrant is saturated (wave-uniform)
        // compact the batch into this wave's list (ascending index = depth order)
        int n_w = 0;
#pragma unroll
        for (int k = 0; k < BLEND_THREADS / GVF_WAVE; ++k) {
            const int idx = k * GVF_WAVE + lane;
            const bool hit = idx < cnt && ((sMask[idx] >> wave) & 1u);
            const uint64_t bal = __ballot(hit);
            if (hit) sList[wave][n_w + __popcll(bal & lt_mask)] = (unsigned)idx * 16u;
            n_w += __popcll(bal);
        }
        __builtin_amdgcn_wave_barrier();
        BT(5);
#ifdef BLEND_TIMING
        bt_acc[8] += n_w;
#endif
        // Branch-free compositing step (upstream's `continue`s become predicates: a skipped splat gets weight 0,
        // which leaves C and T bit-identical), two splats per trip so that the second one's LDS reads and alpha
        // arithmetic overlap the first one's serial T update.
#define GVF_BLEND_STEP(J)                                                                              \
        {                                                                                              \
            const float4 a = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(sA) + (J));          \
            const float4 b = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(sB) + (J));          \
            const float4 c4_ = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(sC) + (J));        \
            const float2 c = DEPTH ? make_float2(c4_.x, c4_.y) : make_float2(c4_.x, 0.f);              \
            const float nlog = splat_neg_exponent(a.x, a.y, a.z, a.w, b.x, pxr, pyr, b.y);   /* -log2(opacity * G) */ \
            const float alpha = fminf(0.99f, __builtin_amdgcn_exp2f(-nlog));                           \
            const float w_raw = alpha * T;                                                             \
            const float test_T = T - w_raw;        /* = T (1 - alpha) up to one rounding; one op less */ \
            BL_STEP_PREDICATES(alpha, test_T)      /* ok = !done && !(alpha < 1/255); stop = ok && test_T < 1e-4; done |= stop; acc = ok && !stop */ \
            const float wgt = acc ? w_raw : 0.0f;                                                      \
            C0 = __builtin_fmaf(b.z, wgt, C0);                                                         \
            C1 = __builtin_fmaf(b.w, wgt, C1);                                                         \
            C2 = __builtin_fmaf(c.x, wgt, C2);                                                         \
            if (DEPTH) Dacc = __builtin_fmaf(c.y, wgt, Dacc);                                          \
            T = acc ? test_T : T;                                                                      \
        }
        int jj = 0;
        for (; jj + 3 < n_w; jj += 4) {
            if (BL_WAVE_DONE()) break;
            const unsigned j0 = sList[wave][jj], j1 = sList[wave][jj + 1], j2 = sList[wave][jj + 2], j3 = sList[wave][jj + 3];
            GVF_BLEND_STEP(j0)
            GVF_BLEND_STEP(j1)
            BL_PAIR_FENCE();
            GVF_BLEND_STEP(j2)
            GVF_BLEND_STEP(j3)
        }
        for (; jj < n_w; ++jj) {
            if (BL_WAVE_DONE()) break;
            const unsigned j0 = sList[wave][jj];
            GVF_BLEND_STEP(j0)
        }
#undef GVF_BLEND_STEP
    }
#ifdef BLEND_CONSUMED
    if (t == 0) {
        const unsigned n = rng.y - rng.x, used = min(n, (unsigned)rounds_done * BLEND_THREADS);
        const int cls = n <= 2048u ? 0 : n <= 4096u ? 1 : n <= 16384u ? 2 : 3;
        atomicAdd(&g_blend_cons[cls][0], 1ull); atomicAdd(&g_blend_cons[cls][1], (unsigned long long)n); atomicAdd(&g_blend_cons[cls][2], (unsigned long long)used);
    }
#endif
    if (inside) {
        const size_t hw = (size_t)H * W;
        const float r0 = __builtin_fmaf(T, bg0, C0), r1 = __builtin_fmaf(T, bg1, C1), r2 = __builtin_fmaf(T, bg2, C2);
#ifdef BLEND_ABL_NO_U8      // timing experiment only (variant build): the epilogue without the uint8 form
        if (false) {
#else
        if (out_u8 != nullptr) {                   // (uniform) round 6: no fp32 frame in HBM at all when the caller wants the uint8 one
#endif
            unsigned char* ob = out_u8 + (size_t)f * 3 * hw;
            ob[0 * hw + pid] = (unsigned char)(fminf(fmaxf(r0, 0.f), 1.f) * 255.0f);
            ob[1 * hw + pid] = (unsigned char)(fminf(fmaxf(r1, 0.f), 1.f) * 255.0f);
            ob[2 * hw + pid] = (unsigned char)(fminf(fmaxf(r2, 0.f), 1.f) * 255.0f);
        } else {
            float* oc = out_color + (size_t)f * 3 * hw;
            oc[0 * hw + pid] = r0;
            oc[1 * hw + pid] = r1;
            oc[2 * hw + pid] = r2;
        }
        if (out_alpha != nullptr) out_alpha[(size_t)f * hw + pid] = 1.0f - T;
        if (out_depth != nullptr) out_depth[(size_t)f * hw + pid] = Dacc;
    }
#ifdef BLEND_TIMING
    BT(6);
    BT_VMWAIT(); BT(11);
    bt_acc[9] = 1; bt_acc[10] = bt_last - bt_first;
    const unsigned long long wid = ((unsigned long long)f * gridDim.x + tile) * 4 + wave;
    if (lane == 0 && g_blend_buf != nullptr && wid < g_blend_cap)
        for (int i = 0; i < 12; ++i) g_blend_buf[wid * 12 + i] = bt_acc[i];
#endif
}

// (Round 5 measured two more variants of the compositing step and dropped both: the two predicates as EXEC masks (if / else instead of v_cndmask:
// one vector instruction less, blend 0.82 ms against 0.70) and the colour accumulation on the idle matrix pipe (one v_mfma_f32_4x4x1_16b_f32 per
// splat and wave -- weights as A, channel (lane & 3) of {r, g, b, depth} as B -- instead of three v_fma_f32: 14 vector instructions + 1 MFMA per step
// against 16, blend 0.74 ms against 0.69).  profiles/r05_blend_branchy_ab.txt, r05_blend_mfma_accumulate_ab.txt; git history has both.)
// (Round 4 measured a matrix-pipe variant of this kernel -- the exponents of 32 splats x 64 pixels from v_mfma_f32_32x32x2_f32 on the expanded
// quadratic form, or from ONE v_mfma_f32_32x32x16_f16 with hi / lo split coefficients, software-pipelined under the compositing steps;
// 22.3 -> 14.4 vector instructions per step, images within 1e-6 of this kernel's -- and dropped it: 0.92 ms (fp16 split) / 1.03 ms (f32)
// against 0.80 ms.  profiles/r04_blend_matrix_pipe.txt has the numbers, git history (round 4) the kernel.)
// C1 post-process: rgb float -> uint8 exactly as utils/inference_utils.py:280-286 does on the host
// (clamp(0,1) * 255, truncating cast), so frames leave the device at 1 byte per channel.
__global__ __launch_bounds__(256) void rgb_to_u8_kernel(const float4* __restrict__ src, uchar4* __restrict__ dst,
                                                        long long n4, const float* __restrict__ tail_src,
                                                        unsigned char* __restrict__ tail_dst, int tail) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const float4 v = src[i];
        uchar4 o;
        o.x = (unsigned char)(fminf(fmaxf(v.x, 0.f), 1.f) * 255.0f);
        o.y = (unsigned char)(fminf(fmaxf(v.y, 0.f), 1.f) * 255.0f);
        o.z = (unsigned char)(fminf(fmaxf(v.z, 0.f), 1.f) * 255.0f);
        o.w = (unsigned char)(fminf(fmaxf(v.w, 0.f), 1.f) * 255.0f);
        dst[i] = o;
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < tail)
        tail_dst[threadIdx.x] = (unsigned char)(fminf(fmaxf(tail_src[threadIdx.x], 0.f), 1.f) * 255.0f);
}

// Camera blocks travel as kernel arguments (16 per launch): no host buffer has to outlive the call
// and the upload is capturable in a hipGraph.
struct FrameChunk { GvfRastFrame f[16]; };
// The first upload launch of a call also clears the call's tables (one launch instead of six memsets, each a ~4 us bubble in a
// 1.5 ms step): blocks >= 1 zero the tile ranges, the tile counters, the Morton histogram and the sort-class counters, and
// set the bounding-box accumulators (min slots to all-ones, max slots to zero).
struct CallTables {
    uint32_t* ranges; uint32_t n_ranges;             // words
    uint32_t* tile_count; uint32_t n_tile_count;
    uint32_t* mhist; uint32_t n_mhist;
    uint32_t* cls;                                    // 2 words
    uint32_t* mm;                                     // 6 words
};
__global__ void upload_frames_kernel(FrameChunk c, int count, GvfRastFrame* __restrict__ dst, CallTables tab) {
    if (blockIdx.x == 0) {
        const int words = (int)(sizeof(GvfRastFrame) / 4) * count;
        const uint32_t* src = reinterpret_cast<const uint32_t*>(&c);
        for (int k = threadIdx.x; k < words; k += blockDim.x) reinterpret_cast<uint32_t*>(dst)[k] = src[k];
        if (tab.cls != nullptr && threadIdx.x < 2) tab.cls[threadIdx.x] = 0u;
        if (tab.mm != nullptr && threadIdx.x < 6) tab.mm[threadIdx.x] = threadIdx.x < 3 ? 0xffffffffu : 0u;
        return;
    }
    const uint32_t t = (blockIdx.x - 1) * blockDim.x + threadIdx.x, nt = (gridDim.x - 1) * blockDim.x;
    for (uint32_t k = t; k < tab.n_ranges; k += nt) tab.ranges[k] = 0u;
    for (uint32_t k = t; k < tab.n_tile_count; k += nt) tab.tile_count[k] = 0u;
    for (uint32_t k = t; k < tab.n_mhist; k += nt) tab.mhist[k] = 0u;
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
// Opt-in stage timing (HIP events on the caller's stream), used by bench.py for the roofline figure.
// The only process-global state in this library; off by default.
constexpr int PROF_MAX_CALLS = 256;
constexpr int PROF_EVENTS = GVF_RAST_NSTAGES + 1;
struct Profiler {
    bool on = false;
    std::atomic<int> calls{0};          // slots are handed out to concurrent callers (one host thread per sample in flight)
    hipEvent_t ev[PROF_MAX_CALLS][PROF_EVENTS];
};
Profiler g_prof;
inline void prof_mark(hipStream_t s, int slot, int k) {
    if (slot >= 0) (void)hipEventRecord(g_prof.ev[slot][k], s);
}

struct Workspace {
    GvfRastFrame* frames;
    float4* splats;   // [F*P][4]: 64-byte records
    uint32_t* tiles_touched; int32_t* radii;
    uint32_t* block_sums; uint32_t* frame_base; uint32_t* total;
    uint64_t* keys; uint64_t* keys_alt; uint32_t* vals; uint32_t* vals_alt; uint32_t* ids;
    uint2* ranges; uint32_t* cls;
    uint32_t* tile_count; uint32_t* cursor;            // bucket binning: [F*ntiles*NSLAB] each
    uint32_t* partial; float2* zrange;
    uint32_t* order; uint32_t* order_alt; uint32_t* mhist; uint32_t* mm;
    uint4* binrec;                                     // [F*P] {x0|y0<<16, x1|y1<<16, depth bits, -}
    void* sort_tmp; size_t sort_tmp_bytes;
    size_t bytes; bool ok;
};

int key_end_bit(int F, int ntiles) {
    uint64_t m = (uint64_t)F * (uint64_t)ntiles;
    int bits = 0;
    while (((uint64_t)1 << bits) < m) ++bits;
    return 32 + bits;
}

Workspace carve(void* ws, size_t bytes, int P, int F, int H, int W, int64_t max_rendered) {
    Workspace w;
    GvfCarver c(ws, bytes);
    const int nb = (P + PRE_THREADS - 1) / PRE_THREADS;
    const int ntiles = ((W + TILE - 1) / TILE) * ((H + TILE - 1) / TILE);
    const size_t FP = (size_t)F * (size_t)(P > 0 ? P : 1);
    const size_t D = (size_t)(max_rendered > 0 ? max_rendered : 1);
    w.frames = c.take<GvfRastFrame>(F);
    w.splats = c.take<float4>(4 * FP);
    w.tiles_touched = c.take<uint32_t>(FP);
    w.radii = c.take<int32_t>(FP);
    w.block_sums = c.take<uint32_t>((size_t)F * (nb > 0 ? nb : 1));
    w.frame_base = c.take<uint32_t>(F + 1);
    w.total = c.take<uint32_t>(1);
    w.keys = c.take<uint64_t>(D);
    w.keys_alt = c.take<uint64_t>(D);
    w.vals = c.take<uint32_t>(D);
    w.vals_alt = c.take<uint32_t>(D);
    w.ids = c.take<uint32_t>(D);
    w.ranges = c.take<uint2>((size_t)F * ntiles * NSLAB);
    w.cls = c.take<uint32_t>(2 + 2 * (size_t)F * ntiles * NSLAB);
    w.tile_count = c.take<uint32_t>((size_t)F * ntiles * NSLAB);
    w.cursor = c.take<uint32_t>((size_t)F * ntiles * NSLAB);
    w.partial = c.take<uint32_t>(((size_t)F * ntiles * NSLAB + SCAN_CHUNK - 1) / SCAN_CHUNK + 1);
    w.zrange = c.take<float2>(F);
    const size_t Pp = (size_t)(P > 0 ? P : 1);
    w.order = c.take<uint32_t>(Pp);
    w.order_alt = c.take<uint32_t>(Pp);
    w.mhist = c.take<uint32_t>(1 << 15);
    w.mm = c.take<uint32_t>(8);
    w.binrec = c.take<uint4>(FP);
    w.sort_tmp_bytes = gvf_sort_tmp_bytes((int64_t)D);
    w.sort_tmp = c.take<char>(w.sort_tmp_bytes);
    w.bytes = gvf_align_up(c.off, 256);
    w.ok = c.ok;
    return w;
}

std::atomic<long long> g_shared_calls{0};

int run_pipeline(const GvfRastSettings& st, const GvfRastFrame* frames_host, int F, int P, int M, bool fused,
                 const GvfGaussianActivation* act, const float* a0, const float* a1, const float* a2,
                 const float* a3, const float* sh, const float* colors_precomp, const float* cov3D_precomp,
                 const float* delta, int n_delta, const float* subpixel_offset, void* workspace,
                 size_t workspace_bytes, int64_t max_rendered, float* out_color, float* out_alpha,
                 float* out_depth, int32_t* out_radii, uint32_t* out_num_rendered, hipStream_t stream,
                 unsigned char* out_u8 = nullptr /* instead of out_color: uint8 frames (gvf_rast_forward_batched_u8) */) {
    const int H = st.image_height, W = st.image_width;
    if (H <= 0 || W <= 0 || P < 0 || F <= 0 || max_rendered < 0 || max_rendered > 0xFFFFFFFFll) return GVF_EINVAL;
    if (st.sh_degree < 0 || st.sh_degree > 3) return GVF_EINVAL;
    if (st.mode != GVF_RAST_MODE_MIP && st.mode != GVF_RAST_MODE_DILATE) return GVF_EINVAL;
    if ((out_color == nullptr) == (out_u8 == nullptr) || !out_num_rendered || !frames_host || !workspace) return GVF_EINVAL;
    if (P > 0 && colors_precomp == nullptr) {
        if (sh == nullptr || M < (st.sh_degree + 1) * (st.sh_degree + 1) || M > MAX_SH_COEFFS) return GVF_EINVAL;
    }
    // hipGetLastError() is process-wide: clear what other users of the runtime (e.g. an event query that
    // returned hipErrorNotReady) left behind, so GVF_CHECK_LAUNCH reports only this call's launches.
    (void)hipGetLastError();
    if ((((uintptr_t)sh) & 15) != 0 || (((uintptr_t)workspace) & 255) != 0) return GVF_EINVAL;  // 16-B SH rows, 256-B workspace
    Workspace w = carve(workspace, workspace_bytes, P, F, H, W, max_rendered);
    if (!w.ok) return GVF_ENOSPC;

    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE, ntiles = gx * gy;
    const int nb = (P + PRE_THREADS - 1) / PRE_THREADS;
    int slot = g_prof.on ? g_prof.calls.fetch_add(1) : -1;
    if (slot >= PROF_MAX_CALLS) slot = -1;

    const bool morton = st.bin_algo != GVF_RAST_BIN_RADIX && F >= 4 && P >= 4096;     // spatial order of the Gaussians (see below)
    // ---- shared activation (activate_cov_kernel): the call's distinct delta slices, if they are few.  The records live in keys_alt, which only
    // the radix binning uses (max_rendered x 8 bytes: room for max_rendered / (8 P) slices; a sizing call with max_rendered = 0 takes the fused path,
    // whose outputs are the same bits).  GVF_RAST_SHARED_ACT=0: measurement / test switch.
    ActSlices slices; slices.n = 0;
    bool shared = false;
    std::vector<int> frame_slice((size_t)F, 0);
    const char* shared_env = getenv("GVF_RAST_SHARED_ACT");          // read per call: tests switch it inside one process
    const bool shared_on = !(shared_env && shared_env[0] == '0');
    if (shared_on && fused && cov3D_precomp == nullptr && st.bin_algo != GVF_RAST_BIN_RADIX && P > 0 && F >= 2) {
        shared = true;
        for (int f = 0; f < F && shared; ++f) {
            const int di = (delta != nullptr && frames_host[f].delta_index >= 0) ? frames_host[f].delta_index : -1;
            int k = 0;
            while (k < slices.n && slices.di[k] != di) ++k;
            if (k == slices.n) {
                if (slices.n == ACT_MAX_SLICES) { shared = false; break; }
                slices.di[slices.n++] = di;
            }
            frame_slice[(size_t)f] = k;
        }
        // worth it from two frames per slice on; the records must fit into keys_alt
        if (shared && (F < 2 * slices.n || (size_t)slices.n * (size_t)P * 64u > (size_t)(max_rendered > 0 ? max_rendered : 0) * 8u)) shared = false;
    }
    for (int f0 = 0; f0 < F; f0 += 16) {
        FrameChunk ch;
        const int cnt = F - f0 < 16 ? F - f0 : 16;
        for (int k = 0; k < cnt; ++k) {
            ch.f[k] = frames_host[f0 + k];
            ch.f[k].reserved[0] = shared ? frame_slice[(size_t)(f0 + k)] : 0;       // the device copy's slice index (the caller's block is not touched)
        }
        CallTables tab = {};
        if (f0 == 0) {
            const size_t nseg_all = (size_t)F * ntiles * NSLAB;
            tab.ranges = reinterpret_cast<uint32_t*>(w.ranges); tab.n_ranges = (uint32_t)(2 * nseg_all);
            tab.tile_count = w.tile_count; tab.n_tile_count = (uint32_t)nseg_all;
            tab.mhist = w.mhist; tab.n_mhist = morton ? (uint32_t)MORTON_BINS : 0u;
            tab.cls = w.cls; tab.mm = w.mm;
        }
        const size_t clear_words = (size_t)tab.n_ranges + tab.n_tile_count + tab.n_mhist;
        const int zb = f0 == 0 ? (int)((clear_words + 4095) / 4096 < 256 ? (clear_words + 4095) / 4096 : 256) : 0;
        hipLaunchKernelGGL(upload_frames_kernel, dim3(1 + zb), dim3(256), 0, stream, ch, cnt, w.frames + f0, tab);
    }
    GVF_CHECK_LAUNCH();

    int nslab_blend = 1;
    const uint32_t* blend_rec_of = nullptr;           // slot order: Gaussian id -> index of its splat record inside a frame
    if (P == 0 || nb == 0) {
        // nothing to splat: background only
        if (hipMemsetAsync(out_num_rendered, 0, sizeof(uint32_t) * F, stream) != hipSuccess) return GVF_ELAUNCH;
    } else {
        const bool bucket = st.bin_algo != GVF_RAST_BIN_RADIX;
        prof_mark(stream, slot, 0);
        // ---- spatial order of the Gaussians (bucket binning, several frames to amortise it over) ----
        const uint32_t* order = nullptr;
        if (morton) {                                  // (bounding box accumulators and histogram: cleared by the upload launch)
            int bb = (P + 255) / 256; if (bb > 128) bb = 128;
            const int pb = (P + 255) / 256;
            hipLaunchKernelGGL(bbox_kernel, dim3(bb), dim3(256), 0, stream, P, a0, w.mm);
            hipLaunchKernelGGL(morton_count_kernel, dim3(pb), dim3(256), 0, stream, P, a0, w.mm, w.order_alt, w.mhist);
            hipLaunchKernelGGL(morton_scan_kernel, dim3(1), dim3(1024), 0, stream, w.mhist);
            hipLaunchKernelGGL(morton_scatter_kernel, dim3(pb), dim3(256), 0, stream, P, w.order_alt, w.mhist, w.order);
            order = w.order;
            if (NSLAB > 1)
                hipLaunchKernelGGL(frame_zrange_kernel, dim3((F + 63) / 64), dim3(64), 0, stream, w.frames, F, w.mm, fused ? 1 : 0,
                                   fused ? *act : GvfGaussianActivation{}, w.zrange);
            GVF_CHECK_LAUNCH();
        }
        // depth slabs need the scene's depth range, which comes with the Morton stage: one slab otherwise
        const int nslab = order != nullptr ? NSLAB : 1;
        nslab_blend = nslab;
        const unsigned nseg = (unsigned)((size_t)F * ntiles * nslab);
        prof_mark(stream, slot, 1);
        PreParams pp;
        pp.P = P; pp.M = colors_precomp ? 0 : M; pp.deg = st.sh_degree; pp.H = H; pp.W = W; pp.mode = st.mode;
        pp.gx = gx; pp.gy = gy; pp.kernel_size = st.kernel_size; pp.scale_modifier = st.scale_modifier;
        pp.fused = fused ? 1 : 0; pp.n_delta = n_delta; pp.F = F;
        // per-pixel sub-pixel offsets move the sample positions: no box culling then (as in the blend)
        pp.upstream_binning = (st.upstream_binning != 0 || subpixel_offset != nullptr) ? 1 : 0;
        if (fused) pp.act = *act; else pp.act = GvfGaussianActivation{};
        const size_t sh_lds_bytes = gvf_align_up((size_t)PRE_THREADS * pp.M * 3 * sizeof(float), 16) + 16;
        const int pre_fy = (F + PRE_FB - 1) / PRE_FB;
        const dim3 pre_grid = GVF_PRE_XCD ? dim3((nb + 7) / 8 * 8 * pre_fy) : dim3(nb, pre_fy);
        if (shared) {
            g_shared_calls.fetch_add(1, std::memory_order_relaxed);
            float4* rec3d = reinterpret_cast<float4*>(w.keys_alt);
            // slot order (see activate_cov_kernel): needs the Morton order, SH input and room for the slot-ordered SH copy behind the records.
            // GVF_RAST_SLOT_ORDER=0: measurement / test switch
            const char* slot_env = getenv("GVF_RAST_SLOT_ORDER");
            const size_t sh_floats = (size_t)pp.M * 3;
            const bool slot_mode = !(slot_env && slot_env[0] == '0') && order != nullptr && colors_precomp == nullptr && sh != nullptr &&
                                   (size_t)slices.n * (size_t)P * 64u + (size_t)P * sh_floats * 4u <= (size_t)max_rendered * 8u;
            float* sh_by_slot = slot_mode ? reinterpret_cast<float*>(rec3d + 4 * (size_t)slices.n * (size_t)P) : nullptr;
            hipLaunchKernelGGL(activate_cov_kernel, dim3((P + 255) / 256, slices.n), dim3(256), 0, stream, *act, st.scale_modifier, P, slices,
                               a0, a1, a2, a3, delta, rec3d, slot_mode ? w.order_alt : nullptr, sh, (int)sh_floats, sh_by_slot);
            if (slot_mode) {
                // thread i = slot i: records, SH rows, splat records and bin records are all read / written at i; the sorted lists keep Gaussian
                // ids (ties break by index, as upstream's stable sort does) and the blend looks the record index up (rec_of)
                blend_rec_of = w.order_alt;
                hipLaunchKernelGGL(preprocess_kernel<true>, pre_grid, dim3(PRE_THREADS), sh_lds_bytes, stream, pp,
                                   w.frames, reinterpret_cast<const float*>(rec3d), reinterpret_cast<const float*>(order), nullptr, nullptr, sh_by_slot,
                                   nullptr, nullptr, nullptr, w.splats, nullptr, out_radii == nullptr ? nullptr : w.radii, nullptr, w.binrec,
                                   nullptr, nslab > 1 ? w.zrange : nullptr);
            } else
            hipLaunchKernelGGL(preprocess_kernel<true>, pre_grid, dim3(PRE_THREADS), sh_lds_bytes, stream, pp,
                               w.frames, reinterpret_cast<const float*>(rec3d), nullptr, nullptr, nullptr, colors_precomp ? nullptr : sh, colors_precomp,
                               nullptr, nullptr, w.splats, nullptr, out_radii == nullptr ? nullptr : w.radii, nullptr, w.binrec,
                               order != nullptr ? w.order_alt : nullptr, nslab > 1 ? w.zrange : nullptr);
        } else
        hipLaunchKernelGGL(preprocess_kernel<false>, pre_grid, dim3(PRE_THREADS), sh_lds_bytes, stream, pp,
                           w.frames, a0, a1, a2, a3, colors_precomp ? nullptr : sh, colors_precomp, cov3D_precomp, delta,
                           w.splats, bucket ? nullptr : w.tiles_touched, (bucket && out_radii == nullptr) ? nullptr : w.radii,
                           bucket ? nullptr : w.block_sums, bucket ? w.binrec : nullptr, order != nullptr ? w.order_alt : nullptr,
                           (bucket && nslab > 1) ? w.zrange : nullptr);
        const int bnb = (P + BIN_SLOTS - 1) / BIN_SLOTS;
        if (bucket)
            hipLaunchKernelGGL(bin_kernel<false>, dim3(bnb, F), dim3(PRE_THREADS), 0, stream, P, gx, gy, w.binrec, order,
                               w.tile_count, w.cursor, w.total, w.keys, nslab, nullptr, nullptr);
        GVF_CHECK_LAUNCH();
        prof_mark(stream, slot, 2);
        if (bucket)
        {
            const int sblocks = (int)((nseg + SCAN_CHUNK - 1) / SCAN_CHUNK);
            hipLaunchKernelGGL(seg_sums_kernel, dim3(sblocks), dim3(1024), 0, stream, w.tile_count, (int)nseg, w.partial);
            hipLaunchKernelGGL(seg_scan_kernel, dim3(sblocks), dim3(1024), 0, stream, w.tile_count, (int)nseg, ntiles * nslab, F,
                               w.partial, sblocks, w.ranges, w.cursor, w.frame_base, out_num_rendered, w.total,
                               (uint32_t)max_rendered, w.cls);
            if (max_rendered <= 0)                   // otherwise the scatter pass writes the per-frame counts on its way
                hipLaunchKernelGGL(frame_counts_kernel, dim3((F + 63) / 64), dim3(64), 0, stream, w.frame_base, F, out_num_rendered);
        }
        else
            hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(1024), 0, stream, w.block_sums, nb, F, w.frame_base,
                               out_num_rendered, w.total, (uint32_t)max_rendered);
        GVF_CHECK_LAUNCH();
        prof_mark(stream, slot, 3);
        if (max_rendered > 0) {
            if (bucket)
                hipLaunchKernelGGL(bin_kernel<true>, dim3(bnb, F), dim3(PRE_THREADS), 0, stream, P, gx, gy, w.binrec, order,
                                   w.tile_count, w.cursor, w.total, w.keys, nslab, w.frame_base, out_num_rendered);
            else
                hipLaunchKernelGGL(duplicate_kernel, dim3(nb, F), dim3(PRE_THREADS), 0, stream, P, gx, gy, w.splats,
                                   w.tiles_touched, w.radii, w.block_sums, w.keys, w.vals, (uint32_t)max_rendered,
                                   pp.upstream_binning);
            GVF_CHECK_LAUNCH();
        }
        if (out_radii != nullptr) {
            if (hipMemcpyAsync(out_radii, w.radii, sizeof(int32_t) * (size_t)F * P, hipMemcpyDeviceToDevice, stream) != hipSuccess)
                return GVF_ELAUNCH;
        }
        prof_mark(stream, slot, 4);
        if (max_rendered > 0) {
            uint64_t* keys_sorted = w.keys;
            uint32_t* vals_by_tile = nullptr;
            if (!bucket) {
                // (a) stable radix sort on the (frame, tile) bits only: segments become contiguous; (b) tile ranges
                vals_by_tile = w.vals;
                int in_alt = 0;
                int rc = gvf_sort_pairs_device_n(w.keys, w.keys_alt, w.vals, w.vals_alt, w.total, max_rendered, 32,
                                                 key_end_bit(F, ntiles), w.sort_tmp, w.sort_tmp_bytes, stream, &in_alt);
                if (rc != GVF_OK) return rc;
                if (in_alt) { keys_sorted = w.keys_alt; vals_by_tile = w.vals_alt; }
                int rblocks = (int)((max_rendered + 255) / 256);
                if (rblocks > 4096) rblocks = 4096;
                hipLaunchKernelGGL(ranges_kernel, dim3(rblocks), dim3(256), 0, stream, keys_sorted, w.total,
                                   (uint32_t)max_rendered, w.ranges, (uint32_t)nseg);
                GVF_CHECK_LAUNCH();
            }
            prof_mark(stream, slot, 5);
            // per-tile on-chip sort by (depth, id)
            const int rc = launch_tile_sort(stream, w.ranges, keys_sorted, vals_by_tile, w.ids, w.cls, nseg, bucket ? 2 : 1);
            if (rc != GVF_OK) return rc;
        } else {
            prof_mark(stream, slot, 5);
        }
    }
    uint32_t* vals_sorted = w.ids;
    // heaviest tiles first (blend_order_kernel; the scatter pass is done with its cursors: their array takes the order).  Worth a launch when the
    // frames' workgroups outnumber the chip's resident slots; GVF_RAST_BLEND_ORDER=0: measurement switch
    const char* order_env = getenv("GVF_RAST_BLEND_ORDER");
    const uint32_t* tile_order = nullptr;
    if (!(order_env && order_env[0] == '0') && P > 0 && nb > 0 && max_rendered > 0 && nslab_blend == 1 && (size_t)F * ntiles >= 2048) {
        hipLaunchKernelGGL(blend_order_kernel, dim3(F), dim3(1024), 0, stream, w.ranges, ntiles, w.cursor);
        tile_order = w.cursor;
    }
    prof_mark(stream, slot, 6);
    if (out_depth != nullptr)
        hipLaunchKernelGGL(blend_kernel<true>, dim3(ntiles, F), dim3(BLEND_THREADS), 0, stream, P, H, W, gx, gy, st.bg[0],
                           st.bg[1], st.bg[2], w.ranges, vals_sorted, w.splats, subpixel_offset,
                           out_color, out_alpha, out_depth, nslab_blend, tile_order, blend_rec_of, out_u8);
    else
        hipLaunchKernelGGL(blend_kernel<false>, dim3(ntiles, F), dim3(BLEND_THREADS), 0, stream, P, H, W, gx, gy, st.bg[0],
                           st.bg[1], st.bg[2], w.ranges, vals_sorted, w.splats, subpixel_offset,
                           out_color, out_alpha, out_depth, nslab_blend, tile_order, blend_rec_of, out_u8);
    GVF_CHECK_LAUNCH();
    prof_mark(stream, slot, 7);
    return GVF_OK;
}


// ---------------------------------------------------------------------------------------------
// R7: backward of the operator (SURVEY.md section 8f NEXT #4; upstream backward.cu restated from its published
// algorithm, checked against oracle/rast_bwd_oracle.c which is pinned by finite differences).
// Conventions taken over from upstream: the gradient passes THROUGH alpha = min(0.99, .); a clamped EWA view
// coordinate gets no gradient; the screen-space mean's gradient is reported in NDC units.
// ---------------------------------------------------------------------------------------------
constexpr int BWD_ACC = 10;   // per Gaussian: d/dx, d/dy [pixels], d/d(conic a, b, c), d/d(opacity_eff), d/d(r, g, b), d/d(depth)

// Sums over the lanes of a wave without LDS round trips (a __shfl_xor butterfly is six ds_bpermute / ds_swizzle per
// value): rows of 16 lanes by DPP (quad permutes, then the two mirror patterns: after each step a lane holds the sum
// of a group twice as large); the four rows and the two halves by the gfx950 lane-swap instructions.
__device__ __forceinline__ float row_sum(float v) {          // every lane: sum of its row of 16 lanes
    int x;
#define GVF_DPP_ADD(ctrl_)                                                                                   \
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl_, 0xf, 0xf, false);                          \
    v += __int_as_float(x);
    GVF_DPP_ADD(0xB1)      // quad_perm [1,0,3,2]
    GVF_DPP_ADD(0x4E)      // quad_perm [2,3,0,1]
    GVF_DPP_ADD(0x141)     // row_half_mirror
    GVF_DPP_ADD(0x140)     // row_mirror
#undef GVF_DPP_ADD
    return v;
}
__device__ __forceinline__ float across_rows_sum(float v) {  // every lane: sum of the lanes at its position in the 4 rows
    {   // rows (r0, r1, r2, r3) -> r0 + r1 resp. r2 + r3
        const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
    {   // halves
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
    return v;
}

// One workgroup per 16x16 tile, the 4 waves own its four 8x8 quadrants and walk per-wave lists of the splats whose
// alpha >= 1/255 box reaches the quadrant (exactly blend_kernel's culling, so the same splats are evaluated).
// Phase A replays the forward compositing (same arithmetic as blend_kernel: same skip / stop decisions) to get
// each pixel's final transmittance and the list position after its last contributor; phase B walks the lists back
// to front, forms the per-(pixel, splat) gradients, sums them over the 64 pixels of the wave and adds the wave
// sums to the per-Gaussian accumulators with hardware fp32 atomics.
// AUX: the depth / alpha outputs carry gradients (diff_gauss); the mip path has three channels and nine partials.
template <bool AUX>
__global__ __launch_bounds__(BLEND_THREADS) void blend_backward_kernel(
    int P, int H, int W, int gx, float bg0, float bg1, float bg2, const uint2* __restrict__ ranges,
    const uint32_t* __restrict__ point_list, const float4* __restrict__ splats, const float* __restrict__ subpixel_offset,
    const float* __restrict__ dL_dcolor, const float* __restrict__ dL_dalpha, const float* __restrict__ dL_ddepth,
    float* __restrict__ acc) {
    __shared__ float4 sA[BLEND_THREADS];
    __shared__ float4 sB[BLEND_THREADS];
    __shared__ float2 sC[BLEND_THREADS];
    __shared__ uint32_t sId[BLEND_THREADS];
    __shared__ float4 sL[BLEND_THREADS];                  // the forward's Cholesky form of the exponent (splat_cholesky): l11, l12, l22, c1
    __shared__ float2 sL2[BLEND_THREADS];                 // c2, log2(opacity)
    __shared__ unsigned char sMask[BLEND_THREADS];
    __shared__ unsigned char sList[4][BLEND_THREADS];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int tile = blockIdx.x;
    const int tx = tile % gx, ty = tile / gx;
    const int px = tx * TILE + (wave & 1) * 8 + (lane & 7), py = ty * TILE + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < W && py < H;
    const size_t pid = (size_t)py * W + px, hw = (size_t)H * W;
    float pxf = (float)px, pyf = (float)py;
    if (subpixel_offset != nullptr && inside) { pxf += subpixel_offset[2 * pid]; pyf += subpixel_offset[2 * pid + 1]; }
    const uint2 rng = ranges[tile];
    const int n = (int)(rng.y - rng.x);
    const int rounds = (n + BLEND_THREADS - 1) / BLEND_THREADS;
    const uint64_t lt_mask = (1ull << lane) - 1ull;

// stage batch r_ (records, ids, quadrant masks) and compact it into this wave's list (ascending = depth order)
#define GVF_BWD_STAGE(r_, n_w_)                                                                     \
    {                                                                                               \
        const int k_ = (r_) * BLEND_THREADS + t;                                                    \
        if (k_ < n) {                                                                               \
            const uint32_t id_ = point_list[rng.x + (uint32_t)k_];                                  \
            const float4* rec_ = splats + 4 * (size_t)id_;                                          \
            const float4 a_ = rec_[0];                                                              \
            const float4 c_ = rec_[2];                                                              \
            const float4 b_ = rec_[1];                                                              \
            const SplatChol ch_ = splat_cholesky(a_.x, a_.y, a_.z, a_.w, b_.x, (float)(tx * TILE), (float)(ty * TILE));   \
            sA[t] = a_; sB[t] = make_float4(b_.x, ch_.ok ? b_.y : 0.f, b_.z, b_.w); sC[t] = make_float2(c_.x, c_.y); sId[t] = id_;   \
            sL[t] = make_float4(ch_.l11, ch_.l12, ch_.l22, ch_.c1);                                  \
            sL2[t] = make_float2(ch_.c2, ch_.ok ? __builtin_amdgcn_logf(b_.y) : -__builtin_inff());  \
            sMask[t] = (unsigned char)quadrant_mask(a_.x, a_.y, a_.z, a_.w, b_.x, b_.y, c_.z, (float)(tx * TILE), (float)(ty * TILE), subpixel_offset != nullptr); \
        }                                                                                           \
        __syncthreads();                                                                            \
        const int cnt_ = min(BLEND_THREADS, n - (r_) * BLEND_THREADS);                              \
        n_w_ = 0;                                                                                   \
        _Pragma("unroll") for (int q_ = 0; q_ < BLEND_THREADS / GVF_WAVE; ++q_) {                   \
            const int idx_ = q_ * GVF_WAVE + lane;                                                  \
            const bool hit_ = idx_ < cnt_ && ((sMask[idx_] >> wave) & 1u);                          \
            const uint64_t bal_ = __ballot(hit_);                                                   \
            if (hit_) sList[wave][n_w_ + __popcll(bal_ & lt_mask)] = (unsigned char)idx_;           \
            n_w_ += __popcll(bal_);                                                                 \
        }                                                                                           \
        __builtin_amdgcn_wave_barrier();                                                            \
    }
    const float pxr = pxf - (float)(tx * TILE), pyr = pyf - (float)(ty * TILE);
    // ---- phase A: forward replay
    bool done = !inside;
    float T = 1.0f;
    int last = 0;
    for (int r = 0; r < rounds; ++r) {
        if (__syncthreads_count(done) == BLEND_THREADS) break;
        int n_w;
        GVF_BWD_STAGE(r, n_w)
        for (int jj = 0; jj < n_w; ++jj) {
            if (__all(done)) break;
            const int j = sList[wave][jj];
            const float4 L = sL[j];
            const float2 L2 = sL2[j];
            const float alpha = fminf(0.99f, __builtin_amdgcn_exp2f(-splat_neg_exponent(L.x, L.y, L.z, L.w, L2.x, pxr, pyr, L2.y)));   // the forward's arithmetic: same decisions
            const bool ok = !done && !(alpha < 1.0f / 255.0f);
            const float test_T = T - alpha * T;            // the forward's form
            const bool stop = ok && test_T < 0.0001f;
            done = done || stop;
            if (ok && !stop) { T = test_T; last = r * BLEND_THREADS + j + 1; }
        }
        __syncthreads();                                   // the batch is restaged next round
    }
    const float T_final = T;
    constexpr int NCH = AUX ? 5 : 3;                        // channels r, g, b (, depth, one)
    constexpr int NACC = AUX ? BWD_ACC : BWD_ACC - 1;       // without AUX the depth partial is identically zero
    float dch[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (inside) {
        dch[0] = dL_dcolor[pid]; dch[1] = dL_dcolor[hw + pid]; dch[2] = dL_dcolor[2 * hw + pid];
        if (AUX && dL_ddepth != nullptr) dch[3] = dL_ddepth[pid];
        if (AUX && dL_dalpha != nullptr) dch[4] = dL_dalpha[pid];
    }
    // what lies behind the current splat, per channel (r, g, b, depth, one); backgrounds (bg, 0, 0)
    float suf[5] = {T_final * bg0, T_final * bg1, T_final * bg2, 0.f, 0.f};
    int max_last = last;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) max_last = max(max_last, __shfl_xor(max_last, o, 64));
    // ---- phase B: back to front
    __syncthreads();
    for (int r = rounds - 1; r >= 0; --r) {
        int n_w;
        GVF_BWD_STAGE(r, n_w)
        if (r * BLEND_THREADS < max_last) {                // else: nothing of this batch reached this wave's pixels
            for (int jj = n_w - 1; jj >= 0; --jj) {
                const int j = sList[wave][jj];
                const int k = r * BLEND_THREADS + j;
                if (k >= max_last) continue;               // wave-uniform
                const float4 a = sA[j];
                const float4 b = sB[j];
                const float2 c = sC[j];
                const float dx = a.x - pxf, dy = a.y - pyf;
                const float4 L = sL[j];
                const float2 L2 = sL2[j];
                const float G = __builtin_amdgcn_exp2f(-splat_neg_exponent(L.x, L.y, L.z, L.w, L2.x, pxr, pyr));
                const float alpha = fminf(0.99f, __builtin_amdgcn_exp2f(-splat_neg_exponent(L.x, L.y, L.z, L.w, L2.x, pxr, pyr, L2.y)));   // (the forward's alpha)
                const bool on = inside && k < last && !(alpha < 1.0f / 255.0f);
                if (!__any(on)) continue;
                float g[BWD_ACC];
#pragma unroll
                for (int e = 0; e < BWD_ACC; ++e) g[e] = 0.f;
                if (on) {
                    T = T / (1.f - alpha);                 // transmittance in front of this splat
                    const float cch[5] = {b.z, b.w, c.x, c.y, 1.0f};
                    const float inv1ma = 1.0f / (1.f - alpha);
                    float dL_da = 0.f;
#pragma unroll
                    for (int ch = 0; ch < NCH; ++ch) {
                        dL_da += (cch[ch] * T - suf[ch] * inv1ma) * dch[ch];
                        suf[ch] += cch[ch] * alpha * T;
                    }
                    const float w = alpha * T;
                    g[6] = w * dch[0]; g[7] = w * dch[1]; g[8] = w * dch[2];
                    if (AUX) g[9] = w * dch[3];
                    g[5] = G * dL_da;
                    const float dG = b.y * dL_da * G;      // dL/dpower (gradient passes through the 0.99 clamp)
                    const float ca = a.z * CONIC_IK1, cb = a.w * CONIC_IK2, cc = b.x * CONIC_IK1;   // the conic itself
                    g[0] = dG * (-ca * dx - cb * dy);
                    g[1] = dG * (-cc * dy - cb * dx);
                    g[2] = dG * (-0.5f * dx * dx);
                    g[3] = dG * (-dx * dy);
                    g[4] = dG * (-0.5f * dy * dy);
                }
                // row sums of the ten components, then position e of every row keeps component e, so that ONE
                // cross-row reduction finishes all ten; lanes 0-9 add them with one atomic instruction
#pragma unroll
                for (int e = 0; e < NACC; ++e) g[e] = row_sum(g[e]);
                float mine = g[0];
#pragma unroll
                for (int e = 1; e < NACC; ++e) mine = (lane & 15) == e ? g[e] : mine;
                mine = across_rows_sum(mine);
                if (lane < NACC) unsafeAtomicAdd(acc + (size_t)sId[j] * BWD_ACC + lane, mine);
            }
        }
        __syncthreads();                                   // everyone is done with this batch
    }
#undef GVF_BWD_STAGE
}

struct BwdParams {
    int P, M, deg, H, W, mode;
    float kernel_size, scale_modifier;
    GvfRastFrame fr;
};

// Per-Gaussian chain rule from the blend's accumulators to the operator's inputs (the forward intermediates are
// recomputed: 3D covariance, EWA Jacobian, 2D covariance, mip coefficient, SH basis).
template <int DEG>      // SH degree: compile-time trip counts keep the basis arrays in registers
__global__ __launch_bounds__(PRE_THREADS) void preprocess_backward_kernel(
    BwdParams bp, const float* __restrict__ means3D, const float* __restrict__ shs, const float* __restrict__ colors_precomp,
    const float* __restrict__ opacities, const float* __restrict__ scales, const float* __restrict__ rotations,
    const float* __restrict__ cov3D_precomp, const float* __restrict__ acc, float* __restrict__ g_means3D,
    float* __restrict__ g_means2D, float* __restrict__ g_shs, float* __restrict__ g_colors, float* __restrict__ g_opac,
    float* __restrict__ g_scales, float* __restrict__ g_rots, float* __restrict__ g_cov3D) {
    const int i = blockIdx.x * PRE_THREADS + threadIdx.x;
    if (i >= bp.P) return;
    const GvfRastFrame& fr = bp.fr;
    const int M = bp.M;
    constexpr int deg = DEG;
    float gm[3] = {0.f, 0.f, 0.f}, gsc[3] = {0.f, 0.f, 0.f}, gq[4] = {0.f, 0.f, 0.f, 0.f}, gc6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float gop = 0.f, gcol[3] = {0.f, 0.f, 0.f}, gm2[2] = {0.f, 0.f};
    bool vis = false;
    float dirv[3] = {0.f, 0.f, 0.f}, gcol_sh[3] = {0.f, 0.f, 0.f};

    float a[BWD_ACC];
#pragma unroll
    for (int e = 0; e < BWD_ACC; ++e) a[e] = acc[(size_t)i * BWD_ACC + e];
    float p[3] = {means3D[3 * (size_t)i], means3D[3 * (size_t)i + 1], means3D[3 * (size_t)i + 2]};
    float pv[3];
    xform43(fr.viewmatrix, p, pv);
    if (pv[2] > 0.2f) {
        float ph[4];
        xform44(fr.projmatrix, p, ph);
        const float pw = 1.0f / (ph[3] + 0.0000001f);
        float c6[6], s[3] = {0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
        if (cov3D_precomp != nullptr) {
#pragma unroll
            for (int k = 0; k < 6; ++k) c6[k] = cov3D_precomp[6 * (size_t)i + k];
        } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) s[k] = scales[3 * (size_t)i + k];
#pragma unroll
            for (int k = 0; k < 4; ++k) q[k] = rotations[4 * (size_t)i + k];
            cov3d_from_scale_rot(s, bp.scale_modifier, q, c6);
        }
        const float fx = (float)bp.W / (2.0f * fr.tanfovx), fy = (float)bp.H / (2.0f * fr.tanfovy);
        const float limx = 1.3f * fr.tanfovx, limy = 1.3f * fr.tanfovy;
        const float txtz = pv[0] / pv[2], tytz = pv[1] / pv[2];
        const float xmul = (txtz < -limx || txtz > limx) ? 0.f : 1.f, ymul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
        const float tx = fminf(limx, fmaxf(-limx, txtz)) * pv[2], ty = fminf(limy, fmaxf(-limy, tytz)) * pv[2], tz = pv[2];
        const float J00 = fx / tz, J02 = -(fx * tx) / (tz * tz), J11 = fy / tz, J12 = -(fy * ty) / (tz * tz);
        float A0[3], A1[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float w0 = fr.viewmatrix[c * 4 + 0], w1 = fr.viewmatrix[c * 4 + 1], w2 = fr.viewmatrix[c * 4 + 2];
            A0[c] = J00 * w0 + J02 * w2;
            A1[c] = J11 * w1 + J12 * w2;
        }
        const float S[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
        float SA0[3], SA1[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            SA0[r] = S[r][0] * A0[0] + S[r][1] * A0[1] + S[r][2] * A0[2];
            SA1[r] = S[r][0] * A1[0] + S[r][1] * A1[1] + S[r][2] * A1[2];
        }
        const float cxx = A0[0] * SA0[0] + A0[1] * SA0[1] + A0[2] * SA0[2];
        const float cxy = A0[0] * SA1[0] + A0[1] * SA1[1] + A0[2] * SA1[2];
        const float cyy = A1[0] * SA1[0] + A1[1] * SA1[1] + A1[2] * SA1[2];
        const float kf = bp.mode == GVF_RAST_MODE_MIP ? bp.kernel_size : 0.3f;
        float coef = 1.0f;
        const float det0r = cxx * cyy - cxy * cxy, det1r = (cxx + kf) * (cyy + kf) - cxy * cxy;
        if (bp.mode == GVF_RAST_MODE_MIP) {
            const float det0 = fmaxf(1e-6f, det0r), det1 = fmaxf(1e-6f, det1r);
            coef = sqrtf(det0 / (det1 + 1e-6f) + 1e-6f);
            if (det0 <= 1e-6f || det1 <= 1e-6f) coef = 0.0f;
        }
        const float ap = cxx + kf, bq = cxy, cp = cyy + kf;
        const float det = ap * cp - bq * bq;
        if (det != 0.0f) {
            vis = true;
            // screen-space mean (NDC units) and its path into the 3D mean
            gm2[0] = a[0] * 0.5f * (float)bp.W; gm2[1] = a[1] * 0.5f * (float)bp.H;
            const float* m = fr.projmatrix;
            const float mul1 = ph[0] * pw * pw, mul2 = ph[1] * pw * pw;
            gm[0] += (m[0] * pw - m[3] * mul1) * gm2[0] + (m[1] * pw - m[3] * mul2) * gm2[1];
            gm[1] += (m[4] * pw - m[7] * mul1) * gm2[0] + (m[5] * pw - m[7] * mul2) * gm2[1];
            gm[2] += (m[8] * pw - m[11] * mul1) * gm2[0] + (m[9] * pw - m[11] * mul2) * gm2[1];
            // depth output
            gm[0] += fr.viewmatrix[2] * a[9]; gm[1] += fr.viewmatrix[6] * a[9]; gm[2] += fr.viewmatrix[10] * a[9];
            // colour: precomputed colours directly; SH below (needs the per-coefficient loop)
            if (colors_precomp != nullptr) { gcol[0] = a[6]; gcol[1] = a[7]; gcol[2] = a[8]; }
            else { gcol_sh[0] = a[6]; gcol_sh[1] = a[7]; gcol_sh[2] = a[8]; }
            // opacity and the mip coefficient
            gop = a[5] * coef;
            float gcxx = 0.f, gcxy = 0.f, gcyy = 0.f;
            if (bp.mode == GVF_RAST_MODE_MIP && coef > 0.0f) {
                const float dcoef = a[5] * opacities[i];
                const float dr = dcoef * 0.5f / coef;
                const float d1e = det1r + 1e-6f;
                const float dd0 = dr / d1e, dd1 = -dr * det0r / (d1e * d1e);
                gcxx += dd0 * cyy + dd1 * (cyy + kf);
                gcyy += dd0 * cxx + dd1 * (cxx + kf);
                gcxy += -2.0f * cxy * (dd0 + dd1);
            }
            {
                const float d2 = 1.0f / (det * det);
                const float gA = a[2], gB = a[3], gC = a[4];
                gcxx += d2 * (-cp * cp * gA + bq * cp * gB - bq * bq * gC);
                gcxy += d2 * (2.f * bq * cp * gA - (det + 2.f * bq * bq) * gB + 2.f * ap * bq * gC);
                gcyy += d2 * (-bq * bq * gA + ap * bq * gB - ap * ap * gC);
            }
            float Gm[3][3];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) Gm[r][c] = gcxx * A0[r] * A0[c] + gcxy * A0[r] * A1[c] + gcyy * A1[r] * A1[c];
            gc6[0] = Gm[0][0]; gc6[3] = Gm[1][1]; gc6[5] = Gm[2][2];
            gc6[1] = Gm[0][1] + Gm[1][0]; gc6[2] = Gm[0][2] + Gm[2][0]; gc6[4] = Gm[1][2] + Gm[2][1];
            float dJ00 = 0.f, dJ02 = 0.f, dJ11 = 0.f, dJ12 = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float dA0 = 2.f * gcxx * SA0[c] + gcxy * SA1[c], dA1 = 2.f * gcyy * SA1[c] + gcxy * SA0[c];
                const float w0 = fr.viewmatrix[c * 4 + 0], w1 = fr.viewmatrix[c * 4 + 1], w2 = fr.viewmatrix[c * 4 + 2];
                dJ00 += dA0 * w0; dJ02 += dA0 * w2; dJ11 += dA1 * w1; dJ12 += dA1 * w2;
            }
            const float tz2 = 1.0f / (tz * tz), tz3 = tz2 / tz;
            const float dtx = xmul * (-fx * tz2 * dJ02), dty = ymul * (-fy * tz2 * dJ12);
            const float dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + 2.f * fx * tx * tz3 * dJ02 + 2.f * fy * ty * tz3 * dJ12;
            const float* v = fr.viewmatrix;
            gm[0] += v[0] * dtx + v[1] * dty + v[2] * dtz;
            gm[1] += v[4] * dtx + v[5] * dty + v[6] * dtz;
            gm[2] += v[8] * dtx + v[9] * dty + v[10] * dtz;
            if (cov3D_precomp == nullptr) {
                const float r = q[0], x = q[1], y = q[2], z = q[3];
                const float R[3][3] = {{1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y)},
                                       {2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x)},
                                       {2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)}};
                const float sc[3] = {bp.scale_modifier * s[0], bp.scale_modifier * s[1], bp.scale_modifier * s[2]};
                const float Gs[3][3] = {{gc6[0], 0.5f * gc6[1], 0.5f * gc6[2]}, {0.5f * gc6[1], gc6[3], 0.5f * gc6[4]},
                                        {0.5f * gc6[2], 0.5f * gc6[4], gc6[5]}};
                float dR[3][3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float acc_s = 0.f;
#pragma unroll
                    for (int r2 = 0; r2 < 3; ++r2) {
                        float dl = 0.f;
#pragma unroll
                        for (int kk = 0; kk < 3; ++kk) dl += 2.f * Gs[r2][kk] * R[kk][c] * sc[c];
                        acc_s += dl * R[r2][c];
                        dR[r2][c] = dl * sc[c];
                    }
                    gsc[c] = bp.scale_modifier * acc_s;
                }
                gq[0] = 2.f * (-z * dR[0][1] + y * dR[0][2] + z * dR[1][0] - x * dR[1][2] - y * dR[2][0] + x * dR[2][1]);
                gq[1] = 2.f * (y * dR[0][1] + z * dR[0][2] + y * dR[1][0] - 2.f * x * dR[1][1] - r * dR[1][2] + z * dR[2][0] + r * dR[2][1] - 2.f * x * dR[2][2]);
                gq[2] = 2.f * (-2.f * y * dR[0][0] + x * dR[0][1] + r * dR[0][2] + x * dR[1][0] + z * dR[1][2] - r * dR[2][0] + z * dR[2][1] - 2.f * y * dR[2][2]);
                gq[3] = 2.f * (-2.f * z * dR[0][0] - r * dR[0][1] + x * dR[0][2] + r * dR[1][0] - 2.f * z * dR[1][1] + y * dR[1][2] + x * dR[2][0] + y * dR[2][1]);
            }
        }
    }
    // SH colours: d/d(coefficients) = basis * d/d(rgb) where the +0.5 / clamp let it through; d/d(direction) -> mean
    if (shs != nullptr && g_shs != nullptr) {
        float* gs = g_shs + (size_t)i * M * 3;
        if (!vis) {
            for (int k = 0; k < M * 3; ++k) gs[k] = 0.f;
        } else {
            const float dxc = p[0] - fr.campos[0], dyc = p[1] - fr.campos[1], dzc = p[2] - fr.campos[2];
            const float len = sqrtf(dxc * dxc + dyc * dyc + dzc * dzc);
            const float x = dxc / len, y = dyc / len, z = dzc / len;
            dirv[0] = x; dirv[1] = y; dirv[2] = z;
            float bas[16], db[16][3];
#pragma unroll
            for (int k = 0; k < 16; ++k) { bas[k] = 0.f; db[k][0] = 0.f; db[k][1] = 0.f; db[k][2] = 0.f; }
            bas[0] = SH_C0;
            if (deg > 0) {
                bas[1] = -SH_C1 * y; bas[2] = SH_C1 * z; bas[3] = -SH_C1 * x;
                db[1][1] = -SH_C1; db[2][2] = SH_C1; db[3][0] = -SH_C1;
                if (deg > 1) {
                    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                    bas[4] = SH_C2[0] * xy; bas[5] = SH_C2[1] * yz; bas[6] = SH_C2[2] * (2.f * zz - xx - yy);
                    bas[7] = SH_C2[3] * xz; bas[8] = SH_C2[4] * (xx - yy);
                    db[4][0] = SH_C2[0] * y; db[4][1] = SH_C2[0] * x;
                    db[5][1] = SH_C2[1] * z; db[5][2] = SH_C2[1] * y;
                    db[6][0] = SH_C2[2] * -2.f * x; db[6][1] = SH_C2[2] * -2.f * y; db[6][2] = SH_C2[2] * 4.f * z;
                    db[7][0] = SH_C2[3] * z; db[7][2] = SH_C2[3] * x;
                    db[8][0] = SH_C2[4] * 2.f * x; db[8][1] = SH_C2[4] * -2.f * y;
                    if (deg > 2) {
                        bas[9] = SH_C3[0] * y * (3.f * xx - yy); bas[10] = SH_C3[1] * xy * z; bas[11] = SH_C3[2] * y * (4.f * zz - xx - yy);
                        bas[12] = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy); bas[13] = SH_C3[4] * x * (4.f * zz - xx - yy);
                        bas[14] = SH_C3[5] * z * (xx - yy); bas[15] = SH_C3[6] * x * (xx - 3.f * yy);
                        db[9][0] = SH_C3[0] * 6.f * xy; db[9][1] = SH_C3[0] * (3.f * xx - 3.f * yy);
                        db[10][0] = SH_C3[1] * yz; db[10][1] = SH_C3[1] * xz; db[10][2] = SH_C3[1] * xy;
                        db[11][0] = SH_C3[2] * -2.f * xy; db[11][1] = SH_C3[2] * (4.f * zz - xx - 3.f * yy); db[11][2] = SH_C3[2] * 8.f * yz;
                        db[12][0] = SH_C3[3] * -6.f * xz; db[12][1] = SH_C3[3] * -6.f * yz; db[12][2] = SH_C3[3] * (6.f * zz - 3.f * xx - 3.f * yy);
                        db[13][0] = SH_C3[4] * (4.f * zz - 3.f * xx - yy); db[13][1] = SH_C3[4] * -2.f * xy; db[13][2] = SH_C3[4] * 8.f * xz;
                        db[14][0] = SH_C3[5] * 2.f * xz; db[14][1] = SH_C3[5] * -2.f * yz; db[14][2] = SH_C3[5] * (xx - yy);
                        db[15][0] = SH_C3[6] * (3.f * xx - 3.f * yy); db[15][1] = SH_C3[6] * -6.f * xy;
                    }
                }
            }
            constexpr int nb = (DEG + 1) * (DEG + 1);
            const float* sh = shs + (size_t)i * M * 3;
            float ddir[3] = {0.f, 0.f, 0.f};
            float shc[nb][3];
            float res[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < nb; ++k)
#pragma unroll
                for (int c = 0; c < 3; ++c) { shc[k][c] = sh[k * 3 + c]; res[c] += bas[k] * shc[k][c]; }
            float gr[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) gr[c] = (res[c] + 0.5f < 0.f) ? 0.f : gcol_sh[c];
#pragma unroll
            for (int k = 0; k < nb; ++k)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    gs[k * 3 + c] = bas[k] * gr[c];
                    const float sg = shc[k][c] * gr[c];
                    ddir[0] += db[k][0] * sg; ddir[1] += db[k][1] * sg; ddir[2] += db[k][2] * sg;
                }
            for (int k = nb * 3; k < M * 3; ++k) gs[k] = 0.f;      // coefficients above the active degree
            const float dot = ddir[0] * dirv[0] + ddir[1] * dirv[1] + ddir[2] * dirv[2];
            gm[0] += (ddir[0] - dirv[0] * dot) / len; gm[1] += (ddir[1] - dirv[1] * dot) / len; gm[2] += (ddir[2] - dirv[2] * dot) / len;
        }
    }
#pragma unroll
    for (int e = 0; e < 3; ++e) g_means3D[3 * (size_t)i + e] = gm[e];
    if (g_means2D != nullptr) { g_means2D[2 * (size_t)i] = gm2[0]; g_means2D[2 * (size_t)i + 1] = gm2[1]; }
    if (g_colors != nullptr) { g_colors[3 * (size_t)i] = gcol[0]; g_colors[3 * (size_t)i + 1] = gcol[1]; g_colors[3 * (size_t)i + 2] = gcol[2]; }
    g_opac[i] = gop;
    if (g_scales != nullptr) { g_scales[3 * (size_t)i] = gsc[0]; g_scales[3 * (size_t)i + 1] = gsc[1]; g_scales[3 * (size_t)i + 2] = gsc[2]; }
    if (g_rots != nullptr) { g_rots[4 * (size_t)i] = gq[0]; g_rots[4 * (size_t)i + 1] = gq[1]; g_rots[4 * (size_t)i + 2] = gq[2]; g_rots[4 * (size_t)i + 3] = gq[3]; }
    if (g_cov3D != nullptr) {
#pragma unroll
        for (int e = 0; e < 6; ++e) g_cov3D[6 * (size_t)i + e] = gc6[e];
    }
}

}  // namespace

extern "C" int gvf_rast_workspace_bytes(int P, int F, int H, int W, int64_t max_rendered, size_t* bytes) {
    if (!bytes || P < 0 || F <= 0 || H <= 0 || W <= 0 || max_rendered < 0) return GVF_EINVAL;
    Workspace w = carve(nullptr, (size_t)-1, P, F, H, W, max_rendered);
    *bytes = w.bytes + 256;
    (void)tile_sort_set_lds_limit();        // best effort here (no device in a CPU-only process); launch_tile_sort insists
    return GVF_OK;
}

extern "C" int gvf_rast_forward(const GvfRastSettings* st, const GvfRastFrame* frame_host, int P, int M,
                                const float* means3D, const float* shs, const float* colors_precomp,
                                const float* opacities, const float* scales, const float* rotations,
                                const float* cov3D_precomp, const float* subpixel_offset, void* workspace,
                                size_t workspace_bytes, int64_t max_rendered, float* out_color, float* out_alpha,
                                float* out_depth, int32_t* out_radii, uint32_t* out_num_rendered, void* stream) {
    if (!st || !frame_host) return GVF_EINVAL;
    if (P > 0) {
        if (!means3D || !opacities) return GVF_EINVAL;
        // exactly one colour source and exactly one covariance source (upstream wrapper contract)
        if ((shs == nullptr) == (colors_precomp == nullptr)) return GVF_EINVAL;
        const bool have_sr = scales != nullptr && rotations != nullptr;
        if (have_sr == (cov3D_precomp != nullptr)) return GVF_EINVAL;
    }
    return run_pipeline(*st, frame_host, 1, P, M, false, nullptr, means3D, scales, rotations, opacities, shs,
                        colors_precomp, cov3D_precomp, nullptr, 0, subpixel_offset, workspace, workspace_bytes,
                        max_rendered, out_color, out_alpha, out_depth, out_radii, out_num_rendered,
                        (hipStream_t)stream);
}

extern "C" int gvf_rast_forward_batched(const GvfRastSettings* st, const GvfRastFrame* frames_host, int F,
                                        const GvfGaussianActivation* act, int P, int M, const float* xyz_raw,
                                        const float* features_dc, const float* scaling_raw,
                                        const float* rotation_raw, const float* opacity_raw, const float* delta,
                                        int n_delta, void* workspace, size_t workspace_bytes,
                                        int64_t max_rendered, float* out_color, float* out_alpha, float* out_depth,
                                        int32_t* out_radii, uint32_t* out_num_rendered, void* stream) {
    if (!st || !frames_host || !act || F <= 0) return GVF_EINVAL;
    if (P > 0 && (!xyz_raw || !features_dc || !scaling_raw || !rotation_raw || !opacity_raw)) return GVF_EINVAL;
    if (act->scaling_activation != 0 && act->scaling_activation != 1) return GVF_EINVAL;
    for (int f = 0; f < F; ++f) {
        int di = frames_host[f].delta_index;
        if (di >= 0 && (delta == nullptr || di >= n_delta)) return GVF_EINVAL;
    }
    return run_pipeline(*st, frames_host, F, P, M, true, act, xyz_raw, scaling_raw, rotation_raw, opacity_raw,
                        features_dc, nullptr, nullptr, delta, n_delta, nullptr, workspace, workspace_bytes,
                        max_rendered, out_color, out_alpha, out_depth, out_radii, out_num_rendered,
                        (hipStream_t)stream);
}

// the batched call with the frames leaving as uint8 (the reference's post-process, utils/inference_utils.py:280-286, in the blend's epilogue)
extern "C" int gvf_rast_forward_batched_u8(const GvfRastSettings* st, const GvfRastFrame* frames_host, int F,
                                           const GvfGaussianActivation* act, int P, int M, const float* xyz_raw,
                                           const float* features_dc, const float* scaling_raw,
                                           const float* rotation_raw, const float* opacity_raw, const float* delta,
                                           int n_delta, void* workspace, size_t workspace_bytes,
                                           int64_t max_rendered, uint8_t* out_rgb_u8, uint32_t* out_num_rendered, void* stream) {
    if (!st || !frames_host || !act || F <= 0 || !out_rgb_u8) return GVF_EINVAL;
    if (P > 0 && (!xyz_raw || !features_dc || !scaling_raw || !rotation_raw || !opacity_raw)) return GVF_EINVAL;
    if (act->scaling_activation != 0 && act->scaling_activation != 1) return GVF_EINVAL;
    for (int f = 0; f < F; ++f) {
        int di = frames_host[f].delta_index;
        if (di >= 0 && (delta == nullptr || di >= n_delta)) return GVF_EINVAL;
    }
    return run_pipeline(*st, frames_host, F, P, M, true, act, xyz_raw, scaling_raw, rotation_raw, opacity_raw,
                        features_dc, nullptr, nullptr, delta, n_delta, nullptr, workspace, workspace_bytes,
                        max_rendered, nullptr, nullptr, nullptr, nullptr, out_num_rendered, (hipStream_t)stream, out_rgb_u8);
}

extern "C" int gvf_rast_backward_scratch_bytes(int P, size_t* bytes) {
    if (!bytes || P < 0) return GVF_EINVAL;
    *bytes = gvf_align_up((size_t)(P > 0 ? P : 1) * BWD_ACC * sizeof(float), 256);
    return GVF_OK;
}

extern "C" int gvf_rast_backward(const GvfRastSettings* st, const GvfRastFrame* frame_host, int P, int M,
                                 const float* means3D, const float* shs, const float* colors_precomp,
                                 const float* opacities, const float* scales, const float* rotations,
                                 const float* cov3D_precomp, const float* subpixel_offset, const void* workspace,
                                 size_t workspace_bytes, int64_t max_rendered, const float* dL_dcolor,
                                 const float* dL_dalpha, const float* dL_ddepth, void* scratch, size_t scratch_bytes,
                                 float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dshs, float* dL_dcolors,
                                 float* dL_dopacities, float* dL_dscales, float* dL_drotations, float* dL_dcov3D,
                                 void* stream_) {
    if (!st || !frame_host || !workspace || !dL_dcolor || P < 0) return GVF_EINVAL;
    const int H = st->image_height, W = st->image_width;
    if (H <= 0 || W <= 0 || st->sh_degree < 0 || st->sh_degree > 3) return GVF_EINVAL;
    if (st->mode != GVF_RAST_MODE_MIP && st->mode != GVF_RAST_MODE_DILATE) return GVF_EINVAL;
    if (P == 0) return GVF_OK;
    if (!means3D || !opacities || !scratch || !dL_dmeans3D || !dL_dopacities) return GVF_EINVAL;
    if ((shs == nullptr) == (colors_precomp == nullptr)) return GVF_EINVAL;
    const bool have_sr = scales != nullptr && rotations != nullptr;
    if (have_sr == (cov3D_precomp != nullptr)) return GVF_EINVAL;
    if (shs != nullptr && (M < (st->sh_degree + 1) * (st->sh_degree + 1) || M > MAX_SH_COEFFS || !dL_dshs)) return GVF_EINVAL;
    if (colors_precomp != nullptr && !dL_dcolors) return GVF_EINVAL;
    if (have_sr && (!dL_dscales || !dL_drotations)) return GVF_EINVAL;
    if (!have_sr && !dL_dcov3D) return GVF_EINVAL;
    size_t need = 0;
    gvf_rast_backward_scratch_bytes(P, &need);
    if (scratch_bytes < need) return GVF_ENOSPC;
    if ((((uintptr_t)workspace) & 255) != 0 || (((uintptr_t)scratch) & 15) != 0) return GVF_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    (void)hipGetLastError();
    // the layout of the forward call's workspace: same (P, F = 1, H, W, max_rendered) => same carve
    Workspace w = carve(const_cast<void*>(workspace), workspace_bytes, P, 1, H, W, max_rendered);
    if (!w.ok) return GVF_ENOSPC;
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE, ntiles = gx * gy;
    float* acc = (float*)scratch;
    if (hipMemsetAsync(acc, 0, (size_t)P * BWD_ACC * sizeof(float), stream) != hipSuccess) return GVF_ELAUNCH;
    if (max_rendered > 0) {
        if (dL_dalpha != nullptr || dL_ddepth != nullptr)
            hipLaunchKernelGGL(blend_backward_kernel<true>, dim3(ntiles), dim3(BLEND_THREADS), 0, stream, P, H, W, gx, st->bg[0],
                               st->bg[1], st->bg[2], w.ranges, w.ids, w.splats, subpixel_offset, dL_dcolor, dL_dalpha, dL_ddepth, acc);
        else
            hipLaunchKernelGGL(blend_backward_kernel<false>, dim3(ntiles), dim3(BLEND_THREADS), 0, stream, P, H, W, gx, st->bg[0],
                               st->bg[1], st->bg[2], w.ranges, w.ids, w.splats, subpixel_offset, dL_dcolor, dL_dalpha, dL_ddepth, acc);
    }
    GVF_CHECK_LAUNCH();
    BwdParams bp;
    bp.P = P; bp.M = M; bp.deg = st->sh_degree; bp.H = H; bp.W = W; bp.mode = st->mode;
    bp.kernel_size = st->kernel_size; bp.scale_modifier = st->scale_modifier; bp.fr = *frame_host;
#define GVF_PRE_BWD(D_)                                                                                                         \
    hipLaunchKernelGGL(preprocess_backward_kernel<D_>, dim3((P + PRE_THREADS - 1) / PRE_THREADS), dim3(PRE_THREADS), 0, stream, bp, \
                       means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, acc, dL_dmeans3D, dL_dmeans2D,    \
                       dL_dshs, dL_dcolors, dL_dopacities, dL_dscales, dL_drotations, dL_dcov3D)
    switch (shs != nullptr ? st->sh_degree : 0) {
        case 0: GVF_PRE_BWD(0); break;
        case 1: GVF_PRE_BWD(1); break;
        case 2: GVF_PRE_BWD(2); break;
        default: GVF_PRE_BWD(3); break;
    }
#undef GVF_PRE_BWD
    GVF_CHECK_LAUNCH();
    return GVF_OK;
}

extern "C" int gvf_gaussian_activate(const GvfGaussianActivation* act, int P, int M, const float* xyz_raw,
                                     const float* features_dc, const float* scaling_raw, const float* rotation_raw,
                                     const float* opacity_raw, const float* delta, float* means3D, float* scales,
                                     float* rotations, float* shs, float* opacities, void* stream) {
    if (!act || P < 0 || M < 1) return GVF_EINVAL;
    if (P == 0) return GVF_OK;
    if (!xyz_raw || !features_dc || !scaling_raw || !rotation_raw || !opacity_raw || !means3D || !scales ||
        !rotations || !shs || !opacities)
        return GVF_EINVAL;
    (void)hipGetLastError();
    hipLaunchKernelGGL(activate_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, *act, P, M, xyz_raw,
                       features_dc, scaling_raw, rotation_raw, opacity_raw, delta, means3D, scales, rotations, shs,
                       opacities);
    GVF_CHECK_LAUNCH();
    return GVF_OK;
}

extern "C" int gvf_tile_sort_u64(uint64_t* keys, const uint32_t* ranges, int nseg, uint32_t* ids, uint32_t* scratch, void* stream) {
    if (nseg < 0) return GVF_EINVAL;
    if (nseg == 0) return GVF_OK;
    if (!keys || !ranges || !ids || !scratch) return GVF_EINVAL;
    (void)hipGetLastError();
    return launch_tile_sort((hipStream_t)stream, reinterpret_cast<const uint2*>(ranges), keys, nullptr, ids, scratch, (uint32_t)nseg, 0);
}

extern "C" int gvf_rgb_to_u8(const float* rgb, uint8_t* out, int64_t n, void* stream) {
    if (n < 0) return GVF_EINVAL;
    if (n == 0) return GVF_OK;
    if (!rgb || !out || (((uintptr_t)rgb) & 15) || (((uintptr_t)out) & 3)) return GVF_EINVAL;
    (void)hipGetLastError();
    const long long n4 = n / 4;
    long long blocks = (n4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(rgb_to_u8_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(rgb), reinterpret_cast<uchar4*>(out), n4, rgb + n4 * 4,
                       out + n4 * 4, (int)(n - n4 * 4));
    GVF_CHECK_LAUNCH();
    return GVF_OK;
}

extern "C" int gvf_rast_profile_enable(int on) {
    if (on && !g_prof.on) {
        for (int c = 0; c < PROF_MAX_CALLS; ++c)
            for (int k = 0; k < PROF_EVENTS; ++k)
                if (hipEventCreate(&g_prof.ev[c][k]) != hipSuccess) return GVF_ELAUNCH;
        g_prof.on = true;
        g_prof.calls = 0;
    } else if (!on && g_prof.on) {
        for (int c = 0; c < PROF_MAX_CALLS; ++c)
            for (int k = 0; k < PROF_EVENTS; ++k) (void)hipEventDestroy(g_prof.ev[c][k]);
        g_prof.on = false;
        g_prof.calls = 0;
    }
    return GVF_OK;
}

extern "C" int64_t gvf_rast_shared_activation_calls(void) { return (int64_t)g_shared_calls.load(std::memory_order_relaxed); }

extern "C" int gvf_rast_profile_read(float* ms_sum, int* calls) {
    if (!ms_sum || !calls) return GVF_EINVAL;
    for (int k = 0; k < GVF_RAST_NSTAGES; ++k) ms_sum[k] = 0.f;
    *calls = 0;
    if (!g_prof.on) return GVF_OK;
    const int n_calls = g_prof.calls.load() < PROF_MAX_CALLS ? g_prof.calls.load() : PROF_MAX_CALLS;
    for (int c = 0; c < n_calls; ++c) {
        if (hipEventSynchronize(g_prof.ev[c][PROF_EVENTS - 1]) != hipSuccess) return GVF_ELAUNCH;
        for (int k = 0; k < GVF_RAST_NSTAGES; ++k) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, g_prof.ev[c][k], g_prof.ev[c][k + 1]) != hipSuccess) return GVF_ELAUNCH;
            ms_sum[k] += ms;
        }
    }
    *calls = n_calls;
    g_prof.calls = 0;
    return GVF_OK;
}

extern "C" int gvf_rast_sort_class_counts(const void* workspace, size_t workspace_bytes, int P, int F, int H, int W, int64_t max_rendered,
                                          uint32_t* counts_host, void* stream) {
    if (!workspace || !counts_host || H <= 0 || W <= 0 || P < 0 || F <= 0 || max_rendered < 0) return GVF_EINVAL;
    if ((((uintptr_t)workspace) & 255) != 0) return GVF_EINVAL;
    Workspace w = carve(const_cast<void*>(workspace), workspace_bytes, P, F, H, W, max_rendered);
    if (!w.ok) return GVF_ENOSPC;
    // test-only diagnostic.  The copy rides on the CALLER's stream (no blocking hipMemcpy on the legacy stream: that is illegal while another
    // host thread captures a hipGraph, the hazard utils/in_flight.py documents), then that stream is waited for.
    if (hipMemcpyAsync(counts_host, w.cls, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) return GVF_ELAUNCH;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return GVF_ELAUNCH;
    return GVF_OK;
}

extern "C" const char* gvf_version(void) { return "gvf_hip 0.1.0 gfx950"; }
