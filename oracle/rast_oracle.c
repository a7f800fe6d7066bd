/*
 * rast_oracle.c -- CPU restatement (plain C) of the tile-based 3D-Gaussian-splatting forward pass.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under gvfdiffusion_amd/ may import, link or call this file;
 * it is the checker used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 *
 * PARITY STATUS: "parity unpinned" at the pixel level.  The rasteriser arithmetic lives in two
 * third-party CUDA packages that are absent from /root/reference and installed from unpinned
 * git HEADs (setup.sh:111  slothfulxtx/diff-gaussian-rasterization -> `diff_gauss`;
 * setup.sh:220-227 autonomousvision/mip-splatting submodules/diff-gaussian-rasterization ->
 * `diff_gaussian_rasterization`).  The reference holds no test, golden image or fixture for them
 * (SURVEY.md section 4).  This file restates their published forward algorithm (3DGS, Kerbl et al.
 * 2023, forward.cu/rasterizer_impl.cu; mip-splatting 2D filter, Yu et al. 2024) anchored on the
 * reference's own call sites:
 *   renderers/gaussian_render.py:110-143  settings (tanfov, kernel_size, bg, V^T, (PV)^T, campos)
 *   renderers/gaussian_render.py:198-220  operator call and the two return arities
 *   renderers/gaussian_render.py:176-181  SH -> RGB: eval_sh(...) + 0.5, clamp >= 0
 *   renderers/sh_utils.py:57-112          SH basis (pinned by tests/golden/sh_golden.npz)
 *   representations/gaussian/gaussian_model.py:18-22 + general_utils.py:78-110  cov3D = (R S)(R S)^T
 *   representations/gaussian/gaussian_model.py:84-114                         activations / deltas
 * Sub-results that the reference's Python mirrors pin (SH, rotation, projection, activations) are
 * checked against golden vectors generated from the reference (tests/golden/make_golden.py).
 *
 * Floating-point contract shared with the HIP kernels (so that every discrete decision -- cull,
 * radius ceil, tile rect, sort order -- is bit-identical): IEEE binary32, no contraction
 * (compile with -ffp-contract=off), fused multiply-adds only where fmaf() is written, division
 * and sqrt correctly rounded (the activations' exp / log1p included: act_expf / act_log1pf below, the same
 * sequence on both sides since round 6).  The only non-shared primitive is exp() in the blend (libm expf here,
 * v_exp_f32 on the device): pixels where a discrete blend decision sits within float noise of its
 * threshold are reported in `out_flags` so tests can account for them explicitly.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define TILE 16
#define MODE_MIP 0
#define MODE_DILATE 1

/* SH constants: renderers/sh_utils.py:26-45 */
static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                               0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                               -0.5900435899266435f};

typedef struct {
    float depth;
    float x, y;          /* pixel-space mean */
    float ca, cb, cc;    /* conic */
    float op;            /* opacity (x mip coef) */
    float r, g, b;
    int radius;
    int x0, y0, x1, y1;  /* tile rect [x0,x1) x [y0,y1) of the 3-sigma radius (what upstream bins) */
    float hx, hy;        /* half extents of the alpha >= 1/255 box (see cull_extent) */
    int tx0, ty0, tx1, ty1; /* (x0..y1) intersected with the tiles that box reaches */
} Geom;

/* ---- instance culling (an optimisation of the HIP path, restated here so that its instance counts can be
 * checked bit for bit; OFF by default = upstream binning).  A (Gaussian, tile) instance is dropped when
 * alpha = op * exp(power) < 1/255 at every pixel of the tile -- the blend would skip it at each of them, so the
 * image is unchanged (tests/test_oracle_rast.py asserts bit-equal images with the toggle on and off).
 * Everything is built from correctly rounded +,*,/,sqrt,floor,ceil and bit operations (no libm log), so the
 * device reproduces hx, hy and the tile set exactly. */
static int g_tight_binning = 0;
void gvfo_set_tight_binning(int on) { g_tight_binning = on; }

/* ln(z) upper bound for normal z > 0: exponent + min of the tangents of log2 at 1, 1.25, 1.5, 1.75, 2 */
static float ln_upper(float z) {
    uint32_t b; memcpy(&b, &z, 4);
    int e = (int)(b >> 23) - 127;
    uint32_t mb = (b & 0x7fffffu) | 0x3f800000u;
    float m; memcpy(&m, &mb, 4);
    float L = (m - 1.0f) * 1.4426951f;
    L = fminf(L, 0.32192809f + (m - 1.25f) * 1.1541561f);
    L = fminf(L, 0.5849625f + (m - 1.5f) * 0.96179669f);
    L = fminf(L, 0.80735492f + (m - 1.75f) * 0.8243972f);
    L = fminf(L, 1.0f + (m - 2.0f) * 0.72134752f);
    return ((float)e + (L + 1e-5f)) * 0.69314724f;
}

/* hx < 0: opacity < 1/255, never visible; +inf: degenerate conic, keep everywhere */
static void cull_extent(float ca, float cb, float cc, float op, float* hx, float* hy) {
    if (op < 1.0f / 255.0f) { *hx = -1.0f; *hy = -1.0f; return; }
    float det = ca * cc - cb * cb;
    if (!(det > 0.0f) || !(ca > 0.0f) || !(cc > 0.0f)) { *hx = INFINITY; *hy = INFINITY; return; }
    float tau = 2.0f * ln_upper(255.0f * op) * 1.001f + 1e-3f;
    float inv = 1.0f / det;
    *hx = sqrtf(tau * cc * inv) * 1.001f + 0.01f;
    *hy = sqrtf(tau * ca * inv) * 1.001f + 0.01f;
}

/* tiles t with a pixel p in [16t, 16t+15] such that |p - c| <= h, clipped to [lo, hi) */
static void tight_range(float c, float h, int lo, int hi, int n, int* t0, int* t1) {
    if (h < 0.0f) { *t0 = lo; *t1 = lo; return; }
    float a = fminf(fmaxf(ceilf((c - h - (float)(TILE - 1)) / (float)TILE), 0.0f), (float)n);
    float b = fminf(fmaxf(floorf((c + h) / (float)TILE) + 1.0f, 0.0f), (float)n);
    int x0 = (int)a > lo ? (int)a : lo, x1 = (int)b < hi ? (int)b : hi;
    if (x1 < x0) x1 = x0;
    *t0 = x0; *t1 = x1;
}

/* p' = M p with M given as the 16 floats of V^T row-major (column-major V): upstream transformPoint4x3 */
static void xform43(const float* m, const float* p, float* o) {
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}
static void xform44(const float* m, const float* p, float* o) {
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
    o[3] = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}

/* Sigma = R S S^T R^T, R from quaternion (r,x,y,z) used UN-normalised (upstream computeCov3D;
 * reference mirror: general_utils.py:78-110 build_scaling_rotation, which normalises first --
 * GaussianModel.get_rotation already returns unit quaternions, so both agree on that input). */
static void cov3d_from_scale_rot(const float* s, float mod, const float* q, float* c6) {
    float sx = mod * s[0], sy = mod * s[1], sz = mod * s[2];
    float r = q[0], x = q[1], y = q[2], z = q[3];
    float R00 = 1.f - 2.f * (y * y + z * z), R01 = 2.f * (x * y - r * z), R02 = 2.f * (x * z + r * y);
    float R10 = 2.f * (x * y + r * z), R11 = 1.f - 2.f * (x * x + z * z), R12 = 2.f * (y * z - r * x);
    float R20 = 2.f * (x * z - r * y), R21 = 2.f * (y * z + r * x), R22 = 1.f - 2.f * (x * x + y * y);
    /* L = R diag(s): L[i][j] = R[i][j] * s[j];  Sigma = L L^T */
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

/* colour = clamp_min(eval_sh(deg, sh, normalize(p - campos)) + 0.5, 0)
 * (renderers/gaussian_render.py:176-181; sh_utils.py:57-112; shs laid out [M][3]) */
static void sh_to_rgb(int deg, int M, const float* sh, const float* p, const float* cam, float* rgb) {
    float dx = p[0] - cam[0], dy = p[1] - cam[1], dz = p[2] - cam[2];
    float len = sqrtf(dx * dx + dy * dy + dz * dz);
    float x = dx / len, y = dy / len, z = dz / len;
    (void)M;
    for (int c = 0; c < 3; ++c) {
        float res = SH_C0 * sh[0 * 3 + c];
        if (deg > 0) {
            res = res - SH_C1 * y * sh[1 * 3 + c] + SH_C1 * z * sh[2 * 3 + c] - SH_C1 * x * sh[3 * 3 + c];
            if (deg > 1) {
                float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                res = res + SH_C2[0] * xy * sh[4 * 3 + c] + SH_C2[1] * yz * sh[5 * 3 + c] +
                      SH_C2[2] * (2.0f * zz - xx - yy) * sh[6 * 3 + c] + SH_C2[3] * xz * sh[7 * 3 + c] +
                      SH_C2[4] * (xx - yy) * sh[8 * 3 + c];
                if (deg > 2) {
                    res = res + SH_C3[0] * y * (3.0f * xx - yy) * sh[9 * 3 + c] +
                          SH_C3[1] * xy * z * sh[10 * 3 + c] +
                          SH_C3[2] * y * (4.0f * zz - xx - yy) * sh[11 * 3 + c] +
                          SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * sh[12 * 3 + c] +
                          SH_C3[4] * x * (4.0f * zz - xx - yy) * sh[13 * 3 + c] +
                          SH_C3[5] * z * (xx - yy) * sh[14 * 3 + c] +
                          SH_C3[6] * x * (xx - 3.0f * yy) * sh[15 * 3 + c];
                }
            }
        }
        res += 0.5f;
        rgb[c] = res < 0.f ? 0.f : res;
    }
}

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

/* R1: per-Gaussian preprocess.  Returns 1 if the Gaussian is visible (radius > 0). */
static int preprocess_one(int i, int M, int deg, const float* means3D, const float* shs,
                          const float* colors_precomp, const float* opacities, const float* scales,
                          const float* rotations, const float* cov3D_precomp, int H, int W,
                          float tanfovx, float tanfovy, float kernel_size, float scale_modifier, int mode,
                          const float* view, const float* proj, const float* campos, Geom* g) {
    memset(g, 0, sizeof(*g));
    const float* p = means3D + 3 * (size_t)i;
    float pv[3];
    xform43(view, p, pv);
    if (pv[2] <= 0.2f) return 0; /* near cull (upstream in_frustum) */

    float ph[4];
    xform44(proj, p, ph);
    float pw = 1.0f / (ph[3] + 0.0000001f);
    float projx = ph[0] * pw, projy = ph[1] * pw;

    float c6[6];
    if (cov3D_precomp) {
        for (int k = 0; k < 6; ++k) c6[k] = cov3D_precomp[6 * (size_t)i + k];
    } else {
        cov3d_from_scale_rot(scales + 3 * (size_t)i, scale_modifier, rotations + 4 * (size_t)i, c6);
    }

    /* EWA projection (upstream computeCov2D): cov2D = J W Sigma W^T J^T */
    float focal_x = (float)W / (2.0f * tanfovx);
    float focal_y = (float)H / (2.0f * tanfovy);
    float limx = 1.3f * tanfovx, limy = 1.3f * tanfovy;
    float txtz = pv[0] / pv[2], tytz = pv[1] / pv[2];
    float tx = fminf(limx, fmaxf(-limx, txtz)) * pv[2];
    float ty = fminf(limy, fmaxf(-limy, tytz)) * pv[2];
    float tz = pv[2];
    float J00 = focal_x / tz, J02 = -(focal_x * tx) / (tz * tz);
    float J11 = focal_y / tz, J12 = -(focal_y * ty) / (tz * tz);
    /* A = J * Wrot, Wrot[r][c] = view[c*4 + r] (rotation rows of V) */
    float A0[3], A1[3];
    for (int c = 0; c < 3; ++c) {
        float w0 = view[c * 4 + 0], w1 = view[c * 4 + 1], w2 = view[c * 4 + 2];
        A0[c] = J00 * w0 + J02 * w2;
        A1[c] = J11 * w1 + J12 * w2;
    }
    /* B = A * Sigma (2x3), cov = B * A^T */
    float S[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
    float B0[3], B1[3];
    for (int c = 0; c < 3; ++c) {
        B0[c] = A0[0] * S[0][c] + A0[1] * S[1][c] + A0[2] * S[2][c];
        B1[c] = A1[0] * S[0][c] + A1[1] * S[1][c] + A1[2] * S[2][c];
    }
    float cxx = B0[0] * A0[0] + B0[1] * A0[1] + B0[2] * A0[2];
    float cxy = B0[0] * A1[0] + B0[1] * A1[1] + B0[2] * A1[2];
    float cyy = B1[0] * A1[0] + B1[1] * A1[1] + B1[2] * A1[2];

    float coef = 1.0f;
    if (mode == MODE_MIP) {
        /* mip-splatting 2D filter: opacity compensation by sqrt(det0/det1) */
        float det0 = fmaxf(1e-6f, cxx * cyy - cxy * cxy);
        float det1 = fmaxf(1e-6f, (cxx + kernel_size) * (cyy + kernel_size) - cxy * cxy);
        coef = sqrtf(det0 / (det1 + 1e-6f) + 1e-6f);
        if (det0 <= 1e-6f || det1 <= 1e-6f) coef = 0.0f;
        cxx += kernel_size;
        cyy += kernel_size;
    } else {
        cxx += 0.3f;
        cyy += 0.3f;
    }

    float det = cxx * cyy - cxy * cxy;
    if (det == 0.0f) return 0;
    float det_inv = 1.f / det;
    float ca = cyy * det_inv, cb = -cxy * det_inv, cc = cxx * det_inv;

    float mid = 0.5f * (cxx + cyy);
    float lam1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
    float lam2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
    float my_radius = ceilf(3.f * sqrtf(fmaxf(lam1, lam2)));
    float px = ((projx + 1.0f) * (float)W - 1.0f) * 0.5f;
    float py = ((projy + 1.0f) * (float)H - 1.0f) * 0.5f;

    int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    int x0 = imin(gx, imax(0, (int)((px - my_radius) / (float)TILE)));
    int y0 = imin(gy, imax(0, (int)((py - my_radius) / (float)TILE)));
    int x1 = imin(gx, imax(0, (int)((px + my_radius + (float)(TILE - 1)) / (float)TILE)));
    int y1 = imin(gy, imax(0, (int)((py + my_radius + (float)(TILE - 1)) / (float)TILE)));
    if ((x1 - x0) * (y1 - y0) == 0) return 0;

    if (colors_precomp) {
        g->r = colors_precomp[3 * (size_t)i + 0];
        g->g = colors_precomp[3 * (size_t)i + 1];
        g->b = colors_precomp[3 * (size_t)i + 2];
    } else {
        float rgb[3];
        sh_to_rgb(deg, M, shs + (size_t)i * M * 3, p, campos, rgb);
        g->r = rgb[0]; g->g = rgb[1]; g->b = rgb[2];
    }
    g->depth = pv[2];
    g->x = px; g->y = py;
    g->ca = ca; g->cb = cb; g->cc = cc;
    g->op = opacities[i] * coef;
    g->radius = (int)my_radius;
    g->x0 = x0; g->y0 = y0; g->x1 = x1; g->y1 = y1;
    cull_extent(ca, cb, cc, g->op, &g->hx, &g->hy);
    tight_range(px, g->hx, x0, x1, gx, &g->tx0, &g->tx1);
    tight_range(py, g->hy, y0, y1, gy, &g->ty0, &g->ty1);
    if (g_tight_binning) { g->x0 = g->tx0; g->x1 = g->tx1; g->y0 = g->ty0; g->y1 = g->ty1; }
    return 1;
}

/* Exported: per-Gaussian geometry, for stage-level parity tests.
 * geom_out[P][24]: depth,x,y,ca,cb,cc,op,r,g,b,radius,x0,y0,x1,y1,visible, tx0,ty0,tx1,ty1, hx,hy, 0,0 (all as float);
 * x0..y1 follow the binning toggle, tx0..ty1 are always the culled rect. */
int gvfo_preprocess(int P, int M, int deg, const float* means3D, const float* shs,
                    const float* colors_precomp, const float* opacities, const float* scales,
                    const float* rotations, const float* cov3D_precomp, int H, int W, float tanfovx,
                    float tanfovy, float kernel_size, float scale_modifier, int mode, const float* view,
                    const float* proj, const float* campos, float* geom_out) {
    for (int i = 0; i < P; ++i) {
        Geom g;
        int vis = preprocess_one(i, M, deg, means3D, shs, colors_precomp, opacities, scales, rotations,
                                 cov3D_precomp, H, W, tanfovx, tanfovy, kernel_size, scale_modifier, mode,
                                 view, proj, campos, &g);
        float* o = geom_out + 24 * (size_t)i;
        o[0] = g.depth; o[1] = g.x; o[2] = g.y; o[3] = g.ca; o[4] = g.cb; o[5] = g.cc; o[6] = g.op;
        o[7] = g.r; o[8] = g.g; o[9] = g.b; o[10] = (float)g.radius;
        o[11] = (float)g.x0; o[12] = (float)g.y0; o[13] = (float)g.x1; o[14] = (float)g.y1;
        o[15] = (float)vis;
        o[16] = (float)g.tx0; o[17] = (float)g.ty0; o[18] = (float)g.tx1; o[19] = (float)g.ty1;
        o[20] = g.hx; o[21] = g.hy; o[22] = 0.0f; o[23] = 0.0f;
    }
    return 0;
}

typedef struct { uint64_t key; uint32_t id; } Inst;

/* (tile, depth-bits, emission order) -- the order a stable radix sort of (tile<<32 | depth_bits)
 * keys produces (R3 + R4) */
static int inst_cmp(const void* a, const void* b) {
    const Inst* x = (const Inst*)a; const Inst* y = (const Inst*)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->id < y->id ? -1 : (x->id > y->id ? 1 : 0);
}

/* flag bits reported per pixel */
#define FLAG_ALPHA_SKIP 1   /* some alpha within 2e-4 (relative) of the 1/255 skip threshold */
#define FLAG_T_STOP     2   /* some test_T within 2e-4 (relative) of the 1e-4 termination threshold */
#define FLAG_POWER_POS  4   /* some power within 1e-6 of 0 */

/* R6 for one pixel over a depth-sorted splat list */
static void blend_pixel(const Geom* geom, const uint32_t* ids, int n, float pxf, float pyf, const float* bg,
                        float* out3, float* out_alpha, float* out_depth, uint8_t* flag) {
    float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, Dacc = 0.f;
    uint8_t fl = 0;
    for (int k = 0; k < n; ++k) {
        const Geom* g = geom + ids[k];
        float dx = g->x - pxf, dy = g->y - pyf;
        float power = -0.5f * (g->ca * dx * dx + g->cc * dy * dy) - g->cb * dx * dy;
        if (fabsf(power) < 1e-6f) fl |= FLAG_POWER_POS;
        if (power > 0.0f) continue;
        float alpha = fminf(0.99f, g->op * expf(power));
        if (fabsf(alpha - 1.0f / 255.0f) < 2e-4f * (1.0f / 255.0f)) fl |= FLAG_ALPHA_SKIP;
        if (alpha < 1.0f / 255.0f) continue;
        float test_T = T * (1.f - alpha);
        if (fabsf(test_T - 0.0001f) < 2e-4f * 0.0001f) fl |= FLAG_T_STOP;
        if (test_T < 0.0001f) break;
        float w = alpha * T;
        C0 = fmaf(g->r, w, C0);
        C1 = fmaf(g->g, w, C1);
        C2 = fmaf(g->b, w, C2);
        Dacc = fmaf(g->depth, w, Dacc);
        T = test_T;
    }
    out3[0] = fmaf(T, bg[0], C0);
    out3[1] = fmaf(T, bg[1], C1);
    out3[2] = fmaf(T, bg[2], C2);
    if (out_alpha) *out_alpha = 1.0f - T;
    if (out_depth) *out_depth = Dacc;
    if (flag) *flag = fl;
}

/* Full forward: R1 preprocess -> R2/R3 instance emission -> R4 (tile,depth) stable sort -> R5 ranges
 * -> R6 per-tile front-to-back blend.  out_color[3][H][W]; optional out_alpha/out_depth [H][W],
 * out_radii[P], out_num_rendered[1], out_flags[H][W] (uint8).  nthreads<=0: all cores. */
int gvfo_render(int P, int M, int deg, const float* means3D, const float* shs, const float* colors_precomp,
                const float* opacities, const float* scales, const float* rotations,
                const float* cov3D_precomp, const float* subpixel_offset, int H, int W, float tanfovx,
                float tanfovy, float kernel_size, float scale_modifier, int mode, const float* view,
                const float* proj, const float* campos, const float* bg, float* out_color, float* out_alpha,
                float* out_depth, int32_t* out_radii, uint32_t* out_num_rendered, uint8_t* out_flags,
                int nthreads) {
    if (P < 0 || H <= 0 || W <= 0 || deg < 0 || deg > 3) return -1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    Geom* geom = (Geom*)malloc(sizeof(Geom) * (size_t)(P > 0 ? P : 1));
    if (!geom) return -2;
    int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;

#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; ++i) {
        preprocess_one(i, M, deg, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                       H, W, tanfovx, tanfovy, kernel_size, scale_modifier, mode, view, proj, campos,
                       &geom[i]);
    }
    size_t D = 0;
    for (int i = 0; i < P; ++i) {
        if (out_radii) out_radii[i] = geom[i].radius;
        D += (size_t)(geom[i].x1 - geom[i].x0) * (size_t)(geom[i].y1 - geom[i].y0);
    }
    if (out_num_rendered) *out_num_rendered = (uint32_t)D;

    Inst* inst = (Inst*)malloc(sizeof(Inst) * (D > 0 ? D : 1));
    if (!inst) { free(geom); return -2; }
    size_t o = 0;
    for (int i = 0; i < P; ++i) {
        const Geom* g = &geom[i];
        if (g->radius <= 0) continue;
        uint32_t dbits;
        memcpy(&dbits, &g->depth, 4);
        for (int y = g->y0; y < g->y1; ++y)
            for (int x = g->x0; x < g->x1; ++x) {
                uint64_t tile = (uint64_t)(y * gx + x);
                inst[o].key = (tile << 32) | dbits;
                inst[o].id = (uint32_t)i;
                ++o;
            }
    }
    qsort(inst, D, sizeof(Inst), inst_cmp);

    int ntiles = gx * gy;
    uint32_t* rng = (uint32_t*)calloc((size_t)ntiles * 2, sizeof(uint32_t));
    uint32_t* ids = (uint32_t*)malloc(sizeof(uint32_t) * (D > 0 ? D : 1));
    for (size_t k = 0; k < D; ++k) {
        ids[k] = inst[k].id;
        uint32_t t = (uint32_t)(inst[k].key >> 32);
        if (k == 0 || t != (uint32_t)(inst[k - 1].key >> 32)) rng[2 * t] = (uint32_t)k;
        if (k == D - 1 || t != (uint32_t)(inst[k + 1].key >> 32)) rng[2 * t + 1] = (uint32_t)(k + 1);
    }
    free(inst);

#pragma omp parallel for schedule(dynamic, 1)
    for (int t = 0; t < ntiles; ++t) {
        int tx = t % gx, ty = t / gx;
        uint32_t s = rng[2 * t], e = rng[2 * t + 1];
        for (int ly = 0; ly < TILE; ++ly)
            for (int lx = 0; lx < TILE; ++lx) {
                int px = tx * TILE + lx, py = ty * TILE + ly;
                if (px >= W || py >= H) continue;
                size_t pid = (size_t)py * W + px;
                float pxf = (float)px, pyf = (float)py;
                if (subpixel_offset) { pxf += subpixel_offset[2 * pid]; pyf += subpixel_offset[2 * pid + 1]; }
                float c3[3];
                blend_pixel(geom, ids + s, (int)(e - s), pxf, pyf, bg, c3, out_alpha ? out_alpha + pid : NULL,
                            out_depth ? out_depth + pid : NULL, out_flags ? out_flags + pid : NULL);
                out_color[0 * (size_t)H * W + pid] = c3[0];
                out_color[1 * (size_t)H * W + pid] = c3[1];
                out_color[2 * (size_t)H * W + pid] = c3[2];
            }
    }
    free(rng); free(ids); free(geom);
    return 0;
}

typedef struct { float depth; uint32_t id; } DepthId;
static int depthid_cmp(const void* a, const void* b) {
    const DepthId* x = (const DepthId*)a; const DepthId* y = (const DepthId*)b;
    uint32_t xb, yb;
    memcpy(&xb, &x->depth, 4); memcpy(&yb, &y->depth, 4);
    if (xb != yb) return xb < yb ? -1 : 1;
    return x->id < y->id ? -1 : (x->id > y->id ? 1 : 0);
}

/* Independent cross-check of the tile pipeline: for every pixel, gather every Gaussian whose tile
 * rect covers the pixel's tile, depth-sort that list on its own, composite.  O(pixels x P): small
 * cases only. */
int gvfo_render_brute(int P, int M, int deg, const float* means3D, const float* shs,
                      const float* colors_precomp, const float* opacities, const float* scales,
                      const float* rotations, const float* cov3D_precomp, int H, int W, float tanfovx,
                      float tanfovy, float kernel_size, float scale_modifier, int mode, const float* view,
                      const float* proj, const float* campos, const float* bg, float* out_color,
                      float* out_alpha, float* out_depth) {
    Geom* geom = (Geom*)malloc(sizeof(Geom) * (size_t)(P > 0 ? P : 1));
    for (int i = 0; i < P; ++i)
        preprocess_one(i, M, deg, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                       H, W, tanfovx, tanfovy, kernel_size, scale_modifier, mode, view, proj, campos,
                       &geom[i]);
#pragma omp parallel
    {
        DepthId* list = (DepthId*)malloc(sizeof(DepthId) * (size_t)(P > 0 ? P : 1));
        uint32_t* ids = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(P > 0 ? P : 1));
#pragma omp for schedule(dynamic, 8)
        for (int py = 0; py < H; ++py)
            for (int px = 0; px < W; ++px) {
                int tx = px / TILE, ty = py / TILE, n = 0;
                for (int i = 0; i < P; ++i) {
                    const Geom* g = &geom[i];
                    if (g->radius > 0 && tx >= g->x0 && tx < g->x1 && ty >= g->y0 && ty < g->y1) {
                        list[n].depth = g->depth; list[n].id = (uint32_t)i; ++n;
                    }
                }
                qsort(list, (size_t)n, sizeof(DepthId), depthid_cmp);
                for (int k = 0; k < n; ++k) ids[k] = list[k].id;
                size_t pid = (size_t)py * W + px;
                float c3[3];
                blend_pixel(geom, ids, n, (float)px, (float)py, bg, c3, out_alpha ? out_alpha + pid : NULL,
                            out_depth ? out_depth + pid : NULL, NULL);
                out_color[0 * (size_t)H * W + pid] = c3[0];
                out_color[1 * (size_t)H * W + pid] = c3[1];
                out_color[2 * (size_t)H * W + pid] = c3[2];
            }
        free(list); free(ids);
    }
    free(geom);
    return 0;
}

/* ---- exp / log1p of the activations: ONE arithmetic, written here and in gvfdiffusion_amd/csrc/rast.hip (act_expf / act_log1pf) --------
 * gaussian_model.py:84-114 activates with torch's exp / softplus (= log1p(exp(x)) below its threshold of 20) / sigmoid.  Up to round 5 the
 * oracle called libm (expf, log1pf) and the device its own math library: the two agree to an ulp or two, and at 262 144 Gaussians x 24 frames
 * a few radii and tile rects flipped, so the full-size delta-frame tests could not hold the exact-radii rule.  Now both
 * sides evaluate the SAME sequence of correctly rounded operations -- fmaf where written, + - * /, float <-> int conversions, bit operations,
 * no contraction -- so scales, opacities and everything derived from them are bit-identical on the CPU and on the device.  Accuracy against
 * libm / float64 is asserted in tests/test_oracle_rast.py::test_shared_activation_arithmetic_stays_within_2ulp_of_libm (measured: < 1 ulp).
 *   act_expf:   k = round(x log2 e); r = x - k ln2 (two-constant Cody-Waite, fused); e^r by its degree-7 Taylor polynomial in Horner form
 *               (|r| <= 0.347: truncation 5e-9 relative); scaled by 2^k in two exact steps.
 *   act_log1pf: for y >= 0 (y = e^x): u = 1 + y = 2^k m with m in [sqrt 1/2, sqrt 2), the rounding error of 1 + y carried as c = (y - (u - 1)) / u;
 *               log m from s = f / (2 + f), f = m - 1, and an even polynomial in s (the classic fdlibm decomposition of log1p). */
static float bits_to_f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static uint32_t f_to_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

static float act_expf(float x) {
    if (x != x) return x;
    if (x > 88.72283f) return INFINITY;
    if (x < -103.97208f) return 0.0f;
    const float kf = x * 1.44269502f + (x < 0.0f ? -0.5f : 0.5f);
    const int k = (int)kf;                                   /* truncation toward zero = round half away of x log2 e */
    const float t = (float)k;
    float r = fmaf(t, -0.693145751953125f, x);               /* ln 2 = 0.693145751953125 (16 bits: t * it is exact) + 1.42860677e-6 */
    r = fmaf(t, -1.42860677e-6f, r);
    float p = 1.98412698e-4f;                                /* 1/7! */
    p = fmaf(p, r, 1.38888889e-3f);                          /* 1/6! */
    p = fmaf(p, r, 8.33333377e-3f);                          /* 1/5! */
    p = fmaf(p, r, 4.16666679e-2f);                          /* 1/4! */
    p = fmaf(p, r, 1.66666672e-1f);                          /* 1/3! */
    p = fmaf(p, r, 0.5f);
    p = fmaf(p, r, 1.0f);
    p = fmaf(p, r, 1.0f);
    const int k1 = k / 2, k2 = k - k1;                       /* k in [-150, 128]: both factors are normal powers of two */
    return (p * bits_to_f((uint32_t)(k1 + 127) << 23)) * bits_to_f((uint32_t)(k2 + 127) << 23);
}

static float act_log1pf(float y) {                           /* y >= 0 (or NaN) */
    if (!(y >= 5.9604645e-8f)) return y;                     /* < 2^-24: log1p(y) = y to the last bit (and NaN) */
    if (y > 3.4028235e38f) return y;                         /* +inf */
    int k = 0;
    float c = 0.0f, f = y;
    if (y >= 0.41421354f) {                                  /* 1 + y >= sqrt 2: split off the exponent */
        const float u = 1.0f + y;
        uint32_t iu = f_to_bits(u) + (0x3f800000u - 0x3f3504f3u);
        k = (int)(iu >> 23) - 127;
        if (k < 25) c = (k >= 2 ? 1.0f - (u - y) : y - (u - 1.0f)) / u;
        iu = (iu & 0x007fffffu) + 0x3f3504f3u;
        f = bits_to_f(iu) - 1.0f;
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

/* test hook: the two functions over arrays (tests/test_oracle_rast.py compares them with libm / float64) */
void gvfo_act_math(int n, const float* x, float* out_exp, float* out_log1p) {
    for (int i = 0; i < n; ++i) {
        if (out_exp) out_exp[i] = act_expf(x[i]);
        if (out_log1p) out_log1p[i] = act_log1pf(x[i]);
    }
}

/* G1: GaussianModel activations with optional delta (gaussian_model.py:84-114; delta layout
 * [xyz3|scale3|rot4|rgb3|op1], gaussian_render.py:155-160).  scaling_activation 0=exp 1=softplus
 * (torch softplus: beta 1, threshold 20). */
int gvfo_activate(int P, int M, const float* aabb, float scale_bias, float opacity_bias, float min_kernel,
                  int scaling_activation, const float* xyz_raw, const float* features_dc,
                  const float* scaling_raw, const float* rotation_raw, const float* opacity_raw,
                  const float* delta, float* means3D, float* scales, float* rotations, float* shs,
                  float* opacities) {
    for (int i = 0; i < P; ++i) {
        const float* d = delta ? delta + 14 * (size_t)i : NULL;
        for (int k = 0; k < 3; ++k) {
            float v = xyz_raw[3 * (size_t)i + k] * aabb[3 + k] + aabb[k];
            means3D[3 * (size_t)i + k] = d ? v + d[k] : v;
        }
        for (int k = 0; k < 3; ++k) {
            float x = scaling_raw[3 * (size_t)i + k] + scale_bias;
            if (d) x = x + d[3 + k];
            float s = scaling_activation == 0 ? act_expf(x) : (x > 20.0f ? x : act_log1pf(act_expf(x)));
            scales[3 * (size_t)i + k] = sqrtf(s * s + min_kernel * min_kernel);
        }
        float q[4];
        for (int k = 0; k < 4; ++k) {
            q[k] = rotation_raw[4 * (size_t)i + k] + (k == 0 ? 1.0f : 0.0f);
            if (d) q[k] = q[k] + d[6 + k];
        }
        /* F.normalize: x / max(||x||, 1e-12) */
        float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        n = fmaxf(n, 1e-12f);
        for (int k = 0; k < 4; ++k) rotations[4 * (size_t)i + k] = q[k] / n;
        for (int m = 0; m < M; ++m)
            for (int c = 0; c < 3; ++c) {
                float v = features_dc[((size_t)i * M + m) * 3 + c];
                /* delta (P,1,3) broadcasts over the M coefficients (get_features_with_delta) */
                shs[((size_t)i * M + m) * 3 + c] = d ? v + d[10 + c] : v;
            }
        float x = opacity_raw[i] + opacity_bias;
        if (d) x = x + d[13];
        opacities[i] = 1.0f / (1.0f + act_expf(-x));
    }
    return 0;
}

int gvfo_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
