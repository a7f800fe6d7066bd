/*
 * rast_bwd_oracle.c -- CPU restatement (plain C, DOUBLE precision) of the 3D-Gaussian-splatting rasteriser's
 * forward AND backward pass: the checker of gvf_rast_backward() (csrc/rast_bwd.hip).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under gvfdiffusion_amd/ may import, link or call this file.
 *
 * PARITY STATUS: "parity unpinned" against the reference -- the operator's backward lives in the same two
 * third-party CUDA packages as its forward (see rast_oracle.c; setup.sh:111,220-227), absent from /root/reference,
 * which holds no gradient test for them.  The anchor is the reference's use of the operator as a differentiable
 * one: renderers/gaussian_render.py:198-220 (autograd through GaussianRasterizer), train_vae.py:321-352 (render
 * loss back-propagated into the Gaussians).  What pins THIS file is mathematics: tests/test_oracle_rast_bwd.py
 * checks every returned gradient against central finite differences of gvfo64_forward() below (double precision,
 * so the comparison is tight), and gvfo64_forward() against the float forward oracle (rast_oracle.c).
 *
 * Conventions taken over from the published upstream backward (3DGS backward.cu; they matter for parity of the
 * numbers a training run sees, and are the only places where the returned gradient is not the exact derivative):
 *   - alpha = min(0.99, opacity * G): the gradient passes THROUGH the clamp (treated as identity);
 *   - the view-space position (tx, ty) used in the EWA Jacobian is clamped to 1.3 tan(fov) * tz: a clamped
 *     coordinate gets no gradient, and the dependence of the clamp bound on tz is ignored;
 *   - the gradient of the screen-space mean is reported in NDC units (d/d(ndc) = d/d(pixel) * 0.5 * W resp. H),
 *     the quantity `screenspace_points.grad` holds upstream.
 * mip mode: the opacity compensation coef = sqrt(det0 / (det1 + 1e-6) + 1e-6) is differentiated exactly.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define TILE 16
#define MODE_MIP 0
#define MODE_DILATE 1

static const double SH_C0 = 0.28209479177387814;
static const double SH_C1 = 0.4886025119029199;
static const double SH_C2[5] = {1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792,
                                0.5462742152960396};
static const double SH_C3[7] = {-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
                                -0.4570457994644658, 1.445305721320277, -0.5900435899266435};

typedef struct {
    int visible;
    double depth, x, y, ca, cb, cc, op, rgb[3];
    int x0, y0, x1, y1;
    /* intermediates kept for the backward pass */
    double pv[3], c6[6], A0[3], A1[3], cxx, cxy, cyy; /* cxx.. BEFORE the filter */
    double coef, tx, ty, tz, xmul, ymul, pw, ph[4];
    int clamped[3];
    double dir[3], dlen;
} G64;

typedef struct {
    int P, M, deg, H, W, mode;
    const double *means3D, *shs, *colors_precomp, *opac, *scales, *rots, *cov3D;
    double tanfovx, tanfovy, kernel_size, scale_mod;
    const double *view, *proj, *campos, *bg;
} Scene;

static void cov3d(const double* s, double mod, const double* q, double* c6) {
    double sx = mod * s[0], sy = mod * s[1], sz = mod * s[2];
    double r = q[0], x = q[1], y = q[2], z = q[3];
    double R[3][3] = {{1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)},
                      {2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)},
                      {2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)}};
    double sc[3] = {sx, sy, sz}, L[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) L[i][j] = R[i][j] * sc[j];
    double S[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
        S[i][j] = 0;
        for (int k = 0; k < 3; ++k) S[i][j] += L[i][k] * L[j][k];
    }
    c6[0] = S[0][0]; c6[1] = S[0][1]; c6[2] = S[0][2]; c6[3] = S[1][1]; c6[4] = S[1][2]; c6[5] = S[2][2];
}

/* SH basis values b[16] for direction (x,y,z) */
static void sh_basis(int deg, double x, double y, double z, double* b) {
    b[0] = SH_C0;
    if (deg > 0) {
        b[1] = -SH_C1 * y; b[2] = SH_C1 * z; b[3] = -SH_C1 * x;
        if (deg > 1) {
            double xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            b[4] = SH_C2[0] * xy; b[5] = SH_C2[1] * yz; b[6] = SH_C2[2] * (2 * zz - xx - yy); b[7] = SH_C2[3] * xz;
            b[8] = SH_C2[4] * (xx - yy);
            if (deg > 2) {
                b[9] = SH_C3[0] * y * (3 * xx - yy); b[10] = SH_C3[1] * xy * z; b[11] = SH_C3[2] * y * (4 * zz - xx - yy);
                b[12] = SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy); b[13] = SH_C3[4] * x * (4 * zz - xx - yy);
                b[14] = SH_C3[5] * z * (xx - yy); b[15] = SH_C3[6] * x * (xx - 3 * yy);
            }
        }
    }
}
/* d basis / d(x,y,z): db[k][3] */
static void sh_basis_grad(int deg, double x, double y, double z, double db[16][3]) {
    memset(db, 0, sizeof(double) * 48);
    if (deg > 0) {
        db[1][1] = -SH_C1; db[2][2] = SH_C1; db[3][0] = -SH_C1;
        if (deg > 1) {
            db[4][0] = SH_C2[0] * y; db[4][1] = SH_C2[0] * x;
            db[5][1] = SH_C2[1] * z; db[5][2] = SH_C2[1] * y;
            db[6][0] = SH_C2[2] * -2 * x; db[6][1] = SH_C2[2] * -2 * y; db[6][2] = SH_C2[2] * 4 * z;
            db[7][0] = SH_C2[3] * z; db[7][2] = SH_C2[3] * x;
            db[8][0] = SH_C2[4] * 2 * x; db[8][1] = SH_C2[4] * -2 * y;
            if (deg > 2) {
                double xx = x * x, yy = y * y, zz = z * z;
                db[9][0] = SH_C3[0] * 6 * x * y; db[9][1] = SH_C3[0] * (3 * xx - 3 * yy);
                db[10][0] = SH_C3[1] * y * z; db[10][1] = SH_C3[1] * x * z; db[10][2] = SH_C3[1] * x * y;
                db[11][0] = SH_C3[2] * -2 * x * y; db[11][1] = SH_C3[2] * (4 * zz - xx - 3 * yy); db[11][2] = SH_C3[2] * 8 * y * z;
                db[12][0] = SH_C3[3] * -6 * x * z; db[12][1] = SH_C3[3] * -6 * y * z; db[12][2] = SH_C3[3] * (6 * zz - 3 * xx - 3 * yy);
                db[13][0] = SH_C3[4] * (4 * zz - 3 * xx - yy); db[13][1] = SH_C3[4] * -2 * x * y; db[13][2] = SH_C3[4] * 8 * x * z;
                db[14][0] = SH_C3[5] * 2 * x * z; db[14][1] = SH_C3[5] * -2 * y * z; db[14][2] = SH_C3[5] * (xx - yy);
                db[15][0] = SH_C3[6] * (3 * xx - 3 * yy); db[15][1] = SH_C3[6] * -6 * x * y;
            }
        }
    }
}

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

static void pre_one(const Scene* s, int i, G64* g) {
    memset(g, 0, sizeof(*g));
    const double* p = s->means3D + 3 * (size_t)i;
    const double* m = s->view;
    g->pv[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    g->pv[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    g->pv[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
    if (g->pv[2] <= 0.2) return;
    m = s->proj;
    for (int k = 0; k < 4; ++k) g->ph[k] = m[k] * p[0] + m[4 + k] * p[1] + m[8 + k] * p[2] + m[12 + k];
    g->pw = 1.0 / (g->ph[3] + 0.0000001);
    double projx = g->ph[0] * g->pw, projy = g->ph[1] * g->pw;
    if (s->cov3D) memcpy(g->c6, s->cov3D + 6 * (size_t)i, 6 * sizeof(double));
    else cov3d(s->scales + 3 * (size_t)i, s->scale_mod, s->rots + 4 * (size_t)i, g->c6);
    int H = s->H, W = s->W;
    double fx = (double)W / (2.0 * s->tanfovx), fy = (double)H / (2.0 * s->tanfovy);
    double limx = 1.3 * s->tanfovx, limy = 1.3 * s->tanfovy;
    double txtz = g->pv[0] / g->pv[2], tytz = g->pv[1] / g->pv[2];
    g->xmul = (txtz < -limx || txtz > limx) ? 0.0 : 1.0;
    g->ymul = (tytz < -limy || tytz > limy) ? 0.0 : 1.0;
    g->tx = fmin(limx, fmax(-limx, txtz)) * g->pv[2];
    g->ty = fmin(limy, fmax(-limy, tytz)) * g->pv[2];
    g->tz = g->pv[2];
    double J00 = fx / g->tz, J02 = -(fx * g->tx) / (g->tz * g->tz), J11 = fy / g->tz, J12 = -(fy * g->ty) / (g->tz * g->tz);
    for (int c = 0; c < 3; ++c) {
        double w0 = s->view[c * 4 + 0], w1 = s->view[c * 4 + 1], w2 = s->view[c * 4 + 2];
        g->A0[c] = J00 * w0 + J02 * w2;
        g->A1[c] = J11 * w1 + J12 * w2;
    }
    const double* c6 = g->c6;
    double S[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
    double B0[3], B1[3];
    for (int c = 0; c < 3; ++c) {
        B0[c] = g->A0[0] * S[0][c] + g->A0[1] * S[1][c] + g->A0[2] * S[2][c];
        B1[c] = g->A1[0] * S[0][c] + g->A1[1] * S[1][c] + g->A1[2] * S[2][c];
    }
    g->cxx = B0[0] * g->A0[0] + B0[1] * g->A0[1] + B0[2] * g->A0[2];
    g->cxy = B0[0] * g->A1[0] + B0[1] * g->A1[1] + B0[2] * g->A1[2];
    g->cyy = B1[0] * g->A1[0] + B1[1] * g->A1[1] + B1[2] * g->A1[2];
    double cxx = g->cxx, cxy = g->cxy, cyy = g->cyy, k = s->mode == MODE_MIP ? s->kernel_size : 0.3;
    g->coef = 1.0;
    if (s->mode == MODE_MIP) {
        double det0 = fmax(1e-6, cxx * cyy - cxy * cxy);
        double det1 = fmax(1e-6, (cxx + k) * (cyy + k) - cxy * cxy);
        g->coef = sqrt(det0 / (det1 + 1e-6) + 1e-6);
        if (det0 <= 1e-6 || det1 <= 1e-6) g->coef = 0.0;
    }
    cxx += k; cyy += k;
    double det = cxx * cyy - cxy * cxy;
    if (det == 0.0) return;
    g->ca = cyy / det; g->cb = -cxy / det; g->cc = cxx / det;
    double mid = 0.5 * (cxx + cyy);
    double lam1 = mid + sqrt(fmax(0.1, mid * mid - det)), lam2 = mid - sqrt(fmax(0.1, mid * mid - det));
    double rad = ceil(3.0 * sqrt(fmax(lam1, lam2)));
    g->x = ((projx + 1.0) * (double)W - 1.0) * 0.5;
    g->y = ((projy + 1.0) * (double)H - 1.0) * 0.5;
    int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    g->x0 = imin(gx, imax(0, (int)((g->x - rad) / (double)TILE)));
    g->y0 = imin(gy, imax(0, (int)((g->y - rad) / (double)TILE)));
    g->x1 = imin(gx, imax(0, (int)((g->x + rad + (double)(TILE - 1)) / (double)TILE)));
    g->y1 = imin(gy, imax(0, (int)((g->y + rad + (double)(TILE - 1)) / (double)TILE)));
    if ((g->x1 - g->x0) * (g->y1 - g->y0) == 0) return;
    if (s->colors_precomp) {
        for (int c = 0; c < 3; ++c) g->rgb[c] = s->colors_precomp[3 * (size_t)i + c];
    } else {
        double d[3] = {p[0] - s->campos[0], p[1] - s->campos[1], p[2] - s->campos[2]};
        g->dlen = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        for (int c = 0; c < 3; ++c) g->dir[c] = d[c] / g->dlen;
        double b[16];
        sh_basis(s->deg, g->dir[0], g->dir[1], g->dir[2], b);
        int nb = (s->deg + 1) * (s->deg + 1);
        const double* sh = s->shs + (size_t)i * s->M * 3;
        for (int c = 0; c < 3; ++c) {
            double r = 0;
            for (int kk = 0; kk < nb; ++kk) r += b[kk] * sh[kk * 3 + c];
            r += 0.5;
            g->clamped[c] = r < 0.0;
            g->rgb[c] = r < 0.0 ? 0.0 : r;
        }
    }
    g->depth = g->pv[2];
    g->op = s->opac[i] * g->coef;
    g->visible = 1;
}

typedef struct { double depth; int id; } DI;
static int di_cmp(const void* a, const void* b) {
    const DI* x = (const DI*)a; const DI* y = (const DI*)b;
    /* the device orders by the float32 depth bits, then by id */
    float fx = (float)x->depth, fy = (float)y->depth;
    if (fx != fy) return fx < fy ? -1 : 1;
    return x->id < y->id ? -1 : (x->id > y->id ? 1 : 0);
}

/* per-pixel ordered candidate list: every visible Gaussian whose 3-sigma tile rect covers the pixel's tile */
static int pixel_list(const G64* g, int P, int px, int py, DI* list) {
    int tx = px / TILE, ty = py / TILE, n = 0;
    for (int i = 0; i < P; ++i)
        if (g[i].visible && tx >= g[i].x0 && tx < g[i].x1 && ty >= g[i].y0 && ty < g[i].y1) { list[n].depth = g[i].depth; list[n].id = i; ++n; }
    qsort(list, (size_t)n, sizeof(DI), di_cmp);
    return n;
}

static void make_scene(Scene* s, int P, int M, int deg, const double* means3D, const double* shs, const double* colors_precomp,
                       const double* opac, const double* scales, const double* rots, const double* cov3D, int H, int W,
                       double tanfovx, double tanfovy, double kernel_size, double scale_mod, int mode, const double* view,
                       const double* proj, const double* campos, const double* bg) {
    s->P = P; s->M = M; s->deg = deg; s->H = H; s->W = W; s->mode = mode;
    s->means3D = means3D; s->shs = shs; s->colors_precomp = colors_precomp; s->opac = opac; s->scales = scales; s->rots = rots;
    s->cov3D = cov3D; s->tanfovx = tanfovx; s->tanfovy = tanfovy; s->kernel_size = kernel_size; s->scale_mod = scale_mod;
    s->view = view; s->proj = proj; s->campos = campos; s->bg = bg;
}

int gvfo64_forward(int P, int M, int deg, const double* means3D, const double* shs, const double* colors_precomp,
                   const double* opac, const double* scales, const double* rots, const double* cov3D, int H, int W,
                   double tanfovx, double tanfovy, double kernel_size, double scale_mod, int mode, const double* view,
                   const double* proj, const double* campos, const double* bg, double* out_color, double* out_alpha,
                   double* out_depth) {
    Scene s;
    make_scene(&s, P, M, deg, means3D, shs, colors_precomp, opac, scales, rots, cov3D, H, W, tanfovx, tanfovy, kernel_size,
               scale_mod, mode, view, proj, campos, bg);
    G64* g = (G64*)malloc(sizeof(G64) * (size_t)(P > 0 ? P : 1));
    DI* list = (DI*)malloc(sizeof(DI) * (size_t)(P > 0 ? P : 1));
    for (int i = 0; i < P; ++i) pre_one(&s, i, &g[i]);
    for (int py = 0; py < H; ++py)
        for (int px = 0; px < W; ++px) {
            int n = pixel_list(g, P, px, py, list);
            double T = 1.0, C[3] = {0, 0, 0}, D = 0;
            for (int k = 0; k < n; ++k) {
                const G64* q = &g[list[k].id];
                double dx = q->x - (double)px, dy = q->y - (double)py;
                double power = -0.5 * (q->ca * dx * dx + q->cc * dy * dy) - q->cb * dx * dy;
                if (power > 0.0) continue;
                double alpha = fmin(0.99, q->op * exp(power));
                if (alpha < 1.0 / 255.0) continue;
                double test_T = T * (1.0 - alpha);
                if (test_T < 0.0001) break;
                for (int c = 0; c < 3; ++c) C[c] += q->rgb[c] * alpha * T;
                D += q->depth * alpha * T;
                T = test_T;
            }
            size_t pid = (size_t)py * W + px;
            for (int c = 0; c < 3; ++c) out_color[(size_t)c * H * W + pid] = C[c] + T * bg[c];
            if (out_alpha) out_alpha[pid] = 1.0 - T;
            if (out_depth) out_depth[pid] = D;
        }
    free(list); free(g);
    return 0;
}

/* Gradients of  L = sum(dL_dcolor * color) + sum(dL_dalpha * alpha) + sum(dL_ddepth * depth)  w.r.t. every input.
 * Outputs may be NULL.  g_means2D[P][2]: NDC convention (see header). */
int gvfo64_backward(int P, int M, int deg, const double* means3D, const double* shs, const double* colors_precomp,
                    const double* opac, const double* scales, const double* rots, const double* cov3D, int H, int W,
                    double tanfovx, double tanfovy, double kernel_size, double scale_mod, int mode, const double* view,
                    const double* proj, const double* campos, const double* bg, const double* dL_dcolor,
                    const double* dL_dalpha, const double* dL_ddepth, double* g_means3D, double* g_means2D, double* g_shs,
                    double* g_colors, double* g_opac, double* g_scales, double* g_rots, double* g_cov3D) {
    Scene s;
    make_scene(&s, P, M, deg, means3D, shs, colors_precomp, opac, scales, rots, cov3D, H, W, tanfovx, tanfovy, kernel_size,
               scale_mod, mode, view, proj, campos, bg);
    size_t Pn = (size_t)(P > 0 ? P : 1);
    G64* g = (G64*)malloc(sizeof(G64) * Pn);
    DI* list = (DI*)malloc(sizeof(DI) * Pn);
    /* accumulators of the blend's backward: d/d(x,y) [pixels], d/d(conic a,b,c) [true derivatives], d/d(op), d/d(rgb), d/d(depth) */
    double* acc = (double*)calloc(Pn * 10, sizeof(double));
    double* alphas = (double*)malloc(sizeof(double) * Pn);
    double* Gs = (double*)malloc(sizeof(double) * Pn);
    int* used = (int*)malloc(sizeof(int) * Pn);
    for (int i = 0; i < P; ++i) pre_one(&s, i, &g[i]);
    for (int py = 0; py < H; ++py)
        for (int px = 0; px < W; ++px) {
            int n = pixel_list(g, P, px, py, list);
            size_t pid = (size_t)py * W + px;
            /* channels: r, g, b, depth, one (alpha_out = sum alpha_k T_k) with backgrounds bg, 0, 0 */
            double dch[5] = {dL_dcolor[pid], dL_dcolor[(size_t)H * W + pid], dL_dcolor[(size_t)2 * H * W + pid],
                             dL_ddepth ? dL_ddepth[pid] : 0.0, dL_dalpha ? dL_dalpha[pid] : 0.0};
            double bgc[5] = {bg[0], bg[1], bg[2], 0.0, 0.0};
            /* forward replay */
            double T = 1.0;
            int m = 0;
            for (int k = 0; k < n; ++k) {
                const G64* q = &g[list[k].id];
                double dx = q->x - (double)px, dy = q->y - (double)py;
                double power = -0.5 * (q->ca * dx * dx + q->cc * dy * dy) - q->cb * dx * dy;
                if (power > 0.0) continue;
                double Gv = exp(power);
                double alpha = fmin(0.99, q->op * Gv);
                if (alpha < 1.0 / 255.0) continue;
                double test_T = T * (1.0 - alpha);
                if (test_T < 0.0001) break;
                used[m] = list[k].id; alphas[m] = alpha; Gs[m] = Gv; ++m;
                T = test_T;
            }
            const double T_final = T;
            /* back to front.  suffix[ch] = sum_{j>k} c_j alpha_j T_j + T_final bg  (what lies behind splat k) */
            double suffix[5];
            for (int ch = 0; ch < 5; ++ch) suffix[ch] = T_final * bgc[ch];
            for (int k = m - 1; k >= 0; --k) {
                const int id = used[k];
                const G64* q = &g[id];
                const double alpha = alphas[k];
                T = T / (1.0 - alpha);                 /* transmittance in front of splat k */
                double cch[5] = {q->rgb[0], q->rgb[1], q->rgb[2], q->depth, 1.0};
                double dL_dalpha_k = 0.0;
                for (int ch = 0; ch < 5; ++ch) {
                    /* out = ... + c_k alpha T + (1 - alpha) * [behind / (1 - alpha)] : d out / d alpha = c_k T - suffix / (1 - alpha) */
                    dL_dalpha_k += (cch[ch] * T - suffix[ch] / (1.0 - alpha)) * dch[ch];
                    suffix[ch] += cch[ch] * alpha * T;
                }
                double* a = acc + 10 * (size_t)id;
                for (int c = 0; c < 3; ++c) a[6 + c] += alpha * T * dch[c];
                a[9] += alpha * T * dch[3];
                /* alpha = min(0.99, op G): gradient passes through the clamp (upstream) */
                const double Gv = Gs[k];
                const double dL_dG = q->op * dL_dalpha_k;
                a[5] += Gv * dL_dalpha_k;
                const double dx = q->x - (double)px, dy = q->y - (double)py;
                /* power = -0.5 (A dx^2 + C dy^2) - B dx dy */
                a[0] += dL_dG * Gv * (-q->ca * dx - q->cb * dy);
                a[1] += dL_dG * Gv * (-q->cc * dy - q->cb * dx);
                a[2] += dL_dG * Gv * (-0.5 * dx * dx);
                a[3] += dL_dG * Gv * (-dx * dy);
                a[4] += dL_dG * Gv * (-0.5 * dy * dy);
            }
        }
    /* per-Gaussian chain rule */
    for (int i = 0; i < P; ++i) {
        const G64* q = &g[i];
        const double* a = acc + 10 * (size_t)i;
        double gm[3] = {0, 0, 0}, gsc[3] = {0, 0, 0}, gq[4] = {0, 0, 0, 0}, gc6[6] = {0, 0, 0, 0, 0, 0}, gop = 0, gcol[3] = {0, 0, 0};
        double gm2[2] = {0, 0};
        if (q->visible) {
            const double* p = means3D + 3 * (size_t)i;
            /* screen-space mean: px = ((ndc + 1) W - 1) / 2 */
            gm2[0] = a[0] * 0.5 * (double)W; gm2[1] = a[1] * 0.5 * (double)H;
            {
                const double* m = proj;
                double mw = q->pw;
                double mul1 = q->ph[0] * mw * mw, mul2 = q->ph[1] * mw * mw;
                gm[0] += (m[0] * mw - m[3] * mul1) * gm2[0] + (m[1] * mw - m[3] * mul2) * gm2[1];
                gm[1] += (m[4] * mw - m[7] * mul1) * gm2[0] + (m[5] * mw - m[7] * mul2) * gm2[1];
                gm[2] += (m[8] * mw - m[11] * mul1) * gm2[0] + (m[9] * mw - m[11] * mul2) * gm2[1];
            }
            /* depth output: depth = view row 2 . p */
            gm[0] += view[2] * a[9]; gm[1] += view[6] * a[9]; gm[2] += view[10] * a[9];
            /* colour */
            if (colors_precomp) {
                for (int c = 0; c < 3; ++c) gcol[c] = a[6 + c];
            } else {
                double b[16], db[16][3];
                sh_basis(deg, q->dir[0], q->dir[1], q->dir[2], b);
                sh_basis_grad(deg, q->dir[0], q->dir[1], q->dir[2], db);
                int nb = (deg + 1) * (deg + 1);
                const double* sh = shs + (size_t)i * M * 3;
                double ddir[3] = {0, 0, 0};
                for (int c = 0; c < 3; ++c) {
                    double gr = q->clamped[c] ? 0.0 : a[6 + c];
                    for (int kk = 0; kk < nb; ++kk) {
                        if (g_shs) g_shs[((size_t)i * M + kk) * 3 + c] = b[kk] * gr;
                        for (int e = 0; e < 3; ++e) ddir[e] += db[kk][e] * sh[kk * 3 + c] * gr;
                    }
                }
                /* dir = d / |d| */
                double dot = ddir[0] * q->dir[0] + ddir[1] * q->dir[1] + ddir[2] * q->dir[2];
                for (int e = 0; e < 3; ++e) gm[e] += (ddir[e] - q->dir[e] * dot) / q->dlen;
            }
            /* opacity and the mip coefficient */
            gop = a[5] * q->coef;
            double gcxx = 0, gcxy = 0, gcyy = 0;   /* d/d(cov2D before the filter) */
            const double k = mode == MODE_MIP ? kernel_size : 0.3;
            if (mode == MODE_MIP && q->coef > 0.0) {
                double dcoef = a[5] * opac[i];
                double det0 = q->cxx * q->cyy - q->cxy * q->cxy, det1 = (q->cxx + k) * (q->cyy + k) - q->cxy * q->cxy;
                /* coef = sqrt(r + 1e-6), r = det0 / (det1 + 1e-6); neither det is clamped here (coef > 0) */
                double dr = dcoef * 0.5 / q->coef;
                double dd0 = dr / (det1 + 1e-6), dd1 = -dr * det0 / ((det1 + 1e-6) * (det1 + 1e-6));
                gcxx += dd0 * q->cyy + dd1 * (q->cyy + k);
                gcyy += dd0 * q->cxx + dd1 * (q->cxx + k);
                gcxy += -2.0 * q->cxy * (dd0 + dd1);
            }
            /* conic = inverse of the filtered cov2D (a', b', c') */
            {
                double ap = q->cxx + k, bp = q->cxy, cp = q->cyy + k;
                double det = ap * cp - bp * bp, d2 = 1.0 / (det * det);
                double gA = a[2], gB = a[3], gC = a[4];
                gcxx += d2 * (-cp * cp * gA + bp * cp * gB - bp * bp * gC);
                gcxy += d2 * (2 * bp * cp * gA - (det + 2 * bp * bp) * gB + 2 * ap * bp * gC);
                gcyy += d2 * (-bp * bp * gA + ap * bp * gB - ap * ap * gC);
            }
            /* cov2D = [A0; A1] Sigma [A0; A1]^T */
            const double* c6 = q->c6;
            double S[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
            double Gm[3][3];
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c)
                Gm[r][c] = gcxx * q->A0[r] * q->A0[c] + gcxy * q->A0[r] * q->A1[c] + gcyy * q->A1[r] * q->A1[c];
            gc6[0] = Gm[0][0]; gc6[3] = Gm[1][1]; gc6[5] = Gm[2][2];
            gc6[1] = Gm[0][1] + Gm[1][0]; gc6[2] = Gm[0][2] + Gm[2][0]; gc6[4] = Gm[1][2] + Gm[2][1];
            double dA0[3], dA1[3];
            for (int r = 0; r < 3; ++r) {
                double SA0 = S[r][0] * q->A0[0] + S[r][1] * q->A0[1] + S[r][2] * q->A0[2];
                double SA1 = S[r][0] * q->A1[0] + S[r][1] * q->A1[1] + S[r][2] * q->A1[2];
                dA0[r] = 2 * gcxx * SA0 + gcxy * SA1;
                dA1[r] = 2 * gcyy * SA1 + gcxy * SA0;
            }
            double dJ00 = 0, dJ02 = 0, dJ11 = 0, dJ12 = 0;
            for (int c = 0; c < 3; ++c) {
                double w0 = view[c * 4 + 0], w1 = view[c * 4 + 1], w2 = view[c * 4 + 2];
                dJ00 += dA0[c] * w0; dJ02 += dA0[c] * w2; dJ11 += dA1[c] * w1; dJ12 += dA1[c] * w2;
            }
            double fx = (double)W / (2.0 * tanfovx), fy = (double)H / (2.0 * tanfovy);
            double tz = q->tz, tz2 = 1.0 / (tz * tz), tz3 = tz2 / tz;
            double dtx = q->xmul * (-fx * tz2 * dJ02), dty = q->ymul * (-fy * tz2 * dJ12);
            double dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + 2 * fx * q->tx * tz3 * dJ02 + 2 * fy * q->ty * tz3 * dJ12;
            /* t = W p + ... */
            gm[0] += view[0] * dtx + view[1] * dty + view[2] * dtz;
            gm[1] += view[4] * dtx + view[5] * dty + view[6] * dtz;
            gm[2] += view[8] * dtx + view[9] * dty + view[10] * dtz;
            /* Sigma = L L^T, L = R diag(mod s) */
            if (!cov3D) {
                const double* sv = scales + 3 * (size_t)i;
                const double* qq = rots + 4 * (size_t)i;
                double r = qq[0], x = qq[1], y = qq[2], z = qq[3];
                double R[3][3] = {{1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)},
                                  {2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)},
                                  {2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)}};
                double sc[3] = {scale_mod * sv[0], scale_mod * sv[1], scale_mod * sv[2]};
                double Gs2[3][3] = {{gc6[0], 0.5 * gc6[1], 0.5 * gc6[2]}, {0.5 * gc6[1], gc6[3], 0.5 * gc6[4]}, {0.5 * gc6[2], 0.5 * gc6[4], gc6[5]}};
                double dLm[3][3], dR[3][3];
                for (int r2 = 0; r2 < 3; ++r2) for (int c = 0; c < 3; ++c) {
                    double v = 0;
                    for (int kk = 0; kk < 3; ++kk) v += 2 * Gs2[r2][kk] * R[kk][c] * sc[c];
                    dLm[r2][c] = v;
                }
                for (int c = 0; c < 3; ++c) {
                    double v = 0;
                    for (int r2 = 0; r2 < 3; ++r2) { v += dLm[r2][c] * R[r2][c]; dR[r2][c] = dLm[r2][c] * sc[c]; }
                    gsc[c] = scale_mod * v;
                }
                gq[0] = 2 * (-z * dR[0][1] + y * dR[0][2] + z * dR[1][0] - x * dR[1][2] - y * dR[2][0] + x * dR[2][1]);
                gq[1] = 2 * (y * dR[0][1] + z * dR[0][2] + y * dR[1][0] - 2 * x * dR[1][1] - r * dR[1][2] + z * dR[2][0] + r * dR[2][1] - 2 * x * dR[2][2]);
                gq[2] = 2 * (-2 * y * dR[0][0] + x * dR[0][1] + r * dR[0][2] + x * dR[1][0] + z * dR[1][2] - r * dR[2][0] + z * dR[2][1] - 2 * y * dR[2][2]);
                gq[3] = 2 * (-2 * z * dR[0][0] - r * dR[0][1] + x * dR[0][2] + r * dR[1][0] - 2 * z * dR[1][1] + y * dR[1][2] + x * dR[2][0] + y * dR[2][1]);
            }
            (void)p;
        }
        if (g_means3D) for (int e = 0; e < 3; ++e) g_means3D[3 * (size_t)i + e] = gm[e];
        if (g_means2D) { g_means2D[2 * (size_t)i] = gm2[0]; g_means2D[2 * (size_t)i + 1] = gm2[1]; }
        if (g_colors) for (int e = 0; e < 3; ++e) g_colors[3 * (size_t)i + e] = gcol[e];
        if (g_opac) g_opac[i] = gop;
        if (g_scales) for (int e = 0; e < 3; ++e) g_scales[3 * (size_t)i + e] = gsc[e];
        if (g_rots) for (int e = 0; e < 4; ++e) g_rots[4 * (size_t)i + e] = gq[e];
        if (g_cov3D) for (int e = 0; e < 6; ++e) g_cov3D[6 * (size_t)i + e] = gc6[e];
        if (g_shs && (!q->visible || colors_precomp)) for (int kk = 0; kk < M * 3; ++kk) g_shs[(size_t)i * M * 3 + kk] = 0.0;
        if (g_shs && q->visible && !colors_precomp) {
            int nb = (deg + 1) * (deg + 1);
            for (int kk = nb; kk < M; ++kk) for (int c = 0; c < 3; ++c) g_shs[((size_t)i * M + kk) * 3 + c] = 0.0;
        }
    }
    free(used); free(Gs); free(alphas); free(acc); free(list); free(g);
    return 0;
}
