/*
 * vox2seq_oracle.c -- CPU restatement of the reference's Z-order / Hilbert voxel serialisation.
 * TEST INFRASTRUCTURE ONLY (see rast_oracle.c header).
 *
 * Follows model/sparse_voxel_diffusion/vox2seq/src/z_order.cu:35-66 (10-bit Morton interleave,
 * code = x bits at 3k+2, y at 3k+1, z at 3k) and src/hilbert.cu:35-133 (Skilling's transpose
 * <-> axes algorithm on 3 x 10 bits, then the same interleave).  Pinned by the known answers the
 * reference's own fallback (the vox2seq/pytorch sources, which vox2seq/test.py:5-24 asserts the CUDA
 * extension is bit-equal to) produces: tests/golden/vox2seq_golden.npz.
 */
#include <stdint.h>

#define NBITS 10

static uint32_t spread3(uint32_t v) { /* put bit k of v at bit 3k */
    uint32_t r = 0;
    for (int k = 0; k < NBITS; ++k) r |= ((v >> k) & 1u) << (3 * k);
    return r;
}
static uint32_t compact3(uint32_t v) {
    uint32_t r = 0;
    for (int k = 0; k < NBITS; ++k) r |= ((v >> (3 * k)) & 1u) << k;
    return r;
}

void gvfo_z_order_encode(int64_t n, const int32_t* x, const int32_t* y, const int32_t* z, int32_t* code) {
    for (int64_t i = 0; i < n; ++i)
        code[i] = (int32_t)((spread3((uint32_t)x[i]) << 2) | (spread3((uint32_t)y[i]) << 1) | spread3((uint32_t)z[i]));
}
void gvfo_z_order_decode(int64_t n, const int32_t* code, int32_t* x, int32_t* y, int32_t* z) {
    for (int64_t i = 0; i < n; ++i) {
        uint32_t c = (uint32_t)code[i];
        x[i] = (int32_t)compact3(c >> 2); y[i] = (int32_t)compact3(c >> 1); z[i] = (int32_t)compact3(c);
    }
}

/* Skilling, "Programming the Hilbert curve" (2004): AxestoTranspose / TransposetoAxes, n = 3 */
void gvfo_hilbert_encode(int64_t n, const int32_t* x, const int32_t* y, const int32_t* z, int32_t* code) {
    for (int64_t i = 0; i < n; ++i) {
        uint32_t X[3] = {(uint32_t)x[i], (uint32_t)y[i], (uint32_t)z[i]};
        uint32_t M = 1u << (NBITS - 1), P, Q, t;
        for (Q = M; Q > 1; Q >>= 1) {       /* inverse undo */
            P = Q - 1;
            for (int k = 0; k < 3; ++k) {
                if (X[k] & Q) X[0] ^= P;
                else { t = (X[0] ^ X[k]) & P; X[0] ^= t; X[k] ^= t; }
            }
        }
        for (int k = 1; k < 3; ++k) X[k] ^= X[k - 1];   /* Gray encode */
        t = 0;
        for (Q = M; Q > 1; Q >>= 1) if (X[2] & Q) t ^= Q - 1;
        for (int k = 0; k < 3; ++k) X[k] ^= t;
        code[i] = (int32_t)((spread3(X[0]) << 2) | (spread3(X[1]) << 1) | spread3(X[2]));
    }
}
void gvfo_hilbert_decode(int64_t n, const int32_t* code, int32_t* x, int32_t* y, int32_t* z) {
    for (int64_t i = 0; i < n; ++i) {
        uint32_t c = (uint32_t)code[i];
        uint32_t X[3] = {compact3(c >> 2), compact3(c >> 1), compact3(c)};
        uint32_t N = 2u << (NBITS - 1), P, Q, t;
        t = X[2] >> 1;                                   /* Gray decode */
        for (int k = 2; k > 0; --k) X[k] ^= X[k - 1];
        X[0] ^= t;
        for (Q = 2; Q != N; Q <<= 1) {                   /* undo excess work */
            P = Q - 1;
            for (int k = 2; k >= 0; --k) {
                if (X[k] & Q) X[0] ^= P;
                else { t = (X[0] ^ X[k]) & P; X[0] ^= t; X[k] ^= t; }
            }
        }
        x[i] = (int32_t)X[0]; y[i] = (int32_t)X[1]; z[i] = (int32_t)X[2];
    }
}
