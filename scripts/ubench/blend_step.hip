// blend_step.hip -- VERDICT r5 item 9: a LOOP-ONLY benchmark of blend_kernel's compositing step (csrc/rast.hip GVF_BLEND_STEP), 4 waves per
// workgroup = four 8 x 8 quadrants, 8 workgroups per CU (the launch's residency), over a synthetic per-wave list of 256 entries in LDS:
//   MODE 0  today's step: 3 ds_read_b128 + 16 vector instructions per list entry (exponent from the Cholesky factor, v_exp_f32, two predicates,
//           3 colour fmas, T update), 4 entries per trip
//   MODE 1  the one shape not yet priced: the exponents of 4 splats x 64 pixels from six v_mfma_f32_4x4x1_16b_f32 on the monomials
//           (x^2, xy, y^2, x, y, 1) -- lane = pixel, the accumulator's four registers = the pixel's four splats --, computed one group AHEAD of
//           the compositing, which is then 11 vector instructions per entry
//   MODE 2  today's arithmetic with the step's predicates kept as explicit 64-bit lane masks (ballot / inverse ballot): 6 scalar instructions fewer per
//           trip of four entries -- what csrc/rast.hip runs since the end of round 6 (GVF_BLEND_LANE_MASKS)
//   MODE 3  MODE 2 + the 0.99 clamp on v_exp_f32's output modifier (15 vector instructions; not bit-identical): no further gain
//   MODE 4/5  MODE 2 with the trip's LDS reads first and ONE s_waitcnt (/ the next trip's list words ahead): slower -- a s_waitcnt is not an issue slot
//   MODE 6  MODE 2 with the trip's two exits kept apart by a non-speculatable asm statement: the structuriser rebuilds the same 9-10 scalar instructions
//   MODE 7  no done mask: a stopped pixel keeps its transmittance with the sign flipped
//   MODE 9  MODE 2 with eight entries per trip (half the loop control per entry)
//   MODE 8  MODE 2 with upstream's test_T = T * (1 - alpha): one vector instruction off the pixel's serial chain
// Prints wave-cycles per list entry per wave and per SIMD (8 waves per SIMD resident).  Bar to build MODE 1 into the kernel: <= 26 per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form blend_step.hip -o blend_step.bin
// (inner loops as compiled, per trip of four entries: MODE 0 65 vector + 31 scalar + 13 LDS instructions, MODE 2 65 + 25 + 13, MODE 3 61 + 25 + 13;
//  MODE 1 12.0 vector + 1.5 MFMA per entry)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int N = 256;
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float neg_exponent(float l11, float l12, float l22, float c1, float c2, float px, float py, float lo) {
    const float s1 = __builtin_fmaf(-l11, px, __builtin_fmaf(-l12, py, c1));
    const float s2 = __builtin_fmaf(-l22, py, c2);
    return __builtin_fmaf(s2, s2, __builtin_fmaf(s1, s1, -lo));
}

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int reps) {
    __shared__ float4 sA[N];            // MODE 0: {l11, l12, l22, c1}          MODE 1: {K_xx, K_xy, K_yy, K_x}
    __shared__ float4 sB[N];            // MODE 0: {c2, log2 op, r, g}          MODE 1: {K_y, K_1, -, -}
    __shared__ float4 sC[N];            // MODE 0: {b, depth, -, -}             MODE 1: {r, g, b, -}
    __shared__ unsigned sList[4][N];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    {   // a splat somewhere around the tile, sigma 3-12 px (the step is branch-free: its cost does not depend on how many pairs contribute)
        const float cx = -6.f + 28.f * ((t * 37) & 255) / 255.f, cy = -6.f + 28.f * ((t * 101) & 255) / 255.f, sg = 3.f + 9.f * ((t * 13) & 255) / 255.f;
        const float l11 = 0.85f / sg, l12 = 0.1f / sg, l22 = 0.8f / sg, c1 = l11 * cx + l12 * cy, c2 = l22 * cy, lo = -8.6f + 2.4f * ((t * 7) & 255) / 255.f;       // opacity 2^-8.6 .. 2^-6.2: some pairs pass 1/255, no pixel ever saturates (the loop must not exit early)
        if (MODE != 1) {
            sA[t] = make_float4(l11, l12, l22, c1); sB[t] = make_float4(c2, lo, 0.3f, 0.5f); sC[t] = make_float4(0.7f, 1.0f, 0.f, 0.f);
        } else {
            sA[t] = make_float4(l11 * l11, 2.f * l11 * l12, l12 * l12 + l22 * l22, -2.f * c1 * l11);
            sB[t] = make_float4(-2.f * c1 * l12 - 2.f * c2 * l22, c1 * c1 + c2 * c2 - lo, 0.f, 0.f); sC[t] = make_float4(0.3f, 0.5f, 0.7f, 0.f);
        }
        for (int w = 0; w < 4; ++w) sList[w][t] = (unsigned)((t * 5 + w * 64) & 255) * 16u;
    }
    __syncthreads();
    const float pxr = (float)((wave & 1) * 8 + (lane & 7)), pyr = (float)((wave >> 1) * 8 + (lane >> 3));
    bool done = false;
    float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f;
    const long long t0 = __builtin_readcyclecounter();
    if (MODE == 0) {
#define STEP(J) {                                                                                                              \
            const float4 a = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(sA) + (J));               \
            const float4 b = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(sB) + (J));               \
            const float4 c = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(sC) + (J));               \
            const float nlog = neg_exponent(a.x, a.y, a.z, a.w, b.x, pxr, pyr, b.y);                                           \
            const float alpha = fminf(0.99f, __builtin_amdgcn_exp2f(-nlog));                                                   \
            const bool ok = !done && !(alpha < 1.0f / 255.0f);                                                                 \
            const float w_raw = alpha * T;                                                                                     \
            const float test_T = T - w_raw;                                                                                    \
            const bool stop = ok && test_T < 0.0001f;                                                                          \
            done = done || stop;                                                                                               \
            const bool acc = ok != stop;                                                                                       \
            const float wgt = acc ? w_raw : 0.0f;                                                                              \
            C0 = __builtin_fmaf(b.z, wgt, C0); C1 = __builtin_fmaf(b.w, wgt, C1); C2 = __builtin_fmaf(c.x, wgt, C2);           \
            T = acc ? test_T : T; }
        for (int r = 0; r < reps; ++r) {
            for (int jj = 0; jj < N; jj += 4) {
                if (__all(done)) break;
                const unsigned j0 = sList[wave][jj], j1 = sList[wave][jj + 1], j2 = sList[wave][jj + 2], j3 = sList[wave][jj + 3];
                STEP(j0) STEP(j1) STEP(j2) STEP(j3)
            }
            T = T * 0.5f + 0.5f;        // keep the pixel unsaturated across repetitions (one instruction per 256 entries)
        }
#undef STEP
    } else if (MODE == 2 || MODE == 3) {
        // MODE 2: today's arithmetic with the step's predicates kept as explicit 64-bit lane masks (ballot / inverse ballot): the
        //         saturation test at the head of a trip is a scalar compare (today: v_cndmask + v_cmp per trip), one scalar operation less per entry
        // MODE 3: MODE 2 + the 0.99 clamp on v_exp_f32's output modifier: alpha / 0.99 = clamp(exp2(-(nlog - log2(1 / 0.99)))), state T99 = 0.99 T,
        //         w = (alpha / 0.99) T99, T99 -= 0.99 w: 15 vector instructions per entry (NOT bit-identical to today's min(0.99, .): two more roundings)
        unsigned long long dm = 0ull;
        const unsigned long long all = __builtin_amdgcn_ballot_w64(true);
        const float thr_a = MODE == 3 ? (1.0f / 255.0f) / 0.99f : 1.0f / 255.0f, thr_t = MODE == 3 ? 0.0001f * 0.99f : 0.0001f;
#define STEP(J) {                                                                                                              \
            const float4 a = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(sA) + (J));               \
            const float4 b = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(sB) + (J));               \
            const float4 c = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(sC) + (J));               \
            const float nlog = neg_exponent(a.x, a.y, a.z, a.w, b.x, pxr, pyr, b.y);                                           \
            const float e_ = __builtin_amdgcn_exp2f(-nlog);                                                                    \
            const float alpha = MODE == 3 ? __builtin_amdgcn_fmed3f(e_, 0.0f, 1.0f) : fminf(0.99f, e_);                        \
            const unsigned long long okm = __builtin_amdgcn_ballot_w64(!(alpha < thr_a)) & ~dm;                                \
            const float w_raw = alpha * T;                                                                                     \
            const float test_T = MODE == 3 ? __builtin_fmaf(-0.99f, w_raw, T) : T - w_raw;                                     \
            const unsigned long long stopm = okm & __builtin_amdgcn_ballot_w64(test_T < thr_t);                                \
            dm |= stopm;                                                                                                       \
            const bool acc = __builtin_amdgcn_inverse_ballot_w64(okm ^ stopm);                                                 \
            const float wgt = acc ? w_raw : 0.0f;                                                                              \
            C0 = __builtin_fmaf(b.z, wgt, C0); C1 = __builtin_fmaf(b.w, wgt, C1); C2 = __builtin_fmaf(c.x, wgt, C2);           \
            T = acc ? test_T : T; }
        if (MODE == 3) T = 0.99f;
        for (int r = 0; r < reps; ++r) {
            for (int jj = 0; jj < N; jj += 4) {
                if (dm == all) break;
                const unsigned j0 = sList[wave][jj], j1 = sList[wave][jj + 1], j2 = sList[wave][jj + 2], j3 = sList[wave][jj + 3];
                STEP(j0) STEP(j1) STEP(j2) STEP(j3)
            }
            T = T * 0.5f + (MODE == 3 ? 0.495f : 0.5f);
        }
        if (MODE == 3) T = T / 0.99f;
        done = __builtin_amdgcn_inverse_ballot_w64(dm);
        if (done) C0 += 1.0f;
#undef STEP
    } else if (MODE == 4 || MODE == 5) {
        // MODE 4: MODE 2 with the trip's 13 LDS reads issued first and ONE s_waitcnt lgkmcnt(0) (the compiler's ~10 graded waits per trip are
        //         instructions of the wave's stream too; the other 7 waves of the SIMD cover the exposed latency)
        // MODE 5: MODE 4 + the list words of the NEXT trip requested before this trip's arithmetic
        unsigned long long dm = 0ull;
        const unsigned long long all = __builtin_amdgcn_ballot_w64(true);
#define LOADS(J, A, B, C) const float4 A = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(sA) + (J));  \
                          const float4 B = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(sB) + (J));  \
                          const float C = reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(sC) + (J))->x;
#define STEP(a, b, cx) {                                                                                                       \
            const float nlog = neg_exponent(a.x, a.y, a.z, a.w, b.x, pxr, pyr, b.y);                                           \
            const float alpha = fminf(0.99f, __builtin_amdgcn_exp2f(-nlog));                                                   \
            const unsigned long long okm = __builtin_amdgcn_ballot_w64(!(alpha < 1.0f / 255.0f)) & ~dm;                        \
            const float w_raw = alpha * T;                                                                                     \
            const float test_T = T - w_raw;                                                                                    \
            const unsigned long long stopm = okm & __builtin_amdgcn_ballot_w64(test_T < 0.0001f);                              \
            dm |= stopm;                                                                                                       \
            const bool acc = __builtin_amdgcn_inverse_ballot_w64(okm ^ stopm);                                                 \
            const float wgt = acc ? w_raw : 0.0f;                                                                              \
            C0 = __builtin_fmaf(b.z, wgt, C0); C1 = __builtin_fmaf(b.w, wgt, C1); C2 = __builtin_fmaf(cx, wgt, C2);            \
            T = acc ? test_T : T; }
        for (int r = 0; r < reps; ++r) {
            uint4 jn = *reinterpret_cast<const uint4*>(&sList[wave][0]);
            for (int jj = 0; jj < N; jj += 4) {
                if (dm == all) break;
                uint4 j;
                if (MODE == 5) { j = jn; } else { j = *reinterpret_cast<const uint4*>(&sList[wave][jj]); }
                LOADS(j.x, a0, b0, c0) LOADS(j.y, a1, b1, c1) LOADS(j.z, a2, b2, c2) LOADS(j.w, a3, b3, c3)
                if (MODE == 5) jn = *reinterpret_cast<const uint4*>(&sList[wave][(jj + 4) & (N - 1)]);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0)
                __builtin_amdgcn_sched_barrier(0);
                STEP(a0, b0, c0) STEP(a1, b1, c1) STEP(a2, b2, c2) STEP(a3, b3, c3)
            }
            T = T * 0.5f + 0.5f;
        }
        done = __builtin_amdgcn_inverse_ballot_w64(dm);
        if (done) C0 += 1.0f;
#undef STEP
#undef LOADS
    } else if (MODE == 6) {
        // MODE 6: MODE 2 with the saturation exit at the END of the trip, kept apart from the loop condition by a non-speculatable asm statement (the
        //         compiler otherwise folds the two exits into one branch through two s_cselect_b64, an s_or_b64 and an s_and_b64 with exec; a select
        //         on the trip limit is folded back into that form; an asm goto crashes hipcc 7.2's loop canonicalisation)
        unsigned long long dm = 0ull;
#define STEP(J) {                                                                                                              \
            const float4 a = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(sA) + (J));               \
            const float4 b = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(sB) + (J));               \
            const float4 c = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(sC) + (J));               \
            const float nlog = neg_exponent(a.x, a.y, a.z, a.w, b.x, pxr, pyr, b.y);                                           \
            const float alpha = fminf(0.99f, __builtin_amdgcn_exp2f(-nlog));                                                   \
            const unsigned long long okm = __builtin_amdgcn_ballot_w64(!(alpha < 1.0f / 255.0f)) & ~dm;                        \
            const float w_raw = alpha * T;                                                                                     \
            const float test_T = T - w_raw;                                                                                    \
            const unsigned long long stopm = okm & __builtin_amdgcn_ballot_w64(test_T < 0.0001f);                              \
            dm |= stopm;                                                                                                       \
            const bool acc = __builtin_amdgcn_inverse_ballot_w64(okm ^ stopm);                                                 \
            const float wgt = acc ? w_raw : 0.0f;                                                                              \
            C0 = __builtin_fmaf(b.z, wgt, C0); C1 = __builtin_fmaf(b.w, wgt, C1); C2 = __builtin_fmaf(c.x, wgt, C2);           \
            T = acc ? test_T : T; }
        for (int r = 0; r < reps; ++r) {
            for (int jj = 0; jj < N; jj += 4) {
                const unsigned j0 = sList[wave][jj], j1 = sList[wave][jj + 1], j2 = sList[wave][jj + 2], j3 = sList[wave][jj + 3];
                STEP(j0) STEP(j1) STEP(j2) STEP(j3)
                if (dm == ~0ull) break;
                asm volatile("" ::: "memory");          // not speculatable: the two exits stay two scalar compare-and-branch pairs
            }
            T = T * 0.5f + 0.5f;
        }
        done = __builtin_amdgcn_inverse_ballot_w64(dm);
        if (done) C0 += 1.0f;
#undef STEP
    } else if (MODE == 7) {
        // MODE 7: no `done` mask at all: a pixel that stops keeps its transmittance with the SIGN flipped (T < 0 = done; |T| = the final value).  A done
        //         pixel's test_T = T (1 - alpha) is negative, i.e. "stops" again and never accumulates; per entry two scalar instructions (acc = c1 & ~c2,
        //         stop = c1 & c2) instead of four, one v_cndmask (with neg / abs modifiers) more; the trip's exit test is ballot(T < 0)
#define STEP(J) {                                                                                                              \
            const float4 a = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(sA) + (J));               \
            const float4 b = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(sB) + (J));               \
            const float4 c = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(sC) + (J));               \
            const float nlog = neg_exponent(a.x, a.y, a.z, a.w, b.x, pxr, pyr, b.y);                                           \
            const float alpha = fminf(0.99f, __builtin_amdgcn_exp2f(-nlog));                                                   \
            const unsigned long long c1m = __builtin_amdgcn_ballot_w64(!(alpha < 1.0f / 255.0f));                              \
            const float w_raw = alpha * T;                                                                                     \
            const float test_T = T - w_raw;                                                                                    \
            const unsigned long long c2m = __builtin_amdgcn_ballot_w64(test_T < 0.0001f);                                      \
            const bool acc = __builtin_amdgcn_inverse_ballot_w64(c1m & ~c2m);                                                  \
            const float wgt = acc ? w_raw : 0.0f;                                                                              \
            C0 = __builtin_fmaf(b.z, wgt, C0); C1 = __builtin_fmaf(b.w, wgt, C1); C2 = __builtin_fmaf(c.x, wgt, C2);           \
            const float T1 = acc ? test_T : T;                                                                                 \
            T = __builtin_amdgcn_inverse_ballot_w64(c1m & c2m) ? -__builtin_fabsf(T1) : T1; }
        for (int r = 0; r < reps; ++r) {
            for (int jj = 0; jj < N; jj += 4) {
                if (__builtin_amdgcn_ballot_w64(T < 0.0f) == ~0ull) break;
                const unsigned j0 = sList[wave][jj], j1 = sList[wave][jj + 1], j2 = sList[wave][jj + 2], j3 = sList[wave][jj + 3];
                STEP(j0) STEP(j1) STEP(j2) STEP(j3)
            }
            T = T * 0.5f + 0.5f;
        }
        T = __builtin_fabsf(T);
#undef STEP
    } else if (MODE == 8) {
        // MODE 8: MODE 2 with upstream's test_T = T * (1 - alpha): (1 - alpha) is off the pixel's serial chain, the chain T -> test_T -> compare -> masks -> T
        //         is one vector instruction shorter (NOT the product's rounding: the product and its oracle use T - alpha T)
        unsigned long long dm = 0ull;
#define STEP(J) {                                                                                                              \
            const float4 a = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(sA) + (J));               \
            const float4 b = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(sB) + (J));               \
            const float4 c = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(sC) + (J));               \
            const float nlog = neg_exponent(a.x, a.y, a.z, a.w, b.x, pxr, pyr, b.y);                                           \
            const float alpha = fminf(0.99f, __builtin_amdgcn_exp2f(-nlog));                                                   \
            const float om = 1.0f - alpha;                                                                                     \
            const unsigned long long okm = __builtin_amdgcn_ballot_w64(!(alpha < 1.0f / 255.0f)) & ~dm;                        \
            const float test_T = T * om;                                                                                       \
            const float w_raw = alpha * T;                                                                                     \
            const unsigned long long stopm = okm & __builtin_amdgcn_ballot_w64(test_T < 0.0001f);                              \
            dm |= stopm;                                                                                                       \
            const bool acc = __builtin_amdgcn_inverse_ballot_w64(okm ^ stopm);                                                 \
            const float wgt = acc ? w_raw : 0.0f;                                                                              \
            C0 = __builtin_fmaf(b.z, wgt, C0); C1 = __builtin_fmaf(b.w, wgt, C1); C2 = __builtin_fmaf(c.x, wgt, C2);           \
            T = acc ? test_T : T; }
        for (int r = 0; r < reps; ++r) {
            for (int jj = 0; jj < N; jj += 4) {
                if (dm == ~0ull) break;
                const unsigned j0 = sList[wave][jj], j1 = sList[wave][jj + 1], j2 = sList[wave][jj + 2], j3 = sList[wave][jj + 3];
                STEP(j0) STEP(j1) STEP(j2) STEP(j3)
            }
            T = T * 0.5f + 0.5f;
        }
        done = __builtin_amdgcn_inverse_ballot_w64(dm);
        if (done) C0 += 1.0f;
#undef STEP
    } else if (MODE == 9) {
        // MODE 9: MODE 2 with EIGHT entries per trip (a fence between the pairs as in the product): the trip's ~10 scalar instructions of loop control per 8 entries
        unsigned long long dm = 0ull;
#define STEP(J) {                                                                                                              \
            const float4 a = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(sA) + (J));               \
            const float4 b = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(sB) + (J));               \
            const float4 c = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(sC) + (J));               \
            const float nlog = neg_exponent(a.x, a.y, a.z, a.w, b.x, pxr, pyr, b.y);                                           \
            const float alpha = fminf(0.99f, __builtin_amdgcn_exp2f(-nlog));                                                   \
            const unsigned long long okm = __builtin_amdgcn_ballot_w64(!(alpha < 1.0f / 255.0f)) & ~dm;                        \
            const float w_raw = alpha * T;                                                                                     \
            const float test_T = T - w_raw;                                                                                    \
            const unsigned long long stopm = okm & __builtin_amdgcn_ballot_w64(test_T < 0.0001f);                              \
            dm |= stopm;                                                                                                       \
            const bool acc = __builtin_amdgcn_inverse_ballot_w64(okm ^ stopm);                                                 \
            const float wgt = acc ? w_raw : 0.0f;                                                                              \
            C0 = __builtin_fmaf(b.z, wgt, C0); C1 = __builtin_fmaf(b.w, wgt, C1); C2 = __builtin_fmaf(c.x, wgt, C2);           \
            T = acc ? test_T : T; }
        for (int r = 0; r < reps; ++r) {
            for (int jj = 0; jj < N; jj += 8) {
                if (dm == ~0ull) break;
                const uint4 ja = *reinterpret_cast<const uint4*>(&sList[wave][jj]), jb = *reinterpret_cast<const uint4*>(&sList[wave][jj + 4]);
                STEP(ja.x) STEP(ja.y)
                __builtin_amdgcn_sched_barrier(0);
                STEP(ja.z) STEP(ja.w)
                __builtin_amdgcn_sched_barrier(0);
                STEP(jb.x) STEP(jb.y)
                __builtin_amdgcn_sched_barrier(0);
                STEP(jb.z) STEP(jb.w)
            }
            T = T * 0.5f + 0.5f;
        }
        done = __builtin_amdgcn_inverse_ballot_w64(dm);
        if (done) C0 += 1.0f;
#undef STEP
    } else {
        const float m[6] = {pxr * pxr, pxr * pyr, pyr * pyr, pxr, pyr, 1.0f};
        auto exps = [&](int jj) -> f4 {           // the four exponents of the group at jj for this lane's pixel
            const unsigned mine = sList[wave][jj + (lane & 3)];
            const float4 qa = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(sA) + mine);
            const float4 qb = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(sB) + mine);
            f4 acc = {0.f, 0.f, 0.f, 0.f};
            acc = __builtin_amdgcn_mfma_f32_4x4x1f32(qa.x, m[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_4x4x1f32(qa.y, m[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_4x4x1f32(qa.z, m[2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_4x4x1f32(qa.w, m[3], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_4x4x1f32(qb.x, m[4], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_4x4x1f32(qb.y, m[5], acc, 0, 0, 0);
            return acc;
        };
#define STEP(J, NLOG) {                                                                                                        \
            const float4 c = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(sC) + (J));               \
            const float alpha = fminf(0.99f, __builtin_amdgcn_exp2f(-(NLOG)));                                                 \
            const bool ok = !done && !(alpha < 1.0f / 255.0f);                                                                 \
            const float w_raw = alpha * T;                                                                                     \
            const float test_T = T - w_raw;                                                                                    \
            const bool stop = ok && test_T < 0.0001f;                                                                          \
            done = done || stop;                                                                                               \
            const bool acc_ = ok != stop;                                                                                      \
            const float wgt = acc_ ? w_raw : 0.0f;                                                                             \
            C0 = __builtin_fmaf(c.x, wgt, C0); C1 = __builtin_fmaf(c.y, wgt, C1); C2 = __builtin_fmaf(c.z, wgt, C2);           \
            T = acc_ ? test_T : T; }
        for (int r = 0; r < reps; ++r) {
            f4 cur = exps(0);
            for (int jj = 0; jj < N; jj += 4) {
                if (__all(done)) break;
                const f4 nxt = exps((jj + 4) & (N - 1));          // one group ahead (the last one wraps: same work, result unused)
                const unsigned j0 = sList[wave][jj], j1 = sList[wave][jj + 1], j2 = sList[wave][jj + 2], j3 = sList[wave][jj + 3];
                STEP(j0, cur[0]) STEP(j1, cur[1]) STEP(j2, cur[2]) STEP(j3, cur[3])
                cur = nxt;
            }
            T = T * 0.5f + 0.5f;
        }
#undef STEP
    }
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + t] = C0 + C1 + C2 + T;
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int MODE>
void run(const char* name) {
    const int reps = 200, blocks = 256 * 8;
    float* out; long long* cyc;
    (void)hipMalloc(&out, blocks * 256 * sizeof(float));
    (void)hipMalloc(&cyc, blocks * 4 * sizeof(long long));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(out, cyc, 4);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(out, cyc, reps);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks * 4);
    (void)hipMemcpy(h.data(), cyc, blocks * 4 * sizeof(long long), hipMemcpyDeviceToHost);
    std::vector<float> o(256);
    (void)hipMemcpy(o.data(), out, 256 * sizeof(float), hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= h.size();
    const double entries = (double)reps * N;
    // 8 workgroups x 4 waves per CU = 8 waves per SIMD; wall-clock form: ns per entry per SIMD = ms * 1e6 / (entries * 8)
    printf("%-44s %.1f counter ticks per list entry per wave = %.1f per entry per SIMD;  wall %.3f ms = %.2f ns per entry per SIMD   (checksum %.4f)\n",
           name, avg / entries, avg / entries / 8.0, ms, ms * 1e6 / (entries * 8.0), o[5]);
    (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
    run<0>("today's step (16 VALU + 3 ds_read_b128)");
    run<1>("6 x v_mfma_f32_4x4x1 per 4 splats + 11 VALU");
    run<0>("today's step (again)");
    run<1>("MFMA form (again)");
    run<2>("today's arithmetic, predicates as lane masks");
    run<3>("lane masks + 0.99 clamp on v_exp's modifier (15 VALU)");
    run<0>("today's step (third)");
    run<2>("lane masks (again)");
    run<3>("lane masks + clamp modifier (again)");
    run<4>("lane masks, loads first, one s_waitcnt per trip");
    run<5>("... + next trip's list words ahead");
    run<2>("lane masks (third)");
    run<4>("one s_waitcnt per trip (again)");
    run<5>("... + list ahead (again)");
    run<6>("lane masks, exits kept apart by an asm statement");
    run<7>("done = sign of T (2 scalar + 17 vector per entry)");
    run<2>("lane masks (fourth)");
    run<6>("exits kept apart (again)");
    run<7>("done = sign of T (again)");
    run<8>("lane masks, test_T = T * (1 - alpha)");
    run<2>("lane masks (fifth)");
    run<8>("test_T = T * (1 - alpha) (again)");
    run<9>("lane masks, eight entries per trip");
    run<2>("lane masks (sixth)");
    run<9>("eight entries per trip (again)");
    return 0;
}
