// valu_rates.hip -- issue cost (shader cycles per wave64 instruction) of the VALU ops the blend and the
// attention softmax are made of, on gfx950.  Standalone: hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates
// Each kernel runs ITERS x 32 instructions of one kind on 8 independent register chains; W waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)

typedef float f2 __attribute__((ext_vector_type(2)));

template <int OP>
__global__ void k(float* out, long long* cyc, int iters) {
    float a[8];
    f2 p[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 1e-3f + i; p[i] = f2{a[i], a[i] + 1.f}; }
    float b = 0.999f, c = 1e-6f;
    f2 pb = {0.999f, 0.998f}, pc = {1e-6f, 2e-6f};
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (OP == 0) {
#define I(n) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[n]) : "v"(b), "v"(c));
            REP8(I(0) I(1) I(2) I(3)) 
            REP8(I(4) I(5) I(6) I(7))
            REP8(I(0) I(1) I(2) I(3))
            REP8(I(4) I(5) I(6) I(7))
#undef I
        } else if (OP == 1) {
#define I(n) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[n]) : "v"(pb), "v"(pc));
            REP8(I(0) I(1) I(2) I(3)) 
            REP8(I(4) I(5) I(6) I(7))
            REP8(I(0) I(1) I(2) I(3))
            REP8(I(4) I(5) I(6) I(7))
#undef I
        } else if (OP == 2) {
#define I(n) asm volatile("v_exp_f32 %0, %0" : "+v"(a[n]));
            REP8(I(0) I(1) I(2) I(3)) 
            REP8(I(4) I(5) I(6) I(7))
            REP8(I(0) I(1) I(2) I(3))
            REP8(I(4) I(5) I(6) I(7))
#undef I
        } else if (OP == 3) {
#define I(n) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[n]) : "v"(b), "v"(c));
            REP8(I(0) I(1) I(2) I(3)) 
            REP8(I(4) I(5) I(6) I(7))
            REP8(I(0) I(1) I(2) I(3))
            REP8(I(4) I(5) I(6) I(7))
#undef I
        } else if (OP == 4) {
#define I(n) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a[n]) : "v"(b));
            REP8(I(0) I(1) I(2) I(3)) 
            REP8(I(4) I(5) I(6) I(7))
            REP8(I(0) I(1) I(2) I(3))
            REP8(I(4) I(5) I(6) I(7))
#undef I
        } else if (OP == 5) {
#define I(n) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[n]) : "v"(pb));
            REP8(I(0) I(1) I(2) I(3)) 
            REP8(I(4) I(5) I(6) I(7))
            REP8(I(0) I(1) I(2) I(3))
            REP8(I(4) I(5) I(6) I(7))
#undef I
        } else if (OP == 6) {
#define I(n) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[n]) : "v"(pb));
            REP8(I(0) I(1) I(2) I(3)) 
            REP8(I(4) I(5) I(6) I(7))
            REP8(I(0) I(1) I(2) I(3))
            REP8(I(4) I(5) I(6) I(7))
#undef I
        } else if (OP == 7) {
#define I(n) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[n]));
            REP8(I(0) I(1) I(2) I(3)) 
            REP8(I(4) I(5) I(6) I(7))
            REP8(I(0) I(1) I(2) I(3))
            REP8(I(4) I(5) I(6) I(7))
#undef I
        } else if (OP == 8) {
#define I(n) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[n]) : "v"(b));
            REP8(I(0) I(1) I(2) I(3)) 
            REP8(I(4) I(5) I(6) I(7))
            REP8(I(0) I(1) I(2) I(3))
            REP8(I(4) I(5) I(6) I(7))
#undef I
        } else if (OP == 9) {   // exp interleaved 1:3 with fma: does the transcendental pipe overlap plain VALU?
#define I(n) asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(a[n]), "+v"(a[n+1]), "+v"(a[n+2]), "+v"(a[n+3]) : "v"(b), "v"(c));
            REP8(I(0) I(4)) REP8(I(0) I(4)) REP8(I(0) I(4)) REP8(I(0) I(4))   // 64 groups of 4 = 256 instr: counts as 8 x 32
#undef I
        } else if (OP == 10) {
#define I(n) asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %2, %3" : "+v"(a[n]), "+v"(a[n+1]) : "v"(b), "v"(c));
            REP8(I(0) I(2) I(4) I(6)) REP8(I(0) I(2) I(4) I(6))  // 64 pairs = 128 instr
#undef I
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += a[i] + p[i][0] + p[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
void run(const char* name, int instr_per_iter) {
    const int iters = 2000;
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 8 * 1024 * sizeof(float));
    hipMalloc(&cyc, 256 * 8 * sizeof(long long));
    for (int wps = 1; wps <= 4; wps *= 2) {       // waves per SIMD
        const int threads = 256 * wps > 1024 ? 1024 : 256 * wps;
        const int blocks = 256 * (256 * wps / threads);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<OP><<<blocks, threads>>>(out, cyc, 10);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<OP><<<blocks, threads>>>(out, cyc, iters);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> h(blocks);
        hipMemcpy(h.data(), cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
        double avg = 0; for (auto v : h) avg += v; avg /= blocks;
        const double n = (double)iters * instr_per_iter;
        printf("%-28s waves/SIMD %d: %.2f counter ticks per instr per wave (=> %.2f per instr per SIMD), wall %.3f ms => %.2f ns/instr/SIMD\n",
               name, wps, avg / n, avg / n / wps, ms, ms * 1e6 / n / wps);
    }
    hipFree(out); hipFree(cyc);
}

int main() {
    run<0>("v_fma_f32", 32 * 4);
    run<8>("v_mul_f32", 32 * 4);
    run<1>("v_pk_fma_f32", 32 * 4);
    run<5>("v_pk_add_f32", 32 * 4);
    run<6>("v_pk_mul_f32", 32 * 4);
    run<2>("v_exp_f32", 32 * 4);
    run<7>("v_rcp_f32", 32 * 4);
    run<3>("v_max3_f32", 32 * 4);
    run<4>("v_cvt_pk_bf16_f32", 32 * 4);
    run<9>("exp + 3 fma (per 4 instr)", 256);
    run<10>("exp + 1 fma (per 2 instr)", 128);
    return 0;
}
