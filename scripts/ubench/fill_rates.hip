// fill_rates.hip -- what one CU can pull from an L2-resident panel per clock, by path, on gfx950:
//   dma   : global_load_lds_dwordx4 (LDS-DMA, the GEMM's staging path)
//   vgpr  : global_load_dwordx4 into registers (consumed by an OR chain)
//   vgprw : the same + ds_write_b128 of every chunk (the classic staging path)
// Every wave streams its workgroup's private 32 KiB window (re-read `iters` times: L2/L1-resident after the first pass;
// the per-CU footprint is deliberately larger than the 32 KiB L1 so that the loads hit L2, as a GEMM's A/W panels do).
// Build: hipcc --offload-arch=gfx950 -O3 -o fill_rates fill_rates.hip ; run: ./fill_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int WIN = 64 * 1024;            // bytes per workgroup window
constexpr int UNROLL = 8;

__device__ __forceinline__ void dma16(const uint4* g, uint4* l) {
    __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// GEMM-like panel walk: the workgroup owns ROWS rows of a row-major bf16 matrix with a 1 KiB pitch (K = 512) and walks it in
// k-steps of SEG bytes per row (SEG = 64: BK = 32, 16 rows per wave-instruction; SEG = 128: BK = 64, 8 rows per instruction).
template <int SEG>
__global__ __launch_bounds__(256) void panel_kernel(const uint4* __restrict__ src, uint4* __restrict__ sink, int iters, long long* cycles) {
    constexpr int ROWS = 192, PITCH = 1024, CPR = SEG / 16, RPI = 64 / CPR;      // A 64 rows + W 128 rows of a 64x128 tile
    __shared__ uint4 lds[2][ROWS * CPR];
    const char* win = (const char*)src + (size_t)(blockIdx.x % 64) * ROWS * PITCH;   // 12 MiB footprint: L2-resident, shared like a GEMM's panels
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r_in = lane / CPR, c = lane % CPR;
    for (int it = 0; it < iters; ++it) {
        for (int k = 0; k < PITCH / SEG; ++k) {
            const int buf = k & 1;
#pragma unroll
            for (int i = 0; i < ROWS / RPI / 4; ++i) {
                const int row = (i * 4 + wave) * RPI + r_in;
                dma16((const uint4*)(win + (size_t)row * PITCH + k * SEG + c * 16), &lds[buf][(i * 4 + wave) * 64]);
            }
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
        }
    }
    if (lds[0][tid].x == 0x12345678u) sink[tid] = lds[1][tid];
    if (tid == 0) cycles[blockIdx.x] = 0;
}

template <int SEG>
void run_panel(const char* name, const uint4* src, uint4* sink, long long* cyc, int wgs, int iters) {
    panel_kernel<SEG><<<wgs, 256>>>(src, sink, 2, cyc);
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    panel_kernel<SEG><<<wgs, 256>>>(src, sink, iters, cyc);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)wgs * 192 * 1024 * iters;
    printf("%-8s %5d workgroups (%d per CU): %7.2f TB/s chip-wide = %5.1f B/clk/CU at 2.1 GHz\n", name, wgs, wgs / 256, bytes / ms / 1e9,
           bytes / ms / 1e-3 / 256 / 2.1e9);
}

template <int MODE>
__global__ __launch_bounds__(256) void fill_kernel(const uint4* __restrict__ src, uint4* __restrict__ sink, int iters, long long* cycles) {
    __shared__ uint4 lds[UNROLL * 256];
    const uint4* win = src + (size_t)blockIdx.x * (WIN / 16);
    const int tid = threadIdx.x;
    uint4 acc = {0, 0, 0, 0};
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        for (int base = 0; base < WIN / 16; base += UNROLL * 256) {
            if (MODE == 0) {
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) dma16(win + base + u * 256 + tid, &lds[u * 256 + (tid & ~63)]);
                __builtin_amdgcn_s_waitcnt(0);      // vmcnt(0)
            } else {
                uint4 v[UNROLL];
#pragma unroll
                for (int u = 0; u < UNROLL; ++u)       // asm: the compiler would hoist these loop-invariant loads
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[u]) : "v"(win + base + u * 256 + tid) : "memory");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    if (MODE == 2) lds[u * 256 + tid] = v[u];
                    acc.x |= v[u].x; acc.y |= v[u].y; acc.z |= v[u].z; acc.w |= v[u].w;
                }
            }
        }
    }
    const long long t1 = clock64();
    if (MODE == 0 || MODE == 2) { __syncthreads(); acc.x |= lds[tid].x; }
    if (acc.x == 0x12345678u) sink[tid] = acc;
    if (tid == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, const uint4* src, uint4* sink, long long* cyc, int wgs, int iters) {
    fill_kernel<MODE><<<wgs, 256>>>(src, sink, 2, cyc);
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    fill_kernel<MODE><<<wgs, 256>>>(src, sink, iters, cyc);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)wgs * WIN * iters;
    printf("%-6s %5d workgroups (%d per CU): %7.2f TB/s chip-wide = %5.1f B/clk/CU at 2.1 GHz\n", name, wgs, wgs / 256, bytes / ms / 1e9,
           bytes / ms / 1e-3 / 256 / 2.1e9);
}

int main() {
    const int max_wgs = 256 * 8;
    uint4* src; uint4* sink; long long* cyc;
    CHECK(hipMalloc(&src, (size_t)256 << 20));
    CHECK(hipMemset(src, 1, (size_t)256 << 20));
    CHECK(hipMalloc(&sink, 4096 * 16));
    CHECK(hipMalloc(&cyc, max_wgs * 8));
    for (int per_cu : {1, 2, 3, 4}) {
        run_panel<64>("panel64", src, sink, cyc, 256 * per_cu, 100);
        run_panel<128>("panel128", src, sink, cyc, 256 * per_cu, 100);
    }
    for (int per_cu : {1, 2, 4, 8}) {
        const int wgs = 256 * per_cu;
        run<0>("dma", src, sink, cyc, wgs, 200);
        run<1>("vgpr", src, sink, cyc, wgs, 200);
        run<2>("vgprw", src, sink, cyc, wgs, 200);
    }
    return 0;
}
