// coresident_victim.hip -- gvf_dit_modulation_f32 (victim, stream A) beside a synthetic aggressor kernel (stream B) whose waves share its CUs:
// which kind of co-resident work makes the victim's sums come out wrong (scripts/inflight_mod_kernel.py: gvf_gemm does, 1-3 elements per launch)?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 coresident_victim.hip -o coresident_victim.bin
#include "../../gvfdiffusion_amd/csrc/elem.hip"
#include <cstdio>
#include <vector>
#include <random>
#include <cstring>

// aggressor: 256 threads, `lds_bytes` of dynamic LDS, one of several inner loops
template <int KIND>
__global__ __launch_bounds__(256) void aggressor(const uint4* __restrict__ src, uint4* __restrict__ dst, int iters, size_t n16) {
    extern __shared__ __attribute__((aligned(16))) uint4 lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint4 acc = make_uint4(tid, 0, 0, 0);
    size_t base = ((size_t)blockIdx.x * 4099 * 256) % (n16 - 65536);
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {                       // LDS-DMA: 8 x 1 KiB per wave and iteration, then a barrier and a read
#pragma unroll
            for (int j = 0; j < 8; ++j)
                __builtin_amdgcn_global_load_lds((const void*)(src + base + (size_t)it * 2048 % 32768 + (wave * 8 + j) * 64 + lane),
                                                 (__attribute__((address_space(3))) void*)(lds + (wave * 8 + j) * 64), 16, 0, 0);
            __syncthreads();
            acc.x ^= lds[(tid * 7 + it) & 2047].x;
            __syncthreads();
        } else if (KIND == 1) {                // plain LDS traffic
            lds[(tid + it) & 2047] = acc;
            __syncthreads();
            acc.x ^= lds[(tid * 7 + it) & 2047].y;
            __syncthreads();
        } else if (KIND == 2) {                // global loads into registers
            const uint4 v = src[base + (size_t)it * 2048 % 32768 + tid];
            acc.x ^= v.x; acc.y += v.y;
        } else if (KIND == 3) {                // MFMA + VALU only
            typedef __attribute__((ext_vector_type(4))) float f4; typedef __attribute__((ext_vector_type(8))) __bf16 b8;
            f4 c = {0, 0, 0, 0}; b8 a, b;
            for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(tid + e); b[e] = (__bf16)1.0f; }
#pragma unroll
            for (int j = 0; j < 16; ++j) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
            acc.x ^= __float_as_uint(c[0]);
        } else if (KIND == 4) {                // 16-byte global stores
            dst[base + (size_t)it * 2048 % 32768 + tid] = acc;
        }
    }
    if (acc.x == 0x12345u) dst[tid] = acc;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main() {
    const int C = 512, N = 56320;
    std::mt19937 rng(1); std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> hs(C), hW((size_t)N * C), hb(N);
    for (auto& v : hs) v = nd(rng);
    for (auto& v : hW) v = 0.05f * nd(rng);
    for (auto& v : hb) v = nd(rng);
    float *ds, *dW, *db, *dout, *dref; uint4 *src, *dst;
    const size_t n16 = (size_t)8 << 20;          // 128 MB
    CK(hipMalloc(&ds, C * 4)); CK(hipMalloc(&dW, hW.size() * 4)); CK(hipMalloc(&db, N * 4)); CK(hipMalloc(&dout, (size_t)N * 4 * 64)); CK(hipMalloc(&dref, N * 4));
    CK(hipMalloc(&src, n16 * 16)); CK(hipMalloc(&dst, n16 * 16)); CK(hipMemset(src, 1, n16 * 16));
    CK(hipMemcpy(ds, hs.data(), C * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), N * 4, hipMemcpyHostToDevice));
    hipStream_t sa, sb; CK(hipStreamCreate(&sa)); CK(hipStreamCreate(&sb));
    gvf_dit_modulation_f32(ds, 1, C, dW, db, N, dref, sa); CK(hipStreamSynchronize(sa));
    std::vector<float> href(N), hout((size_t)N * 64);
    CK(hipMemcpy(href.data(), dref, N * 4, hipMemcpyDeviceToHost));
    const char* names[] = {"LDS-DMA (global_load_lds_dwordx4) + barriers", "ds_write / ds_read + barriers", "global_load_dwordx4", "MFMA only", "global_store_dwordx4", "nothing"};
    for (int kind = 0; kind < 6; ++kind) {
        for (int ldsb = 32768; ldsb >= 4096; ldsb /= 8) {
            int bad_launches = 0, bad_elems = 0;
            for (int rep = 0; rep < 8; ++rep) {
                const int iters = 4000;
                if (kind == 0) aggressor<0><<<1024, 256, ldsb, sb>>>(src, dst, iters, n16);
                if (kind == 1) aggressor<1><<<1024, 256, ldsb, sb>>>(src, dst, iters, n16);
                if (kind == 2) aggressor<2><<<1024, 256, ldsb, sb>>>(src, dst, iters, n16);
                if (kind == 3) aggressor<3><<<1024, 256, ldsb, sb>>>(src, dst, iters, n16);
                if (kind == 4) aggressor<4><<<1024, 256, ldsb, sb>>>(src, dst, iters, n16);
                for (int j = 0; j < 64; ++j) gvf_dit_modulation_f32(ds, 1, C, dW, db, N, dout + (size_t)j * N, sa);
                CK(hipStreamSynchronize(sa)); CK(hipStreamSynchronize(sb));
                CK(hipMemcpy(hout.data(), dout, hout.size() * 4, hipMemcpyDeviceToHost));
                for (int j = 0; j < 64; ++j) {
                    int e = 0;
                    for (int n = 0; n < N; ++n) e += memcmp(&hout[(size_t)j * N + n], &href[n], 4) != 0;
                    bad_launches += e > 0; bad_elems += e;
                }
            }
            printf("aggressor: %-46s dynamic LDS %6d B : %3d of 512 victim launches wrong (%d elements)\n", names[kind], ldsb, bad_launches, bad_elems);
            if (kind >= 2) break;
        }
    }
    return 0;
}
