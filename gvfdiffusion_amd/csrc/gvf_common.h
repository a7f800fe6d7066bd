// gvf_common.h -- shared host/device helpers for the gfx950 kernels (internal; the public C ABI
// lives in include/*.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define GVF_WAVE 64

#include <stdio.h>
#include <stdlib.h>
// Set GVF_DEBUG=1 to have launch failures name the HIP error and source line on stderr.
#define GVF_CHECK_LAUNCH()                                                                    \
    do {                                                                                      \
        hipError_t gvf_e_ = hipGetLastError();                                                \
        if (gvf_e_ != hipSuccess) {                                                           \
            if (getenv("GVF_DEBUG"))                                                          \
                fprintf(stderr, "[gvf] %s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(gvf_e_)); \
            return GVF_ELAUNCH;                                                               \
        }                                                                                     \
    } while (0)

static inline size_t gvf_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Bump allocator over a caller-owned workspace; every carve is 256-byte aligned.
struct GvfCarver {
    char* base;
    size_t off;
    size_t cap;
    bool ok;
    GvfCarver(void* p, size_t bytes) : base((char*)p), off(0), cap(bytes), ok(true) {}
    template <typename T>
    T* take(size_t count) {
        size_t start = gvf_align_up(off, 256);
        size_t end = start + count * sizeof(T);
        if (end > cap) ok = false;
        off = end;
        return (T*)(base ? base + start : nullptr);
    }
};

__device__ __forceinline__ unsigned gvf_lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// Inclusive wave64 prefix sum via DPP-free shuffles.
__device__ __forceinline__ unsigned gvf_wave_incl_scan(unsigned v, unsigned lane) {
#pragma unroll
    for (int d = 1; d < GVF_WAVE; d <<= 1) {
        unsigned o = __shfl_up(v, d, GVF_WAVE);
        if (lane >= (unsigned)d) v += o;
    }
    return v;
}

// XCD-aware workgroup remap (MI355X: 8 XCDs, each with a private 4 MiB L2; the dispatcher is observed to
// place workgroup b on XCD b % 8).  Returns the logical work index of physical workgroup `bid` such that
// each XCD owns ONE contiguous range of logical indices: neighbours in logical order (which share operand
// panels / K-V sets) then hit the same L2 instead of each XCD re-fetching them over the fabric.
// Bijective for any grid size; a different placement only costs speed, never correctness.
__device__ __forceinline__ unsigned gvf_xcd_remap(unsigned bid, unsigned nwg) {
    const unsigned q = nwg >> 3, r = nwg & 7u;
    const unsigned xcd = bid & 7u, slot = bid >> 3;
    return (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + slot;
}
