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

// hipFuncSetAttribute is PER DEVICE and illegal during stream capture: a kernel that needs more dynamic LDS than the default limit raises it
// once per (process, device), under a lock (callers may be on several host threads), from an entry point that runs outside any capture
// (a pack / workspace-size call) -- and the launch site insists, so a first launch on a second device of the same process is covered too.
#include <mutex>
struct GvfPerDeviceOnce {
    std::mutex m;
    uint64_t done = 0;              // bit d: device d has been configured
};
template <typename F>
static inline bool gvf_once_per_device(GvfPerDeviceOnce& o, F&& configure) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    std::lock_guard<std::mutex> g(o.m);
    if (dev >= 0 && dev < 64 && ((o.done >> dev) & 1ull)) return true;
    if (!configure()) {
        (void)hipGetLastError();
        return false;
    }
    if (dev >= 0 && dev < 64) o.done |= 1ull << dev;
    return true;
}

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

// The same scan / the wave-wide unsigned min, max on the DPP network (no LDS round trips: ~10 VALU instructions instead of six
// dependent ds_bpermute latencies).  ALL 64 lanes must be active.  Within a row of 16 lanes: shifts by 1, 2, 4, 8 (a lane whose
// source falls outside the row keeps `old`: 0 for the sum, its own value for min / max); across rows: row_bcast:15 into rows 1 and 3,
// row_bcast:31 into rows 2 and 3 -- lane 63 then holds the reduction, every lane its inclusive prefix.
#define GVF_DPP_ROW_SHR(n) (0x110 + (n))
#define GVF_DPP_ROW_BCAST15 0x142
#define GVF_DPP_ROW_BCAST31 0x143
__device__ __forceinline__ unsigned gvf_wave_incl_scan_dpp(unsigned v) {
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, GVF_DPP_ROW_SHR(1), 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, GVF_DPP_ROW_SHR(2), 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, GVF_DPP_ROW_SHR(4), 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, GVF_DPP_ROW_SHR(8), 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, GVF_DPP_ROW_BCAST15, 0xa, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, GVF_DPP_ROW_BCAST31, 0xc, 0xf, false);
    return v;
}
#define GVF_WAVE_REDUCE_DPP(OP)                                                                                         \
    v = OP(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, GVF_DPP_ROW_SHR(1), 0xf, 0xf, false));            \
    v = OP(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, GVF_DPP_ROW_SHR(2), 0xf, 0xf, false));            \
    v = OP(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, GVF_DPP_ROW_SHR(4), 0xf, 0xf, false));            \
    v = OP(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, GVF_DPP_ROW_SHR(8), 0xf, 0xf, false));            \
    v = OP(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, GVF_DPP_ROW_BCAST15, 0xa, 0xf, false));           \
    v = OP(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, GVF_DPP_ROW_BCAST31, 0xc, 0xf, false));           \
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63)
__device__ __forceinline__ unsigned gvf_wave_umin(unsigned v) { GVF_WAVE_REDUCE_DPP(min); }
__device__ __forceinline__ unsigned gvf_wave_umax(unsigned v) { GVF_WAVE_REDUCE_DPP(max); }

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
