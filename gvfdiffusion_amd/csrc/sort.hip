// sort.hip -- stable LSD radix sort of (u64 key, u32 value) pairs for gfx950 (wave64).
//
// Rasteriser stage R4: upstream calls cub::DeviceRadixSort::SortPairs on
// (tile<<32 | depth_bits) keys; the reference itself never sees this (external package, call site
// renderers/gaussian_render.py:198-220).  This is an own implementation designed around the
// MI355X: a FIXED grid of at most 1024 workgroups (4 per CU) each owning one contiguous key range,
// so the per-pass spine is 256 x 1024 counters regardless of n; wave64 ballot matching gives the
// stable in-tile rank; tiles are re-ordered through LDS so global writes are coalesced runs.
// The element count is read from device memory (n_ptr) -- the rasteriser never syncs to learn D.
//
// Per 8-bit pass: upsweep (digit histogram per block range) -> spine (one block per digit scans its
// row) -> downsweep (rank + scatter).  HBM traffic per pass = 8 B (upsweep keys) + 12 B read + 12 B
// written per pair.
#include "gvf_common.h"
#include "gvf_sort.h"
#include "../../include/gvf_rast.h"

namespace {

constexpr int SORT_THREADS = 256;
constexpr int SORT_WAVES = SORT_THREADS / GVF_WAVE;
constexpr int SORT_IPT = 16;
constexpr int SORT_TILE = SORT_THREADS * SORT_IPT;       // 4096 pairs per tile
constexpr int SORT_WAVE_CHUNK = SORT_TILE / SORT_WAVES;  // 1024 consecutive pairs per wave
constexpr int SORT_BITS = 8;
constexpr int SORT_BINS = 1 << SORT_BITS;

struct BlockRange {
    uint32_t start, end;
};

__device__ __forceinline__ BlockRange block_range(uint32_t n, int nb, int b) {
    uint32_t per = (n + nb - 1) / nb;
    per = (per + SORT_TILE - 1) / SORT_TILE * SORT_TILE;
    uint64_t s = (uint64_t)per * (uint32_t)b;
    BlockRange r;
    r.start = s < n ? (uint32_t)s : n;
    uint64_t e = s + per;
    r.end = e < n ? (uint32_t)e : n;
    return r;
}

__device__ __forceinline__ uint32_t load_n(const uint32_t* n_ptr, uint32_t n_cap) {
    uint32_t n = *n_ptr;
    return n < n_cap ? n : n_cap;
}

// hist layout: [digit][block]
__global__ __launch_bounds__(SORT_THREADS) void sort_upsweep(const uint64_t* __restrict__ keys,
                                                             const uint32_t* __restrict__ n_ptr,
                                                             uint32_t n_cap, int shift,
                                                             uint32_t* __restrict__ hist, int nb) {
    __shared__ uint32_t h[SORT_BINS];
    const int t = threadIdx.x;
    h[t] = 0;
    __syncthreads();
    const uint32_t n = load_n(n_ptr, n_cap);
    const BlockRange br = block_range(n, nb, blockIdx.x);
    for (uint32_t base = br.start; base < br.end; base += SORT_TILE) {
#pragma unroll 4
        for (int i = 0; i < SORT_IPT; ++i) {
            uint32_t e = base + i * SORT_THREADS + t;
            if (e < br.end) {
                uint32_t d = (uint32_t)(keys[e] >> shift) & (SORT_BINS - 1);
                atomicAdd(&h[d], 1u);
            }
        }
    }
    __syncthreads();
    hist[(size_t)t * nb + blockIdx.x] = h[t];
}

// One block per digit: exclusive scan of hist[d][0..nb) in place, row total -> digit_tot[d].
__global__ __launch_bounds__(SORT_THREADS) void sort_spine(uint32_t* __restrict__ hist,
                                                           uint32_t* __restrict__ digit_tot, int nb) {
    __shared__ uint32_t wsum[SORT_WAVES];
    const int t = threadIdx.x;
    const unsigned lane = t & 63, w = t >> 6;
    uint32_t* row = hist + (size_t)blockIdx.x * nb;
    const int per = (nb + SORT_THREADS - 1) / SORT_THREADS;  // <= 4 for nb <= 1024
    uint32_t v[8];
    uint32_t s = 0;
    for (int i = 0; i < per; ++i) {
        int j = t * per + i;
        v[i] = j < nb ? row[j] : 0u;
        s += v[i];
    }
    uint32_t incl = gvf_wave_incl_scan(s, lane);
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    uint32_t wbase = 0, total = 0;
    for (int k = 0; k < SORT_WAVES; ++k) {
        if ((unsigned)k < w) wbase += wsum[k];
        total += wsum[k];
    }
    uint32_t run = wbase + incl - s;
    for (int i = 0; i < per; ++i) {
        int j = t * per + i;
        if (j < nb) row[j] = run;
        run += v[i];
    }
    if (t == 0) digit_tot[blockIdx.x] = total;
}

__global__ __launch_bounds__(SORT_THREADS) void sort_downsweep(
    const uint64_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
    uint64_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, const uint32_t* __restrict__ n_ptr,
    uint32_t n_cap, int shift, const uint32_t* __restrict__ hist, const uint32_t* __restrict__ digit_tot,
    int nb) {
    __shared__ uint64_t lds_keys[SORT_TILE];             // 32 KiB
    __shared__ uint32_t lds_vals[SORT_TILE];             // 16 KiB
    __shared__ uint32_t wave_hist[SORT_WAVES][SORT_BINS];
    __shared__ uint32_t running[SORT_BINS];              // global write cursor per digit
    __shared__ int32_t glob_delta[SORT_BINS];            // global pos = glob_delta[d] + local index
    __shared__ uint32_t wsum[SORT_WAVES];

    const int t = threadIdx.x;
    const unsigned lane = t & 63, w = t >> 6;
    const uint64_t lt_mask = (1ull << lane) - 1ull;

    // digit base = exclusive scan of the digit totals; + this block's prefix within the digit row
    {
        uint32_t tot = digit_tot[t];
        uint32_t incl = gvf_wave_incl_scan(tot, lane);
        if (lane == 63) wsum[w] = incl;
        __syncthreads();
        uint32_t wbase = 0;
        for (unsigned k = 0; k < w; ++k) wbase += wsum[k];
        running[t] = wbase + incl - tot + hist[(size_t)t * nb + blockIdx.x];
        __syncthreads();
    }

    const uint32_t n = load_n(n_ptr, n_cap);
    const BlockRange br = block_range(n, nb, blockIdx.x);

    for (uint32_t base = br.start; base < br.end; base += SORT_TILE) {
        for (int k = 0; k < SORT_WAVES; ++k) wave_hist[k][t] = 0;
        __syncthreads();

        // A: stable rank of every key among equal digits of its wave chunk (chunk order = round, lane)
        uint64_t key[SORT_IPT];
        uint32_t info[SORT_IPT];  // digit | rank << SORT_BITS
        const uint32_t cbase = base + w * SORT_WAVE_CHUNK;
#pragma unroll
        for (int r = 0; r < SORT_IPT; ++r) {
            uint32_t e = cbase + r * GVF_WAVE + lane;
            bool valid = e < br.end;
            key[r] = valid ? keys_in[e] : 0ull;
            uint32_t d = (uint32_t)(key[r] >> shift) & (SORT_BINS - 1);
            uint64_t peers = __ballot(valid);
#pragma unroll
            for (int b = 0; b < SORT_BITS; ++b) {
                uint64_t m = __ballot((d >> b) & 1u);
                peers &= ((d >> b) & 1u) ? m : ~m;
            }
            uint32_t before = __popcll(peers & lt_mask);
            uint32_t prev = wave_hist[w][d];
            info[r] = d | ((prev + before) << SORT_BITS);
            if (valid && before == 0) wave_hist[w][d] = prev + __popcll(peers);
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();

        // B: per-digit tile counts -> local start (exclusive over digits), per-wave bases, cursors
        {
            uint32_t c[SORT_WAVES], total = 0;
            for (int k = 0; k < SORT_WAVES; ++k) { c[k] = wave_hist[k][t]; total += c[k]; }
            uint32_t incl = gvf_wave_incl_scan(total, lane);
            if (lane == 63) wsum[w] = incl;
            __syncthreads();
            uint32_t wbase = 0;
            for (unsigned k = 0; k < w; ++k) wbase += wsum[k];
            uint32_t local_start = wbase + incl - total;
            uint32_t run = local_start;
            for (int k = 0; k < SORT_WAVES; ++k) { wave_hist[k][t] = run; run += c[k]; }
            glob_delta[t] = (int32_t)(running[t] - local_start);
            running[t] += total;
        }
        __syncthreads();

        // C: re-order the tile through LDS
#pragma unroll
        for (int r = 0; r < SORT_IPT; ++r) {
            uint32_t e = cbase + r * GVF_WAVE + lane;
            if (e < br.end) {
                uint32_t d = info[r] & (SORT_BINS - 1);
                uint32_t li = wave_hist[w][d] + (info[r] >> SORT_BITS);
                lds_keys[li] = key[r];
                lds_vals[li] = vals_in[e];
            }
        }
        __syncthreads();

        // D: coalesced runs to global
        const uint32_t tile_n = (br.end - base) < (uint32_t)SORT_TILE ? (br.end - base) : (uint32_t)SORT_TILE;
#pragma unroll 4
        for (int i = 0; i < SORT_IPT; ++i) {
            uint32_t j = i * SORT_THREADS + t;
            if (j < tile_n) {
                uint64_t k = lds_keys[j];
                uint32_t d = (uint32_t)(k >> shift) & (SORT_BINS - 1);
                uint32_t pos = (uint32_t)(glob_delta[d] + (int32_t)j);
                keys_out[pos] = k;
                vals_out[pos] = lds_vals[j];
            }
        }
        __syncthreads();
    }
}

__global__ void sort_set_u32(uint32_t* p, uint32_t v) { *p = v; }

}  // namespace

int gvf_sort_num_blocks(int64_t n_cap) {
    int64_t nb = (n_cap + SORT_TILE - 1) / SORT_TILE;
    if (nb < 1) nb = 1;
    if (nb > 1024) nb = 1024;
    return (int)nb;
}

size_t gvf_sort_tmp_bytes(int64_t n) {
    int nb = gvf_sort_num_blocks(n);
    return gvf_align_up((size_t)SORT_BINS * nb * sizeof(uint32_t), 256) + gvf_align_up(SORT_BINS * sizeof(uint32_t), 256) + 256;
}

// Internal entry: n on device. Returns (via *result_in_alt) where the sorted data ended up.
int gvf_sort_pairs_device_n(uint64_t* keys, uint64_t* keys_alt, uint32_t* vals, uint32_t* vals_alt,
                            const uint32_t* n_ptr, int64_t n_cap, int begin_bit, int end_bit, void* tmp,
                            size_t tmp_bytes, hipStream_t stream, int* result_in_alt) {
    if (n_cap < 0 || begin_bit < 0 || end_bit < begin_bit || end_bit > 64) return GVF_EINVAL;
    if (tmp_bytes < gvf_sort_tmp_bytes(n_cap)) return GVF_ENOSPC;
    *result_in_alt = 0;
    if (n_cap == 0 || end_bit == begin_bit) return GVF_OK;
    const int nb = gvf_sort_num_blocks(n_cap);
    uint32_t* hist = (uint32_t*)tmp;
    uint32_t* digit_tot = (uint32_t*)((char*)tmp + gvf_align_up((size_t)SORT_BINS * nb * sizeof(uint32_t), 256));
    uint64_t* kin = keys; uint64_t* kout = keys_alt;
    uint32_t* vin = vals; uint32_t* vout = vals_alt;
    int flips = 0;
    for (int shift = begin_bit; shift < end_bit; shift += SORT_BITS) {
        hipLaunchKernelGGL(sort_upsweep, dim3(nb), dim3(SORT_THREADS), 0, stream, kin, n_ptr, (uint32_t)n_cap, shift, hist, nb);
        hipLaunchKernelGGL(sort_spine, dim3(SORT_BINS), dim3(SORT_THREADS), 0, stream, hist, digit_tot, nb);
        hipLaunchKernelGGL(sort_downsweep, dim3(nb), dim3(SORT_THREADS), 0, stream, kin, vin, kout, vout, n_ptr,
                           (uint32_t)n_cap, shift, hist, digit_tot, nb);
        GVF_CHECK_LAUNCH();
        uint64_t* tk = kin; kin = kout; kout = tk;
        uint32_t* tv = vin; vin = vout; vout = tv;
        ++flips;
    }
    *result_in_alt = flips & 1;
    return GVF_OK;
}

extern "C" int gvf_sort_pairs_u64(uint64_t* keys, uint64_t* keys_alt, uint32_t* values, uint32_t* values_alt,
                                  int64_t n, int end_bit, void* tmp, size_t tmp_bytes, void* stream_) {
    if (n < 0 || n > 0xFFFFFFFFll) return GVF_EINVAL;
    if (n > 0 && (!keys || !keys_alt || !values || !values_alt || !tmp)) return GVF_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    (void)hipGetLastError();
    size_t need = gvf_sort_tmp_bytes(n);
    if (tmp_bytes < need) return GVF_ENOSPC;
    if (n == 0) return GVF_OK;
    // n lives in the last 256 bytes of tmp
    uint32_t* n_dev = (uint32_t*)((char*)tmp + need - 256);
    hipLaunchKernelGGL(sort_set_u32, dim3(1), dim3(1), 0, stream, n_dev, (uint32_t)n);
    GVF_CHECK_LAUNCH();
    int in_alt = 0;
    int rc = gvf_sort_pairs_device_n(keys, keys_alt, values, values_alt, n_dev, n, 0, end_bit, tmp, need, stream, &in_alt);
    if (rc != GVF_OK) return rc;
    if (in_alt) {
        if (hipMemcpyAsync(keys, keys_alt, n * sizeof(uint64_t), hipMemcpyDeviceToDevice, stream) != hipSuccess) return GVF_ELAUNCH;
        if (hipMemcpyAsync(values, values_alt, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream) != hipSuccess) return GVF_ELAUNCH;
    }
    return GVF_OK;
}
