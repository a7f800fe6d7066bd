// fps.hip -- farthest point sampling for gfx950 (MI355X): include/gvf_points.h.
//
// K sequential selections over N points is a latency problem (K = 4096 dependent arg-max reductions over N = 262 144
// points for the DiT's static condition): the points must not be re-read from HBM 4096 times and the chip must work on
// every reduction.  So a batch element gets G = ceil(N / 4096) persistent workgroups (64 for 262 144 points), each
// keeping its 4096 points AND their running minimum distances in registers (16 per thread) for the whole call.  Per
// selection: in-register update + arg-max, wave reduction, one 64-bit atomicMax per workgroup on the selection's own
// slot, one arrival count, and a bounded spin on that count -- the hand-off pattern of cdna_hip_programming.md
// Guideline 16 in its counter form (device-scope atomics on both sides, one slot per selection so nothing is reset
// while in use).  The winner's coordinates are re-read from `pos` (read-only data).
#include "gvf_common.h"
#include "../../include/gvf_rast.h"
#include "../../include/gvf_points.h"

namespace {

constexpr int FPS_THREADS = 256;
constexpr int FPS_PT = 16;                          // points per thread
constexpr int FPS_CHUNK = FPS_THREADS * FPS_PT;     // points per workgroup
constexpr unsigned FPS_SPIN_LIMIT = 1u << 24;       // polls before a workgroup gives up (a hang would cost the box)

struct FpsBatch { int row0, n, k, start, wg0, nwg; long long out0, slot0; };
struct FpsParams { FpsBatch b[GVF_FPS_MAX_BATCH]; int n_batches; };

// key: larger distance wins, then the LOWER index (stored inverted)
__device__ __forceinline__ unsigned long long fps_key(float d, unsigned idx) {
    return ((unsigned long long)__float_as_uint(d) << 32) | (0xFFFFFFFFu - idx);
}

__global__ __launch_bounds__(FPS_THREADS) void fps_kernel(FpsParams p, const float* __restrict__ pos,
                                                          unsigned long long* __restrict__ best, unsigned* __restrict__ arrive,
                                                          long long* __restrict__ out_idx, int* __restrict__ status) {
    __shared__ unsigned long long s_wave[FPS_THREADS / 64];
    __shared__ int s_fail;
    // which batch element / which chunk of it
    int bi = 0;
    while (bi + 1 < p.n_batches && (int)blockIdx.x >= p.b[bi + 1].wg0) ++bi;
    const FpsBatch B = p.b[bi];
    const int g = (int)blockIdx.x - B.wg0, tid = threadIdx.x;
    const float* P = pos + 3 * (size_t)B.row0;

    float px[FPS_PT], py[FPS_PT], pz[FPS_PT], dist[FPS_PT];
#pragma unroll
    for (int r = 0; r < FPS_PT; ++r) {
        const int i = g * FPS_CHUNK + r * FPS_THREADS + tid;       // coalesced in r-major order
        const bool in = i < B.n;
        px[r] = in ? P[3 * (size_t)i] : 0.f; py[r] = in ? P[3 * (size_t)i + 1] : 0.f; pz[r] = in ? P[3 * (size_t)i + 2] : 0.f;
        dist[r] = in ? __builtin_inff() : -1.0f;                   // padding can never win
    }
    if (tid == 0) s_fail = 0;
    unsigned sel = (unsigned)B.start;
    for (int it = 0; it < B.k; ++it) {
        if (g == 0 && tid == 0) out_idx[B.out0 + it] = (long long)B.row0 + sel;
        if (it + 1 == B.k) break;
        const float sx = P[3 * (size_t)sel], sy = P[3 * (size_t)sel + 1], sz = P[3 * (size_t)sel + 2];
        unsigned long long mine = 0ull;
#pragma unroll
        for (int r = 0; r < FPS_PT; ++r) {
            const float dx = px[r] - sx, dy = py[r] - sy, dz = pz[r] - sz;
            const float d = (dx * dx + dy * dy) + dz * dz;
            dist[r] = fminf(dist[r], d);
            const unsigned long long key = fps_key(dist[r], (unsigned)(g * FPS_CHUNK + r * FPS_THREADS + tid));
            mine = (dist[r] >= 0.0f && key > mine) ? key : mine;
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const unsigned lo = __shfl_xor((unsigned)mine, o, 64), hi = __shfl_xor((unsigned)(mine >> 32), o, 64);
            const unsigned long long other = ((unsigned long long)hi << 32) | lo;
            mine = other > mine ? other : mine;
        }
        if ((tid & 63) == 0) s_wave[tid >> 6] = mine;
        __syncthreads();
        unsigned long long* slot = best + B.slot0 + it;
        unsigned* cnt = arrive + B.slot0 + it;
        if (tid == 0) {
            unsigned long long m = s_wave[0];
#pragma unroll
            for (int w = 1; w < FPS_THREADS / 64; ++w) m = s_wave[w] > m ? s_wave[w] : m;
            // device-scope atomics on both sides: the slot's value is complete once all nwg arrivals are counted
            __hip_atomic_fetch_max(slot, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            unsigned spins = 0;
            while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)B.nwg) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > FPS_SPIN_LIMIT) { s_fail = 1; break; }
            }
            const unsigned long long win = __hip_atomic_load(slot, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
            s_wave[0] = win;
        }
        __syncthreads();
        if (s_fail) { if (tid == 0 && status != nullptr) *status = 1; return; }
        sel = 0xFFFFFFFFu - (unsigned)(s_wave[0] & 0xFFFFFFFFull);
        __syncthreads();                                           // s_wave is rewritten next round
    }
}

}  // namespace

extern "C" int gvf_fps_scratch_bytes(int n_batches, int max_k, size_t* bytes) {
    if (!bytes || n_batches <= 0 || n_batches > GVF_FPS_MAX_BATCH || max_k <= 0) return GVF_EINVAL;
    const size_t slots = (size_t)n_batches * (size_t)max_k;
    *bytes = gvf_align_up(slots * sizeof(unsigned long long), 256) + gvf_align_up(slots * sizeof(unsigned), 256) + 256;
    return GVF_OK;
}

extern "C" int gvf_fps(const float* pos, const int32_t* ptr_host, int n_batches, const int32_t* k_host,
                       const int32_t* start_host, int64_t* out_idx, void* scratch, size_t scratch_bytes,
                       int32_t* status_out, void* stream_) {
    if (!pos || !ptr_host || !k_host || !start_host || !out_idx || !scratch) return GVF_EINVAL;
    if (n_batches <= 0 || n_batches > GVF_FPS_MAX_BATCH) return GVF_EINVAL;
    if ((((uintptr_t)scratch) & 255) != 0) return GVF_EINVAL;
    FpsParams p;
    p.n_batches = n_batches;
    int wg = 0, max_k = 0;
    long long out0 = 0;
    for (int b = 0; b < n_batches; ++b) {
        const int n = ptr_host[b + 1] - ptr_host[b], k = k_host[b];
        if (n <= 0 || n > GVF_FPS_MAX_POINTS || k <= 0 || k > n || start_host[b] < 0 || start_host[b] >= n) return GVF_EINVAL;
        FpsBatch& B = p.b[b];
        B.row0 = ptr_host[b]; B.n = n; B.k = k; B.start = start_host[b];
        B.wg0 = wg; B.nwg = (n + FPS_CHUNK - 1) / FPS_CHUNK; wg += B.nwg;
        B.out0 = out0; out0 += k;
        max_k = k > max_k ? k : max_k;
    }
    // every workgroup of the call must be resident at once (they wait for each other): 256 CUs x >= 4 of these
    if (wg > 1024) return GVF_EINVAL;
    size_t need = 0;
    gvf_fps_scratch_bytes(n_batches, max_k, &need);
    if (scratch_bytes < need) return GVF_ENOSPC;
    const size_t slots = (size_t)n_batches * (size_t)max_k;
    unsigned long long* best = (unsigned long long*)scratch;
    unsigned* arrive = (unsigned*)((char*)scratch + gvf_align_up(slots * sizeof(unsigned long long), 256));
    for (int b = 0; b < n_batches; ++b) p.b[b].slot0 = (long long)b * max_k;
    hipStream_t stream = (hipStream_t)stream_;
    (void)hipGetLastError();
    if (hipMemsetAsync(scratch, 0, need, stream) != hipSuccess) return GVF_ELAUNCH;
    if (status_out != nullptr && hipMemsetAsync(status_out, 0, sizeof(int32_t), stream) != hipSuccess) return GVF_ELAUNCH;
    hipLaunchKernelGGL(fps_kernel, dim3(wg), dim3(FPS_THREADS), 0, stream, p, pos, best, arrive, (long long*)out_idx, (int*)status_out);
    GVF_CHECK_LAUNCH();
    return GVF_OK;
}
