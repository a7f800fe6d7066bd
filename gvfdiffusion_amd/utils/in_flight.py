"""Independent jobs in flight on separate HIP streams.

The denoise step alternates launches with complementary bounds (tiled attention: VALU issue; row-block projections: the L2 -> CU weight
stream) that each fill the chip on their own, the attention with a half-empty last round of workgroups, and ~110 launch boundaries per
forward; a 4D sample's rasterisation alternates HBM-bound and VALU-bound stages.  Nothing inside ONE sample can overlap them (the chain
is sequential), but independent samples can: two of them in flight, each on its own stream, fill each other's tails and boundaries --
measured on one MI355X: DiT 157 -> 189-195 denoise steps/s aggregate (scripts/dit_two_streams.py; three in flight: 174), rasteriser
13.9 -> 15.4 k frames/s (scripts/rast_two_streams.py).  Results are bit-identical to running the jobs one after the other.

Each job runs on its own Python thread (ctypes calls and torch ops release the GIL) inside `torch.cuda.stream(its stream)`.  A job must own
its mutable state: for the DiT that means one `DiT` instance per job in flight (the condition cache and the hipGraph of an instance belong to
one sample at a time; instances may share parameters).  hipGraph captures of DiT instances are serialised and thread-local
(DiT._forward_graphed), so an instance may capture while the other slots run; results are bit-identical to serial runs either way
(tests/test_inference_script_gpu.py; the round-3 "capture in flight differs in the last bits" was packed-fp32 arithmetic beside another
kernel's MFMAs, profiles/r04_inflight_root_cause.txt, fixed in the build).
"""
import threading
from typing import Callable, List, Sequence

import torch

_POOL = {}


def streams_for(device, n: int) -> List["torch.cuda.Stream"]:
    dev = torch.device(device)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), n)
    if key not in _POOL:
        _POOL[key] = [torch.cuda.Stream(device=dev) for _ in range(n)]
    return _POOL[key]


def run_in_flight(jobs: Sequence[Callable[[int], object]], device, in_flight: int = 2) -> list:
    """Run `jobs` (callables taking their slot index 0 .. in_flight-1) with at most `in_flight` of them running, slot k on stream k; returns
    their results in job order.  The caller's current stream is waited for before the first job starts and waits for every slot's stream at
    the end, so tensors produced before / consumed after the call need no extra synchronisation; an exception in a job is re-raised."""
    dev = torch.device(device)
    in_flight = max(1, min(in_flight, len(jobs)))
    streams = streams_for(dev, in_flight)
    cur = torch.cuda.current_stream(dev)
    results, errors = [None] * len(jobs), []
    next_job = [0]
    lock = threading.Lock()

    # Library handles are created lazily per thread (torch: hipBLASLt / rocBLAS on a thread's first GEMM) and their creation touches the legacy
    # stream, which is illegal while ANOTHER slot captures a hipGraph (hipErrorStreamCaptureImplicit; hipBLASLt exits the process on it).  The
    # package's own paths launch no library GEMM, but a job is free to: every worker makes its first GEMM here, and nobody starts a job (and so
    # a capture) before all of them have.
    ready = threading.Barrier(in_flight) if in_flight > 1 else None

    def worker(slot):
        # Everything a worker does sits inside the try: an exception before the first job (set_device, wait_stream, the warm-up GEMM) used to
        # kill the thread silently -- all workers failing that way returned a list of None without an error, and one failing in front of the
        # barrier left the others waiting in it for ever.  Now it lands in `errors`, and the barrier is ABORTED so that the
        # workers parked in it wake up (BrokenBarrierError) and leave.
        try:
            torch.cuda.set_device(dev)
            s = streams[slot]
            s.wait_stream(cur)
            if ready is not None:
                with torch.cuda.stream(s):
                    w_ = torch.zeros((8, 8), device=dev)
                    torch.mm(w_, w_)
                    s.synchronize()
                ready.wait(timeout=600)
        except threading.BrokenBarrierError:
            return                                      # another worker failed before the jobs started: its exception is in `errors`
        except BaseException as e:                      # noqa: BLE001 -- handed to the caller
            errors.append(e)
            if ready is not None:
                ready.abort()
            return
        while True:
            with lock:
                j = next_job[0]
                next_job[0] += 1
            if j >= len(jobs) or errors:
                break
            try:
                with torch.cuda.stream(s):
                    results[j] = jobs[j](slot)
            except BaseException as e:      # noqa: BLE001 -- handed to the caller
                errors.append(e)
                break

    if in_flight == 1:
        worker(0)
    else:
        threads = [threading.Thread(target=worker, args=(k,)) for k in range(in_flight)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    for s in streams:
        cur.wait_stream(s)
    if errors:
        raise errors[0]
    if next_job[0] < len(jobs):                         # (cannot happen with live workers; a list of None must never look like success)
        raise RuntimeError(f"run_in_flight: only {next_job[0]} of {len(jobs)} jobs were started")
    _record_stream(results, cur)
    return results


def _record_stream(obj, stream):
    """Tensors a job allocated on its side stream and handed back are used on the caller's stream from now on: tell the caching allocator,
    so that the block is not recycled on the side stream while the caller's stream still reads it."""
    if torch.is_tensor(obj):
        if obj.is_cuda:
            obj.record_stream(stream)
    elif isinstance(obj, (list, tuple)):
        for o in obj:
            _record_stream(o, stream)
    elif isinstance(obj, dict):
        for o in obj.values():
            _record_stream(o, stream)
