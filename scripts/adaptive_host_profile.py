"""Host-side profile of the adaptive sampler on the full-size DiT (GPU only): where the time between the error-norm read and the next launch goes."""
import os, sys, time, torch, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
w = bench.DiTWorkload(dev, input_seed=1)
w.solver.verbose = False
run = lambda: w.solver.sample(w.x, steps=100, t_start=1.0, t_end=1 / 1000, order=2, skip_type="time_uniform", method="adaptive")
run(); run()
torch.cuda.synchronize(); t0 = time.perf_counter(); run(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
st = w.solver.spec_stats
calls = w.solver.last_nfe - st["rejected"]
print("adaptive: %.1f ms, %d evaluations (%d reported), %.3f ms per evaluation, %d steps" % (dt * 1e3, calls, w.solver.last_nfe, dt * 1e3 / calls, st["steps"]))
pr = cProfile.Profile(); pr.enable(); run(); pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
