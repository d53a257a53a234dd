"""cProfile of the host side of the as-executed step at config 1 (2 x 1 s: launch-bound): where does the Python launch path spend its time?"""
import cProfile, io, os, pstats, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
t, batch, cfg = bench.build_trainer(int(os.environ.get("HP_B", "2")), float(os.environ.get("HP_S", "1")), "bf16", "cuda:0")
t.audio_encoder.model.cfg.update(bench.HF_REGULARIZERS)
torch.manual_seed(1234); np.random.seed(1234)
for _ in range(5):
    t.train_step(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    t.train_step(batch)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue time per step {(t1 - t0) / 10 * 1e3:.2f} ms, wall per step {(t2 - t0) / 10 * 1e3:.2f} ms")
torch.autograd.set_multithreading_enabled(False)      # backward functions in this thread: visible to cProfile
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    t.train_step(batch)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(70)
print(s.getvalue()[:16000])
