"""cProfile of the host side of the training step at BASELINE configs[0] (2 x 1 s clips, as executed): the launch-bound case."""
import cProfile, io, os, pstats, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
t, batch, cfg = bench.build_trainer(2, 1.0, "bf16", "cuda:0")
t.audio_encoder.model.cfg.update(bench.HF_REGULARIZERS)
torch.manual_seed(1234)
for _ in range(6):
    t.train_step(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    t.train_step(batch)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue time per step {(t1 - t0) / 20 * 1e3:.2f} ms, wall per step {(t2 - t0) / 20 * 1e3:.2f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    t.train_step(batch)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45)
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(60)
print(s.getvalue())
