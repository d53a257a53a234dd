"""cProfile of the host side of the bench's training step with the one-rank RCCL leg (--force-dp): what does the data-parallel machinery cost the host?"""
import cProfile, io, os, pstats, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29517")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
dp = bench.imp("parallel.dp")
red = dp.GradBucketReducer(always_collective=True)
t, batch, cfg = bench.build_trainer(64, 4.0, "bf16", "cuda:0", reducer=red)
t.audio_encoder.model.cfg.update(bench.HF_REGULARIZERS)
torch.manual_seed(1234)
for _ in range(4):
    t.train_step(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(6):
    t.train_step(batch)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"host enqueue time per step {(t1 - t0) / 6 * 1e3:.2f} ms, wall per step {(t2 - t0) / 6 * 1e3:.2f} ms")
pr = cProfile.Profile(); pr.enable()
for _ in range(6):
    t.train_step(batch)
torch.cuda.synchronize(); pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(25)
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(40)
print(s.getvalue())
dist.destroy_process_group()
