"""Where do the 3-4 ms per step of the one-rank data-parallel leg come from?  The bench's step with (a) no reducer, (b) a reducer that issues no collective
(world 1: buckets, arenas, hooks and joins only), (c) a reducer that really calls torch.distributed all_reduce on a one-rank RCCL group.  One process, one box."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29519")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
dp = bench.imp("parallel.dp")


def run(name, reducer):
    t, batch, cfg = bench.build_trainer(64, 4.0, "bf16", "cuda:0", reducer=reducer)
    t.audio_encoder.model.cfg.update(bench.HF_REGULARIZERS)
    torch.manual_seed(1234)
    for _ in range(5):
        t.train_step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        t.train_step(batch)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    print(f"{name:58s}: {ms:6.2f} ms per step ({64 / ms * 1e3:6.1f} utt/s)", flush=True)
    del t, batch
    torch.cuda.empty_cache()


for rep in range(2):
    run("(a) no reducer", None)
    run("(b) reducer, no collective (world 1, always_collective=False)", dp.GradBucketReducer(always_collective=False))
    run("(c) reducer + all_reduce on a one-rank RCCL group", dp.GradBucketReducer(always_collective=True))
dist.destroy_process_group()
