"""Is the training step host-bound anywhere?  The bench's trainer and batch (as executed: regularisers on, two audio passes), steps split at the
phase boundaries of MultimodalTrainer.train_step: host enqueue time per phase, and - right after each phase is enqueued - whether the GPU has
ALREADY finished it (event.query() true = the device ran dry there and waits for the host).  No tracer, no synchronisation inside the loop.
usage (GPU box): python tools/host_gap.py [steps]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
t, batch, cfg = bench.build_trainer(64, 4.0, "bf16", "cuda:0")
t.audio_encoder.model.cfg.update(bench.HF_REGULARIZERS)
torch.manual_seed(1234)
for _ in range(4):
    t.train_step(batch)
torch.cuda.synchronize()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rows = []
w0 = time.perf_counter()
for s in range(N):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    h = [time.perf_counter()]
    ev[0].record()
    t.optimizer.zero_grad(set_to_none=True); t._head_arena.begin_step()
    begin = getattr(t.audio_encoder.model, "begin_grad_step", None)
    if begin is not None:
        begin()
    out = t.forward_losses(batch)
    h.append(time.perf_counter()); ev[1].record(); q1 = ev[1].query()
    t.scaler.scale(out["total"]).backward()
    h.append(time.perf_counter()); ev[2].record(); q2 = ev[2].query()
    t._head_arena.finalize()
    t.scaler.step(t.optimizer); t.scaler.update()
    h.append(time.perf_counter()); ev[3].record(); q3 = ev[3].query()
    rows.append((h, ev, (q1, q2, q3)))
torch.cuda.synchronize()
w1 = time.perf_counter()
print(f"wall {1e3 * (w1 - w0) / N:.2f} ms per step ({64 * N / (w1 - w0):.1f} utt/s)")
print("step | host ms: forward backward optimizer | device ms between the phase events: forward backward optimizer | device already idle after: fwd bwd opt")
for i, (h, ev, q) in enumerate(rows):
    hd = [1e3 * (h[k + 1] - h[k]) for k in range(3)]
    gd = [ev[k].elapsed_time(ev[k + 1]) for k in range(3)]
    print(f"{i:3d} | {hd[0]:7.2f} {hd[1]:7.2f} {hd[2]:7.2f} | {gd[0]:7.2f} {gd[1]:7.2f} {gd[2]:7.2f} | {q}")
