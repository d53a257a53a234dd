"""Host-side (enqueue) time of the phases of a training step at the bench configuration, without synchronising: where can the GPU run dry?"""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_profile.py")).read().split("for _ in range(3):")[0])
for _ in range(3):
    t.train_step(batch)
torch.cuda.synchronize()
acc = {"zero": 0.0, "fwd": 0.0, "bwd": 0.0, "adam": 0.0}
N = 8
w0 = time.perf_counter()
for _ in range(N):
    a = time.perf_counter(); t.optimizer.zero_grad(set_to_none=True)
    b = time.perf_counter(); out = t.forward_losses(batch)
    c = time.perf_counter(); out["total"].backward()
    d = time.perf_counter(); t.optimizer.step()
    e = time.perf_counter()
    acc["zero"] += b - a; acc["fwd"] += c - b; acc["bwd"] += d - c; acc["adam"] += e - d
torch.cuda.synchronize()
w1 = time.perf_counter()
print({k: round(v / N * 1e3, 3) for k, v in acc.items()}, "ms per step (host); wall", round((w1 - w0) / N * 1e3, 2))
