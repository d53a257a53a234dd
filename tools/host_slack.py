"""Is the host ever the bottleneck of the as-executed step (batch 64)?  Adds a host-side sleep at chosen points of the step and reports
how much of it shows up in the wall time per step: ~0 = the host has that much slack there, ~all of it = the GPU waits for the host."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_time.py")).read().split("for _ in range(3):")[0]
exec(src)


def run(where, ms, n=8):
    def step():
        if where == "start": time.sleep(ms / 1e3)
        t.optimizer.zero_grad(set_to_none=True)
        out = t.forward_losses(batch)
        if where == "pre_bwd": time.sleep(ms / 1e3)
        t.scaler.scale(out["total"]).backward()
        if where == "pre_adam": time.sleep(ms / 1e3)
        t.scaler.step(t.optimizer); t.scaler.update()
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


base = run("none", 0)
print(f"baseline {base:.2f} ms/step", flush=True)
for where in ("start", "pre_bwd", "pre_adam"):
    for ms in (2.0, 5.0):
        w = run(where, ms)
        print(f"sleep {ms:.0f} ms at {where:9s}: {w:.2f} ms/step (+{w - base:.2f})", flush=True)
print(f"baseline again {run('none', 0):.2f} ms/step")
