"""Diagnostic for the divergence of the as-executed batch-64 step: the seeded bench sequence, with per step
  * finiteness of the loss, of every trainable fp32 parameter / gradient / Adam moment and of its bf16 shadow,
  * max |shadow - bf16(master)| per step (a stale or torn shadow shows here before it shows in the loss),
  * the LayerDrop lists of both passes and which optimizer path ran.
usage: python tools/nan_hunt.py [--steps 30] [--batch 64] [--sync-each-pass]   (writes one line per step)"""
import argparse, importlib, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--seconds", type=float, default=4.0)
    ap.add_argument("--loss-scaling", action="store_true")
    a = ap.parse_args()
    t, batch, cfg = bench.build_trainer(a.batch, a.seconds, "bf16", "cuda:0", loss_scaling=a.loss_scaling)
    shadow = importlib.import_module(bench.PKG + ".utils.shadow")
    m = t.audio_encoder.model
    m.cfg.update(bench.HF_REGULARIZERS)
    torch.manual_seed(1234); np.random.seed(1234)
    m.dropped_log = []
    named = [(n, p) for mod, pre in ((m, "w2v2."), (t.fusion_module, "fusion."), (t.decoder1, "dec.")) for n, p in
             ((pre + n, p) for n, p in mod.named_parameters()) if p.requires_grad]
    first_bad = None
    for step in range(a.steps):
        m.dropped_log.clear()
        calls = {"multi": 0, "single": 0}
        out = t.train_step(batch)
        torch.cuda.synchronize()
        loss = float(out["total"])
        bad, nograd, worst = [], [], (0.0, "")
        for n, p in named:
            st = t.optimizer.state.get(p, {})
            if p.grad is None:
                nograd.append(n)
            for tag, x in (("p", p.data), ("g", p.grad), ("m", st.get("exp_avg")), ("v", st.get("exp_avg_sq"))):
                if x is not None and not bool(torch.isfinite(x).all()):
                    bad.append(f"{tag}:{n}")
            sh = shadow.lookup(p)
            if sh is not None:
                d = float((sh.float() - p.data.reshape(-1).to(torch.bfloat16).float()).abs().max())
                if not np.isfinite(d) or d > worst[0]:
                    worst = (d, n)
                if not bool(torch.isfinite(sh).all()):
                    bad.append(f"shadow:{n}")
        steps_set = sorted({int(st["step"]) for st in t.optimizer.state.values() if "step" in st})
        print(f"step {step:3d} loss {loss:10.4f} dropped {m.dropped_log} nograd {len(nograd)} adam_steps {steps_set} "
              f"shadow_err {worst[0]:.3e} {worst[1] if worst[0] else ''} bad {bad[:6]}{'...' if len(bad) > 6 else ''}", flush=True)
        if (bad or not np.isfinite(loss)) and first_bad is None:
            first_bad = step
    print("first non-finite step:", first_bad)


if __name__ == "__main__":
    main()
