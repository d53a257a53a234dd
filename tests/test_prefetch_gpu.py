"""The host -> device leg of the reference's step (model/trainer.py:66-75) moved one step ahead onto a prefetch stream
(dataset/prefetch.py): steps fed through the double-buffered prefetcher must equal steps on HBM-resident batches bit for bit, and
bench.py --h2d / --force-dp must report what DESIGN.md section 4 / 5 promise."""
import json
import os
import subprocess
import sys

import pytest
import torch

from conftest import ROOT, pkg

pytestmark = pytest.mark.gpu


def _run(prefetch: bool):
    from test_step_gpu import build
    init = pkg("utils.init"); synth = pkg("dataset.synthetic"); pre = pkg("dataset.prefetch")
    cfg = dict(init.W2V2_TINY)
    t = build(cfg, "bf16")
    t.visual_encoder.train(); t.audio_encoder.train(); t.fusion_module.train(); t.decoder1.train()
    T_enc = None
    host = []
    for s in range(4):                                      # four different batches: both buffer slots are reused
        cb = synth.make_batch(2, 1.0, seed=50 + s, ragged=(s % 2 == 1))
        if T_enc is None:
            T_enc = int(pkg("model.w2v2").conv_out_lengths(cfg, cb["audio"].shape[1]))
        hb = dict(cb); hb.update(t.host_metadata(cb, T_enc))
        host.append(hb)
    torch.manual_seed(7)
    losses = []
    if prefetch:
        pf = pre.DevicePrefetcher("cuda")
        pinned = [pre.pin_batch(h) for h in host]
        slot = pf.stage(pinned[0])
        for i in range(len(host)):
            cur = pf.get(slot)
            nxt = pf.stage(pinned[i + 1]) if i + 1 < len(host) else None
            out = t.train_step(cur)
            pf.release(slot)
            slot = nxt
            losses.append(out["total"].detach().clone())
        assert pf.bytes_last == sum(v.numel() * v.element_size() for k, v in host[-1].items() if torch.is_tensor(v) and not k.startswith("_"))
    else:
        for h in host:
            dev = {k: (v.cuda() if torch.is_tensor(v) and not k.startswith("_") else v) for k, v in h.items()}
            out = t.train_step(dev)
            losses.append(out["total"].detach().clone())
    torch.cuda.synchronize()
    params = [p.detach().clone() for m in (t.audio_encoder, t.fusion_module, t.decoder1) for p in m.parameters() if p.requires_grad]
    bufs = [b.detach().clone() for b in t.visual_encoder.buffers()]
    return torch.stack(losses).cpu(), params, bufs


def test_prefetched_steps_equal_resident_steps_bitwise():
    la, pa, ba = _run(False)
    lb, pb, bb = _run(True)
    assert torch.equal(la, lb), (la, lb)
    assert all(torch.equal(x, y) for x, y in zip(pa, pb))
    assert all(torch.equal(x, y) for x, y in zip(ba, bb))


def _bench(*flags):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", "4", "--seconds", "1", "--steps", "3", "--warmup", "2",
                          "--no-probe", "--no-cpu-baseline", "--single-variant", *flags], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])


def test_bench_h2d_leg_reports_the_copy_inside_the_timed_region():
    d = _bench("--h2d")
    h = d["h2d"]
    assert d["metric"].endswith("at batch 4, full training step")
    assert h["overlapped"] is True and h["bytes_per_step"] > 4 * 2 * 25 * 96 * 96 * 4 and h["value"] > 0
    assert abs(h["final_loss"] - d["config"]["final_loss"]) < 5e-3 and h["exposed_ms"] == pytest.approx(h["ms_per_step"] - d["ms_per_step"], abs=2e-3)


def test_bench_force_dp_single_rank_rccl_leg():
    """One rank, real RCCL process group: every bucket after the first step is all-reduced IN PLACE (no packing copy), the exposed
    all-reduce time is reported, and the line carries the observed world size and the all-reduce checksum."""
    d = _bench("--force-dp")
    dp = d["config"]["data_parallel"]
    assert dp["rccl_ranks"] == 1 and dp["allreduce_checksum"] == dp["allreduce_checksum_expected"] == 1024.0
    assert dp["buckets_packed_by_copy"] == 0 and dp["buckets_reduced_in_place"] >= 3      # timed steps come after the layout-discovery step
    assert dp["exposed_allreduce_ms_per_step"] >= 0.0 and dp["gradient_bytes_per_step"] > 0
