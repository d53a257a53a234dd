"""Data-parallel training step, world size 2 (two processes sharing the one GPU of the box, gloo transport: the collective
sequence, bucket packing, head-bucket hand-over and the 1/world folded into Adam are the same code RCCL runs).

SURVEY §8e: the all-reduced gradient must equal the MEAN of the per-shard gradients (per-replica BatchNorm statistics and
per-replica contrastive loss), and the post-Adam parameters a single-process Adam step on that mean.  The expected values come
from the CPU oracle run once per shard."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import PKG, ROOT, pkg

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, passes, regularize, outdir):
    import importlib
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ["AVAMD_LSTM_PERSISTENT"] = "0"          # two processes on one GPU: persistent kernels of both cannot all be resident
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    imp = lambda s: importlib.import_module(PKG + "." + s)
    init = imp("utils.init"); synth = imp("dataset.synthetic"); dp = imp("parallel.dp"); tr = imp("model.trainer")
    from test_step_gpu import build
    cfg = dict(init.W2V2_TINY)
    if regularize:
        cfg.update(hidden_dropout=0.1, attention_dropout=0.0, activation_dropout=0.1, layerdrop=0.5, mask_time_prob=0.05,
                   mask_time_length=2, mask_time_min_masks=1)
    b = build(cfg, "bf16" if regularize else "fp32")
    red = dp.GradBucketReducer()
    assert red.world == world
    t = tr.MultimodalTrainer(b.visual_encoder, b.audio_encoder, b.fusion_module, b.decoder1, b.tokenizer, learning_rate=1e-4,
                             device="cuda", lambda_=0.1, reducer=red, audio_passes=passes)
    t.fixed_projection = b.fixed_projection
    t.visual_encoder.train(); t.audio_encoder.train(); t.fusion_module.train(); t.decoder1.train()
    full = synth.make_batch(2 * world, 1.0, seed=21, ragged=False)
    shard = dp.shard_batch(full, rank, world)
    torch.manual_seed(100 + rank)                      # ranks deliberately disagree on the global host RNG
    np.random.seed(100 + rank)
    steps = 15 if regularize else 1
    for _ in range(steps):
        out = t.train_step(shard)
    torch.cuda.synchronize()
    flat_n, cat_n = red.flat_reduces, red.cat_reduces
    mods = {"audio": t.audio_encoder, "fusion": t.fusion_module, "decoder": t.decoder1}
    params = {m + "." + k: p.detach().cpu() for m, mod in mods.items() for k, p in mod.named_parameters()}
    grads = {m + "." + k: (p.grad.detach().float().cpu() / world) for m, mod in mods.items() for k, p in mod.named_parameters()
             if p.grad is not None}
    torch.save(dict(params=params, grads=grads, loss=float(out["total"]), flat_n=flat_n, cat_n=cat_n, steps=steps,
                    ld_state=t.audio_encoder.model.layerdrop_generator.get_state(),
                    dr_state=t.audio_encoder.model.dropout_generator.get_state()), os.path.join(outdir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _run(world, passes, regularize, outdir):
    ctx = mp.get_context("spawn")
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, passes, regularize, outdir)) for r in range(world)]
    for p in ps:
        p.start()
    for p in ps:
        p.join(600)
        assert p.exitcode == 0, p.exitcode
    return [torch.load(os.path.join(outdir, f"rank{r}.pt"), weights_only=True) for r in range(world)]


@pytest.mark.parametrize("passes", [1, 2])
def test_two_rank_step_equals_mean_of_oracle_shard_gradients(tmp_path, passes):
    from oracle import av_oracle as O
    init = pkg("utils.init"); synth = pkg("dataset.synthetic"); dp = pkg("parallel.dp")
    world = 2
    got = _run(world, passes, False, str(tmp_path))
    cfg = init.W2V2_TINY
    full = synth.make_batch(2 * world, 1.0, seed=21, ragged=False)
    proj = init.projection_params(cfg["hidden_size"])
    shard_grads = []
    for r in range(world):
        sds = [init.visual_state_dict(), init.w2v2_state_dict(cfg), init.fusion_state_dict(512, cfg["hidden_size"], 512),
               init.decoder_state_dict(1024, 800)]
        # oracle state dicts carry the reference's "model." prefix on the audio keys; grads come back as audio./fusion./decoder.
        _, g = O.train_step(*sds, cfg, dp.shard_batch(full, r, world), proj, {}, dedup_audio=(passes == 1))
        shard_grads.append(g)
    mean = {k: (sum(g[k] for g in shard_grads) / world) for k in shard_grads[0] if shard_grads[0][k] is not None}
    # expected post-Adam parameters: one Adam step on the mean gradient from the common initial weights
    sds = [init.visual_state_dict(), init.w2v2_state_dict(cfg), init.fusion_state_dict(512, cfg["hidden_size"], 512),
           init.decoder_state_dict(1024, 800)]
    tk = O.trainable_keys(sds[1], sds[2], sds[3])
    st = {}
    for name, sd, lr in (("audio", sds[1], 2e-5), ("fusion", sds[2], 1e-4), ("decoder", sds[3], 1e-4)):
        O.adam_step({f"{name}.{k}": sd[k] for k in tk[name]}, mean, st, lr)
    want_params = {f"{name}.{k}": sd[k] for name, sd in (("audio", sds[1]), ("fusion", sds[2]), ("decoder", sds[3])) for k in tk[name]}
    r0, r1 = got
    # both ranks hold identical reduced gradients and parameters
    for k in r0["grads"]:
        assert torch.equal(r0["grads"][k], r1["grads"][k]), k
    for k in r0["params"]:
        assert torch.equal(r0["params"][k], r1["params"][k]), k
    assert sorted(k for k in mean) == sorted(r0["grads"]), "set of parameters with a gradient differs from the oracle's"
    worst = 0.0
    for k, g in mean.items():
        if "k_proj.bias" in k:                  # exactly-zero gradient in exact arithmetic: rounding noise only
            continue
        d = float((r0["grads"][k] - g).abs().max()); sc = float(g.abs().max())
        worst = max(worst, d / (sc + 1e-12))
        assert d <= 2e-3 * sc + 1e-7, (k, d, sc)
    print("2-rank reduced gradient vs mean of oracle shard gradients: worst rel", worst)
    for k, w in want_params.items():
        if "k_proj.bias" in k or k not in mean:
            continue
        lr = 2e-5 if k.startswith("audio.") else 1e-4
        d = (r0["params"][k] - w).abs()
        # first Adam step moves by lr * g / (|g| + eps): entries whose gradient is far above rounding noise must agree tightly
        big = mean[k].abs() > 1e-3 * mean[k].abs().max()
        assert float(d[big].max() if big.any() else 0.0) < 0.02 * lr, (k, float(d.max()), lr)


def test_two_rank_regularized_two_pass_steps_stay_in_lockstep(tmp_path):
    """HF-style regularisers on (LayerDrop 0.5 so that trainable layers ARE dropped), two audio passes, ranks with different global
    host RNG states: the LayerDrop schedule must be identical on both ranks (or the per-layer buckets would not pair up: hang or
    mismatched sizes), dropout seeds must differ, and parameters must stay bitwise equal across ranks after 15 steps.  From the second
    step on every bucket is all-reduced in place in its flat gradient buffer (parallel/dp.py: GradArena): no packing copy."""
    r0, r1 = _run(2, 2, True, str(tmp_path))
    for r in (r0, r1):
        # the first step discovers the layouts (packed buckets); a later step packs only when a wav2vec2 layer's bucket is seen for the
        # first time (LayerDrop 0.5 can hide a trainable layer from the first steps): at most 1 head + 4 layer buckets in total
        assert r["cat_n"] <= 5 and r["flat_n"] >= r["steps"] - 1, (r["cat_n"], r["flat_n"])
    assert torch.equal(r0["ld_state"], r1["ld_state"])
    assert not torch.equal(r0["dr_state"], r1["dr_state"])
    for k in r0["params"]:
        assert torch.equal(r0["params"][k], r1["params"][k]), k
    assert np.isfinite(r0["loss"]) and np.isfinite(r1["loss"])


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` as typed (the driver's scaling runs): it must start two fresh ranks itself, run both
    variants under the gradient reducer and relay rank 0's ONE JSON line.  Two ranks share this box's one GPU over gloo (the RCCL
    transport needs one GPU per rank); small batch, 1 s clips."""
    import json
    import subprocess
    env = dict(os.environ, MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--same-device", "--steps", "2",
                        "--warmup", "1", "--batch", "4", "--seconds", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600,
                       env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["config"]["global_batch"] == 8 and d["config"]["parallelism"] == "dp2" and d["config"]["audio_passes"] == 2
    assert d["value"] > 0 and np.isfinite(d["config"]["final_loss"]) and d["other_variant"]["audio_passes"] == 1
