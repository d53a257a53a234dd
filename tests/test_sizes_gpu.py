"""The benchmarked mode (bf16 MFMA operands, fp32 accumulation / residual stream / master weights) at the BASELINE.json sizes.

* full training step (visual + audio + fusion + decoder forward, backward, losses) at configs[1] (B=32 x 4 s), configs[3]'s
  per-GPU batch (B=64) and configs[4]'s (B=128) on one GPU, checked through size-independent properties:
  bit-exact batch-permutation equivariance of the log-probs, equality of the pair-batched path (one fusion / decoder / CTC call
  for both speakers) with the reference's two-call form, gradients present for exactly the complement of the reference's
  ``none_grads`` list and finite, and agreement of the bf16 step with the fp32 (exact-fp32 MFMA) step of the same kernels;
* a bf16 run of the full-size reference fixture ``c1.npz`` with tight gates (log-probs, losses, gradient direction);
* the persistent BiLSTM kernels against the fp32 CPU oracle at config 3's T = 375 and at 256 rows (configs[4]: 2 x 128).
"""
import os

import numpy as np
import pytest
import torch

from conftest import pkg
from test_step_gpu import GOLD, build, maxdiff

pytestmark = pytest.mark.gpu


def _cos(a, b):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return float((a @ b) / (a.norm() * b.norm() + 1e-300))


def _grads(t):
    mods = {"audio": t.audio_encoder, "fusion": t.fusion_module, "decoder": t.decoder1}
    return {m + "." + k: p.grad for m, mod in mods.items() for k, p in mod.named_parameters()}


def _fwd_bwd(t, batch):
    t.optimizer.zero_grad(set_to_none=True)
    out = t.forward_losses(batch)
    out["total"].backward()
    torch.cuda.synchronize()
    return out, {k: (None if g is None else g.detach().clone()) for k, g in _grads(t).items()}


@pytest.mark.parametrize("B,seconds,ragged", [(32, 4.0, True), (64, 4.0, False), (128, 4.0, True), (8, 15.0, True)])
def test_full_step_properties_at_config_sizes(B, seconds, ragged):
    """configs[1] (32 x 4 s), configs[3] / configs[4] per-GPU batches (64, 128 x 4 s) and configs[2] (8 x 15 s long form: T_enc 749 -> the
    T > 256 attention kernels, 375 lip frames -> 375 BiLSTM steps and the unfused cross-attention path)."""
    init = pkg("utils.init"); synth = pkg("dataset.synthetic"); w2 = pkg("model.w2v2")
    cfg = init.W2V2_LARGE
    fx = np.load(os.path.join(GOLD, "c1.npz"))
    cpu_batch = synth.make_batch(B, seconds, seed=11, ragged=ragged)
    n_audio = int(round(16000 * seconds))
    assert cpu_batch["audio"].shape == (B, n_audio) and cpu_batch["lip1"].shape[1] == int(round(25 * seconds))
    T_enc = int(w2.conv_out_lengths(cfg, n_audio))
    t = build(cfg, "bf16")
    batch = {k: v.cuda() for k, v in cpu_batch.items()}
    batch.update(t.host_metadata(cpu_batch, T_enc))

    # (1) eval-mode forward: every kernel on the path is item-wise, so permuting the batch permutes the log-probs bit for bit
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(5))
    pb = {k: v[perm].contiguous() for k, v in cpu_batch.items()}
    pbatch = {k: v.cuda() for k, v in pb.items()}
    pbatch.update(t.host_metadata(pb, T_enc))
    for m in (t.visual_encoder, t.audio_encoder, t.fusion_module, t.decoder1):
        m.eval()
    lam, t.lambda_ = t.lambda_, 0.0
    with torch.no_grad():
        o = t.forward_losses(batch); op = t.forward_losses(pbatch)
    t.lambda_ = lam
    pc = perm.cuda()
    for k in ("log_probs1", "log_probs2", "audio_last", "visual_feat1"):
        assert torch.equal(o[k][pc], op[k]), k
    assert torch.equal(o["input_lengths1"][pc], op["input_lengths1"]) and torch.equal(o["input_lengths2"][pc], op["input_lengths2"])

    # (2) train-mode forward + backward, pair-batched (default) vs one fusion / decoder / CTC call per speaker (the reference's form)
    for m in (t.visual_encoder, t.audio_encoder, t.fusion_module, t.decoder1):
        m.train()
    out_a, g_a = _fwd_bwd(t, batch)
    t.pair_batched = False
    out_b, g_b = _fwd_bwd(t, batch)
    t.pair_batched = True
    for k in ("loss1", "loss2", "contrast1", "contrast2", "total"):
        a, b = float(out_a[k]), float(out_b[k])
        assert np.isfinite(a) and abs(a - b) <= 2e-5 * max(1.0, abs(a)), (k, a, b)
    assert maxdiff(out_a["log_probs1"].detach().cpu(), out_b["log_probs1"].detach().cpu()) < 1e-5
    # (3) gradients: None exactly where the reference has none (frozen layers, the unused cross_attn_visual), finite elsewhere
    none = sorted(k for k, g in g_a.items() if g is None)
    assert none == fx["none_grads"].tolist()
    for k, g in g_a.items():
        if g is None:
            continue
        assert bool(torch.isfinite(g).all()), k
        if "k_proj.bias" in k:
            continue
        assert _cos(g, g_b[k]) > 0.9999, (k, _cos(g, g_b[k]))
    lp16 = out_a["log_probs1"].detach().float().cpu(); tot16 = float(out_a["total"])
    l16 = {k: float(out_a[k]) for k in ("loss1", "loss2", "contrast1", "contrast2")}
    keep = {k: g.float().cpu() for k, g in g_a.items() if g is not None and any(s in k for s in
            ("decoder.net.0.weight", "temporal_model.weight_hh_l1", "cross_attn_audio.in_proj_weight", "layers.9.feed_forward.output_dense.weight",
             "layers.6.attention.q_proj.weight"))}
    del t, out_a, out_b, g_a, g_b, o, op
    torch.cuda.empty_cache()

    # (4) the same step with exact-fp32 MFMA kernels (the parity mode, itself pinned against the reference at C1): the benchmarked
    # bf16 mode must agree with it at THIS size
    t32 = build(cfg, "fp32")
    for m in (t32.visual_encoder, t32.audio_encoder, t32.fusion_module, t32.decoder1):
        m.train()
    out32, g32 = _fwd_bwd(t32, batch)
    e_lp = maxdiff(lp16, out32["log_probs1"].detach().cpu())
    e_tot = abs(tot16 - float(out32["total"]))
    print(f"B={B}: bf16 vs fp32 step  max|dlogp|={e_lp:.4f}  |dtotal|={e_tot:.5f} (total {float(out32['total']):.3f})  "
          + "  ".join(f"{k}:{abs(l16[k] - float(out32[k])):.5f}" for k in l16))
    assert e_lp <= 0.05 and e_tot <= 0.05
    for k in l16:
        assert abs(l16[k] - float(out32[k])) <= 0.05, k
    for k, g in keep.items():
        c = _cos(g, g32[k].float().cpu())
        print(f"   grad cosine {k}: {c:.5f}")
        assert c >= 0.99, (k, c)
    pkg("precision").set_precision("bf16")


def test_c1_fixture_bf16_tight_gates():
    """Full-size wav2vec2-large step of the reference fixture (B=2 x 1 s) in the benchmarked bf16 mode."""
    fx = np.load(os.path.join(GOLD, "c1.npz"))
    init = pkg("utils.init"); synth = pkg("dataset.synthetic")
    t = build(init.W2V2_LARGE, "bf16")
    batch = synth.make_batch(int(fx["batch"]), float(fx["seconds"]), seed=int(fx["seed_batch"]), ragged=bool(fx["ragged"]))
    t.visual_encoder.train(); t.audio_encoder.train(); t.fusion_module.train(); t.decoder1.train()
    t.projection_layer = None
    out = t.train_step(batch)
    e1 = maxdiff(out["log_probs1"].detach().cpu(), fx["train_log_probs1"]); e2 = maxdiff(out["log_probs2"].detach().cpu(), fx["train_log_probs2"])
    print(f"c1 bf16: max|dlogp| = {e1:.4f} / {e2:.4f}")
    assert e1 <= 0.05 and e2 <= 0.05
    for k in ("loss1", "loss2", "contrast1", "contrast2", "total"):
        d = abs(float(out[k]) - float(fx["train_" + k]))
        print(f"   {k}: |d| = {d:.5f} of {float(fx['train_' + k]):.4f}")
        assert d <= 0.05, (k, d)
    assert np.array_equal(out["input_lengths1"].cpu().numpy(), fx["train_input_lengths1"])
    mods = {"audio": t.audio_encoder, "fusion": t.fusion_module, "decoder": t.decoder1}
    worst = 1.0
    for key in fx.files:
        if key.startswith("gradslice/"):
            m, k = key[10:].split(".", 1)
            if "k_proj.bias" in k:
                continue
            g = dict(mods[m].named_parameters())[k].grad.reshape(-1)
            sl = g[:: max(1, g.numel() // 2048)][:2048].float().cpu()
            c = _cos(sl, torch.from_numpy(fx[key]))
            worst = min(worst, c)
            assert c >= 0.99, (key, c)
        if key.startswith("gradnorm/"):
            m, k = key[9:].split(".", 1)
            if "k_proj.bias" in k:
                continue
            g = dict(mods[m].named_parameters())[k].grad
            ref = float(fx[key])
            assert abs(float(g.norm()) - ref) <= 0.03 * ref + 1e-9, (key, float(g.norm()), ref)
    print("c1 bf16: worst gradient-slice cosine", worst)


def test_tiny_fixture_bf16_tight_gates():
    fx = np.load(os.path.join(GOLD, "tiny.npz"))
    init = pkg("utils.init"); synth = pkg("dataset.synthetic")
    t = build(init.W2V2_TINY, "bf16")
    batch = synth.make_batch(int(fx["batch"]), float(fx["seconds"]), seed=int(fx["seed_batch"]), ragged=False)
    t.visual_encoder.train()
    out = t.train_step(batch)
    e_lp = maxdiff(out["log_probs1"].detach().cpu(), fx["train_log_probs1"])
    e_loss = abs(float(out["total"]) - float(fx["train_total"]))
    print(f"tiny bf16 step: max|dlogp| = {e_lp:.4f}, |dloss| = {e_loss:.4f}")
    assert e_lp <= 0.05 and e_loss <= 0.05


@pytest.mark.parametrize("T,B", [(375, 16), (100, 256), (30, 3)])
def test_persistent_bilstm_vs_fp32_oracle(T, B):
    """lstm_persistent.hip through the module path (2 layers, both directions, forward + backward) against the CPU oracle's fp32
    BiLSTM and its autograd gradients at config 3's sequence length (375 steps; 2 x 8 rows) and at configs[4]'s row count (2 x 128)."""
    from oracle import av_oracle as O
    init = pkg("utils.init"); fm = pkg("model.fusion_module")
    pkg("precision").set_precision("bf16")
    assert fm.PERSISTENT_LSTM
    sd = init.fusion_state_dict(512, 1024, 512)
    mod = fm.CrossAttentionFusion(512, 1024, 512).cuda(); mod.load_state_dict(sd)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, T, 512, generator=g) * 0.7
    w = torch.randn(B, T, 1024, generator=g)
    xr = x.clone().requires_grad_(True)
    names = [k for k in sd if k.startswith("temporal_model.")]
    leaves = {k: sd[k].clone().requires_grad_(True) for k in names}
    ref = O.bilstm2({**sd, **leaves}, "temporal_model.", xr)
    gr = torch.autograd.grad((ref * w).sum(), [xr] + [leaves[k] for k in names])
    x_tm = x.transpose(0, 1).contiguous().cuda().to(torch.bfloat16)
    out_bt, lctx = fm.lstm_forward(mod, x_tm, True)
    grads = {}
    dx_tm = fm.lstm_backward(mod, lctx, w.cuda().contiguous(), grads)
    torch.cuda.synchronize()
    for c in mod._lstm_flags:
        assert int(c[2]) == 0, "persistent LSTM: inter-workgroup wait timed out"
    e = maxdiff(out_bt.float().cpu(), ref.detach())
    print(f"T={T} B={B}: BiLSTM out max|d| = {e:.4f}")
    assert e <= 0.03                                            # outputs in (-1, 1); bf16 operands through 2 layers x T steps
    c = _cos(dx_tm.float().cpu().transpose(0, 1), gr[0])
    assert c >= 0.995, ("dx", c)
    for k, gref in zip(names, gr[1:]):
        got = grads[k].float().cpu()
        c = _cos(got, gref)
        rn = abs(float(got.norm()) - float(gref.norm())) / float(gref.norm())
        assert c >= 0.995 and rn <= 0.03, (k, c, rn)
