"""GPU parity of the HIP wav2vec2 path against the CPU oracle (fp32 mode: 1e-3 gate; bf16: reported tolerance)."""
import numpy as np
import pytest
import torch

from conftest import pkg

pytestmark = pytest.mark.gpu


def _setup(cfg_name, B, seconds, ragged, precision):
    init = pkg("utils.init"); synth = pkg("dataset.synthetic"); enc = pkg("model.encoder"); prec = pkg("precision")
    from oracle import av_oracle as O
    cfg = getattr(init, cfg_name)
    prec.set_precision(precision)
    batch = synth.make_batch(B, seconds, seed=42, ragged=ragged)
    sd = init.w2v2_state_dict(cfg)
    ae = enc.AudioEncoder(dict(cfg), freeze=True).cuda()
    ae.load_state_dict(sd)
    return cfg, batch, sd, ae, O


@pytest.mark.parametrize("precision,tol", [("fp32", 1e-3), ("bf16", 0.15)])
@pytest.mark.parametrize("cfg_name,ragged", [("W2V2_TINY", False), ("W2V2_TINY", True)])
def test_audio_forward_vs_oracle(cfg_name, ragged, precision, tol):
    cfg, batch, sd, ae, O = _setup(cfg_name, 3, 1.2, ragged, precision)
    mask = batch["mask1"] != 3
    with torch.no_grad():
        ref_last, ref_mid = O.audio_forward({k: v.clone() for k, v in sd.items()}, cfg, batch["audio"], mask)
        last, mid = ae(batch["audio"].cuda(), attention_mask=mask.cuda())
    assert last.shape == ref_last.shape
    assert float((last.cpu() - ref_last).abs().max()) < tol
    assert float((mid.cpu() - ref_mid).abs().max()) < tol


@pytest.mark.parametrize("precision,tol", [("fp32", 2e-3), ("bf16", 0.1)])
def test_audio_backward_vs_oracle(precision, tol):
    cfg, batch, sd, ae, O = _setup("W2V2_TINY", 3, 1.2, True, precision)
    for n, p in ae.model.named_parameters():
        p.requires_grad = any(f"encoder.layers.{i}." in n for i in range(6, 10))
    mask = batch["mask1"] != 3
    g = torch.Generator().manual_seed(7)
    osd = {k: v.clone() for k, v in sd.items()}
    tk = [k for k in osd if any(f"encoder.layers.{i}." in k for i in range(6, 10))]
    for k in tk:
        osd[k].requires_grad_(True)
    ref_last, ref_mid = O.audio_forward(osd, cfg, batch["audio"], mask)
    wl = torch.randn(ref_last.shape, generator=g); wm = torch.randn(ref_mid.shape, generator=g)
    ((ref_last * wl).sum() + (ref_mid * wm).sum()).backward()
    last, mid = ae(batch["audio"].cuda(), attention_mask=mask.cuda())
    ((last * wl.cuda()).sum() + (mid * wm.cuda()).sum()).backward()
    worst = 0.0
    for n, p in ae.named_parameters():
        if p.requires_grad:
            ref = osd[n].grad
            if "k_proj.bias" in n:      # mathematically zero gradient
                # only rounding noise may remain: sum over keys of bf16-rounded dK (fp32 mode: ~1e-6)
                assert float(p.grad.abs().max()) < (1e-4 if precision == "fp32" else 5e-2) * max(1.0, float(wl.abs().max()))
                continue
            rel = float((p.grad.cpu() - ref).norm() / (ref.norm() + 1e-12))
            worst = max(worst, rel)
            assert rel < tol, (n, rel)
        else:
            assert p.grad is None
    print("worst rel grad error", worst)


@pytest.mark.parametrize("precision,tol", [("fp32", 1e-3)])
def test_audio_c1_golden(precision, tol):
    """Full-size wav2vec2-large against the fixture captured from the reference itself (tests/golden/c1.npz)."""
    import os
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "c1.npz"))
    cfg, batch, sd, ae, O = _setup("W2V2_LARGE", 2, 1.0, False, precision)
    ae.eval()
    with torch.no_grad():
        last, mid = ae(batch["audio"].cuda(), attention_mask=(batch["mask1"] != 3).cuda())
    assert float(np.abs(last.cpu().numpy()[..., ::8] - fx["eval_audio_last"]).max()) < tol
    assert float(np.abs(mid.cpu().numpy()[..., ::8] - fx["eval_audio_mid"]).max()) < tol


def test_two_passes_share_the_conv_feature_extractor():
    """The trainer opens a window (model._feat_cache) around the reference's two audio passes over the SAME waveform: the frozen,
    deterministic conv feature extractor runs once, results are unchanged; outside the window nothing is cached."""
    init = pkg("utils.init"); enc = pkg("model.encoder"); synth = pkg("dataset.synthetic")
    pkg("precision").set_precision("bf16")
    cfg = init.W2V2_TINY
    ae = enc.AudioEncoder(dict(cfg), freeze=True).cuda(); ae.load_state_dict(init.w2v2_state_dict(cfg)); ae.eval()
    batch = synth.make_batch(3, 1.0, seed=8, ragged=True)
    wav, m1, m2 = batch["audio"].cuda(), (batch["mask1"] != 3).cuda(), (batch["mask2"] != 3).cuda()
    calls = []
    orig = ae.model.features
    ae.model.features = lambda w, dt: (calls.append(1), orig(w, dt))[1]
    with torch.no_grad():
        ref1, _ = ae(wav, m1); ref2, _ = ae(wav, m2)
        assert len(calls) == 2
        ae.model._feat_cache = {}
        try:
            a1, _ = ae(wav, m1); a2, _ = ae(wav, m2)
        finally:
            ae.model._feat_cache = None
        assert len(calls) == 3                                     # one extractor run for the two passes inside the window
        a3, _ = ae(wav, m1)
        assert len(calls) == 4
    assert torch.equal(a1, ref1) and torch.equal(a2, ref2) and torch.equal(a3, ref1)


@pytest.mark.parametrize("precision,tol", [("fp32", 2e-5), ("bf16", 2e-2)])
def test_forward_pair_equals_two_calls(precision, tol):
    """AudioEncoder.forward_pair (the reference's two audio passes as one autograd node, backward interleaved per layer with in-place
    accumulation of the weight gradients) against two separate calls whose gradients autograd sums: same outputs bit for bit, same
    parameter gradients up to the summation order."""
    init = pkg("utils.init"); enc = pkg("model.encoder"); synth = pkg("dataset.synthetic")
    pkg("precision").set_precision(precision)
    cfg = init.W2V2_TINY
    batch = synth.make_batch(3, 1.0, seed=8, ragged=True)
    wav = batch["audio"].cuda()
    m1 = (batch["mask1"] != 3).cuda()
    m2 = m1.clone(); m2[0, 9000:] = False                          # a different padding pattern for the second pass
    g = torch.Generator().manual_seed(4)
    res = []
    for pair in (True, False):
        ae = enc.AudioEncoder(dict(cfg), freeze=True).cuda(); ae.load_state_dict(init.w2v2_state_dict(cfg)); ae.train()
        for n, p in ae.model.named_parameters():
            p.requires_grad = any(f"encoder.layers.{i}." in n for i in range(6, 10))
        if pair:
            a1, mid1, a2, mid2 = ae.forward_pair(wav, m1, m2)
        else:
            a1, mid1 = ae(wav, m1); a2, mid2 = ae(wav, m2)
        if not res:
            ws = [torch.randn(a1.shape, generator=g).cuda() for _ in range(4)]
        ((a1 * ws[0]).sum() + (mid1 * ws[1]).sum() + (a2 * ws[2]).sum() + (mid2 * ws[3]).sum()).backward()
        res.append(((a1, mid1, a2, mid2), {n: p.grad.clone() for n, p in ae.model.named_parameters() if p.grad is not None}))
    (o_p, g_p), (o_s, g_s) = res
    for x, y in zip(o_p, o_s):
        assert torch.equal(x, y)
    assert sorted(g_p) == sorted(g_s) and len(g_p) == 4 * 16
    for n in g_p:
        if "k_proj.bias" in n:
            continue
        d = float((g_p[n] - g_s[n]).abs().max()); sc = float(g_s[n].abs().max())
        assert d <= tol * sc + 1e-7, (n, d, sc)
