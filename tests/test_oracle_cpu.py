"""CPU: the oracle (oracle/av_oracle.py) re-checked against the fixtures captured from the reference itself."""
import os

import numpy as np
import pytest
import torch

from conftest import pkg

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _run(name, cfg_name):
    from oracle import av_oracle as O
    init = pkg("utils.init"); synth = pkg("dataset.synthetic")
    fx = np.load(os.path.join(GOLD, name + ".npz"))
    cfg = getattr(init, cfg_name)
    batch = synth.make_batch(int(fx["batch"]), float(fx["seconds"]), seed=int(fx["seed_batch"]), ragged=bool(fx["ragged"]))
    sds = [init.visual_state_dict(), init.w2v2_state_dict(cfg), init.fusion_state_dict(512, cfg["hidden_size"], 512),
           init.decoder_state_dict(1024, 800)]
    proj = init.projection_params(cfg["hidden_size"])
    return O, fx, cfg, batch, sds, proj


def md(a, b):
    return float(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)).max())


@pytest.mark.parametrize("name,cfg_name", [("tiny", "W2V2_TINY"), ("tiny_ragged", "W2V2_TINY")])
def test_oracle_matches_reference_fixture(name, cfg_name):
    O, fx, cfg, batch, sds, proj = _run(name, cfg_name)
    with torch.no_grad():
        ev = O.forward_losses(*[{k: v.clone() for k, v in sd.items()} for sd in sds], cfg, batch, proj, training=False)
    assert md(ev["visual_feat1"], fx["eval_visual1"]) < 1e-4
    assert md(ev["audio_last"].numpy()[..., ::8], fx["eval_audio_last"]) < 1e-4
    assert md(ev["log_probs1"], fx["eval_log_probs1"]) < 1e-4 and md(ev["log_probs2"], fx["eval_log_probs2"]) < 1e-4
    assert np.array_equal(ev["input_lengths1"].numpy(), fx["eval_input_lengths1"])
    assert abs(float((ev["loss1"] + ev["loss2"]) / 2) - float(fx["eval_loss"])) < 1e-4
    dec = []
    for i in range(int(fx["batch"])):
        dec += [O.greedy_ctc(ev["log_probs1"][i], 3), O.greedy_ctc(ev["log_probs2"][i], 3)]
    assert dec == [[int(x) for x in s.split(",")] if s else [] for s in fx["eval_decoded"].tolist()]
    out, grads = O.train_step(*sds, cfg, batch, proj, {})
    assert md(out["log_probs1"], fx["train_log_probs1"]) < 1e-4
    for k in ("loss1", "loss2", "contrast1", "contrast2", "total"):
        assert abs(float(out[k]) - float(fx["train_" + k])) < 1e-4, k
    none = sorted(k for k, g in grads.items() if g is None)          # oracle tracks the trainable keys only
    fx_none = set(fx["none_grads"].tolist())                        # fixture lists every parameter without a gradient
    assert none == sorted(k for k in fx_none if k in grads) and all(k.startswith("fusion.cross_attn_visual.") for k in none)
    assert all(k in fx_none or g is not None for k, g in grads.items())
    for key in fx.files:
        if key.startswith("gradnorm/") and "k_proj.bias" not in key:
            assert abs(float(grads[key[9:]].norm()) - float(fx[key])) / (float(fx[key]) + 1e-12) < 1e-4, key
    for key in fx.files:
        if key.startswith("bn_after/"):
            assert md(sds[0][key[9:]], fx[key]) < 1e-5, key


def test_oracle_c1_eval_fixture():
    """Full-size wav2vec2-large eval forward of the oracle vs the reference capture (tests/golden/c1.npz)."""
    O, fx, cfg, batch, sds, proj = _run("c1", "W2V2_LARGE")
    with torch.no_grad():
        last, mid = O.audio_forward(sds[1], cfg, batch["audio"], batch["mask1"] != 3)
    assert md(last.numpy()[..., ::8], fx["eval_audio_last"]) < 1e-4
    assert md(mid.numpy()[..., ::8], fx["eval_audio_mid"]) < 1e-4


def test_pipeline_oracle_known_answers():
    """oracle/pipeline_oracle.py (load_pair's per-sample arithmetic; cv2 / librosa absent => pinned by known answers only)."""
    from oracle import pipeline_oracle as po
    rng = np.random.default_rng(3)
    img = rng.random((40, 56)).astype(np.float32)
    assert np.array_equal(po.resize_bilinear(img, 40, 56), img)                                   # identity: every tap lands on a pixel centre
    ramp = np.tile(np.arange(128, dtype=np.float32), (128, 1))
    want = (np.arange(96, dtype=np.float64) + 0.5) * (128 / 96) - 0.5                              # a linear image is reproduced exactly by bilinear taps
    np.testing.assert_allclose(po.resize_bilinear(ramp, 96, 96)[17], want, rtol=0, atol=2e-5)
    np.testing.assert_allclose(po.resize_bilinear(ramp.T.copy(), 96, 96)[:, 5], want, rtol=0, atol=2e-5)
    up = po.resize_bilinear(np.array([[0.0, 1.0]], dtype=np.float32), 1, 4)                        # pixel-centre law + clamped borders
    np.testing.assert_allclose(up[0], [0.0, 0.25, 0.75, 1.0], atol=1e-7)
    frames = np.full((2, 128, 128, 3), 255, dtype=np.uint8)
    out = po.lips(frames)
    assert out.shape == (2, 1, 96, 96) and out.dtype == np.float32 and np.all(out == 1.0)
    a1 = np.array([0.5, -1.0, 0.25, 0.5, 0.5], dtype=np.float32); a2 = np.array([0.5, -1.0, 0.25], dtype=np.float32)
    mixed, m1, m2 = po.mix_pair(a1, a2)
    assert m1.tolist() == [1, 1, 1, 2, 2] and m2.tolist() == [1, 1, 1, 0, 0] and m1.dtype == np.int64
    np.testing.assert_allclose(mixed, np.array([1.0, -2.0, 0.5, 0.5, 0.5]) / (2.0 + 1e-6), rtol=1e-6)
    mixed, m1, m2 = po.mix_pair(a2, a1)
    assert m1.tolist() == [1, 1, 1, 0, 0] and m2.tolist() == [1, 1, 1, 2, 2]
    mixed, m1, m2 = po.mix_pair(a2, a2)
    assert m1.tolist() == [1, 1, 1] == m2.tolist()


def test_legacy_oracle_vs_reference_fixture():
    """oracle/legacy_oracle.py against tests/golden/legacy_tiny.npz = outputs of the reference's legacy model."""
    from oracle import legacy_oracle as LO
    init = pkg("utils.init")
    fx = np.load(os.path.join(GOLD, "legacy_tiny.npz"))
    vocab, hidden, B, T, seed_w, seed_b = (int(x) for x in fx["cfg"])
    sd = init.legacy_state_dict(vocab, hidden, seed_w)
    batch = init.legacy_batch(B, T, vocab, seed_b)
    before = {k: v.clone() for k, v in sd.items()}
    out, grads = LO.train_step(sd, batch, {}, lr=1e-4)
    assert np.abs(out["logits_A"].numpy() - fx["logits_A"]).max() < 1e-4
    assert np.abs(out["logits_B"].numpy() - fx["logits_B"]).max() < 1e-4
    assert abs(float(out["loss"]) - float(fx["loss"])) < 1e-4
    for k, g in grads.items():
        flat = g.flatten()
        idx = torch.linspace(0, flat.numel() - 1, min(48, flat.numel())).long()
        gs = fx["gslice/" + k]
        assert np.abs(flat[idx].numpy() - gs).max() < 1e-3 * max(np.abs(gs).max(), 1e-6) + 1e-8, k
        assert abs(float(flat.norm()) - float(fx["gnorm/" + k])) < 1e-3 * float(fx["gnorm/" + k]) + 1e-8, k
