"""Whole training step on the HIP path vs the fixtures captured FROM THE REFERENCE (tests/golden/*.npz):
logits, CTC / contrastive losses, input_lengths, decoded ids, gradients, post-Adam deltas, BN running statistics."""
import os

import numpy as np
import pytest
import torch

from conftest import pkg

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def build(cfg, precision):
    init = pkg("utils.init"); enc = pkg("model.encoder"); fm = pkg("model.fusion_module"); dm = pkg("model.decoder")
    tr = pkg("model.trainer"); tok = pkg("utils.tokenizer")
    pkg("precision").set_precision(precision)
    ve = enc.VisualEncoder(); ve.load_state_dict(init.visual_state_dict())
    for p in ve.parameters():
        p.requires_grad = False                                        # main.py:100-103
    ae = enc.AudioEncoder(dict(cfg), freeze=True); ae.load_state_dict(init.w2v2_state_dict(cfg))
    for n, p in ae.model.named_parameters():                           # main.py:26-31
        p.requires_grad = any(f"encoder.layers.{i}." in n for i in range(6, 10))
    fu = fm.CrossAttentionFusion(512, cfg["hidden_size"], 512); fu.load_state_dict(init.fusion_state_dict(512, cfg["hidden_size"], 512))
    de = dm.CTCDecoder(1024, 800, 3); de.load_state_dict(init.decoder_state_dict(1024, 800))
    t = tr.MultimodalTrainer(ve, ae, fu, de, tok.SyntheticTokenizer(800), learning_rate=1e-4, device="cuda", lambda_=0.1)
    t.fixed_projection = init.projection_params(cfg["hidden_size"])
    return t


def maxdiff(a, b):
    return float(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)).max())


@pytest.mark.parametrize("name,cfg_name", [("tiny", "W2V2_TINY"), ("tiny_ragged", "W2V2_TINY"), ("c1", "W2V2_LARGE")])
def test_step_fp32_vs_reference_fixture(name, cfg_name):
    fx = np.load(os.path.join(GOLD, name + ".npz"))
    init = pkg("utils.init"); synth = pkg("dataset.synthetic")
    cfg = getattr(init, cfg_name)
    t = build(cfg, "fp32")
    batch = synth.make_batch(int(fx["batch"]), float(fx["seconds"]), seed=int(fx["seed_batch"]), ragged=bool(fx["ragged"]))
    tol = 1e-3                                                         # BASELINE north_star: 1e-3 fp32
    # ---- eval (BN running statistics) ----
    ev_loss, _ = t.evaluate([batch])
    t.visual_encoder.eval()
    with torch.no_grad():
        o = t.forward_losses(batch)
    assert maxdiff(o["visual_feat1"].cpu(), fx["eval_visual1"]) < tol * 10      # features are O(10); relative 1e-4
    assert maxdiff(o["audio_last"].cpu().numpy()[..., ::8], fx["eval_audio_last"]) < tol
    assert maxdiff(o["fused1"].cpu(), fx["eval_fused1"]) < tol
    assert maxdiff(o["log_probs1"].cpu(), fx["eval_log_probs1"]) < tol
    assert maxdiff(o["log_probs2"].cpu(), fx["eval_log_probs2"]) < tol
    assert np.array_equal(o["input_lengths1"].cpu().numpy(), fx["eval_input_lengths1"])
    assert np.array_equal(o["input_lengths2"].cpu().numpy(), fx["eval_input_lengths2"])
    assert abs(ev_loss - float(fx["eval_loss"])) < tol * 5
    hyp1, hyp2 = t.last_decoded
    bs = pkg("beam_search")
    dec = []
    for i in range(len(hyp1)):
        dec += [bs.simple_beam_search(o["log_probs1"][i], 5, 3), bs.simple_beam_search(o["log_probs2"][i], 5, 3)]
    want = [[int(x) for x in s.split(",")] if s else [] for s in fx["eval_decoded"].tolist()]
    assert dec == want, "decoded token ids differ from the reference's beam search"
    # ---- one training step ----
    before = {n: p.detach().clone() for m, mod in (("audio", t.audio_encoder), ("fusion", t.fusion_module), ("decoder", t.decoder1))
              for n, p in ((m + "." + k, v) for k, v in mod.named_parameters())}
    t.visual_encoder.train(); t.audio_encoder.train(); t.fusion_module.train(); t.decoder1.train()
    t.projection_layer = None
    out = t.train_step(batch)
    assert maxdiff(out["visual_feat1"].detach().cpu(), fx["train_visual1"]) < tol * 10
    assert maxdiff(out["visual_feat2"].detach().cpu(), fx["train_visual2"]) < tol * 10
    assert maxdiff(out["audio_last"].detach().cpu().numpy()[..., ::8], fx["train_audio_last"]) < tol
    assert maxdiff(out["audio_mid"].detach().cpu().numpy()[..., ::8], fx["train_audio_mid"]) < tol
    assert maxdiff(out["fused1"].detach().cpu(), fx["train_fused1"]) < tol
    assert maxdiff(out["log_probs1"].detach().cpu(), fx["train_log_probs1"]) < tol
    assert maxdiff(out["log_probs2"].detach().cpu(), fx["train_log_probs2"]) < tol
    for k in ("loss1", "loss2", "contrast1", "contrast2", "total"):
        assert abs(float(out[k]) - float(fx["train_" + k])) < tol * 5, k
    assert np.array_equal(out["input_lengths1"].cpu().numpy(), fx["train_input_lengths1"])
    # gradients: None-ness, norms and slices
    mods = {"audio": t.audio_encoder, "fusion": t.fusion_module, "decoder": t.decoder1}
    none = sorted(m + "." + k for m, mod in mods.items() for k, p in mod.named_parameters() if p.grad is None)
    assert none == fx["none_grads"].tolist()
    worst = 0.0
    for key in fx.files:
        if key.startswith("gradnorm/"):
            m, k = key[9:].split(".", 1)
            g = dict(mods[m].named_parameters())[k].grad
            ref = float(fx[key])
            if "k_proj.bias" in k:
                continue
            rel = abs(float(g.norm()) - ref) / (ref + 1e-12)
            worst = max(worst, rel)
            assert rel < 2e-3, (key, rel)
        if key.startswith("gradslice/"):
            m, k = key[10:].split(".", 1)
            g = dict(mods[m].named_parameters())[k].grad.reshape(-1)
            sl = g[:: max(1, g.numel() // 2048)][:2048].cpu().numpy()
            assert maxdiff(sl, fx[key]) < 2e-3 * max(1e-6, float(np.abs(fx[key]).max())) + 1e-7, key
    print("worst grad-norm rel err", worst)
    # post-Adam parameter deltas: |delta| <= lr and sign-exact where the gradient is not tiny
    for key in fx.files:
        if key.startswith("adamdelta/"):
            m, k = key[10:].split(".", 1)
            pnow = dict(mods[m].named_parameters())[k].detach()
            d = (pnow - before[m + "." + k]).reshape(-1)
            sl = d[:: max(1, d.numel() // 2048)][:2048].cpu().numpy()
            ref = fx[key]
            close = np.abs(sl - ref) < 2e-6
            assert close.mean() > 0.98, (key, close.mean())
    # BatchNorm running statistics after the step (two updates: lip1 call, lip2 call)
    vs = t.visual_encoder.state_dict()
    for key in fx.files:
        if key.startswith("bn_after/"):
            assert maxdiff(vs[key[9:]].cpu(), fx[key]) < 2e-4, key
    assert int(vs["frontend3D.1.num_batches_tracked"]) == int(fx["bn_num_batches_tracked"])


# the bf16 (benchmarked) mode is gated in tests/test_sizes_gpu.py: tiny + full-size c1 fixtures and the BASELINE batch sizes


def test_host_metadata_gives_the_same_step():
    """A device-resident batch that carries host_metadata() (CTC lengths / class counts from the host copy: no device read-back
    inside the step) must produce exactly the losses of the plain batch, and the host lengths must equal the device ones."""
    init = pkg("utils.init"); synth = pkg("dataset.synthetic"); w2 = pkg("model.w2v2")
    cfg = init.W2V2_TINY
    t = build(cfg, "fp32")
    cpu_batch = synth.make_batch(3, 1.0, seed=77, ragged=True)
    T_enc = int(w2.conv_out_lengths(cfg, cpu_batch["audio"].shape[1]))
    dev_batch = {k: v.cuda() for k, v in cpu_batch.items()}
    t.visual_encoder.train(); t.audio_encoder.train(); t.fusion_module.train(); t.decoder1.train()
    with torch.no_grad():
        plain = t.forward_losses(dict(dev_batch))
        md = t.host_metadata(cpu_batch, T_enc)
        fast = t.forward_losses({**dev_batch, **md})
    il = torch.cat([plain["input_lengths1"], plain["input_lengths2"]]).cpu()
    assert torch.equal(il, md["_ctc_input_lengths"])
    for k in ("loss1", "loss2", "contrast1", "contrast2", "total"):
        assert float(plain[k]) == float(fast[k]), k


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_checkpoint_resume_continues_identically(tmp_path, precision):
    """checkpoint.py (main.py:47-64 layout): a trainer that has diverged (other batches: parameters, Adam moments, BN running
    statistics, cached bf16 weight shadows all differ) and then loads the checkpoint must take the same next step as the trainer
    that wrote it."""
    init = pkg("utils.init"); synth = pkg("dataset.synthetic"); ck = pkg("checkpoint")
    cfg = init.W2V2_TINY
    b0 = synth.make_batch(2, 1.0, seed=5, ragged=False); b1 = synth.make_batch(2, 1.0, seed=6, ragged=True)
    other = synth.make_batch(2, 1.0, seed=9, ragged=False)
    a = build(cfg, precision)
    a.train_step(b0)
    path = str(tmp_path / "resume.pt")
    ck.save_checkpoint(3, a, path)
    b = build(cfg, precision)
    b.train_step(other); b.train_step(other)
    assert ck.load_checkpoint(b, path, audio_encoder=True, optimizer=True) == 4
    oa, ob = a.train_step(b1), b.train_step(b1)
    for k in ("loss1", "loss2", "contrast1", "contrast2", "total"):
        assert abs(float(oa[k].detach()) - float(ob[k].detach())) <= 1e-6 * max(1.0, abs(float(oa[k].detach()))), k
    for ma, mb in ((a.visual_encoder, b.visual_encoder), (a.audio_encoder, b.audio_encoder), (a.fusion_module, b.fusion_module), (a.decoder1, b.decoder1)):
        sa, sb = ma.state_dict(), mb.state_dict()
        for k in sa:
            # the key-projection bias has an exactly-zero gradient in exact arithmetic (softmax is shift-invariant per query): its
            # rounding-noise gradient depends on the atomics' order in the bf16 column sums and Adam turns its sign into +-lr
            tol = 6e-5 if k.endswith("k_proj.bias") else 2e-5
            assert maxdiff(sa[k].float().cpu(), sb[k].float().cpu()) < tol, k


@pytest.mark.parametrize("pair", [True, False])
def test_bucketed_gradient_path_equals_plain_step(pair):
    """The data-parallel plumbing at world size 1 (buckets packed into flat buffers, head bucket handed over when the wav2vec2
    backward starts, Adam reading the bucket views) must be arithmetically invisible: two steps with a reducer == two steps without."""
    init = pkg("utils.init"); synth = pkg("dataset.synthetic"); dp = pkg("parallel.dp"); tr = pkg("model.trainer")
    cfg = init.W2V2_TINY
    batch = synth.make_batch(2, 1.0, seed=5, ragged=True)
    a = build(cfg, "bf16")
    b = build(cfg, "bf16")
    red = dp.GradBucketReducer()
    assert red.world == 1
    b2 = tr.MultimodalTrainer(b.visual_encoder, b.audio_encoder, b.fusion_module, b.decoder1, b.tokenizer, learning_rate=1e-4, device="cuda",
                              lambda_=0.1, reducer=red, pair_batched=pair)
    b2.fixed_projection = b.fixed_projection
    a.pair_batched = pair
    for _ in range(2):
        oa, ob = a.train_step(batch), b2.train_step(batch)
        assert abs(float(oa["total"].detach()) - float(ob["total"].detach())) <= 1e-6 * max(1.0, abs(float(oa["total"].detach())))
    for ma, mb in ((a.audio_encoder, b2.audio_encoder), (a.fusion_module, b2.fusion_module), (a.decoder1, b2.decoder1)):
        sa, sb = ma.state_dict(), mb.state_dict()
        for k in sa:
            tol = 6e-5 if k.endswith("k_proj.bias") else 2e-5
            assert maxdiff(sa[k].float().cpu(), sb[k].float().cpu()) < tol, k
    assert (b2.audio_encoder.model.grad_pre is not None) == pair
