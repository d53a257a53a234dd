"""Long regularised runs of the step AS THE REFERENCE EXECUTES IT (model/trainer.py:62-125: two wav2vec2 passes, HF-default dropout /
LayerDrop / SpecAugment) and the optimizer laws they rest on.

Round 2's headline run diverged to NaN between step 14 and 25 (profiles/r03_nan_hunt_before_fix.txt): the first step in which
LayerDrop left a trainable layer without gradients demoted the optimizer to per-tensor launches that wrote no bf16 shadows, so every
later forward re-cast the trainable weights on the main stream while the second audio pass read them on its own stream without
ordering.  These tests pin the three laws that remove it:
  * AvAdam keeps torch.optim.Adam's PER-PARAMETER step counts inside the one fused launch (with and without loss scaling);
  * every step of a long run stays on the fused launch, shadows stay coherent with the fp32 master weights;
  * 40 as-executed steps at the BASELINE batch (64 x 4 s) and on the tiny config with LayerDrop 0.5 keep loss and weights finite,
    including steps in which a trainable layer is dropped in both passes, with and without loss scaling.
"""
import numpy as np
import pytest
import torch

from conftest import pkg
from test_step_gpu import build

pytestmark = pytest.mark.gpu

HF_REG = dict(hidden_dropout=0.1, attention_dropout=0.1, activation_dropout=0.1, feat_proj_dropout=0.0, layerdrop=0.1,
              mask_time_prob=0.05, mask_time_length=10, mask_time_min_masks=2)


def test_avadam_per_tensor_step_counts_match_torch():
    """Parameters that miss a gradient in some steps (LayerDrop) keep their own step count - and bias corrections - as
    torch.optim.Adam does (state[p]['step'] per tensor); every step is ONE fused launch."""
    optim = pkg("optim")
    g = torch.Generator().manual_seed(21)
    shapes = [(300, 64), (70001,), (5,), (64, 64), (9, 7)]
    ref = [torch.randn(*s, generator=g).requires_grad_(True) for s in shapes]
    mine = [r.detach().clone().cuda().requires_grad_(True) for r in ref]
    o_ref = torch.optim.Adam([{"params": ref[:2], "lr": 1e-3}, {"params": ref[2:], "lr": 2e-4, "betas": (0.8, 0.99), "eps": 1e-6}])
    o_mine = optim.AvAdam([{"params": mine[:2], "lr": 1e-3}, {"params": mine[2:], "lr": 2e-4, "betas": (0.8, 0.99), "eps": 1e-6}])
    missing = {1: [0, 3], 2: [3], 4: [1, 2, 4], 6: [0]}
    for step in range(8):
        for i, (r, m) in enumerate(zip(ref, mine)):
            if i in missing.get(step, []):
                r.grad = None; m.grad = None
            else:
                gr = torch.randn(r.shape, generator=g)
                r.grad = gr.clone(); m.grad = gr.cuda()
        o_ref.step(); o_mine.step()
        for r, m in zip(ref, mine):
            torch.testing.assert_close(m.detach().cpu(), r.detach(), rtol=2e-6, atol=2e-7)
    assert o_mine.fused_launches == 8
    for r, m in zip(ref, mine):
        assert int(o_ref.state[r]["step"]) == o_mine.state[m]["step"]
    assert sorted({st["step"] for st in o_mine.state.values()}) == [6, 7]
    # the device-side table agrees with the host mirror
    host = dict((id(p), st["step"]) for p, st in o_mine.state.items())
    o_mine.sync_steps()
    assert all(o_mine.state[p]["step"] == host[id(p)] for p in mine)
    # resume: a fresh optimizer loaded from the state dict continues identically
    o2 = optim.AvAdam([{"params": mine[:2], "lr": 1e-3}, {"params": mine[2:], "lr": 2e-4, "betas": (0.8, 0.99), "eps": 1e-6}])
    o2.load_state_dict(o_mine.state_dict())
    for r, m in zip(ref, mine):
        gr = torch.randn(r.shape, generator=g)
        r.grad = gr.clone(); m.grad = gr.cuda()
    o_ref.step(); o2.step()
    for r, m in zip(ref, mine):
        torch.testing.assert_close(m.detach().cpu(), r.detach(), rtol=2e-6, atol=2e-7)


def test_grad_scaler_with_gradientless_parameters_matches_torch_amp():
    """Loss scaling + parameters without gradients + injected overflows, against torch.amp.GradScaler + torch.optim.Adam: skipped steps
    advance no counter, a parameter without a gradient keeps its own count, the scale trajectory is torch's."""
    optim = pkg("optim")
    torch.manual_seed(4)
    shapes = [(64, 33), (1000,), (7, 5, 3), (130,)]
    p_ref = [torch.randn(s, device="cuda").requires_grad_(True) for s in shapes]
    p_av = [p.detach().clone().requires_grad_(True) for p in p_ref]
    o_ref = torch.optim.Adam([{"params": p_ref[:2], "lr": 1e-3}, {"params": p_ref[2:], "lr": 3e-4}])
    o_av = optim.AvAdam([{"params": p_av[:2], "lr": 1e-3}, {"params": p_av[2:], "lr": 3e-4}])
    s_ref = torch.amp.GradScaler("cuda", init_scale=1024.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=3)
    s_av = optim.AvGradScaler(init_scale=1024.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=3)
    missing = {1: [3], 2: [0], 4: [1, 3], 5: [3], 8: [2]}
    overflow = {2: 1, 6: 0}                 # step -> parameter that receives an inf (step 2: a step with a missing parameter AND an overflow)
    for it in range(10):
        sc_ref = float(s_ref.scale(torch.ones((), device="cuda")))
        assert sc_ref == s_av.get_scale(), it
        for i, (pr, pa) in enumerate(zip(p_ref, p_av)):
            if i in missing.get(it, []):
                pr.grad = None; pa.grad = None
            else:
                g = torch.randn(pr.shape, device="cuda")
                pr.grad = g * sc_ref; pa.grad = (g * sc_ref).clone()
        if it in overflow:
            p_ref[overflow[it]].grad.view(-1)[3] = float("inf"); p_av[overflow[it]].grad.view(-1)[3] = float("inf")
        before = [p.detach().clone() for p in p_av]
        s_ref.step(o_ref); s_ref.update()
        s_av.step(o_av); s_av.update()
        for pr, pa, b in zip(p_ref, p_av, before):
            if it in overflow:
                assert torch.equal(pa.detach(), b)
            torch.testing.assert_close(pa.detach(), pr.detach(), rtol=2e-6, atol=2e-7)
    # a plain state_dict() (not only checkpoint_dict) must carry the DEVICE-side counts: under a scaler the host entries never move
    sd = o_av.state_dict()
    want = [int(o_ref.state[pr]["step"]) for pr in p_ref]
    assert [int(sd["state"][i]["step"]) for i in range(len(p_av))] == want and want != [0] * len(want)
    for pr, pa in zip(p_ref, p_av):
        assert int(o_ref.state[pr]["step"]) == o_av.state[pa]["step"]
    assert s_av.steps_taken() == 8 and o_av.fused_launches == 10
    # adding a parameter group rebuilds the device table: it must be re-seeded from the current counts, not from stale host values
    extra = torch.randn(17, device="cuda").requires_grad_(True)
    o_av.add_param_group({"params": [extra], "lr": 1e-3})
    for pa in p_av:
        pa.grad = torch.randn(pa.shape, device="cuda") * s_av.get_scale()
    extra.grad = torch.randn(17, device="cuda") * s_av.get_scale()
    s_av.step(o_av); s_av.update()
    o_av.sync_steps(s_av)
    assert [o_av.state[pa]["step"] for pa in p_av] == [w + 1 for w in want] and o_av.state[extra]["step"] == 1


def _run_steps(t, batch, n, check_every=1):
    shadow = pkg("utils.shadow")
    m = t.audio_encoder.model
    m.dropped_log = []
    named = [(n_, p) for mod in (m, t.fusion_module, t.decoder1) for n_, p in mod.named_parameters() if p.requires_grad]
    both_dropped = 0
    losses = []
    for step in range(n):
        m.dropped_log.clear()
        out = t.train_step(batch)
        loss = float(out["total"].detach())
        losses.append(loss)
        assert np.isfinite(loss), (step, loss, m.dropped_log)
        tr = [li for li, f in enumerate(m.trainable_layers()) if f]
        if len(m.dropped_log) == 2 and any(li in m.dropped_log[0] and li in m.dropped_log[1] for li in tr):
            both_dropped += 1
        if step % check_every == 0 or step == n - 1:
            sums = torch.stack([p.detach().float().sum() for _, p in named])
            assert bool(torch.isfinite(sums).all()), (step, [n_ for (n_, _), ok in zip(named, torch.isfinite(sums).tolist()) if not ok][:4])
            for n_, p in named:                                     # bf16 shadows written by the fused Adam stay equal to a cast of the master
                sh = shadow.lookup(p)
                if sh is not None:
                    assert torch.equal(sh, p.detach().reshape(-1).to(torch.bfloat16)), (step, n_)
    assert t.optimizer.fused_launches == n                          # never demoted: every step is the one fused launch
    return losses, both_dropped


@pytest.mark.parametrize("loss_scaling", [False, True])
def test_forty_as_executed_steps_at_batch_64(loss_scaling):
    """BASELINE.json's metric configuration (batch 64 x 4 s, bf16) as executed, 40 consecutive steps with the bench's seeds (the seed-1234
    sequence drops trainable layer 6 in both passes at step 11): loss and weights finite at every step, optimizer always fused."""
    import bench
    t, batch, cfg = bench.build_trainer(64, 4.0, "bf16", "cuda:0", loss_scaling=loss_scaling)
    t.audio_encoder.model.cfg.update(bench.HF_REGULARIZERS)
    torch.manual_seed(1234); np.random.seed(1234)
    losses, both = _run_steps(t, batch, 40, check_every=4)
    assert both >= 1, "the seeded sequence must contain a step with a trainable layer dropped in both passes"
    assert losses[-1] < losses[0]                                   # it trains (34 -> ~7 on random labels)
    if loss_scaling:
        assert t.scaler.steps_taken() + int(round(np.log2(65536.0 / t.scaler.get_scale()))) == 40   # every step either counted or backed off


@pytest.mark.parametrize("precision,loss_scaling", [("bf16", False), ("bf16", True), ("fp32", True)])
def test_forty_steps_tiny_config_layerdrop_half(precision, loss_scaling):
    """Tiny config, LayerDrop 0.5 (a trainable layer is without gradients in ~1 of 4 steps per layer), dropout and SpecAugment on."""
    init = pkg("utils.init"); synth = pkg("dataset.synthetic"); tr = pkg("model.trainer")
    cfg = dict(init.W2V2_TINY)
    t0 = build(cfg, precision)
    t = tr.MultimodalTrainer(t0.visual_encoder, t0.audio_encoder, t0.fusion_module, t0.decoder1, t0.tokenizer, learning_rate=1e-4,
                             device="cuda", lambda_=0.1, loss_scaling=loss_scaling)
    t.fixed_projection = t0.fixed_projection
    t.visual_encoder.train(); t.audio_encoder.train(); t.fusion_module.train(); t.decoder1.train()
    t.audio_encoder.model.cfg.update(dict(HF_REG, layerdrop=0.5))
    batch = synth.make_batch(3, 1.0, seed=9, ragged=True)
    torch.manual_seed(7); np.random.seed(7)
    losses, both = _run_steps(t, batch, 40)
    assert both >= 3
    steps = sorted({st["step"] for st in t.optimizer.state.values() if "step" in st}) if not loss_scaling else None
    if steps is not None:
        assert len(steps) > 1 and steps[-1] == 40                    # per-parameter counts really diverged, the head never skipped


def test_config5_batch_128_loss_scaling_as_executed():
    """BASELINE.json configs[4] in its stated mode: batch 128 x 4 s, contrastive loss on, "fp16 + fp32 master" = the reference's
    GradScaler law (model/trainer.py:40,65,121-123; bf16 MFMA operands here, fp32 master weights), HF-default regularisers, two audio
    passes.  8 steps with an overflow injected into one gradient at steps 2 and 5: the scale trajectory equals torch.amp.GradScaler's on
    the same found-inf pattern, an overflowing step leaves every parameter untouched and advances no step counter, clean steps update."""
    import bench
    optim = pkg("optim")
    t, batch, cfg = bench.build_trainer(128, 4.0, "bf16", "cuda:0", loss_scaling=True)
    t.scaler = optim.AvGradScaler(init_scale=65536.0, growth_interval=3, device="cuda:0")       # short interval: growth is exercised too
    t.audio_encoder.model.cfg.update(bench.HF_REGULARIZERS)
    torch.manual_seed(77); np.random.seed(77)
    # torch's scaler driven by the same found-inf pattern (a one-element dummy parameter)
    dummy = torch.zeros(1, device="cuda", requires_grad=True)
    o_ref = torch.optim.SGD([dummy], lr=0.0)
    s_ref = torch.amp.GradScaler("cuda", init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=3)
    named = [(n, p) for mod in (t.audio_encoder.model, t.fusion_module, t.decoder1) for n, p in mod.named_parameters() if p.requires_grad]
    probe = dict(named)["encoder.layers.8.feed_forward.output_dense.weight"]
    inject_at = (2, 5)
    taken = 0
    for step in range(8):
        sc_ref = float(s_ref.scale(torch.ones((), device="cuda")))
        assert t.scaler.get_scale() == sc_ref, (step, t.scaler.get_scale(), sc_ref)
        t.optimizer.zero_grad(set_to_none=True)
        out = t.forward_losses(batch)
        assert np.isfinite(float(out["total"].detach())), step
        t.scaler.scale(out["total"]).backward()
        with_grad = [p for _, p in named if p.grad is not None]
        assert all(bool(torch.isfinite(p.grad).all()) for p in with_grad[:8])
        g0 = float(probe.grad.abs().max()) if probe.grad is not None else None
        if g0 is not None:
            assert g0 > 0.0                                          # the scaled gradients are real (scale folded out only inside Adam)
        dummy.grad = torch.ones(1, device="cuda")
        if step in inject_at:
            with_grad[-1].grad.view(-1)[11] = float("inf")
            dummy.grad[0] = float("inf")
        before = [p.detach().clone() for p in with_grad]
        t.scaler.step(t.optimizer); t.scaler.update()
        s_ref.step(o_ref); s_ref.update()
        changed = [not torch.equal(p.detach(), b) for p, b in zip(with_grad, before)]
        if step in inject_at:
            assert not any(changed), step                            # skipped: nothing moved
        else:
            assert sum(changed) >= len(changed) - 2, step          # (a tensor whose gradient is exactly zero would not move)
            taken += 1
        assert t.scaler.steps_taken() == taken
    assert t.scaler.get_scale() == s_ref.get_scale()
    t.optimizer.sync_steps(t.scaler)
    counts = sorted({st["step"] for st in t.optimizer.state.values() if "step" in st})
    assert counts[-1] == taken and t.optimizer.fused_launches == 8
    sums = torch.stack([p.detach().float().sum() for _, p in named])
    assert bool(torch.isfinite(sums).all())
