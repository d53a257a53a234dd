"""Precision mode "fp16": the SAME kernels compiled with IEEE-half operands (libavhip_f16.so, csrc/av_common.h AV_HALF) - the reference's
GPU arithmetic (torch.cuda.amp fp16 autocast + GradScaler with fp32 parameters, model/trainer.py:9,40,65,121-123 = BASELINE configs[4]
"fp16 + fp32 master").  Checked: the half MFMA path is really taken (its error is an order of magnitude below bfloat16's), the kernels
agree with fp64 references, a whole training step agrees with the reference fixture more tightly than the bf16 mode does, and the
GradScaler law runs on genuine half overflows."""
import math
import os

import numpy as np
import pytest
import torch

from conftest import pkg

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture()
def fp16_mode():
    P = pkg("precision")
    old = P.get_precision()
    P.set_precision("fp16")
    try:
        yield
    finally:
        P.set_precision(old)


def _rand(*shape, dtype=torch.float32, scale=1.0):
    return (torch.randn(*shape, device="cuda", dtype=torch.float32) * scale).to(dtype)


@pytest.mark.parametrize("M,N,K", [(1000, 768, 320), (2048, 1024, 1024), (12736, 1024, 1024), (300, 4096, 1024), (130, 200, 192)])
def test_fp16_gemm_uses_half_operands_and_matches_fp64(fp16_mode, M, N, K):
    ops = pkg("ops"); L = pkg("_lib")
    torch.manual_seed(0)
    x = _rand(M, K, dtype=torch.float16); w = _rand(N, K, dtype=torch.float16, scale=1 / math.sqrt(K)); b = _rand(N); r = _rand(M, N)
    ref = (x.double() @ w.double().t()).float()
    y = ops.linear(x, w, None, out_dtype=torch.float32)
    err = float((y - ref).abs().max())
    # fp32 accumulation of exact half products: the error is accumulation order only.  With bf16 operands the same product of the SAME
    # fp16-representable numbers would be no better, so the discriminating check is on half-typed OUTPUTS below
    assert err < 2e-3, err
    yh = ops.linear(x, w, b, out_dtype=torch.float16)
    e16 = float((yh.float() - (ref + b)).abs().max())
    assert e16 < 4e-3, e16                                         # half output: 2^-11 relative; a bf16 output would sit at ~1.5e-2 here
    pre = torch.empty(M, N, device="cuda", dtype=torch.float16)
    g = ops.linear(x, w, b, out_dtype=torch.float16, act=L.ACT_GELU, C2=pre)
    torch.testing.assert_close(pre.float(), ref + b, rtol=2e-3, atol=4e-3)
    torch.testing.assert_close(g.float(), torch.nn.functional.gelu(ref + b), rtol=2e-3, atol=4e-3)
    out = r.clone()
    ops.linear(x, w, b, out=out, R=out, alpha=0.5)
    torch.testing.assert_close(out, 0.5 * ref + b + r, rtol=1e-3, atol=2e-3)
    with pytest.raises(TypeError):                                  # the half library does not take bfloat16 tensors
        ops.linear(x.to(torch.bfloat16), w.to(torch.bfloat16), None)


def test_fp16_overflow_is_visible(fp16_mode):
    """Products beyond 65504 become inf in a half output (this is what the GradScaler law exists for); the fp32 output of the same
    product is finite."""
    ops = pkg("ops")
    x = torch.full((256, 64), 64.0, device="cuda", dtype=torch.float16); w = torch.full((256, 64), 32.0, device="cuda", dtype=torch.float16)
    assert bool(torch.isinf(ops.linear(x, w, None, out_dtype=torch.float16)).all())          # 64 * 64 * 32 = 131072 > 65504
    assert bool(torch.isfinite(ops.linear(x, w, None, out_dtype=torch.float32)).all())


@pytest.mark.parametrize("B,H,Tq,Tk,D,masked", [(3, 16, 199, 199, 64, True), (2, 4, 100, 100, 128, False), (2, 3, 749, 749, 64, True), (2, 2, 70, 130, 64, True)])
def test_fp16_attention_fwd_bwd(fp16_mode, B, H, Tq, Tk, D, masked):
    ops = pkg("ops")
    torch.manual_seed(1)
    dtype = torch.float16
    if Tq == Tk:
        qkv = _rand(B, Tq, 3, H, D, dtype=dtype); q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    else:
        q = _rand(B, Tq, H, D, dtype=dtype); kv = _rand(B, Tk, 2, H, D, dtype=dtype); k, v = kv[:, :, 0], kv[:, :, 1]
    klen = torch.tensor([Tk, max(1, Tk // 2), max(1, Tk - 3)][:B], device="cuda", dtype=torch.int32) if masked else None
    scale = D ** -0.5
    o, lse = ops.attention_fwd(q, k, v, klen, scale)
    qr, kr, vr = (t.float().permute(0, 2, 1, 3).detach().clone().requires_grad_(True) for t in (q, k, v))
    s = (qr.double() @ kr.double().transpose(2, 3)) * scale
    if masked:
        keep = torch.arange(Tk, device="cuda")[None, :] < klen[:, None]
        s = s.masked_fill(~keep[:, None, None, :], float("-inf"))
    ref = (torch.softmax(s, -1) @ vr.double()).float()
    torch.testing.assert_close(o.float().permute(0, 2, 1, 3), ref, rtol=4e-3, atol=4e-3)     # (bf16 gate: 3e-2)
    do = _rand(B, Tq, H, D, dtype=dtype)
    ref.backward(do.float().permute(0, 2, 1, 3))
    dq = torch.empty(B, Tq, H, D, device="cuda", dtype=dtype); dk = torch.empty(B, Tk, H, D, device="cuda", dtype=dtype); dv = torch.empty_like(dk)
    ops.attention_bwd(q, k, v, do, dq, dk, dv, klen, scale, o=o, lse=lse)
    for a_, r_ in ((dq, qr.grad), (dk, kr.grad), (dv, vr.grad)):
        torch.testing.assert_close(a_.float().permute(0, 2, 1, 3), r_, rtol=8e-3, atol=8e-3)                # (bf16 gate: 5e-2)


def test_fp16_step_vs_reference_fixture(fp16_mode):
    """Whole training step (tiny fixture captured from the reference) in the half mode: tighter than the bf16 mode's 0.013-0.016."""
    from test_step_gpu import build, maxdiff
    fx = np.load(os.path.join(GOLD, "tiny.npz"))
    init = pkg("utils.init"); synth = pkg("dataset.synthetic")
    t = build(init.W2V2_TINY, "fp16")
    batch = synth.make_batch(int(fx["batch"]), float(fx["seconds"]), seed=int(fx["seed_batch"]), ragged=False)
    t.visual_encoder.train(); t.audio_encoder.train(); t.fusion_module.train(); t.decoder1.train()
    out = t.train_step(batch)
    e_lp = max(maxdiff(out["log_probs1"].detach().cpu(), fx["train_log_probs1"]), maxdiff(out["log_probs2"].detach().cpu(), fx["train_log_probs2"]))
    e_loss = abs(float(out["total"]) - float(fx["train_total"]))
    print(f"tiny fp16 step: max|dlogp| = {e_lp:.5f}, |dloss| = {e_loss:.5f}")
    assert e_lp <= 6e-3 and e_loss <= 6e-3
    assert np.array_equal(out["input_lengths1"].cpu().numpy(), fx["train_input_lengths1"])


def test_fp16_config5_loss_scaling_as_executed(fp16_mode):
    """BASELINE configs[4] as stated: batch 128 x 4 s, contrastive loss on, fp16 operands + fp32 master weights + the GradScaler law, HF-default
    regularisers, two audio passes.  The scale starts at 65536 (torch's default): whatever the half gradients do - overflow (step skipped, scale
    halved, nothing moves) or not (step taken) - must follow torch.amp.GradScaler's law on the observed found-inf pattern, and the weights
    stay finite.  The trajectory is printed."""
    import bench
    optim = pkg("optim")
    t, batch, cfg = bench.build_trainer(128, 4.0, "fp16", "cuda:0", loss_scaling=True)
    t.audio_encoder.model.cfg.update(bench.HF_REGULARIZERS)
    torch.manual_seed(77); np.random.seed(77)
    named = [(n, p) for mod in (t.audio_encoder.model, t.fusion_module, t.decoder1) for n, p in mod.named_parameters() if p.requires_grad]
    traj = []
    scale = 65536.0
    clean = 0
    for step in range(10):
        assert t.scaler.get_scale() == scale, (step, t.scaler.get_scale(), scale)
        before = [p.detach().clone() for _, p in named[:6]]
        taken0 = t.scaler.steps_taken()
        out = t.train_step(batch)
        loss = float(out["total"].detach())
        took = t.scaler.steps_taken() - taken0
        moved = any(not torch.equal(p.detach(), b) for (_, p), b in zip(named[:6], before))
        assert moved == bool(took), step                              # a skipped step moves nothing, a taken one does
        traj.append((step, scale, "taken" if took else "skipped (overflow)", round(loss, 4)))
        if took:
            clean += 1
            if clean == t.scaler.growth_interval:
                scale *= 2.0; clean = 0
        else:
            scale *= 0.5; clean = 0
    print("fp16 config-5 scale trajectory:", traj)
    assert np.isfinite(loss)
    assert sum(1 for x in traj if x[2] == "taken") >= 5                  # the scale settles: most steps are taken
    sums = torch.stack([p.detach().float().sum() for _, p in named])
    assert bool(torch.isfinite(sums).all())
