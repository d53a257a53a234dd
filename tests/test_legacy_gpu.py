"""the legacy mel + GRU model on the HIP kernels vs the fixture captured FROM THE REFERENCE
(tests/golden/legacy_tiny.npz, made by tests/golden/make_golden_legacy.py) and vs the CPU oracle (oracle/legacy_oracle.py)."""
import os

import numpy as np
import pytest
import torch

from conftest import pkg
from oracle import legacy_oracle as O

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _maxdiff(a, b):
    return float(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)).max())


def _build(vocab, hidden, seed, precision):
    init = pkg("utils.init"); lg = pkg("model.legacy")
    pkg("precision").set_precision(precision)
    m = lg.MultimodalCTCKoreanModel(vocab_size=vocab, hidden_dim=hidden)
    sd = init.legacy_state_dict(vocab, hidden, seed)
    assert set(m.state_dict().keys()) == set(sd.keys())                    # the reference's checkpoint keys
    m.load_state_dict(sd)
    return m.cuda().train(), sd


def _cuda_batch(batch):
    fa, fb, mel, mel_len, la, na, lb, nb = batch
    return (fa.cuda(), fb.cuda(), mel.cuda(), mel_len, la.cuda(), na, lb.cuda(), nb)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_legacy_step_vs_reference_fixture(precision):
    fx = np.load(os.path.join(GOLD, "legacy_tiny.npz"))
    vocab, hidden, B, T, seed_w, seed_b = (int(x) for x in fx["cfg"])
    init = pkg("utils.init"); lg = pkg("model.legacy"); optim = pkg("optim")
    m, sd = _build(vocab, hidden, seed_w, precision)
    batch = init.legacy_batch(B, T, vocab, seed_b)
    opt = optim.AvAdam(m.parameters(), lr=1e-4)                            # train_ctc_korea.py:86
    before = {k: v.detach().clone() for k, v in m.named_parameters()}
    fa, fb, mel, mel_len, la, na, lb, nb = _cuda_batch(batch)
    logits_A, logits_B = m(fa, fb, mel)
    loss = lg.legacy_losses(logits_A, logits_B, mel_len, la, na, lb, nb)
    opt.zero_grad(); loss.backward()
    tol = 1e-3 if precision == "fp32" else 6e-2
    assert _maxdiff(logits_A.detach().cpu(), fx["logits_A"]) < tol
    assert _maxdiff(logits_B.detach().cpu(), fx["logits_B"]) < tol
    assert abs(float(loss) - float(fx["loss"])) < (1e-3 if precision == "fp32" else 0.1)
    for k, p in m.named_parameters():
        assert p.grad is not None, k
        flat = p.grad.detach().flatten().cpu()
        idx = torch.linspace(0, flat.numel() - 1, min(48, flat.numel())).long()
        gs, gn = fx["gslice/" + k], float(fx["gnorm/" + k])
        scale = max(float(np.abs(gs).max()), gn / np.sqrt(flat.numel()))
        assert _maxdiff(flat[idx], gs) < (2e-3 if precision == "fp32" else 8e-2) * scale + 1e-7, k
        assert abs(float(flat.norm()) - gn) < (2e-3 if precision == "fp32" else 5e-2) * gn + 1e-7, k
    opt.step()
    if precision == "fp32":                                               # post-Adam deltas (first step: -lr * sign(g) where |g| >> eps)
        bad = tot = 0
        for k, p in m.named_parameters():
            d = (p.detach() - before[k]).flatten().cpu()
            idx = torch.linspace(0, d.numel() - 1, min(48, d.numel())).long()
            diff = np.abs(d[idx].numpy() - fx["dslice/" + k])
            bad += int((diff > 2e-6).sum()); tot += diff.size
        assert bad <= 0.02 * tot, (bad, tot)


def test_legacy_matches_oracle_at_reference_width():
    """hidden 256 (the reference's default width, :40), batch 2 x 5 steps, vocab 200: logits and every gradient vs the CPU oracle."""
    init = pkg("utils.init"); lg = pkg("model.legacy")
    vocab, hidden, B, T = 200, 256, 2, 5
    m, sd = _build(vocab, hidden, 3, "fp32")
    batch = init.legacy_batch(B, T, vocab, 5)
    fa, fb, mel, mel_len, la, na, lb, nb = _cuda_batch(batch)
    logits_A, logits_B = m(fa, fb, mel)
    loss = lg.legacy_losses(logits_A, logits_B, mel_len, la, na, lb, nb)
    loss.backward()
    osd = {k: v.clone() for k, v in sd.items()}
    out, og = O.train_step(osd, batch, {}, lr=1e-4)
    assert _maxdiff(logits_A.detach().cpu(), out["logits_A"]) < 1e-3 and _maxdiff(logits_B.detach().cpu(), out["logits_B"]) < 1e-3
    assert abs(float(loss) - float(out["loss"])) < 1e-3
    for k, p in m.named_parameters():
        ref = og[k]
        assert _maxdiff(p.grad.cpu(), ref) < 2e-3 * float(ref.abs().max()) + 1e-7, k


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_relu_maxpool_kernels(dtype):
    ops = pkg("ops"); L = pkg("_lib")
    N, H, W, C = 5, 12, 20, 32
    g = torch.Generator().manual_seed(0)
    x = torch.randn(N, C, H, W, generator=g)
    x[0, :, :2, :2] = 0.5                                                  # ties: the first maximum of the window takes the gradient
    xq = x.to(dtype).float()
    xr = xq.clone().requires_grad_(True)
    y_ref = torch.nn.functional.max_pool2d(torch.relu(xr), 2)
    dy = torch.randn(y_ref.shape, generator=g).to(dtype).float()
    y_ref.backward(dy)
    xc = xq.permute(0, 2, 3, 1).contiguous().to(dtype).cuda()             # NHWC
    for nchw in (0, 1):
        y = torch.empty((N, C, H // 2, W // 2) if nchw else (N, H // 2, W // 2, C), dtype=dtype, device="cuda")
        L.check(L.lib().av_relu_maxpool2_fwd(ops.ptr(xc), ops.ptr(y), ops.dt(xc), N, H, W, C, nchw, ops.stream()))
        y_nchw = y.float().cpu() if nchw else y.float().cpu().permute(0, 3, 1, 2)
        assert torch.equal(y_nchw, y_ref.detach())
        dyc = (dy if nchw else dy.permute(0, 2, 3, 1)).contiguous().to(dtype).cuda()
        dx = torch.empty_like(xc)
        L.check(L.lib().av_relu_maxpool2_bwd(ops.ptr(xc), ops.ptr(dyc), ops.ptr(dx), ops.dt(xc), N, H, W, C, nchw, ops.stream()))
        assert torch.equal(dx.float().cpu().permute(0, 3, 1, 2), xr.grad)


@pytest.mark.parametrize("dtype,B,T,H", [(torch.float32, 70, 7, 128), (torch.bfloat16, 70, 7, 256), (torch.float32, 3, 1, 128)])
def test_gru_layers_vs_oracle(dtype, B, T, H):
    """gru_forward / gru_backward (two bidirectional layers, per-step kernels; 70 rows = two 64-row groups) vs the oracle's loop."""
    lg = pkg("model.legacy"); shadow = pkg("utils.shadow")
    torch.manual_seed(1)
    in_f = 96
    rnn = torch.nn.GRU(in_f, H, num_layers=2, batch_first=True, bidirectional=True).cuda()
    x = torch.randn(B, T, in_f)
    sd = {"r." + k: v.detach().cpu().clone().requires_grad_(True) for k, v in rnn.named_parameters()}
    xr = x.to(dtype).float().requires_grad_(True)
    ref = O.bigru2(sd, "r.", xr)
    dout = torch.randn(B, T, 2 * H)
    ref.backward(dout)
    x_tm = x.to(dtype).cuda().transpose(0, 1).contiguous()
    out_bt, ctx = lg.gru_forward(rnn, shadow.ParamCache(), x_tm, True)
    tol = 2e-4 if dtype == torch.float32 else 3e-2
    assert _maxdiff(out_bt.float().cpu(), ref.detach()) < tol
    grads = {}
    dx = lg.gru_backward(rnn, ctx, dout.cuda(), grads, "r.", need_dx=True)
    assert _maxdiff(dx.float().cpu().transpose(0, 1), xr.grad) < tol * max(1.0, float(xr.grad.abs().max()))
    for k, v in sd.items():
        assert _maxdiff(grads[k].cpu(), v.grad) < tol * max(1.0, float(v.grad.abs().max())), k
