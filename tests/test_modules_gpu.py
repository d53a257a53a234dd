"""GPU parity of the HIP modules (visual encoder, fusion incl. BiLSTM, CTC head, contrastive loss, Adam) vs the oracle."""
import numpy as np
import pytest
import torch

from conftest import pkg

pytestmark = pytest.mark.gpu


def _p(precision):
    pkg("precision").set_precision(precision)


def _clone(sd):
    return {k: v.clone() for k, v in sd.items()}


@pytest.mark.parametrize("precision,tol", [("fp32", 1e-3), ("bf16", 0.12)])
@pytest.mark.parametrize("training", [True, False])
def test_visual_encoder(precision, tol, training):
    _p(precision)
    from oracle import av_oracle as O
    init = pkg("utils.init"); enc = pkg("model.encoder")
    sd = init.visual_state_dict()
    g = torch.Generator().manual_seed(3)
    x = torch.rand(2, 1, 5, 96, 96, generator=g)
    osd = _clone(sd)
    with torch.no_grad():
        ref = O.visual_forward(osd, x, training)
    ve = enc.VisualEncoder().cuda()
    ve.load_state_dict(sd)
    for p in ve.parameters():
        p.requires_grad = False
    ve.train(training)
    out = ve(x.cuda())
    assert out.shape == (2, 5, 512)
    err = float((out.cpu() - ref).abs().max())
    print("visual max err", precision, training, err, "ref scale", float(ref.abs().max()))
    assert err < tol * max(1.0, float(ref.abs().max()))
    if training:   # running statistics side effect (model/trainer.py:54)
        now = ve.state_dict()
        for k in ("frontend3D.1.running_mean", "frontend3D.1.running_var", "trunk.layer4.1.bn2.running_var", "trunk.layer2.0.downsample.1.running_mean"):
            assert float((now[k].cpu() - osd[k]).abs().max()) < (2e-4 if precision == "fp32" else 5e-2), k
        assert int(now["frontend3D.1.num_batches_tracked"]) == 1


@pytest.mark.parametrize("training", [True, False])
def test_visual_encoder_position_major_layers_equal_frame_major(training):
    """ResNet layers 2-4 in blocked position-major pixel order (model/encoder.py: POS_MAJOR; av_gemm_args.cNF / cPM: rows of a tile share
    their image position, out-of-image filter taps are skipped as whole K-tiles) against the frame-major path on the same input: 512 frames
    (two blocks of 256), train-mode batch statistics and eval mode.  Skipped taps only drop exact zeros from the fp32 accumulations, so the
    convolution outputs agree to rounding of the batch statistics (summed in another order)."""
    _p("bf16")
    init = pkg("utils.init"); enc = pkg("model.encoder")
    sd = init.visual_state_dict()
    g = torch.Generator().manual_seed(13)
    x = torch.rand(4, 1, 128, 96, 96, generator=g).cuda()
    outs, stats = [], []
    for pm in (False, True):
        enc.POS_MAJOR = pm
        try:
            ve = enc.VisualEncoder().cuda(); ve.load_state_dict(sd)
            for p in ve.parameters():
                p.requires_grad = False
            ve.train(training)
            outs.append(ve(x).float().cpu())
            stats.append({k: v.float().cpu() for k, v in ve.state_dict().items() if "running" in k})
        finally:
            enc.POS_MAJOR = True
    scale = float(outs[0].abs().max())
    err = float((outs[0] - outs[1]).abs().max())
    print("position-major vs frame-major: max |d| =", err, "of", scale)
    assert outs[0].shape == (4, 128, 512) and err <= 2e-2 * scale
    if training:
        for k in stats[0]:
            assert float((stats[0][k] - stats[1][k]).abs().max()) <= 1e-3 * max(1.0, float(stats[0][k].abs().max())), k


@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("slope_sign", [1.0, -1.0])
def test_front_end_pooled_in_the_conv_kernel_equals_the_unfused_path(training, slope_sign):
    """frontend3d.hip POOL form + av_bn_prelu_minmax (window max / min of the raw Conv3d output, BatchNorm + PReLU applied to the two extremes)
    against Conv3d -> HBM -> av_bn_prelu_maxpool on the same input: bit-identical features and running statistics.  Negative BatchNorm scales
    (gamma < 0) and negative PReLU slopes (V-shaped activation) are the cases where the MINIMUM decides; T = 7 frames exercises the temporal
    padding, 96 x 96 frames three column strips (halo column) and six row tiles (carry row)."""
    _p("bf16")
    init = pkg("utils.init"); enc = pkg("model.encoder")
    sd = _clone(init.visual_state_dict())
    g = torch.Generator().manual_seed(23)
    sd["frontend3D.1.weight"] = sd["frontend3D.1.weight"] * torch.where(torch.rand(64, generator=g) < 0.4, -1.0, 1.0)       # some negative BN scales
    sd["frontend3D.2.weight"] = (sd["frontend3D.2.weight"].abs() + 0.1) * torch.where(torch.rand(64, generator=g) < 0.5, slope_sign, 1.0)
    x = torch.rand(3, 1, 7, 96, 96, generator=g).cuda()
    outs, stats = [], []
    for fused in (False, True):
        enc.FRONT_POOL = fused
        try:
            ve = enc.VisualEncoder().cuda(); ve.load_state_dict(sd)
            for p in ve.parameters():
                p.requires_grad = False
            ve.train(training)
            outs.append(ve(x).float().cpu())
            stats.append({k: v.float().cpu() for k, v in ve.state_dict().items() if "running" in k})
        finally:
            enc.FRONT_POOL = True
    assert outs[0].shape == (3, 7, 512) and bool(torch.isfinite(outs[0]).all())
    assert torch.equal(outs[0], outs[1]), float((outs[0] - outs[1]).abs().max())
    for k in stats[0]:
        assert torch.equal(stats[0][k], stats[1][k]), k


@pytest.mark.parametrize("precision,tol,gtol", [("fp32", 1e-3, 2e-3), ("bf16", 6e-2, 0.1)])
@pytest.mark.parametrize("ragged", [False, True])
def test_fusion_fwd_bwd(precision, tol, gtol, ragged):
    _p(precision)
    from oracle import av_oracle as O
    init = pkg("utils.init"); fm = pkg("model.fusion_module")
    B, Tv, Ta, Da = 3, 30, 59, 64
    g = torch.Generator().manual_seed(5)
    vis = torch.randn(B, Tv, 512, generator=g); aud = torch.randn(B, Ta, Da, generator=g)
    mask = torch.ones(B, Ta, dtype=torch.long)
    mask[:, 44:] = 2
    if ragged:
        mask[1, 30:] = 3; mask[1, 20:30] = 0; mask[2, 50:] = 3; mask[2, :5] = 0
    else:
        mask[:, 40:] = 0          # speaker-2 style: 1 then 0
    sd = init.fusion_state_dict(512, Da, 512)
    osd = _clone(sd)
    used = [k for k in osd if not k.startswith("cross_attn_visual.")]
    for k in used:
        osd[k].requires_grad_(True)
    aud_r = aud.clone().requires_grad_(True)
    ref, ref_len = O.fusion_forward(osd, vis, aud_r, mask)
    w = torch.randn(ref.shape, generator=g)
    (ref * w).sum().backward()
    mod = fm.CrossAttentionFusion(512, Da, 512).cuda()
    mod.load_state_dict(sd)
    aud_c = aud.cuda().requires_grad_(True)
    out, lens = mod(vis.cuda(), aud_c, mask.cuda())
    assert torch.equal(lens.cpu(), ref_len), (lens, ref_len)
    err = float((out.cpu() - ref).abs().max())
    print("fusion fwd err", err)
    assert err < tol
    (out * w.cuda()).sum().backward()
    worst = 0.0
    for n, p in mod.named_parameters():
        if n.startswith("cross_attn_visual."):
            assert p.grad is None
            continue
        rel = float((p.grad.cpu() - osd[n].grad).norm() / (osd[n].grad.norm() + 1e-12))
        worst = max(worst, rel)
        assert rel < gtol, (n, rel)
    rel = float((aud_c.grad.cpu() - aud_r.grad).norm() / (aud_r.grad.norm() + 1e-12))
    print("fusion worst param-grad rel", worst, "audio-grad rel", rel)
    assert rel < gtol


@pytest.mark.parametrize("precision,tol", [("fp32", 1e-3), ("bf16", 5e-2)])
def test_decoder_fwd_bwd(precision, tol):
    _p(precision)
    from oracle import av_oracle as O
    init = pkg("utils.init"); dm = pkg("model.decoder")
    sd = init.decoder_state_dict(1024, 800)
    g = torch.Generator().manual_seed(6)
    x = torch.randn(2, 25, 1024, generator=g) * 0.2
    osd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xr = x.clone().requires_grad_(True)
    ref = O.decoder_forward(osd, xr)
    w = torch.randn(ref.shape, generator=g)
    (ref * w).sum().backward()
    d = dm.CTCDecoder(1024, 800, 3).cuda(); d.load_state_dict(sd)
    xc = x.cuda().requires_grad_(True)
    out = d(xc)
    assert float((out.cpu() - ref).abs().max()) < tol
    (out * w.cuda()).sum().backward()
    for n, p in d.named_parameters():
        rel = float((p.grad.cpu() - osd[n].grad).norm() / osd[n].grad.norm())
        assert rel < (2e-4 if precision == "fp32" else 3e-2), (n, rel)
    assert float((xc.grad.cpu() - xr.grad).norm() / xr.grad.norm()) < (2e-4 if precision == "fp32" else 3e-2)
    # CTC branch (stays on PyTorch-ROCm)
    tgt = torch.randint(4, 800, (2, 5)); il = torch.tensor([25, 20]); tl = torch.tensor([5, 3])
    loss = d(xc, tgt.cuda(), il.cuda(), tl.cuda())
    ref_loss = O.ctc_loss(ref.detach(), tgt, il, tl, 3)
    assert abs(float(loss) - float(ref_loss)) < (1e-3 if precision == "fp32" else 0.3)


@pytest.mark.parametrize("precision,tol,gtol", [("fp32", 1e-4, 1e-3), ("bf16", 5e-2, 0.1)])
@pytest.mark.parametrize("with_pad", [False, True])
def test_contrastive_fwd_bwd(precision, tol, gtol, with_pad):
    _p(precision)
    from oracle import av_oracle as O
    init = pkg("utils.init"); con = pkg("contrastive")
    B, T, D = 3, 49, 64
    g = torch.Generator().manual_seed(8)
    mid = torch.randn(B, T, D, generator=g)
    m = torch.ones(B, T, dtype=torch.long); m[:, 30:] = 2; m[1, 20:] = 0
    if with_pad:
        m[2, 40:] = 3
    pw, pb = init.projection_params(D)
    mr = mid.clone().requires_grad_(True)
    ref = O.contrastive(mr, m.reshape(-1), pw, pb)
    ref.backward()
    proj = torch.nn.Linear(D, 128).cuda()
    with torch.no_grad():
        proj.weight.copy_(pw); proj.bias.copy_(pb)
    mc = mid.cuda().requires_grad_(True)
    for counts in (None, (int((m == 1).sum()), int((m == 2).sum()), int((m == 0).sum()))):
        mc.grad = None
        out = con.contrastive_loss_with_mask(mc, m.reshape(-1).cuda(), projection_layer=proj, counts=counts)
        assert abs(float(out) - float(ref)) < tol * max(1.0, abs(float(ref))), (float(out), float(ref))
        (out * 0.05).backward()
        rel = float((mc.grad.cpu() / 0.05 - mr.grad).norm() / mr.grad.norm())
        print("contrastive", precision, float(out), float(ref), "grad rel", rel)
        assert rel < gtol


@pytest.mark.parametrize("precision,tol,gtol", [("fp32", 1e-4, 2e-3), ("bf16", 5e-2, 0.1)])
def test_contrastive_chunked_small_blocks(precision, tol, gtol, monkeypatch):
    """Blocks of 24 rows x 64 columns force several row AND column chunks (running LSE merge, accumulated dP) on a small problem with
    both terms present."""
    _p(precision)
    from oracle import av_oracle as O
    init = pkg("utils.init"); con = pkg("contrastive")
    monkeypatch.setattr(con, "ROW_CHUNK", 24); monkeypatch.setattr(con, "COL_CHUNK", 64)
    B, T, D = 4, 80, 64
    g = torch.Generator().manual_seed(9)
    mid = torch.randn(B, T, D, generator=g)
    m = torch.ones(B, T, dtype=torch.long); m[:, 25:] = 2; m[1:3, 10:] = 0; m[3, 70:] = 3
    pw, pb = init.projection_params(D)
    mr = mid.clone().requires_grad_(True)
    ref = O.contrastive(mr, m.reshape(-1), pw, pb); ref.backward()
    proj = torch.nn.Linear(D, 128).cuda()
    with torch.no_grad():
        proj.weight.copy_(pw); proj.bias.copy_(pb)
    mc = mid.cuda().requires_grad_(True)
    out = con.contrastive_loss_with_mask(mc, m.reshape(-1).cuda(), projection_layer=proj)
    assert abs(float(out) - float(ref)) < tol * max(1.0, abs(float(ref))), (float(out), float(ref))
    out.backward()
    rel = float((mc.grad.cpu() - mr.grad).norm() / mr.grad.norm())
    assert rel < gtol, rel


def test_contrastive_streaming_at_config5_size():
    """configs[4] (B = 128 / GPU x 4 s): 19 104 anchors x 6 368 positives (speaker 1) - the similarity matrix (0.49 GB in fp32) is
    never allocated: peak extra memory of the loss stays near the 100 MB block workspace; value and gradient against the CPU oracle."""
    _p("bf16")
    from oracle import av_oracle as O
    init = pkg("utils.init"); con = pkg("contrastive")
    B, T, D = 128, 199, 1024
    g = torch.Generator().manual_seed(10)
    mid = torch.randn(B, T, D, generator=g) * 0.5
    m = torch.ones(B, T, dtype=torch.long); m[:, 149:] = 2                      # mask1 of the synthetic clips: 0.75 overlap, then speaker alone
    n1, n2 = int((m == 1).sum()), int((m == 2).sum())
    assert (n1, n2) == (19072, 6400)
    pw, pb = init.projection_params(D)
    mr = mid.clone().requires_grad_(True)
    ref = O.contrastive(mr, m.reshape(-1), pw, pb); ref.backward()
    proj = torch.nn.Linear(D, 128).cuda()
    with torch.no_grad():
        proj.weight.copy_(pw); proj.bias.copy_(pb)
    mc = mid.cuda().requires_grad_(True)
    mflat = m.reshape(-1).cuda()
    torch.cuda.synchronize(); torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    out = con.contrastive_loss_with_mask(mc, mflat, projection_layer=proj, counts=(n1, n2, 0))
    out.backward()
    torch.cuda.synchronize()
    peak = torch.cuda.max_memory_allocated() - base
    full_matrix = n1 * n2 * 4
    print(f"config-5 contrastive: loss {float(out):.5f} (oracle {float(ref):.5f}); peak extra memory {peak / 1e6:.0f} MB; a full fp32 S would be {full_matrix / 1e6:.0f} MB")
    assert abs(float(out) - float(ref)) < 5e-2 * max(1.0, abs(float(ref)))
    rel = float((mc.grad.cpu() - mr.grad).norm() / mr.grad.norm())
    assert rel < 0.1, rel
    # O(N D) buffers of the loss itself (gathered rows 52 MB, three fp32 [N, D] gradient buffers 3 x 104 MB) + the 100 MB block workspace;
    # the O(N1 N2) matrices (489 MB fp32 S + 244 MB bf16 dS) must not appear
    assert peak < 600e6, peak


def test_adam_step_matches_torch():
    L = pkg("_lib"); ops = pkg("ops")
    g = torch.Generator().manual_seed(9)
    p0 = torch.randn(1000, 33, generator=g)
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([pr], lr=2e-5)
    pc = p0.clone().cuda(); m = torch.zeros_like(pc); v = torch.zeros_like(pc)
    for step in range(1, 4):
        gr = torch.randn(1000, 33, generator=g)
        pr.grad = gr.clone()
        opt.step()
        L.check(L.lib().av_adam_step(ops.ptr(pc), ops.ptr(gr.cuda()), ops.ptr(m), ops.ptr(v), pc.numel(), 2e-5, 0.9, 0.999, 1e-8, step, 1.0, ops.stream()))
    torch.testing.assert_close(pc.cpu(), pr.detach(), rtol=1e-6, atol=1e-7)


def test_fusion_groups_equals_separate_calls():
    """Pair-batched call (groups=2) == two per-speaker calls: the 'batch maximum' is evaluated per group."""
    _p("fp32")
    init = pkg("utils.init"); fm = pkg("model.fusion_module")
    B, Tv, Ta, Da = 3, 25, 49, 64
    g = torch.Generator().manual_seed(11)
    vis1, vis2 = torch.randn(B, Tv, 512, generator=g).cuda(), torch.randn(B, Tv, 512, generator=g).cuda()
    aud = torch.randn(B, Ta, Da, generator=g).cuda()
    m1 = torch.ones(B, Ta, dtype=torch.long); m1[:, 36:] = 2
    m2 = torch.ones(B, Ta, dtype=torch.long); m2[:, 36:] = 0            # speaker 2 keeps only 36 frames -> different batch max
    m1, m2 = m1.cuda(), m2.cuda()
    mod = fm.CrossAttentionFusion(512, Da, 512).cuda(); mod.load_state_dict(init.fusion_state_dict(512, Da, 512))
    with torch.no_grad():
        o1, l1 = mod(vis1, aud, m1); o2, l2 = mod(vis2, aud, m2)
        o12, l12 = mod(torch.cat([vis1, vis2]), torch.cat([aud, aud]), torch.cat([m1, m2]), groups=2)
    assert torch.equal(l12, torch.cat([l1, l2]))
    torch.testing.assert_close(o12, torch.cat([o1, o2]), rtol=1e-5, atol=1e-5)
    with torch.no_grad():
        bad, _ = mod(torch.cat([vis1, vis2]), torch.cat([aud, aud]), torch.cat([m1, m2]), groups=1)
    assert float((bad[B:] - o2).abs().max()) > 1e-3      # without groups the second speaker would be resampled differently


def test_avadam_multi_tensor_matches_torch():
    optim = pkg("optim")
    g = torch.Generator().manual_seed(12)
    shapes = [(1000, 33), (70000,), (5,), (128, 1024)]
    ref = [torch.randn(*s, generator=g).requires_grad_(True) for s in shapes]
    mine = [r.detach().clone().cuda().requires_grad_(True) for r in ref]
    o_ref = torch.optim.Adam([{"params": ref[:2], "lr": 1e-4}, {"params": ref[2:], "lr": 2e-5}])
    o_mine = optim.AvAdam([{"params": mine[:2], "lr": 1e-4}, {"params": mine[2:], "lr": 2e-5}])
    for step in range(3):
        for r, m in zip(ref, mine):
            gr = torch.randn(r.shape, generator=g)
            r.grad = gr.clone(); m.grad = gr.cuda()
        if step == 1:
            ref[2].grad = None; mine[2].grad = None          # a parameter without a gradient is skipped
        o_ref.step(); o_mine.step()
    for r, m in zip(ref, mine):
        torch.testing.assert_close(m.detach().cpu(), r.detach(), rtol=1e-6, atol=1e-7)
    assert set(o_mine.state[mine[0]].keys()) == {"step", "exp_avg", "exp_avg_sq"}


def test_avadam_writes_bf16_shadows_and_keeps_caches_coherent():
    """The fused step writes the bf16 compute copies of the parameters it updates (utils/shadow.py): after a step the cached
    copy equals a fresh cast of the master weights and is NOT rebuilt; an out-of-band update invalidates it."""
    optim, shadow, ops = pkg("optim"), pkg("utils.shadow"), pkg("ops")
    g = torch.Generator().manual_seed(5)
    a = torch.randn(300, 64, generator=g).cuda().requires_grad_(True)
    b = torch.randn(100, 64, generator=g).cuda().requires_grad_(True)
    c = torch.randn(77, generator=g).cuda().requires_grad_(True)                     # odd size: scalar tail of the kernel
    cache = shadow.ParamCache()
    builds = []

    def cat_ab():
        builds.append("ab")
        return torch.cat([a.data, b.data], 0).to(torch.bfloat16)

    def cast_c():
        builds.append("c")
        return c.data.to(torch.bfloat16)

    ab0 = cache.get("ab", [a, b], torch.bfloat16, cat_ab, flat=True)
    c0 = cache.get("c", [c], torch.bfloat16, cast_c, flat=True)
    opt = optim.AvAdam([a, b, c], lr=1e-2)
    abv = ab0.view(200, 128)                                                         # a second weight-like view of the same shadow (shares its version counter)
    for step in range(3):
        for p in (a, c):                                                             # b never has a gradient: its slice stays valid
            p.grad = torch.randn(p.shape, generator=g).cuda()
        opt.step()
        ab = cache.get("ab", [a, b], torch.bfloat16, cat_ab, flat=True)
        cc = cache.get("c", [c], torch.bfloat16, cast_c, flat=True)
        assert ab.data_ptr() == ab0.data_ptr() and cc.data_ptr() == c0.data_ptr()    # same storage, updated in place
        torch.testing.assert_close(ab, torch.cat([a.data, b.data], 0).to(torch.bfloat16), rtol=0, atol=0)
        torch.testing.assert_close(cc, c.data.to(torch.bfloat16), rtol=0, atol=0)
        # caches DERIVED from a shadow (W^T for the dX products) must follow the in-place update (the kernel writes through raw pointers)
        torch.testing.assert_close(ops.transpose_cached(ab), ab.t().contiguous(), rtol=0, atol=0)
        hot = ops.transpose_cached(abv, hot_ok=True)             # from its third version on a weight is "hot": no copy, k-major operand instead
        assert hot is None if step >= 2 else torch.equal(hot, abv.t().contiguous())
    assert builds == ["ab", "c"]                                                     # never rebuilt by the optimizer steps
    with torch.no_grad():
        a.mul_(2.0)                                                                  # out-of-band update: version bump -> rebuild
    assert shadow.lookup(a) is None
    ab = cache.get("ab", [a, b], torch.bfloat16, cat_ab, flat=True)
    torch.testing.assert_close(ab, torch.cat([a.data, b.data], 0).to(torch.bfloat16), rtol=0, atol=0)
    assert builds == ["ab", "c", "ab"]



def test_grad_scaler_matches_torch_amp_on_injected_infs():
    """AvGradScaler + fused Adam against torch.amp.GradScaler + torch.optim.Adam over 9 steps with overflowing gradients injected at
    steps 2, 3 and 7 (growth interval 3): identical scale trajectory, skipped steps leave parameters, moments and the step count
    untouched, clean steps agree with torch's unscaled Adam (model/trainer.py:40,121-123; torch/amp/grad_scaler.py)."""
    optim = pkg("optim")
    torch.manual_seed(3)
    shapes = [(64, 33), (1000,), (7, 5, 3)]
    p_ref = [torch.randn(s, device="cuda").requires_grad_(True) for s in shapes]
    p_av = [p.detach().clone().requires_grad_(True) for p in p_ref]
    o_ref = torch.optim.Adam([{"params": p_ref[:2], "lr": 1e-3}, {"params": p_ref[2:], "lr": 3e-4}])
    o_av = optim.AvAdam([{"params": p_av[:2], "lr": 1e-3}, {"params": p_av[2:], "lr": 3e-4}])
    s_ref = torch.amp.GradScaler("cuda", init_scale=1024.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=3)
    s_av = optim.AvGradScaler(init_scale=1024.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=3)
    scales = []
    for it in range(9):
        gs = [torch.randn(s, device="cuda") for s in shapes]
        sc_ref = float(s_ref.scale(torch.ones((), device="cuda")))           # also initialises torch's lazy scale tensor
        sc_av = s_av.get_scale()
        assert sc_ref == sc_av, (it, sc_ref, sc_av)
        scales.append(sc_av)
        for pr, pa, g in zip(p_ref, p_av, gs):
            pr.grad = g * sc_ref; pa.grad = (g * sc_av).clone()
        if it in (2, 7):
            p_ref[1].grad[17] = float("inf"); p_av[1].grad[17] = float("inf")
        if it == 3:
            p_ref[2].grad[0, 0, 0] = float("nan"); p_av[2].grad[0, 0, 0] = float("nan")
        before = [p.detach().clone() for p in p_av]
        s_ref.step(o_ref); s_ref.update()
        s_av.step(o_av); s_av.update()
        for pr, pa, b in zip(p_ref, p_av, before):
            if it in (2, 3, 7):
                assert torch.equal(pa.detach(), b)                               # skipped
            torch.testing.assert_close(pa.detach(), pr.detach(), rtol=2e-6, atol=2e-7)
    assert scales == [1024.0, 1024.0, 1024.0, 512.0, 256.0, 256.0, 256.0, 512.0, 256.0]
    assert s_av.steps_taken() == 6
    o_av.sync_steps(s_av)
    assert all(st["step"] == 6 for st in o_av.state.values())
    assert all(int(st["step"]) == 6 for st in o_ref.state.values())
    sd = s_av.state_dict()
    assert sd["scale"] == s_ref.get_scale() and sd["_growth_tracker"] == s_ref.state_dict()["_growth_tracker"]


def test_trainer_loss_scaling_step_equals_plain_step():
    """A clean step with loss scaling on (scale 65536 folded out again inside the Adam kernel) moves the parameters like the plain step."""
    from test_step_gpu import build
    init = pkg("utils.init"); synth = pkg("dataset.synthetic"); tr = pkg("model.trainer")
    cfg = init.W2V2_TINY
    batch = synth.make_batch(2, 1.0, seed=5, ragged=True)
    a = build(cfg, "fp32")
    b0 = build(cfg, "fp32")
    b = tr.MultimodalTrainer(b0.visual_encoder, b0.audio_encoder, b0.fusion_module, b0.decoder1, b0.tokenizer, learning_rate=1e-4, device="cuda",
                             lambda_=0.1, loss_scaling=True)
    b.fixed_projection = b0.fixed_projection
    for _ in range(2):
        oa, ob = a.train_step(batch), b.train_step(batch)
        assert abs(float(oa["total"].detach()) - float(ob["total"].detach())) < 1e-5
    assert b.scaler.get_scale() == 65536.0 and b.scaler.steps_taken() == 2
    for ma, mb in ((a.audio_encoder, b.audio_encoder), (a.fusion_module, b.fusion_module), (a.decoder1, b.decoder1)):
        sa, sb = ma.state_dict(), mb.state_dict()
        for k in sa:
            tol = 2e-4 if k.endswith("k_proj.bias") else 3e-6
            assert float((sa[k].float() - sb[k].float()).abs().max()) < tol, k


@pytest.mark.parametrize("regularize", [False, True])
def test_native_layer_forward_equals_the_per_kernel_path(regularize):
    """av_w2v2_layer_fwd (csrc/w2v2_layer.hip: the seven launches of an encoder layer's forward from one native call) against the per-kernel
    path of model/w2v2.py on the same weights, input, seeds and LayerDrop draws: outputs, mid-layer average and every tensor saved for the
    backward are bit-identical; the backward then gives bit-identical gradients."""
    init = pkg("utils.init"); enc = pkg("model.encoder"); synth = pkg("dataset.synthetic"); w2 = pkg("model.w2v2")
    pkg("precision").set_precision("bf16")
    extra = dict(hidden_dropout=0.1, attention_dropout=0.1, activation_dropout=0.1, layerdrop=0.2, mask_time_prob=0.05) if regularize else {}
    outs = []
    for native in (2, 0):                                       # 2 = always (the default takes it up to NATIVE_MAX_ROWS tokens per pass), 0 = never
        w2.NATIVE_LAYER = native
        try:
            cfg = dict(init.W2V2_TINY, **extra)
            ae = enc.AudioEncoder(cfg, freeze=True).cuda()
            ae.load_state_dict(init.w2v2_state_dict(init.W2V2_TINY))
            for n, p in ae.model.named_parameters():
                p.requires_grad = any(f"encoder.layers.{i}." in n for i in range(6, 10))
            ae.train()
            batch = synth.make_batch(3, 1.2, seed=42, ragged=True)
            torch.manual_seed(7); np.random.seed(11)            # dropout / LayerDrop draws: torch generators; SpecAugment: numpy's global RNG (as HF)
            last, mid = ae(batch["audio"].cuda(), attention_mask=(batch["mask1"] != 3).cuda())
            (last.float().square().mean() + 0.5 * mid.float().square().mean()).backward()
            grads = {n: p.grad.clone() for n, p in ae.model.named_parameters() if p.grad is not None}
            outs.append((last.detach().clone(), mid.detach().clone(), grads))
        finally:
            w2.NATIVE_LAYER = 1
    (l0, m0, g0), (l1, m1, g1) = outs
    assert torch.equal(l0, l1) and torch.equal(m0, m1)
    assert g0.keys() == g1.keys() and len(g0) > 0
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k
