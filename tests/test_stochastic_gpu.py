"""Train-mode stochastic regularisers of wav2vec2 on the HIP path: Philox dropout masks (elementwise, GEMM epilogue,
attention probabilities) are pure functions of (seed, stream, index), so every check is exact against a PyTorch
reference that uses the SAME mask; LayerDrop / SpecAugment / full-model checks use fixed RNG seeds."""
import math

import numpy as np
import pytest
import torch

from conftest import pkg

pytestmark = pytest.mark.gpu


def _mask(n, p, seed, stream):
    L = pkg("_lib"); ops = pkg("ops")
    u = torch.empty(n, device="cuda")
    L.check(L.lib().av_dropout_uniform(ops.ptr(u), n, seed, stream, ops.stream()))
    # survivors are scaled by the inverse of the REALISED keep fraction: the keep test resolves p to ceil(65536 p) sixteen-bit steps
    # (av_common.h: drop_inv_keep), evaluated in fp32 like the kernels do
    thr = np.ceil(np.float32(p) * np.float32(65536.0))
    inv = float(np.float32(65536.0) / (np.float32(65536.0) - np.float32(thr)))
    return (u >= p).float() * inv, u


def test_dropout_kernel_statistics_and_determinism():
    ops = pkg("ops")
    x = torch.randn(1000, 1031, device="cuda")
    m, u = _mask(x.numel(), 0.1, 1234, 7)
    assert 0.0 <= float(u.min()) and float(u.max()) < 1.0 and abs(float(u.mean()) - 0.5) < 2e-3
    y = ops.cast_dropout(x, torch.float32, (0.1, 1234, 7))
    torch.testing.assert_close(y, x * m.view_as(x), rtol=0, atol=0)
    assert abs(float((y == 0).float().mean()) - 0.1) < 3e-3
    torch.testing.assert_close(ops.cast_dropout(x, torch.float32, (0.1, 1234, 7)), y, rtol=0, atol=0)
    assert not torch.equal(ops.cast_dropout(x, torch.float32, (0.1, 1235, 7)), y)
    assert not torch.equal(ops.cast_dropout(x, torch.float32, (0.1, 1234, 8)), y)
    yb = ops.cast_dropout(x, torch.bfloat16, (0.25, 5, 1))
    mb, _ = _mask(x.numel(), 0.25, 5, 1)
    torch.testing.assert_close(yb.float(), (x * mb.view_as(x)).to(torch.bfloat16).float(), rtol=0, atol=0)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_epilogue_dropout_matches_mask(dtype):
    ops = pkg("ops"); L = pkg("_lib")
    M, N, K = 300, 256, 128
    x = (torch.randn(M, K, device="cuda")).to(dtype); w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).to(dtype)
    b = torch.randn(N, device="cuda"); r = torch.randn(M, N, device="cuda")
    m, _ = _mask(M * N, 0.2, 99, 3)
    m = m.view(M, N)
    ref = torch.nn.functional.gelu((x.double() @ w.double().t()).float() + b) * m + r
    out = ops.linear(x, w, b, out_dtype=torch.float32, act=L.ACT_GELU, R=r, drop=(0.2, 99, 3))
    tol = dict(rtol=2e-5, atol=2e-5) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(out, ref, **tol)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])                              # fp32: tiled forward + unfused backward
@pytest.mark.parametrize("B,H,T,D", [(2, 4, 70, 64), (1, 2, 130, 128), (1, 2, 300, 64)])      # 300: the chunked T > 256 kernels with dropout
def test_attention_dropout_fwd_bwd_same_mask(B, H, T, D, dtype):
    ops = pkg("ops")
    p_, seed, stream = 0.15, 4242, 11
    qkv = torch.randn(B, T, 3, H, D, device="cuda").to(dtype)
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    scale = D ** -0.5
    o, lse = ops.attention_fwd(q, k, v, None, scale, drop=(p_, seed, stream))
    T4 = (T + 3) // 4 * 4                                        # Philox row pitch of the attention kernels: keys padded to a multiple of 4
    m, _ = _mask(B * H * T * T4, p_, seed, stream)
    m = m.view(B, H, T, T4)[..., :T].double()
    qr, kr, vr = (t.float().permute(0, 2, 1, 3).detach().clone().requires_grad_(True) for t in (q, k, v))
    P = torch.softmax((qr.double() @ kr.double().transpose(2, 3)) * scale, -1)
    ref = ((P * m) @ vr.double()).float()
    ft = 3e-2 if dtype == torch.bfloat16 else 1e-3
    torch.testing.assert_close(o.float().permute(0, 2, 1, 3), ref, rtol=ft, atol=ft)
    torch.testing.assert_close(lse, torch.logsumexp((qr.double() @ kr.double().transpose(2, 3)) * scale, -1).float(), rtol=2e-2, atol=5e-2)
    do = torch.randn(B, T, H, D, device="cuda").to(dtype)
    ref.backward(do.float().permute(0, 2, 1, 3))
    dq, dk, dv = torch.empty_like(q.contiguous()), torch.empty_like(k.contiguous()), torch.empty_like(v.contiguous())
    ops.attention_bwd(q, k, v, do, dq, dk, dv, None, scale, o=o, lse=lse, drop=(p_, seed, stream))
    tol = 6e-2 if dtype == torch.bfloat16 else 2e-3
    for a, g in ((dq, qr.grad), (dk, kr.grad), (dv, vr.grad)):
        torch.testing.assert_close(a.float().permute(0, 2, 1, 3), g, rtol=tol, atol=tol)


@pytest.mark.parametrize("B,H,T", [(2, 4, 70), (3, 16, 199), (1, 2, 256), (2, 3, 33)])
def test_attention_precomputed_dropout_bits_equal_generated_masks(B, H, T):
    """Keep bits evaluated once (av_attention_dropmask) and read by the whole-sequence forward and by both phases of the backward must
    reproduce the kernels that generate the Philox masks themselves: bit for bit in the backward (same arithmetic), and in the forward
    - where the keep-bit kernel folds 1 / (1 - p) into the normalisation and the scale into the exponent, i.e. rounds differently - to
    bf16 resolution on random data plus an EXACT count of kept keys per residue class on a probe with uniform probabilities."""
    ops = pkg("ops")
    D = 64
    qkv = torch.randn(B, T, 3, H, D, device="cuda").to(torch.bfloat16)
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    klen = torch.tensor([T, max(1, T - 9), max(1, T // 2)][:B], device="cuda", dtype=torch.int32)
    dr = (0.2, 99, 5)
    assert ops.attention_mask_shape_ok(q.dtype, B, T, T, D)
    mask = ops.attention_dropmask(B, H, T, T, dr, q.device)
    assert mask.shape == (B, H, (T + 15) // 16, 64)
    o, lse = ops.attention_fwd(q, k, v, klen, D ** -0.5, drop=dr, drop_mask=mask)
    o2, lse2 = ops.attention_fwd(q, k, v, klen, D ** -0.5, drop=dr)
    torch.testing.assert_close(o.float(), o2.float(), rtol=2e-2, atol=5e-3)       # two bf16 roundings of |o| <= 1 values
    torch.testing.assert_close(lse, lse2, rtol=1e-5, atol=1e-5)
    # mask identity: zero queries give uniform probabilities 1 / klen, V[key][d] = [key % 64 == d] turns the output into
    # (kept keys of residue class d) / (klen (1 - p)): an integer after scaling, equal in both paths iff the masks are the same
    qz = torch.zeros_like(q)
    vp = torch.zeros(B, T, H, D, device="cuda")
    vp[:, torch.arange(T), :, torch.arange(T) % D] = 1.0
    vp = vp.to(torch.bfloat16)
    cnt = []
    for m in (mask, None):
        oz, _ = ops.attention_fwd(qz, k, vp, klen, D ** -0.5, drop=dr, drop_mask=m)
        c = oz.float() * klen.view(B, 1, 1, 1).float() * (1.0 - dr[0])
        assert (c - c.round()).abs().max() < 0.05
        cnt.append(c.round())
    assert torch.equal(cnt[0], cnt[1]) and cnt[0].sum() > 0
    do = torch.randn(B, T, H, D, device="cuda").to(torch.bfloat16)
    outs = []
    for m in (mask, None):
        d = torch.zeros_like(qkv)
        ops.attention_bwd(q, k, v, do, d[:, :, 0], d[:, :, 1], d[:, :, 2], klen, D ** -0.5, o=o, lse=lse, drop=dr, drop_mask=m)
        outs.append(d)
    assert torch.equal(outs[0], outs[1])


def _audio(cfg_extra, precision):
    init = pkg("utils.init"); enc = pkg("model.encoder"); synth = pkg("dataset.synthetic")
    pkg("precision").set_precision(precision)
    cfg = dict(init.W2V2_TINY, **cfg_extra)
    ae = enc.AudioEncoder(cfg, freeze=True).cuda()
    ae.load_state_dict(init.w2v2_state_dict(init.W2V2_TINY))
    for n, p in ae.model.named_parameters():
        p.requires_grad = any(f"encoder.layers.{i}." in n for i in range(6, 10))
    batch = synth.make_batch(3, 1.2, seed=42, ragged=True)
    return ae, batch["audio"].cuda(), (batch["mask1"] != 3).cuda()


def test_eval_mode_ignores_stochastic_knobs_and_train_mode_uses_them():
    knobs = dict(hidden_dropout=0.1, attention_dropout=0.1, activation_dropout=0.1, feat_proj_dropout=0.1, layerdrop=0.2,
                 mask_time_prob=0.3, mask_time_length=5, mask_time_min_masks=1)
    ae, wav, mask = _audio(knobs, "bf16")
    ae0, _, _ = _audio({}, "bf16")
    ae.eval(); ae0.eval()
    with torch.no_grad():
        a, _ = ae(wav, mask); b, _ = ae0(wav, mask)
    assert torch.equal(a, b)                                   # eval: deterministic, identical to the knob-free model
    ae.train()
    torch.manual_seed(1); np.random.seed(1)
    t1, _ = ae(wav, mask)
    torch.manual_seed(1); np.random.seed(1)
    t2, _ = ae(wav, mask)
    torch.manual_seed(2); np.random.seed(2)
    t3, _ = ae(wav, mask)
    assert torch.equal(t1, t2) and not torch.equal(t1, t3) and not torch.equal(t1, a)
    assert torch.isfinite(t1).all()
    (t1.float() ** 2).mean().backward()
    g = [p.grad for p in ae.parameters() if p.requires_grad]
    assert all(x is not None and torch.isfinite(x).all() for x in g)


def test_train_mode_gradients_match_directional_derivative():
    """fp32: with the RNG re-seeded before every forward the masks are identical, so the stochastic model is a fixed
    differentiable function; its hand-written backward must match a central difference."""
    knobs = dict(hidden_dropout=0.15, attention_dropout=0.0, activation_dropout=0.1, feat_proj_dropout=0.1, layerdrop=0.15,
                 mask_time_prob=0.2, mask_time_length=4, mask_time_min_masks=1)
    ae, wav, mask = _audio(knobs, "fp32")
    ae.train()
    g = torch.Generator().manual_seed(0)

    def loss():
        torch.manual_seed(5); np.random.seed(5)
        last, mid = ae(wav, mask)
        return (last * wl).sum() + (mid * wm).sum()
    torch.manual_seed(5); np.random.seed(5)
    with torch.no_grad():
        last, mid = ae(wav, mask)
    wl = torch.randn(last.shape, generator=g).cuda(); wm = torch.randn(mid.shape, generator=g).cuda()
    L0 = loss(); L0.backward()
    params = [p for p in ae.parameters() if p.requires_grad]
    dirs = [torch.randn(p.shape, generator=g).cuda() * 1e-3 for p in params]
    analytic = sum(float((p.grad * d).sum()) for p, d in zip(params, dirs) if p.grad is not None)   # LayerDrop-skipped layers: no grad
    with torch.no_grad():
        for p, d in zip(params, dirs):
            p.add_(d)
        lp = float(loss())
        for p, d in zip(params, dirs):
            p.sub_(2 * d)
        lm = float(loss())
    numeric = (lp - lm) / 2.0
    print("directional derivative: analytic", analytic, "numeric", numeric)
    assert abs(analytic - numeric) < 2e-2 * max(1.0, abs(numeric))


def test_specaugment_mask_matches_hf_function():
    tm = pytest.importorskip("transformers.models.wav2vec2.modeling_wav2vec2")
    w2 = pkg("model.w2v2")
    for seed, (B, T, prob, ln, lens, mn) in enumerate([(3, 49, 0.3, 5, [49, 30, 49], 2), (2, 199, 0.05, 10, [199, 150], 2), (4, 60, 0.5, 3, [60, 10, 2, 33], 0)]):
        np.random.seed(seed); mine = w2.specaugment_mask(B, T, prob, ln, lens, mn)
        am = torch.zeros(B, T, dtype=torch.long)
        for i, l in enumerate(lens):
            am[i, :l] = 1
        np.random.seed(seed); ref = tm._compute_mask_indices((B, T), prob, ln, attention_mask=am, min_masks=mn)
        assert (mine == ref).all()



def test_specaugment_time_and_feature_masks_vs_hf_model():
    """SpecAugment end to end (time axis hf:1272-1296 + feature axis hf:1298-1316) against the installed HF Wav2Vec2Model in train
    mode with every dropout at 0: same numpy seed => same masks (drawn in HF's order: time first, then features) => same
    last_hidden_state and hidden_states[6:10] mean within the fp32 gate."""
    tf = pytest.importorskip("transformers")
    init = pkg("utils.init"); enc = pkg("model.encoder")
    pkg("precision").set_precision("fp32")
    cfg = dict(init.W2V2_TINY, mask_time_prob=0.3, mask_time_length=4, mask_time_min_masks=2, mask_feature_prob=0.25, mask_feature_length=5,
               mask_feature_min_masks=1)
    hc = tf.Wav2Vec2Config(hidden_size=cfg["hidden_size"], num_hidden_layers=cfg["num_hidden_layers"], num_attention_heads=cfg["num_attention_heads"],
                           intermediate_size=cfg["intermediate_size"], conv_dim=tuple(cfg["conv_dim"]), conv_kernel=tuple(cfg["conv_kernel"]),
                           conv_stride=tuple(cfg["conv_stride"]), num_conv_pos_embeddings=cfg["num_conv_pos_embeddings"],
                           num_conv_pos_embedding_groups=cfg["num_conv_pos_embedding_groups"], feat_extract_norm="layer", do_stable_layer_norm=True,
                           conv_bias=True, hidden_dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, feat_proj_dropout=0.0, final_dropout=0.0,
                           layerdrop=0.0, mask_time_prob=0.3, mask_time_length=4, mask_time_min_masks=2, mask_feature_prob=0.25,
                           mask_feature_length=5, mask_feature_min_masks=1, attn_implementation="eager")
    hf = tf.Wav2Vec2Model(hc)
    sd = init.w2v2_state_dict(cfg, seed=4, prefix="")
    hf.load_state_dict(sd)
    hf.train()
    ae = enc.AudioEncoder(dict(cfg), freeze=True).cuda()
    ae.model.load_state_dict(sd)
    ae.train()
    g = torch.Generator().manual_seed(2)
    wav = torch.randn(3, 16000, generator=g) * 0.3
    am = torch.ones(3, 16000, dtype=torch.long); am[1, 12000:] = 0
    np.random.seed(11)
    with torch.no_grad():
        ref = hf(wav, attention_mask=am, output_hidden_states=True)
    ref_last = ref.last_hidden_state
    ref_mid = torch.stack(ref.hidden_states[6:10], 0).mean(0)
    np.random.seed(11)
    with torch.no_grad():
        last, mid = ae(wav.cuda(), attention_mask=am.bool().cuda())
    assert float((last.cpu() - ref_last).abs().max()) < 1e-3
    assert float((mid.cpu() - ref_mid).abs().max()) < 1e-3
    # the masks did something: an unmasked run differs
    ae.model.cfg.update(mask_time_prob=0.0, mask_feature_prob=0.0)
    with torch.no_grad():
        last0, _ = ae(wav.cuda(), attention_mask=am.bool().cuda())
    assert float((last0 - last).abs().max()) > 1e-2


def test_dropout_uniform_generator_statistics():
    """The counter-based generator behind every dropout mask (av_common.h: drop_uniform4): uniform 16-bit marginals, no correlation
    between neighbouring elements, between the two words of a group, or between streams / seeds at the same positions."""
    L = pkg("_lib"); ops = pkg("ops")
    n = 1 << 22

    def draw(seed, stream):
        u = torch.empty(n, device="cuda", dtype=torch.float32)
        L.check(L.lib().av_dropout_uniform(ops.ptr(u), n, seed, stream, ops.stream()))
        return u.double()

    u = draw(1234, 5)
    assert float(u.min()) >= 0.0 and float(u.max()) < 1.0
    assert abs(float(u.mean()) - 0.5) < 1e-3 and abs(float(u.var()) - 1.0 / 12.0) < 1e-3
    hist = torch.histc(u.float(), bins=256, min=0.0, max=1.0).double()
    chi2 = float(((hist - n / 256) ** 2 / (n / 256)).sum())
    assert 150 < chi2 < 400, chi2                                 # 255 degrees of freedom: mean 255, sd 22.6

    def corr(a, b):
        a = a - a.mean(); b = b - b.mean()
        return float((a * b).mean() / (a.std() * b.std()))
    for lag in (1, 2, 3, 4, 5, 8, 199, 1024):                      # inside a group of four, across groups, across rows
        assert abs(corr(u[:-lag], u[lag:])) < 4e-3, lag
    for other in (draw(1234, 6), draw(1235, 5), draw(1234, 5 + 8), draw(99, 5)):
        assert abs(corr(u, other)) < 4e-3
        assert abs(float(((u >= 0.1) & (other >= 0.1)).double().mean()) - 0.81) < 2e-3      # joint keep rate of two p = 0.1 masks
    # the low 16-bit lattice: every uniform is k / 65536
    assert float((u * 65536 - torch.round(u * 65536)).abs().max()) == 0.0
