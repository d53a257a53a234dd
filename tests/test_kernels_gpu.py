"""GPU parity of the individual HIP kernels (through the C-ABI) against plain PyTorch fp64/fp32 references."""
import math
import os
import subprocess
import sys

import numpy as np

import pytest
import torch

from conftest import pkg

pytestmark = pytest.mark.gpu

ops = None
L = None


@pytest.fixture(scope="module", autouse=True)
def _load():
    global ops, L
    ops = pkg("ops")
    L = pkg("_lib")
    L.lib()
    assert torch.cuda.is_available()
    torch.manual_seed(0)


def _tol(dtype):
    return dict(rtol=2e-5, atol=2e-5) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2)


def _rand(*shape, dtype=torch.float32, scale=1.0):
    return (torch.randn(*shape, device="cuda", dtype=torch.float32) * scale).to(dtype)


def _ref_mm(a, b):
    return (a.double() @ b.double()).float()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (100, 70, 50), (257, 129, 95), (64, 2048, 512), (300, 64, 64), (1, 800, 1024),
                                   (199, 1024, 1024), (513, 48, 40)])
def test_linear_shapes(dtype, M, N, K):
    x = _rand(M, K, dtype=dtype); w = _rand(N, K, dtype=dtype, scale=1 / math.sqrt(K)); b = _rand(N)
    y = ops.linear(x, w, b, out_dtype=torch.float32)
    ref = _ref_mm(x, w.t()) + b
    torch.testing.assert_close(y, ref, **_tol(dtype))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_mfma_layout_identity_asymmetric(dtype):
    """A = I with an asymmetric B catches swapped row/col maps (guide §3)."""
    n = 128
    a = torch.eye(n, device="cuda").to(dtype)
    w = (torch.arange(n * n, device="cuda", dtype=torch.float32).reshape(n, n) % 251 - 125).to(dtype)  # w[n][k]
    y = ops.linear(a, w, None, out_dtype=torch.float32)
    torch.testing.assert_close(y, w.float().t().contiguous(), rtol=0, atol=0)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_linear_epilogues(dtype):
    M, N, K = 200, 192, 160
    x = _rand(M, K, dtype=dtype); w = _rand(N, K, dtype=dtype, scale=1 / math.sqrt(K)); b = _rand(N)
    r = _rand(M, N)
    pre = torch.empty(M, N, device="cuda", dtype=dtype)
    y = ops.linear(x, w, b, out_dtype=dtype, act=L.ACT_GELU, C2=pre)
    ref_pre = _ref_mm(x, w.t()) + b
    torch.testing.assert_close(pre.float(), ref_pre, **_tol(dtype))
    torch.testing.assert_close(y.float(), torch.nn.functional.gelu(ref_pre), **_tol(dtype))
    # residual (aliasing C) + alpha
    out = r.clone()
    ops.linear(x, w, b, out=out, R=out, alpha=0.5)
    torch.testing.assert_close(out, 0.5 * _ref_mm(x, w.t()) + b + r, **_tol(dtype))
    # gelu-grad multiply
    u = _rand(M, N, dtype=dtype)
    g = ops.linear(x, w, None, out_dtype=torch.float32, act=L.ACT_MUL_GELU_GRAD, aux=u)
    uu = u.float().requires_grad_(True)
    torch.nn.functional.gelu(uu).sum().backward()
    torch.testing.assert_close(g, _ref_mm(x, w.t()) * uu.grad, **_tol(dtype))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K,p", [(200, 192, 160, 0.0), (200, 192, 160, 0.25), (600, 512, 256, 0.1), (130, 70, 64, 0.1)])
def test_gelu_gradient_factor_epilogues(dtype, M, N, K, p):
    """AV_ACT_GELU_GF writes C = gelu(v) o m and C2 = gelu'(v) o m (m = the dropout multiplier of (seed, stream, m * ldc + n), the
    same mask AV_ACT_GELU + dropout applies); AV_ACT_MUL_AUX multiplies a product by such a saved factor.  Together they equal the
    recomputing pair AV_ACT_GELU / AV_ACT_MUL_GELU_GRAD with the same dropout triple (hf:565-572 and its backward)."""
    x = _rand(M, K, dtype=dtype); w = _rand(N, K, dtype=dtype, scale=1 / math.sqrt(K)); b = _rand(N)
    drop = (p, 4242, 5) if p > 0 else None
    gf = torch.empty(M, N, device="cuda", dtype=dtype)
    y = ops.linear(x, w, b, out_dtype=dtype, act=L.ACT_GELU_GF, C2=gf, drop=drop)
    pre = torch.empty(M, N, device="cuda", dtype=dtype)
    y_ref = ops.linear(x, w, b, out_dtype=dtype, act=L.ACT_GELU, C2=pre, drop=drop)
    torch.testing.assert_close(y.float(), y_ref.float(), **_tol(dtype))
    ref_pre = (_ref_mm(x, w.t()) + b).requires_grad_(True)
    torch.nn.functional.gelu(ref_pre).sum().backward()
    keep = torch.ones(M, N, device="cuda", dtype=torch.bool)
    if p > 0:                                                   # the mask itself: element index = m * ldc + n
        uni = torch.empty(M * N, device="cuda", dtype=torch.float32)
        L.check(L.lib().av_dropout_uniform(ops.ptr(uni), M * N, drop[1], drop[2], ops.stream()))
        keep = (uni >= p).view(M, N)
        torch.testing.assert_close(y.float(), torch.nn.functional.gelu(ref_pre.detach()) * keep.float() / (1.0 - p), **_tol(dtype))
    mult = keep.float() / (1.0 - p)
    torch.testing.assert_close(gf.float(), ref_pre.grad * mult, **_tol(dtype))
    if p > 0:
        assert 0.5 * p < 1.0 - keep.float().mean().item() < 1.5 * p
    # backward: dY W o gf == (dY W o gelu'(pre)) o mask
    dy = _rand(M, K, dtype=dtype)
    du = ops.linear(dy, w, None, out_dtype=torch.float32, act=L.ACT_MUL_AUX, aux=gf)
    du_ref = ops.linear(dy, w, None, out_dtype=torch.float32, act=L.ACT_MUL_GELU_GRAD, aux=pre, drop=drop)
    t = _tol(dtype)
    torch.testing.assert_close(du, du_ref, rtol=max(t["rtol"], 2e-2 if dtype == torch.bfloat16 else 0), atol=t["atol"])
    torch.testing.assert_close(du, _ref_mm(dy, w.t()) * gf.float(), **_tol(dtype))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(130, 96, 200), (64, 64, 64), (257, 1024, 129), (200, 70, 33)])
def test_matmul_nn_tn(dtype, M, N, K):
    a = _rand(M, K, dtype=dtype); b = _rand(K, N, dtype=dtype, scale=1 / math.sqrt(K))
    torch.testing.assert_close(ops.matmul_nn(a, b, out_dtype=torch.float32), _ref_mm(a, b), **_tol(dtype))
    at = _rand(K, M, dtype=dtype, scale=1 / math.sqrt(K))           # [K,M]^T @ [K,N]
    torch.testing.assert_close(ops.matmul_tn(at, b), _ref_mm(at.t(), b), **_tol(dtype))
    acc = _rand(M, N)
    ref = acc + 2.0 * _ref_mm(at.t(), b)
    ops.matmul_tn(at, b, out=acc, alpha=2.0, accumulate=True)
    torch.testing.assert_close(acc, ref, **_tol(dtype))


@pytest.mark.parametrize("K,M,N", [(4200, 2048, 1024), (12736, 1024, 1024), (3000, 1024, 4096), (6368, 3072, 1024)])
def test_matmul_tn_large_split_k(K, M, N):
    """dW-shaped products at the step's sizes (both operands k-major, K = tokens split over the batch: ragged last K slice, ragged last
    K-tile, row-strided operand views, accumulation into an existing gradient)."""
    a = _rand(K, M + 64, dtype=torch.bfloat16, scale=1.0)[:, 32:32 + M]                 # row-strided view, 16-byte aligned
    b = _rand(K, N, dtype=torch.bfloat16, scale=1 / math.sqrt(K))
    ref = _ref_mm(a.t(), b)
    tol = dict(rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(ops.matmul_tn(a, b), ref, **tol)
    acc = _rand(M, N)
    exp = acc + 0.5 * ref
    ops.matmul_tn(a, b, out=acc, alpha=0.5, accumulate=True)
    torch.testing.assert_close(acc, exp, **tol)


def test_kmajor_products_on_the_forced_8_phase_form():
    """The k-major form of the 8-phase kernel (gemm_fast.hip: gemm_nt_bf16_v4_kernel<false, true>) takes dW products of >= 32 tiles by default;
    AVAMD_GEMM_KM8=2 (read once per process) sends every >= 256 x 256 one to it: the ragged / strided / exact-integer cases above again in
    a child process under that switch."""
    if os.environ.get("AVAMD_GEMM_KM8") == "2":
        pytest.skip("already the forced run")
    env = dict(os.environ, AVAMD_GEMM_KM8="2")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-k",
                        "test_fast_gemm_kmajor_operands or test_matmul_tn_large_split_k or test_matmul_nn_tn"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_batched_strided_gemm(dtype):
    """QK^T-style batched product with head-strided operands."""
    Bz, T, H, D = 3, 70, 4, 32
    qkv = _rand(Bz, T, 3 * H * D, dtype=dtype)
    out = torch.empty(Bz * H, T, T, device="cuda", dtype=torch.float32)
    for b in range(Bz):     # batch over heads inside one call per batch item
        ops.gemm(qkv, qkv, out, M=T, N=T, K=D, lda=3 * H * D, ldb=3 * H * D, ldc=T, batch=H, sA=D, sB=D, sC=T * T,
                 a_off=b * T * 3 * H * D, b_off=b * T * 3 * H * D + H * D, c_off=b * H * T * T, alpha=0.125)
    q = qkv[..., : H * D].float().view(Bz, T, H, D).transpose(1, 2)
    k = qkv[..., H * D: 2 * H * D].float().view(Bz, T, H, D).transpose(1, 2)
    ref = (q.double() @ k.double().transpose(2, 3)).float().reshape(Bz * H, T, T) * 0.125
    torch.testing.assert_close(out, ref, **_tol(dtype))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("Cin,Cout,H,stride,k,N_", [(64, 64, 12, 1, 3, 5), (64, 128, 12, 2, 3, 5), (64, 128, 12, 2, 1, 5), (128, 64, 6, 1, 3, 5),
                                                      # ResNet layer3 / layer4 shapes on the 256 x 256 8-phase kernel (bf16, M >= 4096): ragged last
                                                      # row tile (M = 4500 / 4509), one and two column tiles, strided + 1x1 forms
                                                      (128, 256, 12, 2, 3, 125), (256, 256, 6, 1, 3, 125), (256, 512, 6, 2, 3, 501),
                                                      (512, 512, 3, 1, 3, 501), (128, 256, 12, 2, 1, 125),
                                                      # ResNet layer2 shapes (N = 128) at M = 5760 / 5904: many row tiles, ragged last one
                                                      (64, 128, 24, 2, 3, 40), (128, 128, 12, 1, 3, 41)])
def test_conv2d_implicit_gemm(dtype, Cin, Cout, H, stride, k, N_):
    if dtype == torch.float32 and N_ > 5:
        pytest.skip("large shapes exercise the bf16 kernels")
    pad = k // 2
    x = _rand(N_, H, H, Cin, dtype=dtype)                       # NHWC
    w = _rand(Cout, Cin, k, k, dtype=dtype, scale=1 / math.sqrt(Cin * k * k))
    wk = w.permute(0, 2, 3, 1).contiguous().view(Cout, k * k * Cin)
    Ho = (H + 2 * pad - k) // stride + 1
    M = N_ * Ho * Ho
    out = torch.empty(M, Cout, device="cuda", dtype=torch.float32)
    nblk = (M + 127) // 128
    stats = torch.zeros(nblk, 2, Cout, device="cuda")
    conv = dict(cT=1, cH=H, cW=H, cCtot=Cin, cCin=Cin, cCoff=0, cKt=1, cKh=k, cKw=k, cSh=stride, cSw=stride, cPt=0, cPh=pad, cPw=pad,
                cOh=Ho, cOw=Ho)
    ops.gemm(x, wk, out, M=M, N=Cout, K=k * k * Cin, lda=0, ldb=k * k * Cin, ldc=Cout, a_mode=L.A_CONV2D, conv=conv, stats=stats)
    ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2).double(), w.double(), None, stride, pad).float().permute(0, 2, 3, 1).reshape(M, Cout)
    torch.testing.assert_close(out, ref, **_tol(dtype))
    torch.testing.assert_close(stats[:, 0].sum(0), out.sum(0), rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(stats[:, 1].sum(0), (out * out).sum(0), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv3d_frontend_gather(dtype):
    B, T, H = 2, 6, 20
    x = _rand(B, T, H, H, dtype=dtype)
    w = _rand(64, 1, 5, 7, 7, dtype=dtype, scale=1 / math.sqrt(245))
    Ho = (H + 6 - 7) // 2 + 1
    M = B * T * Ho * Ho
    out = torch.empty(M, 64, device="cuda", dtype=torch.float32)
    conv = dict(cT=T, cH=H, cW=H, cCtot=1, cCin=1, cCoff=0, cKt=5, cKh=7, cKw=7, cSh=2, cSw=2, cPt=2, cPh=3, cPw=3, cOh=Ho, cOw=Ho)
    ops.gemm(x, w.view(64, 245).contiguous(), out, M=M, N=64, K=245, lda=0, ldb=245, ldc=64, a_mode=L.A_CONV3D1, conv=conv)
    ref = torch.nn.functional.conv3d(x.float()[:, None].double(), w.double(), None, (1, 2, 2), (2, 3, 3)).float()
    ref = ref.permute(0, 2, 3, 4, 1).reshape(M, 64)
    torch.testing.assert_close(out, ref, **_tol(dtype))


@pytest.mark.parametrize("xdt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cols", [32, 64, 512, 1024, 800])
def test_layernorm_fwd_bwd(xdt, cols):
    rows = 77
    x = _rand(rows, cols, dtype=xdt); g = 1 + 0.1 * _rand(cols); b = 0.1 * _rand(cols)
    y, mean, rstd = ops.layernorm_fwd(x, g, b, out_dtype=torch.float32, save_stats=True)
    xr = x.float().requires_grad_(True); gr = g.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xr, (cols,), gr, br, 1e-5)
    torch.testing.assert_close(y, ref, rtol=1e-5, atol=1e-5)
    yg = ops.layernorm_fwd(x, g, b, out_dtype=torch.float32, act=L.ACT_GELU)
    torch.testing.assert_close(yg, torch.nn.functional.gelu(ref), rtol=1e-5, atol=1e-5)
    dy = _rand(rows, cols); dres = _rand(rows, cols)
    ref.backward(dy)
    dx, dg, db = ops.layernorm_bwd(x, dy, g, mean, rstd, dres, want_param_grads=True)
    dx2, dg2, db2, dxl = ops.layernorm_bwd(x, dy, g, mean, rstd, dres, want_param_grads=True, lp_copy=True)      # + bf16 copy of dx
    torch.testing.assert_close(dx2, dx, rtol=0, atol=0); torch.testing.assert_close(dxl, dx.to(torch.bfloat16), rtol=0, atol=0)
    assert ops.layernorm_bwd(x, dy, g, mean, rstd, dres, lp_copy=True)[1].dtype == torch.bfloat16
    torch.testing.assert_close(dx, xr.grad + dres, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(dg, gr.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(db, br.grad, rtol=1e-4, atol=1e-4)


def test_log_softmax_colsum_cast():
    x = _rand(51, 800, scale=3.0)
    y = ops.log_softmax_fwd(x)
    xr = x.clone().requires_grad_(True)
    ref = torch.log_softmax(xr, -1)
    torch.testing.assert_close(y, ref, rtol=1e-5, atol=1e-5)
    dy = _rand(51, 800)
    ref.backward(dy)
    torch.testing.assert_close(ops.log_softmax_bwd(y, dy, torch.float32), xr.grad, rtol=1e-5, atol=1e-5)
    # K-padded form (CTC head: vocabulary 800 -> 832 columns for the fast GEMM's 64-wide K step): same values, exact zeros in the padding,
    # and the column sums / dW product of the un-padded view (row stride 832) see only the 800 real columns
    dp = ops.log_softmax_bwd(y, dy, torch.bfloat16, pad_to=64)
    assert dp.shape == (51, 832) and float(dp[:, 800:].abs().max()) == 0.0
    torch.testing.assert_close(dp[:, :800].float(), xr.grad, rtol=2e-2, atol=2e-2)
    assert torch.equal(dp[:, :800], ops.log_softmax_bwd(y, dy, torch.bfloat16))
    torch.testing.assert_close(ops.colsum(dp[:, :800]), dp[:, :800].float().sum(0), rtol=1e-3, atol=1e-3)
    big = _rand(1300, 200)
    torch.testing.assert_close(ops.colsum(big), big.sum(0), rtol=1e-4, atol=1e-4)
    odd = _rand(1300, 203)                                                          # cols % 4 != 0: scalar kernel
    torch.testing.assert_close(ops.colsum(odd), odd.sum(0), rtol=1e-4, atol=1e-4)
    for rows, cols in ((1300, 200), (6368, 1024), (5, 8), (777, 4096)):           # bf16: 16-byte-load kernel (cols % 8 == 0)
        hb = _rand(rows, cols, dtype=torch.bfloat16)
        torch.testing.assert_close(ops.colsum(hb), hb.float().sum(0), rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(ops.cast(big, torch.bfloat16), big.to(torch.bfloat16), rtol=0, atol=0)
    z = _rand(1300, 200)
    torch.testing.assert_close(ops.axpby(2.0, big, 0.5, z.clone()), 2 * big + 0.5 * z)
    keep = (torch.rand(1300, device="cuda") > 0.3).to(torch.uint8)
    m = ops.mask_rows_(big.clone(), keep)
    torch.testing.assert_close(m, big * keep[:, None].float())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,Tq,Tk,D,masked", [(2, 4, 49, 49, 16, False), (3, 16, 199, 199, 64, True), (2, 4, 100, 100, 128, False),
                                                (2, 4, 25, 25, 128, False), (1, 2, 300, 300, 64, True), (2, 3, 70, 130, 32, True),
                                                # D = 64, T <= 256 (bf16): whole-sequence kernels of attention_short.hip
                                                (2, 4, 49, 49, 64, True), (2, 2, 256, 256, 64, True), (1, 3, 130, 130, 64, False),
                                                (2, 2, 70, 130, 64, True), (2, 2, 130, 70, 64, True), (1, 1, 1, 1, 64, False),
                                                # D = 64, T > 256 (bf16): chunked forward with an online softmax (config 3: T_enc = 749)
                                                (2, 3, 749, 749, 64, True), (1, 2, 130, 600, 64, False), (1, 2, 257, 257, 64, True)])
def test_attention_fwd_bwd(dtype, B, H, Tq, Tk, D, masked):
    packed = Tq == Tk
    if packed:
        qkv = _rand(B, Tq, 3, H, D, dtype=dtype)
        q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    else:
        q = _rand(B, Tq, H, D, dtype=dtype); kv = _rand(B, Tk, 2, H, D, dtype=dtype)
        k, v = kv[:, :, 0], kv[:, :, 1]
    klen = None
    if masked:
        klen = torch.tensor([Tk, max(1, Tk // 2), max(1, Tk - 3)][:B], device="cuda", dtype=torch.int32)
    scale = D ** -0.5
    o, lse = ops.attention_fwd(q, k, v, klen, scale)
    qr, kr, vr = (t.float().permute(0, 2, 1, 3).detach().clone().requires_grad_(True) for t in (q, k, v))
    s = (qr.double() @ kr.double().transpose(2, 3)) * scale
    if masked:
        keep = torch.arange(Tk, device="cuda")[None, :] < klen[:, None]
        s = s.masked_fill(~keep[:, None, None, :], float("-inf"))
    ref = (torch.softmax(s, -1) @ vr.double()).float()
    tol = dict(rtol=1e-4, atol=1e-4) if dtype == torch.float32 else dict(rtol=3e-2, atol=3e-2)
    torch.testing.assert_close(o.float().permute(0, 2, 1, 3), ref, **tol)
    torch.testing.assert_close(lse, torch.logsumexp(s, -1).float(), rtol=1e-3 if dtype == torch.float32 else 2e-2, atol=1e-3 if dtype == torch.float32 else 5e-2)
    do = _rand(B, Tq, H, D, dtype=dtype)
    ref.backward(do.float().permute(0, 2, 1, 3))
    dq = torch.empty(B, Tq, H, D, device="cuda", dtype=dtype); dk = torch.empty(B, Tk, H, D, device="cuda", dtype=dtype)
    dv = torch.empty(B, Tk, H, D, device="cuda", dtype=dtype)
    ops.attention_bwd(q, k, v, do, dq, dk, dv, klen, scale)
    if dtype == torch.bfloat16:      # fused flash-style backward must agree as well
        dq2, dk2, dv2 = torch.empty_like(dq), torch.empty_like(dk), torch.empty_like(dv)
        ops.attention_bwd(q, k, v, do, dq2, dk2, dv2, klen, scale, o=o, lse=lse)
        for a_, b_, r_ in ((dq2, dq, qr.grad), (dk2, dk, kr.grad), (dv2, dv, vr.grad)):
            torch.testing.assert_close(a_.float().permute(0, 2, 1, 3), r_, rtol=5e-2, atol=5e-2)
    btol = dict(rtol=2e-4, atol=2e-4) if dtype == torch.float32 else dict(rtol=5e-2, atol=5e-2)
    torch.testing.assert_close(dq.float().permute(0, 2, 1, 3), qr.grad, **btol)
    torch.testing.assert_close(dk.float().permute(0, 2, 1, 3), kr.grad, **btol)
    torch.testing.assert_close(dv.float().permute(0, 2, 1, 3), vr.grad, **btol)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (6368, 1024, 1024), (300, 4096, 1024), (1000, 3072, 1024), (257, 800, 1024),
                                   (6368, 1024, 4096), (199, 512, 1536), (130, 200, 192),
                                   # 256 x 256 8-phase kernel: one round; a short last round cut into quadrant jobs (18 x 17 = 306 tiles, ragged M and
                                   # N); a remainder too long for quadrant jobs (15 x 32 = 480 tiles); one K-tile; an odd number of K-tiles
                                   (12736, 1024, 1024), (4400, 4328, 192), (3800, 8192, 128), (2048, 512, 64), (1500, 768, 320)])
def test_fast_bf16_gemm(M, N, K):
    """LDS-DMA fast path (bf16, K-contiguous operands) against fp64, with every epilogue."""
    dtype = torch.bfloat16
    x = _rand(M, K, dtype=dtype); w = _rand(N, K, dtype=dtype, scale=1 / math.sqrt(K)); b = _rand(N); r = _rand(M, N)
    ref = _ref_mm(x, w.t())
    torch.testing.assert_close(ops.linear(x, w, None, out_dtype=torch.float32), ref, rtol=2e-2, atol=2e-2)
    pre = torch.empty(M, N, device="cuda", dtype=dtype)
    y = ops.linear(x, w, b, out_dtype=dtype, act=L.ACT_GELU, C2=pre)
    torch.testing.assert_close(pre.float(), ref + b, rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(y.float(), torch.nn.functional.gelu(ref + b), rtol=2e-2, atol=2e-2)
    out = r.clone()
    ops.linear(x, w, b, out=out, R=out, alpha=0.5)
    torch.testing.assert_close(out, 0.5 * ref + b + r, rtol=2e-2, atol=2e-2)
    u = _rand(M, N, dtype=dtype)
    g = ops.linear(x, w, None, out_dtype=dtype, act=L.ACT_MUL_GELU_GRAD, aux=u)
    uu = u.float().requires_grad_(True)
    torch.nn.functional.gelu(uu).sum().backward()
    torch.testing.assert_close(g.float(), ref * uu.grad, rtol=3e-2, atol=3e-2)


def test_fast_layout_identity_asymmetric():
    n = 256
    a = torch.eye(n, device="cuda").to(torch.bfloat16)
    w = (torch.arange(n * n, device="cuda", dtype=torch.float32).reshape(n, n) % 251 - 125).to(torch.bfloat16)
    y = ops.linear(a, w, None, out_dtype=torch.float32)
    torch.testing.assert_close(y, w.float().t().contiguous(), rtol=0, atol=0)


def test_transpose_and_fast_nn_tn():
    a = _rand(6368, 1024, dtype=torch.bfloat16); w = _rand(1024, 4096, dtype=torch.bfloat16, scale=1 / 32)
    torch.testing.assert_close(ops.transpose(a).float(), a.float().t().contiguous(), rtol=0, atol=0)
    tp = ops.transpose(a[:999], pad_to=64)
    assert tp.shape == (1024, 1024) and float(tp[:, 999:].abs().max()) == 0.0
    torch.testing.assert_close(ops.matmul_nn(a, w, out_dtype=torch.float32), _ref_mm(a, w), rtol=2e-2, atol=2e-2)
    dy = _rand(6368, 4096, dtype=torch.bfloat16, scale=0.05)
    torch.testing.assert_close(ops.matmul_tn(dy, a), _ref_mm(dy.t(), a), rtol=2e-2, atol=5e-2)


def test_conv3d_front_fast_kernel():
    B, T, H = 2, 7, 96
    x = torch.rand(B, T, H, H, device="cuda")
    w = torch.randn(64, 1, 5, 7, 7, device="cuda") / math.sqrt(245)
    wk = torch.zeros(64, 36, 8, device="cuda"); wk[:, :35, :7] = w.reshape(64, 35, 7)
    wk = wk.reshape(64, 288).to(torch.bfloat16).contiguous()
    Ho = H // 2
    y = torch.empty(B * T * Ho * Ho, 64, device="cuda", dtype=torch.bfloat16)
    nblk = B * T * (Ho // 8) * (Ho // 16)
    stats = torch.zeros(nblk, 2, 64, device="cuda")
    L.check(L.lib().av_conv3d_front(ops.ptr(x), ops.ptr(wk), ops.ptr(y), ops.ptr(stats), B, T, H, H, ops.stream()))
    xr = x.to(torch.bfloat16).double()[:, None]
    ref = torch.nn.functional.conv3d(xr, w.to(torch.bfloat16).double(), None, (1, 2, 2), (2, 3, 3)).float()
    ref = ref.permute(0, 2, 3, 4, 1).reshape(-1, 64)
    torch.testing.assert_close(y.float(), ref, rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(stats[:, 0].sum(0), ref.sum(0), rtol=2e-3, atol=0.5)
    torch.testing.assert_close(stats[:, 1].sum(0), (ref * ref).sum(0), rtol=2e-3, atol=0.5)


@pytest.mark.parametrize("Kk,M,N", [(100, 136, 200), (64, 128, 128), (777, 264, 72), (6368, 1024, 1024), (199, 4096, 1024), (777, 264, 520),
                                     (1300, 1000, 264), (12736, 520, 1024)])
def test_fast_gemm_kmajor_operands(Kk, M, N):
    """k-major operands read through ds_read_b64_tr_b16 (dW = dY^T X with both operands [tokens][features], dX = dY W with
    W [out][in]): ragged K, ragged tiles, row-strided views, and an exact integer check that would expose a transposed tile."""
    dt_ = torch.bfloat16
    at = _rand(Kk, M, dtype=dt_, scale=1 / math.sqrt(Kk)); b = _rand(Kk, N, dtype=dt_)
    torch.testing.assert_close(ops.matmul_tn(at, b), _ref_mm(at.t(), b), rtol=2e-2, atol=3e-2)
    wide = _rand(Kk, 3 * M, dtype=dt_, scale=1 / math.sqrt(Kk))                     # row-strided view (a q/k/v slice of a packed tensor)
    torch.testing.assert_close(ops.matmul_tn(wide[:, M:2 * M], b), _ref_mm(wide[:, M:2 * M].t(), b), rtol=2e-2, atol=3e-2)
    # exact: small integers, asymmetric operands
    ai = (torch.arange(Kk * M, device="cuda").reshape(Kk, M) % 7 - 3).to(dt_)
    bi = (torch.arange(Kk * N, device="cuda").reshape(Kk, N) % 5 - 2).to(dt_)
    torch.testing.assert_close(ops.matmul_tn(ai, bi), (ai.float().t() @ bi.float()), rtol=0, atol=0)
    acc = _rand(M, N)                                                                 # alpha / accumulate (through av_sum_slices when K is split)
    ref = acc + 0.5 * (ai.float().t() @ bi.float())
    ops.matmul_tn(ai, bi, out=acc, alpha=0.5, accumulate=True)
    torch.testing.assert_close(acc, ref, rtol=1e-5, atol=1e-3)
    if True:
        K2 = (Kk + 63) // 64 * 64                                                     # dX form: A row-major [M2, K2], B k-major [K2, N]
        a2 = (torch.arange(300 * K2, device="cuda").reshape(300, K2) % 7 - 3).to(dt_)
        b2 = (torch.arange(K2 * N, device="cuda").reshape(K2, N) % 5 - 2).to(dt_)
        torch.testing.assert_close(ops.matmul_nn(a2, b2, out_dtype=torch.float32), a2.float() @ b2.float(), rtol=0, atol=0)


@pytest.mark.parametrize("n_img,H,W", [(5, 24, 24), (3, 10, 7), (1, 3, 3), (40, 24, 24)])
def test_conv3x3_c64_weights_stationary(n_img, H, W):
    """conv3x3_c64.hip (ResNet layer1 shape: 3x3, stride 1, pad 1, 64 -> 64, NHWC bf16) against torch conv2d in fp64 on the same
    bf16 operands, with the BatchNorm partial sums; sizes cover ragged last tiles, images smaller than a window and many tiles
    per workgroup."""
    x = _rand(n_img, H, W, 64, dtype=torch.bfloat16)
    w = _rand(64, 64, 3, 3, dtype=torch.bfloat16, scale=1 / 24.0)
    wk = w.permute(0, 2, 3, 1).reshape(64, 576).contiguous()                          # [Cout][(ky*3+kx)*64 + c]
    M = n_img * H * W
    y = torch.empty(M, 64, device="cuda", dtype=torch.bfloat16)
    nblk = (M + 255) // 256
    stats = torch.empty(nblk, 2, 64, device="cuda")
    L.check(L.lib().av_conv3x3_c64(ops.ptr(x), ops.ptr(wk), ops.ptr(y), ops.ptr(stats), n_img, H, W, None, None, None, ops.stream()), "av_conv3x3_c64")
    ref = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double(), padding=1).permute(0, 2, 3, 1).reshape(M, 64)
    torch.testing.assert_close(y.double(), ref, rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(stats[:, 0].sum(0).double(), ref.sum(0), rtol=1e-3, atol=2e-2)
    torch.testing.assert_close(stats[:, 1].sum(0).double(), (ref * ref).sum(0), rtol=1e-3, atol=2e-2)
    # exact: small integers (any mis-addressed tap or seam would show)
    xi = (torch.arange(M * 64, device="cuda").reshape(n_img, H, W, 64) % 5 - 2).to(torch.bfloat16)
    wi = (torch.arange(64 * 576, device="cuda").reshape(64, 64, 3, 3) % 3 - 1).to(torch.bfloat16)
    wki = wi.permute(0, 2, 3, 1).reshape(64, 576).contiguous()
    L.check(L.lib().av_conv3x3_c64(ops.ptr(xi), ops.ptr(wki), ops.ptr(y), None, n_img, H, W, None, None, None, ops.stream()), "av_conv3x3_c64")
    refi = torch.nn.functional.conv2d(xi.double().permute(0, 3, 1, 2), wi.double(), padding=1).permute(0, 2, 3, 1).reshape(M, 64)
    torch.testing.assert_close(y.double(), refi.to(torch.bfloat16).double(), rtol=0, atol=0)
    # fused input activation (BatchNorm-apply + PReLU on the staged window) == av_bn_act followed by the plain convolution, bit for bit
    sc = torch.rand(64, device="cuda") + 0.5; sh = _rand(64); sl = torch.rand(64, device="cuda") * 0.5
    a = torch.empty_like(x)
    L.check(L.lib().av_bn_act(ops.ptr(x), ops.ptr(sc), ops.ptr(sh), None, None, None, ops.ptr(sl), ops.ptr(a), 1, x.numel(), 64, ops.stream()), "av_bn_act")
    y_ref = torch.empty_like(y); y_fused = torch.empty_like(y); st_ref = torch.empty_like(stats); st_fused = torch.empty_like(stats)
    L.check(L.lib().av_conv3x3_c64(ops.ptr(a), ops.ptr(wk), ops.ptr(y_ref), ops.ptr(st_ref), n_img, H, W, None, None, None, ops.stream()), "av_conv3x3_c64")
    L.check(L.lib().av_conv3x3_c64(ops.ptr(x), ops.ptr(wk), ops.ptr(y_fused), ops.ptr(st_fused), n_img, H, W, ops.ptr(sc), ops.ptr(sh), ops.ptr(sl),
                                   ops.stream()), "av_conv3x3_c64")
    assert torch.equal(y_fused, y_ref) and torch.equal(st_fused, st_ref)


def test_ctc_greedy_on_device():
    """decode.hip against the host restatement of the reference's decode (argmax path, collapse repeats, drop blanks), with ties
    (first maximal index) and per-item lengths."""
    bs = pkg("beam_search")
    B, T, V = 5, 37, 800
    lp = torch.log_softmax(_rand(B, T, V) * 3, -1)
    lp[0, 3, 10] = lp[0, 3].max() + 1; lp[0, 4, 10] = lp[0, 4].max() + 1          # a repeat to collapse
    lp[1, 5, 7] = 5.0; lp[1, 5, 2] = 5.0                                            # a tie: index 2 wins
    lp[2, :, 3] = 9.0                                                               # all blank (blank = 3)
    want = [bs.simple_beam_search(lp[i], 5, 3) for i in range(B)]
    assert bs.greedy_batch(lp, 3) == want
    lens = torch.tensor([37, 20, 5, 0, 11], device="cuda")
    want_l = [bs.simple_beam_search(lp[i, : int(lens[i])], 5, 3) if int(lens[i]) else [] for i in range(B)]
    assert bs.greedy_batch(lp, 3, lens) == want_l
    assert bs.greedy_batch(lp.cpu(), 3) == want                                     # host tensors keep the old path


@pytest.mark.parametrize("T,B", [(1, 1), (9, 3), (6, 16), (4, 17), (5, 50), (12, 64), (100, 64), (7, 70), (5, 130), (3, 200)])
def test_persistent_lstm_matches_step_kernels(T, B):
    """lstm_persistent.hip (one launch per layer, coherent hand-off between workgroups, LDS-DMA streaming) against the per-step
    kernels of lstm.hip on the same buffers: forward h / c / gates, backward dgates / dc; B = 3 .. 50 exercise 1, 2 and 4 row groups
    with ragged last tiles, B > 64 the row-tile loop inside a workgroup."""
    H = 512
    dt_ = torch.bfloat16
    gx = _rand(T, B, 2, 4 * H)
    whh = _rand(2, 4 * H, H, dtype=dt_, scale=1 / 22.0)
    whhT = whh.transpose(1, 2).contiguous()
    dout = _rand(B, T, 2 * H)
    st = ops.stream()

    def run(persistent):
        hseq = torch.zeros(T, B, 2 * H, device="cuda", dtype=dt_); cseq = torch.zeros(T, B, 2, H, device="cuda")
        gates = torch.zeros(T, B, 2, 4 * H, device="cuda", dtype=dt_); dg = torch.zeros_like(gates); dc = torch.zeros(2, B, H, device="cuda")
        cnt = torch.zeros(L.LSTM_COUNTER_INTS, dtype=torch.int32, device="cuda")
        if persistent:
            L.check(L.lib().av_lstm_fwd_layer(ops.ptr(gx), ops.ptr(whh), ops.ptr(hseq), ops.ptr(cseq), ops.ptr(gates), None, ops.ptr(cnt), T, B, H, st), "fwd")
            L.check(L.lib().av_lstm_bwd_layer(ops.ptr(dout), 0, T * 2 * H, 2 * H, ops.ptr(dg), ops.ptr(whhT), ops.ptr(gates), ops.ptr(cseq), ops.ptr(dc),
                                              ops.ptr(cnt), T, B, H, st), "bwd")
            torch.cuda.synchronize()
            assert int(cnt[2]) == 0, "persistent LSTM: inter-workgroup wait timed out"
        else:
            for s in range(T):
                L.check(L.lib().av_lstm_fwd_step(ops.ptr(gx), ops.ptr(whh), ops.ptr(hseq), ops.ptr(cseq), ops.ptr(gates), None, 1, T, B, H, s, st), "fwd step")
            for s in range(T):
                L.check(L.lib().av_lstm_bwd_step(ops.ptr(dout), 0, T * 2 * H, 2 * H, ops.ptr(dg), ops.ptr(whhT), ops.ptr(gates), ops.ptr(cseq), ops.ptr(dc),
                                                 1, T, B, H, s, st), "bwd step")
        return hseq.float(), cseq, gates.float(), dg.float(), dc

    a, b = run(True), run(False)
    for x, y, name in zip(a, b, ("h", "c", "gates", "dgates", "dc")):
        torch.testing.assert_close(x, y, rtol=3e-2, atol=3e-2, msg=lambda m, n=name: f"{n}: {m}")


def test_device_input_pipeline_equals_numpy_restatement():
    """csrc/preprocess.hip against oracle/pipeline_oracle.py (the per-sample arithmetic of the reference's load_pair,
    dataset/multi_speaker_dataset.py:13-59): bit-exact, float32 operation order of the reference.  cv2 is absent from the image, so the
    resize law itself is pinned only by the oracle's known answers (parity with cv2 unpinned)."""
    from oracle import pipeline_oracle as po
    dp = pkg("dataset.device_pipeline"); cf = pkg("dataset.collate_fn").collate_fn
    rng = np.random.default_rng(11)
    for shape, dtype in (((7, 128, 128, 3), np.uint8), ((3, 100, 120, 3), np.float32), ((2, 64, 80, 1), np.float32), ((1, 96, 96, 3), np.uint8)):
        fr = (rng.random(shape) * 255).astype(dtype)
        got = dp.lips_to_device(fr).cpu().numpy()
        want = po.lips(fr)
        assert got.shape == want.shape and np.array_equal(got, want), (shape, float(np.abs(got - want).max()))
    for n1, n2 in ((1000, 700), (700, 1000), (512, 512), (0, 300), (70001, 64000)):
        a1 = rng.standard_normal(n1).astype(np.float32); a2 = rng.standard_normal(n2).astype(np.float32)
        got = dp.mix_pair(a1, a2)
        mixed, m1, m2 = po.mix_pair(a1, a2)
        assert np.array_equal(got["audio"].cpu().numpy(), mixed), (n1, n2)
        assert np.array_equal(got["mask1"].cpu().numpy(), m1) and np.array_equal(got["mask2"].cpu().numpy(), m2)
    with pytest.raises(RuntimeError):
        dp.lips_to_device(np.zeros((0, 128, 128, 3), np.uint8))                                  # empty clip: the reference raises too (:59-60)
    items = []
    for n1, n2, t1, t2 in ((4000, 3000, 6, 5), (2500, 2500, 4, 4)):
        items.append(dp.load_pair_device(rng.standard_normal(n1).astype(np.float32), rng.standard_normal(n2).astype(np.float32),
                                         (rng.random((t1, 128, 128, 3)) * 255).astype(np.uint8), (rng.random((t2, 128, 128, 3)) * 255).astype(np.uint8),
                                         [5, 6, 7], [8, 9]))
    batch = cf(items)                                                                             # collate pads device tensors in place
    assert batch["audio"].is_cuda and batch["audio"].shape == (2, 4000) and batch["lip1"].shape == (2, 6, 1, 96, 96)
    assert batch["mask1"][1, 2500:].eq(3).all() and batch["mask1"][0, 3000:].eq(2).all() and batch["mask2"][0, 3000:].eq(0).all()
    assert batch["lip2_lengths"].tolist() == [5, 4] and batch["text1"].tolist() == [[5, 6, 7], [5, 6, 7]]



@pytest.mark.parametrize("B,T", [(3, 100), (2, 25), (1, 112), (5, 7), (130, 100)])
def test_fused_cross_attention_block(B, T):
    """csrc/fusion_attn.hip (packed in-projection + attention core per (item, head)) against the fp64 statement of
    nn.MultiheadAttention's need-weights path (torch:functional.py:6576-6606: q scaled by 1/sqrt(128), softmax, no masks)."""
    E, nh, hd = 512, 4, 128
    dt_ = torch.bfloat16
    a = _rand(B, T, E, dtype=dt_); v = _rand(B, T, E, dtype=dt_)
    w = _rand(3 * E, E, dtype=dt_, scale=1 / math.sqrt(E)); bias = _rand(3 * E, scale=0.1)
    o, q, kv, lse = ops.fusion_xattn_fwd(a, v, w, bias, nh, hd ** -0.5, True)
    wd, bd = w.double(), bias.double()
    qr = (a.double() @ wd[:E].t() + bd[:E]).view(B, T, nh, hd)
    kr = (v.double() @ wd[E:2 * E].t() + bd[E:2 * E]).view(B, T, nh, hd)
    vr = (v.double() @ wd[2 * E:].t() + bd[2 * E:]).view(B, T, nh, hd)
    torch.testing.assert_close(q.double(), qr, rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(kv[:, :, 0].double(), kr, rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(kv[:, :, 1].double(), vr, rtol=2e-2, atol=2e-2)
    # the attention core consumes the bf16-rounded projections (as the unfused path does)
    qb, kb, vb = q.double().permute(0, 2, 1, 3), kv[:, :, 0].double().permute(0, 2, 1, 3), kv[:, :, 1].double().permute(0, 2, 1, 3)
    s = (qb @ kb.transpose(2, 3)) * hd ** -0.5
    ref = (torch.softmax(s, -1) @ vb).permute(0, 2, 1, 3)
    torch.testing.assert_close(o.double(), ref, rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(lse.double(), torch.logsumexp(s, -1), rtol=1e-3, atol=2e-3)
    o2, q2, kv2, lse2 = ops.fusion_xattn_fwd(a, v, w, bias, nh, hd ** -0.5, False)
    assert q2 is None and kv2 is None and lse2 is None and torch.equal(o2, o)



@pytest.mark.parametrize("B,T", [(3, 100), (2, 25), (1, 112), (4, 7), (130, 100)])
def test_fused_cross_attention_core_backward(B, T):
    """fusion_xattn_bwd_kernel (whole-sequence backward of the 4 x 128 attention core, P recomputed from the LSE) against fp64 autograd."""
    E, nh, hd = 512, 4, 128
    dt_ = torch.bfloat16
    q = _rand(B, T, nh, hd, dtype=dt_); kv = _rand(B, T, 2, nh, hd, dtype=dt_); do = _rand(B, T, nh, hd, dtype=dt_)
    scale = hd ** -0.5
    qr, kr, vr = (x.double().permute(0, 2, 1, 3).detach().clone().requires_grad_(True) for x in (q, kv[:, :, 0], kv[:, :, 1]))
    s = (qr @ kr.transpose(2, 3)) * scale
    oref = torch.softmax(s, -1) @ vr
    oref.backward(do.double().permute(0, 2, 1, 3))
    o = oref.detach().permute(0, 2, 1, 3).contiguous().to(dt_)
    lse = torch.logsumexp(s.detach(), -1).float().contiguous()
    dq, dkv = ops.fusion_xattn_bwd(q, kv, o, do, lse, scale)
    for got, ref in ((dq, qr.grad), (dkv[:, :, 0], kr.grad), (dkv[:, :, 1], vr.grad)):
        torch.testing.assert_close(got.double().permute(0, 2, 1, 3), ref, rtol=4e-2, atol=4e-2)
    # against the tiled kernels it replaces (same inputs)
    dq2 = torch.empty_like(q); dkv2 = torch.empty_like(kv)
    ops.attention_bwd(q, kv[:, :, 0], kv[:, :, 1], do, dq2, dkv2[:, :, 0], dkv2[:, :, 1], None, scale, o=o, lse=lse)
    torch.testing.assert_close(dq.float(), dq2.float(), rtol=3e-2, atol=3e-2)
    torch.testing.assert_close(dkv.float(), dkv2.float(), rtol=3e-2, atol=3e-2)


@pytest.mark.parametrize("n", [1, 7, 1023, 1025, 12736, 25472, 100003])
def test_class_order_equals_stable_argsort_of_the_class_rank(n):
    """av_class_order (contrastive.py:24-26: anchors 1, positives 2, negatives 0, the rest) == torch.argsort(rank, stable=True)."""
    g = torch.Generator().manual_seed(n)
    mask = torch.randint(0, 4, (n,), generator=g)
    if n > 100:
        mask[5] = -2; mask[17] = 9                               # out-of-range values: clamped like the counts (below 0 -> 0, above 3 -> 3)
    m = mask.cuda()
    order = torch.empty(n, dtype=torch.long, device="cuda")
    L.check(L.lib().av_class_order(ops.ptr(m), n, ops.ptr(order), ops.stream()))
    rank = torch.tensor([2, 0, 1, 3])[mask.clamp(0, 3)]
    assert torch.equal(order.cpu(), torch.argsort(rank, stable=True))


def test_four_wavefront_gemm_kernel_behind_its_switch():
    """gemm_nt_bf16_v6_kernel (experimental, AVAMD_GEMM_V6=1; the switch is read once per process, hence the child process): plain, bias +
    GELU and fp32 + residual epilogues at a ragged M / N against fp32 torch."""
    import os, subprocess, sys
    code = r'''
import importlib, sys, torch
sys.path.insert(0, %r)
ops = importlib.import_module("multimodal-av-model_amd.ops"); L = importlib.import_module("multimodal-av-model_amd._lib")
torch.manual_seed(0)
for (M, N, K) in ((1000, 520, 192), (2048, 1024, 1024), (300, 264, 64)):
    a = (torch.rand(M, K, device="cuda") - 0.5).to(torch.bfloat16); w = (torch.rand(N, K, device="cuda") - 0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda"); r = torch.randn(M, N, device="cuda")
    ref = a.float() @ w.float().t()
    y = ops.linear(a, w, out_dtype=torch.float32)
    assert (y - ref).abs().max() < 2e-3 * K ** 0.5, ("plain", M, N, K, float((y - ref).abs().max()))
    y = ops.linear(a, w, b, out_dtype=torch.float32, R=r)
    assert (y - (ref + b + r)).abs().max() < 2e-3 * K ** 0.5, ("bias+res", M, N, K)
    y = ops.linear(a, w, b, act=L.ACT_GELU)
    g = torch.nn.functional.gelu(ref + b)
    assert (y.float() - g).abs().max() < 3e-2 * max(1.0, float(g.abs().max())), ("gelu", M, N, K)
print("ok")
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, AVAMD_GEMM_V6="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout + out.stderr
