"""Torch-tensor wrappers over the C-ABI of libavhip.so (plumbing only: pointers, shapes, current stream).

PyTorch provides device memory and streams; every arithmetic op on the path is a HIP kernel of ours.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

from . import _lib as L
from . import precision
from ._lib import AV_BF16, AV_F32

Tensor = torch.Tensor


def _lp() -> torch.dtype:
    """The 16-bit operand type of the library in use: float16 in precision mode "fp16" (libavhip_f16.so), bfloat16 otherwise."""
    return torch.float16 if precision.get_precision() == "fp16" else torch.bfloat16


def dt(t: Tensor) -> int:
    if t.dtype == torch.float32:
        return AV_F32
    if t.dtype == _lp():
        return AV_BF16                                              # dtype code 1 = "the library's 16-bit type"
    raise TypeError(f"unsupported dtype {t.dtype} (the library of precision mode {precision.get_precision()!r} handles float32 / {_lp()})")


def tdt(code: int) -> torch.dtype:
    return torch.float32 if code == AV_F32 else _lp()


def ptr(t: Optional[Tensor]):
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_dev = getattr(torch._C, "_cuda_getDevice", None) or torch.cuda.current_device


def stream():
    """hipStream_t of torch's current stream on the current device (raw handle: ~10x cheaper than building a Stream object
    per kernel launch, which at ~950 launches per step was 3 ms of host time)."""
    if _raw_stream is not None:
        return _raw_stream(_cur_dev())                      # torch._C._cuda_getDevice: current_device() without the lazy-init checks
    return torch.cuda.current_stream().cuda_stream


def h2d_async(arr, device) -> Tensor:
    """numpy array -> device tensor without draining the stream: staged through the caching pinned-host allocator and copied with
    non_blocking=True (a plain ``torch.tensor(host_data, device=cuda)`` synchronises the stream it runs on)."""
    return torch.from_numpy(arr).pin_memory().to(device, non_blocking=True)


def _req(t: Tensor, name: str):
    if not t.is_cuda:
        raise RuntimeError(f"{name}: tensor must live on the GPU (no CPU fallback in libavhip)")


class GemmProbe:
    """Optional live timing of ONE av_gemm instantiation with events on the launch stream (bench.py roofline leg)."""
    active = None            # dict(key=(in_dtype, a_mode, b_mode, wide), records=[(start, end, flops)])

    @classmethod
    def start(cls, in_dtype: int, a_mode: int, b_mode: int, wide: bool):
        cls.active = dict(key=(in_dtype, a_mode, b_mode, wide), records=[])

    @classmethod
    def stop(cls):
        a, cls.active = cls.active, None
        return a


class AttnProbe:
    """Optional live timing of the attention launches with events on the launch stream (bench.py ``attention`` report).
    Records (start, end, tag, flops, bytes); tag = "fwd"/"bwd" + head_dim (64 = wav2vec2 self-attention K7, 128 = fusion K17)."""
    active = None

    @classmethod
    def start(cls):
        cls.active = []

    @classmethod
    def stop(cls):
        a, cls.active = cls.active, None
        return a


def _probed(tag, flops, nbytes, fn):
    pr = AttnProbe.active
    if pr is None:
        return fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    r = fn()
    e1.record()
    pr.append((e0, e1, tag, float(flops), float(nbytes)))
    return r


def gemm(A: Tensor, B: Tensor, C_: Tensor, *, M: int, N: int, K: int, lda: int, ldb: int, ldc: int,
         a_mode: int = L.A_ROWMAJOR, b_mode: int = L.B_NK, bias: Optional[Tensor] = None, act: int = L.ACT_NONE,
         R: Optional[Tensor] = None, ldr: int = 0, aux: Optional[Tensor] = None, C2: Optional[Tensor] = None,
         stats: Optional[Tensor] = None, alpha: float = 1.0, batch: int = 1, sA: int = 0, sB: int = 0, sC: int = 0,
         sR: int = 0, sBias: int = 0, conv: Optional[dict] = None, a_off: int = 0, b_off: int = 0, c_off: int = 0,
         batch_inner: int = 0, oA: int = 0, oB: int = 0, oC: int = 0, drop: Optional[tuple] = None, k_total: int = 0) -> Tensor:
    """Raw av_gemm call.  Offsets are in elements."""
    _req(A, "gemm A"); _req(B, "gemm B"); _req(C_, "gemm C")
    if A.dtype != B.dtype:
        raise TypeError(f"gemm: A {A.dtype} and B {B.dtype} must match")
    a = L.GemmArgs()
    a.A = A.data_ptr() + a_off * A.element_size()
    a.B = B.data_ptr() + b_off * B.element_size()
    a.C = C_.data_ptr() + c_off * C_.element_size()
    a.C2 = ptr(C2) if C2 is None else C2.data_ptr() + c_off * C2.element_size()
    a.bias = ptr(bias); a.R = ptr(R) if R is None else R.data_ptr() + c_off * 4
    a.aux = ptr(aux); a.stats = ptr(stats)
    a.M, a.N, a.K, a.batch = M, N, K, batch
    a.lda, a.ldb, a.ldc, a.ldr = lda, ldb, ldc, (ldr or ldc)
    a.sA, a.sB, a.sC, a.sR, a.sBias = sA, sB, sC, sR, sBias
    a.batch_inner, a.oA, a.oB, a.oC = batch_inner, oA, oB, oC
    a.k_total = k_total
    if drop is not None and drop[0] > 0:
        a.drop_p, a.drop_seed, a.drop_stream = float(drop[0]), int(drop[1]), int(drop[2])     # (p, seed, stream id)
    a.a_mode, a.b_mode = a_mode, b_mode
    a.in_dtype, a.out_dtype = dt(A), dt(C_)
    a.aux_dtype = dt(aux) if aux is not None else 0
    a.act, a.alpha = act, alpha
    if bias is not None and bias.dtype != torch.float32:
        raise TypeError("gemm: bias must be float32")
    if R is not None and R.dtype != torch.float32:
        raise TypeError("gemm: residual must be float32")
    if conv:
        for k, v in conv.items():
            setattr(a, k, v)
    pr = GemmProbe.active
    if pr is not None and pr["key"] == (a.in_dtype, a_mode, b_mode, N > 64):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        L.check(L.lib().av_gemm(C.byref(a), stream()), "av_gemm")
        e1.record()
        es, oes = A.element_size(), C_.element_size()
        nbytes = batch * ((M * K + N * K) * es + M * N * (oes + (4 if R is not None else 0) + (aux.element_size() if aux is not None else 0)
                                                          + (oes if C2 is not None else 0)))
        pr["records"].append((e0, e1, 2.0 * M * N * K * batch, float(nbytes),
                              (M, N, K, batch, act, bias is not None, R is not None, aux is not None, C2 is not None,
                               drop is not None and drop[0] > 0, bool(conv), stats is not None)))
        return C_
    L.check(L.lib().av_gemm(C.byref(a), stream()), "av_gemm")
    return C_


def linear(x: Tensor, w: Tensor, bias: Optional[Tensor] = None, *, out_dtype: Optional[torch.dtype] = None,
           act: int = L.ACT_NONE, R: Optional[Tensor] = None, out: Optional[Tensor] = None, C2: Optional[Tensor] = None,
           aux: Optional[Tensor] = None, alpha: float = 1.0, drop: Optional[tuple] = None) -> Tensor:
    """y[M,N] = epilogue(x[M,K] @ w[N,K]^T + bias)  (nn.Linear layout).  x may be any [..., K] contiguous tensor."""
    K = x.shape[-1]
    M = x.numel() // K
    N = w.shape[0]
    assert w.shape[1] == K and x.is_contiguous() and w.is_contiguous()
    if out is None:
        out = torch.empty(x.shape[:-1] + (N,), dtype=out_dtype or x.dtype, device=x.device)
    gemm(x, w, out, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=bias, act=act, R=R, aux=aux, C2=C2, alpha=alpha, drop=drop)
    return out


def transpose(x: Tensor, out_dtype: Optional[torch.dtype] = None, pad_to: int = 1) -> Tensor:
    """[R, C] -> [C, Rpad] (Rpad = R rounded up to ``pad_to``, zero-filled)."""
    R, Cc = x.shape
    assert x.stride(1) == 1
    Rp = (R + pad_to - 1) // pad_to * pad_to
    out = torch.empty((Cc, Rp), dtype=out_dtype or x.dtype, device=x.device)
    L.check(L.lib().av_transpose(ptr(x), dt(x), ptr(out), dt(out), R, Cc, x.stride(0), Rp, stream()), "av_transpose")
    return out


_WT_CACHE = {}


def transpose_cached(b: Tensor, hot_ok: bool = False) -> Optional[Tensor]:
    """W^T of a weight-like operand, cached on (storage, version): frozen layers transpose once.  A weight that keeps changing (a
    trained layer: new version after every optimizer step) is *hot*: with ``hot_ok`` the caller gets None from its third version on and
    feeds the k-major operand to the GEMM directly instead of paying a transpose pass per step."""
    key = (b.data_ptr(), tuple(b.shape), b.dtype)
    ver = b._version
    hit = _WT_CACHE.get(key)
    if hit is None or hit[2]() is not b:
        import weakref
        hit = [ver, transpose(b), weakref.ref(b), 0]
        if len(_WT_CACHE) > 512:
            _WT_CACHE.clear()
        _WT_CACHE[key] = hit
    elif hit[0] != ver:
        hit[3] += 1
        if hot_ok and hit[3] >= 2:
            hit[0], hit[1] = None, None                          # hot: no transposed copy is kept
            return None
        hit[0], hit[1] = ver, transpose(b)
    elif hit[1] is None:
        if hot_ok:
            return None
        hit[0], hit[1] = ver, transpose(b)
    return hit[1]


def _fast_ok(t: Tensor, K: int, N: int) -> bool:
    return precision.is_lp(t.dtype) and K >= 64 and N > 64


# k-major operands straight into the fast GEMM (transposed LDS reads) instead of a transpose pass; AVAMD_GEMM_KMAJOR=0 = old path
KMAJOR = os.environ.get("AVAMD_GEMM_KMAJOR", "1") != "0"
KM8 = int(os.environ.get("AVAMD_GEMM_KM8", "1"))         # dW products of >= 32 tiles of 256 x 256 on the 8-phase kernel's k-major form (0: never, 2: from one tile on)
# dX of a weight that changes every step: 0 (default) = transpose the weight once per version (one 12 us pass, shared by both audio passes)
# and run the row-major 8-phase kernel; 1 = feed the k-major weight to the 128 x 128 kernel (no transpose pass)
KMAJOR_DX_HOT = os.environ.get("AVAMD_GEMM_KMAJOR_DX", "0") != "0"


def matmul_nn(a: Tensor, b: Tensor, *, out_dtype: Optional[torch.dtype] = None, act: int = L.ACT_NONE,
              aux: Optional[Tensor] = None, R: Optional[Tensor] = None, out: Optional[Tensor] = None, alpha: float = 1.0,
              b_is_weight: bool = False, drop: Optional[tuple] = None) -> Tensor:
    """y[M,N] = a[M,K] @ b[K,N]  (b row-major, e.g. dX = dY @ W with W [N_out,K_in])."""
    K = a.shape[-1]
    M = a.numel() // K
    N = b.shape[1]
    assert b.shape[0] == K and a.is_contiguous() and b.is_contiguous()
    if out is None:
        out = torch.empty(a.shape[:-1] + (N,), dtype=out_dtype or a.dtype, device=a.device)
    kmajor_ok = KMAJOR and N % 8 == 0
    if _fast_ok(a, K, N) and K % 64 == 0 and (b_is_weight or not kmajor_ok):
        bt = transpose_cached(b, hot_ok=kmajor_ok and KMAJOR_DX_HOT) if b_is_weight else transpose(b)   # [N, K]: K-contiguous operand for the fast kernel
        if bt is not None:
            gemm(a, bt, out, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, act=act, aux=aux, R=R, alpha=alpha, drop=drop)
            return out
    # bf16, K % 64 == 0, N % 8 == 0, b not a (cached) weight: the fast kernel reads the k-major B through transposed LDS reads
    gemm(a, b, out, M=M, N=N, K=K, lda=K, ldb=N, ldc=N, b_mode=L.B_KN, act=act, aux=aux, R=R, alpha=alpha, drop=drop)
    return out


def _split_k(tiles: int, Kk: int, mn: int, slots: int = 512) -> int:
    """Number of K slices for dW-shaped products: minimise rounds(tiles * S over 2 workgroups per CU) * K / S plus the
    partial-sum traffic (S fp32 slices written and read once)."""
    best, best_cost = 1, None
    for S in range(1, 9):
        if S > 1 and Kk // S < 512:
            break
        rounds = (tiles * S + slots - 1) // slots
        cost = rounds * (Kk / S) * 22e-9 + (S * mn * 8 / 5e12 if S > 1 else 0.0)
        if best_cost is None or cost < best_cost * 0.97:
            best, best_cost = S, cost
    return best


def _split_k8(tiles: int, Kk: int, mn: int, slots: int = 256):
    """(slices, k per slice) for the 8-phase kernel's k-major form (256 x 256 tiles, one workgroup per CU): rounds x (K-tiles x 1.55 us + 7 us of
    prologue / epilogue) plus the partial-sum traffic (S fp32 slices written and read once)."""
    best = (1, Kk, None)
    for S in range(1, 33):
        chunk = ((Kk + S - 1) // S + 63) // 64 * 64
        if S > 1 and chunk < 512:
            break
        Se = (Kk + chunk - 1) // chunk
        rounds = (tiles * Se + slots - 1) // slots
        cost = rounds * (chunk / 64 * 1.55e-6 + 7e-6) + (Se * mn * 8 / 5e12 if Se > 1 else 0.0)
        if best[2] is None or cost < best[2] * 0.97:
            best = (Se, chunk, cost)
    return best[0], best[1]


def matmul_tn(a: Tensor, b: Tensor, *, out: Optional[Tensor] = None, alpha: float = 1.0, accumulate: bool = False) -> Tensor:
    """y[M,N] (fp32) = a[K,M]^T @ b[K,N]   (dW = dY^T X: a = dY [tokens, out], b = X [tokens, in]).
    a / b may be row-strided views (last dim contiguous)."""
    Kk, M = a.shape
    N = b.shape[1]
    assert b.shape[0] == Kk and a.stride(1) == 1 and b.stride(1) == 1
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    if _fast_ok(a, Kk, N) and M > 64 and not (KMAJOR and M % 8 == 0 and N % 8 == 0 and N > 64 and a.stride(0) % 8 == 0 and b.stride(0) % 8 == 0):
        at = transpose(a, pad_to=64)             # [M, Kp]  token dimension becomes the contiguous K (zero padded)
        bt = transpose(b, pad_to=64)             # [N, Kp]
        Kp = at.shape[1]
        gemm(at, bt, out, M=M, N=N, K=Kp, lda=Kp, ldb=Kp, ldc=N, alpha=alpha, R=out if accumulate else None)
        return out
    km = KMAJOR and _fast_ok(a, Kk, N) and M > 64 and M % 8 == 0 and N % 8 == 0 and a.stride(0) % 8 == 0 and b.stride(0) % 8 == 0
    t8 = ((M + 255) // 256) * ((N + 255) // 256)
    if km and KM8 and M >= 256 and N >= 256 and (KM8 >= 2 or t8 >= 32):         # 8-phase kernel, k-major form (gemm_fast.hip: the same conditions)
        S, chunk = _split_k8(t8, Kk, M * N)
    else:
        tiles = ((M + 127) // 128) * ((N + 127) // 128)
        S = _split_k(tiles, Kk, M * N) if km else 1  # few output tiles, long K (tokens): split K over the batch dimension
        if S > 1:
            chunk = ((Kk + S - 1) // S + 63) // 64 * 64
            S = (Kk + chunk - 1) // chunk
    if S > 1:
        parts = torch.empty((S, M, N), dtype=torch.float32, device=a.device)
        gemm(a, b, parts, M=M, N=N, K=chunk, lda=a.stride(0), ldb=b.stride(0), ldc=N, a_mode=L.A_TRANS, b_mode=L.B_KN, batch=S,
             sA=chunk * a.stride(0), sB=chunk * b.stride(0), sC=M * N, k_total=Kk)
        L.check(L.lib().av_sum_slices(ptr(parts), S, M * N, M * N, float(alpha), ptr(out), int(accumulate), stream()), "av_sum_slices")
        return out
    gemm(a, b, out, M=M, N=N, K=Kk, lda=a.stride(0), ldb=b.stride(0), ldc=N, a_mode=L.A_TRANS, b_mode=L.B_KN, alpha=alpha,
         R=out if accumulate else None)
    return out


def layernorm_fwd(x: Tensor, gamma: Tensor, beta: Tensor, *, out_dtype: torch.dtype, eps: float = 1e-5, act: int = L.ACT_NONE,
                  save_stats: bool = False):
    cols = x.shape[-1]
    rows = x.numel() // cols
    assert x.is_contiguous()
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    mean = rstd = None
    if save_stats:
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    L.check(L.lib().av_layernorm_fwd(ptr(x), dt(x), ptr(gamma), ptr(beta), ptr(y), dt(y), ptr(mean), ptr(rstd), rows, cols,
                                     eps, act, stream()), "av_layernorm_fwd")
    return (y, mean, rstd) if save_stats else y


def layernorm_bwd(x: Tensor, dy: Tensor, gamma: Tensor, mean: Tensor, rstd: Tensor, dres: Optional[Tensor] = None,
                  want_param_grads: bool = False, lp_copy: bool = False, lp_drop: Optional[tuple] = None, gb_acc: Optional[Tensor] = None,
                  packed_gb: bool = False):
    """dx (fp32) [, dgamma, dbeta]; with ``lp_copy`` the last element returned is a bf16 copy of dx written by the same kernel;
    ``lp_drop`` = (p, seed, stream id) makes that copy dx o dropout-mask / (1 - p).  ``gb_acc`` (a [2 cols] tensor from an earlier call) /
    ``packed_gb``: the parameter gradients come back as ONE packed tensor [dgamma | dbeta], accumulated into ``gb_acc`` when given (an
    earlier pass's tensor, or a zeroed view of a gradient bucket)."""
    cols = x.shape[-1]
    rows = x.numel() // cols
    assert x.is_contiguous() and dy.is_contiguous()
    dx = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    dxl = torch.empty(x.shape, dtype=_lp(), device=x.device) if lp_copy else None
    nblk = max(1, min(512, (rows + 15) // 16))
    part = torch.empty((nblk, 2 * cols), dtype=torch.float32, device=x.device) if want_param_grads else None
    if lp_copy and lp_drop is not None and lp_drop[0] > 0:
        L.check(L.lib().av_layernorm_bwd_drop(ptr(x), dt(x), ptr(dy), dt(dy), ptr(gamma), ptr(mean), ptr(rstd), ptr(dres), ptr(dx),
                                              ptr(part), nblk, rows, cols, ptr(dxl), float(lp_drop[0]), int(lp_drop[1]), int(lp_drop[2]), stream()),
                "av_layernorm_bwd_drop")
    else:
        L.check(L.lib().av_layernorm_bwd(ptr(x), dt(x), ptr(dy), dt(dy), ptr(gamma), ptr(mean), ptr(rstd), ptr(dres), ptr(dx),
                                         ptr(part), nblk, rows, cols, ptr(dxl), stream()), "av_layernorm_bwd")
    res = (dx,)
    if want_param_grads:
        if gb_acc is not None:
            gb = colsum(part, out=gb_acc, accumulate=True)
            res = (dx, gb)
        elif packed_gb:
            res = (dx, colsum(part))
        else:
            gb = colsum(part)
            res = (dx, gb[:cols], gb[cols:])
    if lp_copy:
        return res + (dxl,)
    return res if want_param_grads else dx


def log_softmax_fwd(x: Tensor) -> Tensor:
    cols = x.shape[-1]
    y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    L.check(L.lib().av_log_softmax_fwd(ptr(x), dt(x), ptr(y), x.numel() // cols, cols, stream()), "av_log_softmax_fwd")
    return y


def log_softmax_bwd(y: Tensor, dy: Tensor, out_dtype: torch.dtype, pad_to: int = 0) -> Tensor:
    """dx = dy - exp(y) * sum(dy).  ``pad_to`` > 0: the rows of dx are padded with zero columns to a multiple of it (returned tensor
    [..., ceil(cols / pad_to) * pad_to]: a K-padded operand for the dX product of the layer below)."""
    cols = y.shape[-1]
    ldx = (cols + pad_to - 1) // pad_to * pad_to if pad_to > 0 else cols
    dx = torch.empty(y.shape[:-1] + (ldx,), dtype=out_dtype, device=y.device)
    L.check(L.lib().av_log_softmax_bwd_ld(ptr(y), ptr(dy), ptr(dx), dt(dx), y.numel() // cols, cols, ldx, stream()), "av_log_softmax_bwd")
    return dx


def colsum(x: Tensor, out: Optional[Tensor] = None, accumulate: bool = False) -> Tensor:
    cols = x.shape[-1]
    if x.is_contiguous():
        rows, ld = x.numel() // cols, cols
    else:                                                   # a column slice of a wider matrix (rows strided, last dim contiguous)
        assert x.dim() == 2 and x.stride(1) == 1
        rows, ld = x.shape[0], x.stride(0)
    if out is None:
        out = torch.empty(cols, dtype=torch.float32, device=x.device)
        accumulate = False
    L.check(L.lib().av_colsum(ptr(x), dt(x), ptr(out), rows, cols, ld, int(accumulate), stream()), "av_colsum")
    return out


def colsum_into(x: Tensor, zeroed_view: Optional[Tensor]) -> Tensor:
    """Column sums accumulated into a pre-zeroed view of a gradient bucket (no clear launch), or into a fresh tensor when there is none."""
    return colsum(x, out=zeroed_view, accumulate=True) if zeroed_view is not None else colsum(x)


def cast(x: Tensor, dtype: torch.dtype) -> Tensor:
    if x.dtype == dtype:
        return x
    assert x.is_contiguous()
    y = torch.empty(x.shape, dtype=dtype, device=x.device)
    L.check(L.lib().av_cast(ptr(x), dt(x), ptr(y), dt(y), x.numel(), stream()), "av_cast")
    return y


def cast_dropout(x: Tensor, dtype: torch.dtype, drop: Optional[tuple]) -> Tensor:
    """cast(x * mask): ``drop`` = (p, seed, stream id) or None (plain cast)."""
    if drop is None or drop[0] <= 0:
        return cast(x, dtype)
    assert x.is_contiguous()
    y = torch.empty(x.shape, dtype=dtype, device=x.device)
    L.check(L.lib().av_cast_dropout(ptr(x), dt(x), ptr(y), dt(y), x.numel(), float(drop[0]), int(drop[1]), int(drop[2]), stream()), "av_cast_dropout")
    return y


def axpby(a: float, x: Tensor, b: float, y: Tensor) -> Tensor:
    """y = a*x + b*y  (y fp32, in place)."""
    assert x.is_contiguous() and y.is_contiguous() and y.dtype == torch.float32 and x.numel() == y.numel()
    L.check(L.lib().av_axpby(a, ptr(x), dt(x), b, ptr(y), x.numel(), stream()), "av_axpby")
    return y


def mask_rows_(x: Tensor, keep_u8: Tensor) -> Tensor:
    cols = x.shape[-1]
    L.check(L.lib().av_mask_rows(ptr(x), dt(x), ptr(keep_u8), x.numel() // cols, cols, stream()), "av_mask_rows")
    return x


# ---------------------------------------------------------------------------------------------------------------
# attention.  q/k/v/o are [B, T, H, D] views (last dim contiguous, head stride == D), e.g. slices of a packed
# [B, T, 3, H, D] projection buffer — no copies are made.
# ---------------------------------------------------------------------------------------------------------------
def _chk_view(t: Tensor, name: str):
    if t.dim() != 4 or t.stride(3) != 1 or t.stride(2) != t.shape[3]:
        raise ValueError(f"{name}: expected a [B,T,H,D] view with contiguous D and head stride D, got {tuple(t.shape)} / {t.stride()}")


_ATTN_SHORT = os.environ.get("AVAMD_ATTN_SHORT", "1") != "0"


def attention_mask_shape_ok(dtype, B: int, Tq: int, Tk: int, D: int) -> bool:
    return _ATTN_SHORT and precision.is_lp(dtype) and D == 64 and Tq <= 256 and Tk <= 256 and B <= 65535


def attention_dropmask(B: int, H: int, Tq: int, Tk: int, drop: tuple, device) -> Tensor:
    """Keep bits of the attention-probability dropout of a [B, H, Tq, Tk] problem, evaluated once for the whole-sequence forward and both
    phases of its backward (check ``attention_mask_shape_ok`` first)."""
    mask = torch.empty((B, H, (Tq + 15) // 16, 64), dtype=torch.int64, device=device)
    L.check(L.lib().av_attention_dropmask(ptr(mask), B, H, Tq, Tk, float(drop[0]), int(drop[1]), int(drop[2]), stream()), "av_attention_dropmask")
    return mask


def attention_fwd(q: Tensor, k: Tensor, v: Tensor, klen: Optional[Tensor], scale: float, need_lse: bool = True,
                  drop: Optional[tuple] = None, drop_mask: Optional[Tensor] = None):
    for t, n in ((q, "q"), (k, "k"), (v, "v")):
        _chk_view(t, n)
    B, Tq, H, D = q.shape
    Tk = k.shape[1]
    o = torch.empty((B, Tq, H, D), dtype=q.dtype, device=q.device)
    lse = torch.empty((B, H, Tq), dtype=torch.float32, device=q.device) if need_lse else None
    dp, dseed, dstream = (float(drop[0]), int(drop[1]), int(drop[2])) if (drop is not None and drop[0] > 0) else (0.0, 0, 0)
    es = q.element_size()
    _probed(f"fwd{D}", 4.0 * B * H * Tq * Tk * D, B * H * D * es * (2 * Tq + 2 * Tk),
            lambda: L.check(L.lib().av_attention_fwd_mask(ptr(q), ptr(k), ptr(v), ptr(o), ptr(lse), dt(q), B, H, Tq, Tk, D,
                                                          q.stride(0), q.stride(1), k.stride(0), k.stride(1), v.stride(0), v.stride(1),
                                                          o.stride(0), o.stride(1), ptr(klen), scale, dp, dseed, dstream,
                                                          ptr(drop_mask) if dp > 0 else None, stream()), "av_attention_fwd"))
    return o, lse


def fusion_xattn_fwd(a: Tensor, v: Tensor, w_in: Tensor, b_in: Tensor, nh: int, scale: float, save: bool):
    """Fused in-projection + cross-attention core (csrc/fusion_attn.hip): a, v [B, T, E] bf16 -> (o [B,T,nh,hd], q [B,T,nh,hd] | None,
    kv [B,T,2,nh,hd] | None, lse [B,nh,T] | None)."""
    B, T, E = a.shape
    hd = E // nh
    assert a.is_contiguous() and v.is_contiguous() and w_in.is_contiguous() and w_in.shape == (3 * E, E) and b_in.dtype == torch.float32
    o = torch.empty((B, T, nh, hd), dtype=a.dtype, device=a.device)
    q = torch.empty((B, T, nh, hd), dtype=a.dtype, device=a.device) if save else None
    kv = torch.empty((B, T, 2, nh, hd), dtype=a.dtype, device=a.device) if save else None
    lse = torch.empty((B, nh, T), dtype=torch.float32, device=a.device) if save else None
    fl = B * (6.0 * T * E * E + 4.0 * T * T * E)
    by = a.element_size() * (2 * B * T * E + 3 * E * E + B * T * E * (4 if save else 1))
    _probed("blockfwd", fl, by,
            lambda: L.check(L.lib().av_fusion_xattn_fwd(ptr(a), ptr(v), ptr(w_in), ptr(b_in), ptr(q), ptr(kv), ptr(o), ptr(lse), B, T, E, nh, scale,
                                                        stream()), "av_fusion_xattn_fwd"))
    return o, q, kv, lse


def fusion_xattn_bwd(q: Tensor, kv: Tensor, o: Tensor, do: Tensor, lse: Tensor, scale: float):
    """Backward of the fused cross-attention core (csrc/fusion_attn.hip): q, o, do [B,T,nh,hd], kv [B,T,2,nh,hd] bf16 -> (dq, dkv)."""
    B, T, nh, hd = q.shape
    E = nh * hd
    assert q.is_contiguous() and kv.is_contiguous() and o.is_contiguous() and do.is_contiguous()
    dq = torch.empty_like(q); dkv = torch.empty_like(kv)
    es = q.element_size()
    _probed("blockbwd", 10.0 * B * T * T * E, es * B * T * E * 8,
            lambda: L.check(L.lib().av_fusion_xattn_bwd(ptr(q), ptr(kv), ptr(o), ptr(do), ptr(lse), ptr(dq), ptr(dkv), B, T, E, nh, scale, stream()),
                            "av_fusion_xattn_bwd"))
    return dq, dkv


def attention_bwd(q: Tensor, k: Tensor, v: Tensor, do: Tensor, dq: Tensor, dk: Tensor, dv: Tensor, klen: Optional[Tensor],
                  scale: float, o: Optional[Tensor] = None, lse: Optional[Tensor] = None, drop: Optional[tuple] = None,
                  drop_mask: Optional[Tensor] = None) -> None:
    """Backward of attention_fwd from batched MFMA GEMMs + row kernels (P is re-materialised, T x T is small here):
    P = softmax(scale QK^T); dV = P^T dO; dP = dO V^T; dS = scale P o (dP - rowsum(dP o P)); dQ = dS K; dK = dS^T Q.
    dq/dk/dv are [B,T,H,D] output views (written in place)."""
    for t, n in ((q, "q"), (k, "k"), (v, "v"), (do, "do"), (dq, "dq"), (dk, "dk"), (dv, "dv")):
        _chk_view(t, n)
    B, Tq, H, D = q.shape
    Tk = k.shape[1]
    dp, dseed, dstream = (float(drop[0]), int(drop[1]), int(drop[2])) if (drop is not None and drop[0] > 0) else (0.0, 0, 0)
    if precision.is_lp(q.dtype) and o is not None and lse is not None:     # fused flash-style backward (attention_bwd.hip)
        _chk_view(o, "o")
        st = (C.c_longlong * 16)(*[x for t in (q, k, v, o, do, dq, dk, dv) for x in (t.stride(0), t.stride(1))])
        delta = torch.empty((B, H, Tq), dtype=torch.float32, device=q.device)
        es = q.element_size()
        _probed(f"bwd{D}", 10.0 * B * H * Tq * Tk * D, B * H * D * es * (4 * Tq + 4 * Tk),
                lambda: L.check(L.lib().av_attention_bwd_mask(ptr(q), ptr(k), ptr(v), ptr(o), ptr(do), ptr(lse), ptr(delta), ptr(dq), ptr(dk), ptr(dv),
                                                              B, H, Tq, Tk, D, st, ptr(klen), scale, dp, dseed, dstream,
                                                              ptr(drop_mask) if dp > 0 else None, stream()), "av_attention_bwd"))
        return
    # unfused form (fp32 parity mode).  With dropout the row pitch equals the Philox row pitch of the attention kernels (keys padded
    # to 4, attn_common.h), so the mask of element (b, h, q, k) is the mask of linear index row * ld + k of these buffers
    ld = (Tk + 3) // 4 * 4 if dp > 0 else (Tk + 7) // 8 * 8
    dev = q.device
    drop_t = (dp, dseed, dstream)
    S = torch.empty((B, H, Tq, ld), dtype=torch.float32, device=dev)
    bh = dict(batch=B * H, batch_inner=H)
    gemm(q, k, S, M=Tq, N=Tk, K=D, lda=q.stride(1), ldb=k.stride(1), ldc=ld, sA=D, oA=q.stride(0), sB=D, oB=k.stride(0),
         sC=Tq * ld, oC=H * Tq * ld, **bh)
    P = torch.empty((B, H, Tq, ld), dtype=q.dtype, device=dev)
    L.check(L.lib().av_softmax_rows(ptr(S), ptr(P), dt(P), B * H * Tq, Tk, scale, ptr(klen), H * Tq, ld, stream()), "av_softmax_rows")
    # dV[key, d] = sum_q Pd[q, key] dO[q, d]   (Pd = P o mask / (1 - p): what the forward multiplied V with, hf:457)
    Pd = cast_dropout(P, P.dtype, drop_t) if dp > 0 else P
    gemm(Pd, do, dv, M=Tk, N=D, K=Tq, lda=ld, ldb=do.stride(1), ldc=dv.stride(1), a_mode=L.A_TRANS, b_mode=L.B_KN,
         sA=Tq * ld, oA=H * Tq * ld, sB=D, oB=do.stride(0), sC=D, oC=dv.stride(0), **bh)
    # dP = dO V^T  (into S)
    gemm(do, v, S, M=Tq, N=Tk, K=D, lda=do.stride(1), ldb=v.stride(1), ldc=ld, sA=D, oA=do.stride(0), sB=D, oB=v.stride(0),
         sC=Tq * ld, oC=H * Tq * ld, **bh)
    if dp > 0:
        S = cast_dropout(S, torch.float32, drop_t)           # dP = dPd o mask / (1 - p)
    dS = torch.empty((B, H, Tq, ld), dtype=q.dtype, device=dev)
    L.check(L.lib().av_softmax_bwd_rows(ptr(P), dt(P), ptr(S), ptr(dS), dt(dS), B * H * Tq, Tk, scale, ld, stream()), "av_softmax_bwd_rows")
    # dQ = dS K ; dK = dS^T Q
    gemm(dS, k, dq, M=Tq, N=D, K=Tk, lda=ld, ldb=k.stride(1), ldc=dq.stride(1), b_mode=L.B_KN, sA=Tq * ld, oA=H * Tq * ld,
         sB=D, oB=k.stride(0), sC=D, oC=dq.stride(0), **bh)
    gemm(dS, q, dk, M=Tk, N=D, K=Tq, lda=ld, ldb=q.stride(1), ldc=dk.stride(1), a_mode=L.A_TRANS, b_mode=L.B_KN,
         sA=Tq * ld, oA=H * Tq * ld, sB=D, oB=q.stride(0), sC=D, oC=dk.stride(0), **bh)
