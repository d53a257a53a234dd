"""contrastive_loss_with_mask on the HIP kernels (contrastive.py:8-44).

loss = 1.0 * mean(-log_softmax(A P^T / 0.07)) + 0.3 * mean(-log_softmax(A N^T / 0.07)), A = frames with mask 1
(overlap), P = mask 2 (this speaker alone), N = mask 0 (other speaker), rows with mask 3 dropped; features are
Linear(D,128)-projected and L2-normalised.  The mean runs over ALL matrix entries, so
  mean(-log_softmax(S)) = mean_i LSE_i - mean_ij S_ij     and     dS_ij = (softmax_ij - 1/N_cols) / N_rows.
The class counts are the only data-dependent sizes: pass ``counts=(n1, n2, n0)`` (known on the host from the
CPU batch) to avoid the one host sync.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib as L
from . import ops
from .precision import compute_dtype

TEMPERATURE = 0.07
WEIGHT_POS_ALIGN = 1.0
WEIGHT_NEG_SUPPRESS = 0.3


def _term_fwd(A_t, P_t, n_rows, n_cols):
    """returns (scalar [1] = mean_i lse_i - mean_ij s_ij, sim fp32 [n_rows, ld], lse, ld)"""
    ld = (n_cols + 63) // 64 * 64          # padded columns are written as zeros by av_contrastive_dsim: K of the dA product = ld (fast GEMM)
    dev = A_t.device
    sim = torch.empty((n_rows, ld), dtype=torch.float32, device=dev)
    ops.gemm(A_t, P_t, sim, M=n_rows, N=n_cols, K=A_t.shape[1], lda=A_t.shape[1], ldb=P_t.shape[1], ldc=ld, alpha=1.0 / TEMPERATURE)
    lse = torch.empty(n_rows, dtype=torch.float32, device=dev)
    rsum = torch.empty(n_rows, dtype=torch.float32, device=dev)
    L.check(L.lib().av_lse_rows(ops.ptr(sim), ops.ptr(lse), ops.ptr(rsum), n_rows, n_cols, ld, ops.stream()), "av_lse_rows")
    val = torch.empty(1, dtype=torch.float32, device=dev)
    L.check(L.lib().av_reduce_sum(ops.ptr(lse), n_rows, ops.ptr(val), 1.0 / n_rows, 0, ops.stream()), "av_reduce_sum")
    L.check(L.lib().av_reduce_sum(ops.ptr(rsum), n_rows, ops.ptr(val), -1.0 / (n_rows * n_cols), 1, ops.stream()), "av_reduce_sum")
    return val, sim, lse, ld


_RANK = {}


def _class_rank(dev) -> torch.Tensor:
    """mask value -> sort rank (1, 2, 0, 3), one constant tensor per device (building it per call is a blocking H2D copy)"""
    t = _RANK.get(dev)
    if t is None:
        t = _RANK[dev] = torch.tensor([2, 0, 1, 3], device=dev)
    return t


class _ContrastFn(torch.autograd.Function):
    @staticmethod
    def forward(fctx, mid, flat_mask, pw, pb, counts):
        dtype = compute_dtype()
        dev = mid.device
        B, T, D = mid.shape
        n1, n2, n0 = counts
        x = mid.contiguous().float().view(B * T, D)
        # class order 1 (anchors), 2 (positives), 0 (negatives), 3 (dropped): integer index plumbing only
        key = _class_rank(dev)[flat_mask.clamp(0, 3)]
        order = torch.argsort(key, stable=True)
        n = n1 + n2 + n0
        rows = torch.empty((n, D), dtype=dtype, device=dev)
        L.check(L.lib().av_gather_rows(ops.ptr(x), ops.dt(x), ops.ptr(order), ops.ptr(rows), ops.dt(rows), n, D, ops.stream()), "av_gather_rows")
        E = pw.shape[0]
        proj = ops.linear(rows, ops.cast(pw.data.contiguous(), dtype), pb.data, out_dtype=torch.float32)
        f = torch.empty_like(proj); nrm = torch.empty(n, dtype=torch.float32, device=dev)
        L.check(L.lib().av_l2norm_fwd(ops.ptr(proj), ops.ptr(f), ops.ptr(nrm), n, E, 1e-12, ops.stream()), "av_l2norm_fwd")
        # 64 zero rows behind the last class: the dA product reads ld >= ncols rows of its class block (x zero columns of dS)
        f_t = torch.zeros((n + 64, E), dtype=dtype, device=dev)
        f_t[:n].copy_(f)
        total = torch.zeros(1, dtype=torch.float32, device=dev)
        terms = []
        if n1 > 0 and n2 > 0:
            val, sim, lse, ld = _term_fwd(f_t[:n1], f_t[n1:n1 + n2], n1, n2)
            ops.axpby(WEIGHT_POS_ALIGN, val, 1.0, total)
            terms.append((WEIGHT_POS_ALIGN, n1, n1 + n2, n2, sim, lse, ld))
        if n1 > 0 and n0 > 0:
            val, sim, lse, ld = _term_fwd(f_t[:n1], f_t[n1 + n2:n], n1, n0)
            ops.axpby(WEIGHT_NEG_SUPPRESS, val, 1.0, total)
            terms.append((WEIGHT_NEG_SUPPRESS, n1 + n2, n, n0, sim, lse, ld))
        fctx.s = dict(terms=terms, f=f, f_t=f_t, nrm=nrm, rows=rows, order=order, n=n, n1=n1, shape=(B, T, D), E=E, dtype=dtype)
        fctx.pw = pw
        return total.view(())

    @staticmethod
    def backward(fctx, dloss):
        s = fctx.s
        B, T, D = s["shape"]
        n, n1, E, dtype = s["n"], s["n1"], s["E"], s["dtype"]
        dev = s["f"].device
        df = torch.zeros((n, E), dtype=torch.float32, device=dev)
        f_t = s["f_t"]
        for (wgt, c0, c1, ncols, sim, lse, ld) in s["terms"]:
            dsim = torch.empty((n1, ld), dtype=dtype, device=dev)
            # d/dS of w*(mean lse - mean S) and the 1/temperature of S = A P^T / tau
            L.check(L.lib().av_contrastive_dsim(ops.ptr(sim), ops.ptr(lse), ops.ptr(dsim), ops.dt(dsim), n1, ncols, ld,
                                                wgt / (n1 * TEMPERATURE), ops.stream()), "av_contrastive_dsim")
            Pm = f_t[c0:]                                                     # ld rows are read (rows >= ncols meet zero columns)
            # dA += dS P ; dP = dS^T A
            ops.gemm(dsim, Pm, df, M=n1, N=E, K=ld, lda=ld, ldb=E, ldc=E, b_mode=L.B_KN, R=df, ldr=E)
            ops.matmul_tn(dsim[:, :ncols], f_t[:n1], out=df[c0:c1])          # dP = dS^T A  (transposes + fast GEMM)
        dproj = torch.empty_like(df)
        L.check(L.lib().av_l2norm_bwd(ops.ptr(s["f"]), ops.ptr(df), ops.ptr(s["nrm"]), ops.ptr(dproj), n, E, 1e-12, ops.stream()), "av_l2norm_bwd")
        dproj_t = ops.cast(dproj, dtype)
        drows = ops.matmul_nn(dproj_t, ops.cast(fctx.pw.data.contiguous(), dtype), out_dtype=torch.float32)     # [n, D]
        dmid = torch.zeros((B * T, D), dtype=torch.float32, device=dev)
        L.check(L.lib().av_scatter_rows(ops.ptr(drows), ops.ptr(s["order"]), ops.ptr(dmid), n, D, 1.0, 0, ops.stream()), "av_scatter_rows")
        dmid = dmid.view(B, T, D)
        g = dloss.reshape(1).float()
        out = torch.empty_like(dmid)
        # scale by the incoming scalar gradient on the device (no host sync): out = g * dmid
        L.check(L.lib().av_mul_scalar_dev(ops.ptr(dmid), ops.ptr(g), ops.ptr(out), dmid.numel(), ops.stream()), "av_mul_scalar_dev")
        fctx.s = None
        return out, None, None, None, None


def contrastive_loss_with_mask(middle_feat, flat_mask, projection_layer=None, counts: Optional[Tuple[int, int, int]] = None):
    """Same signature as the reference (contrastive.py:8); ``counts`` = (#mask==1, #mask==2, #mask==0) is optional."""
    if projection_layer is None:
        raise NotImplementedError("the HIP contrastive loss needs the projection layer (the trainer always passes one)")
    if counts is None:
        c = torch.bincount(flat_mask.clamp(0, 3), minlength=4).tolist()      # the single host sync of this loss
        counts = (c[1], c[2], c[0])
    if counts[0] == 0 or (counts[1] == 0 and counts[2] == 0):
        return torch.tensor(0.0, device=middle_feat.device, requires_grad=True)   # contrastive.py:28
    return _ContrastFn.apply(middle_feat, flat_mask, projection_layer.weight, projection_layer.bias, tuple(counts))
