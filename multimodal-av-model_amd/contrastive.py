"""contrastive_loss_with_mask on the HIP kernels (contrastive.py:8-44).

loss = 1.0 * mean(-log_softmax(A P^T / 0.07)) + 0.3 * mean(-log_softmax(A N^T / 0.07)), A = frames with mask 1
(overlap), P = mask 2 (this speaker alone), N = mask 0 (other speaker), rows with mask 3 dropped; features are
Linear(D,128)-projected and L2-normalised.  The mean runs over ALL matrix entries, so
  mean(-log_softmax(S)) = mean_i LSE_i - mean_ij S_ij     and     dS_ij = (softmax_ij - 1/N_cols) / N_rows.
The class counts are the only data-dependent sizes: pass ``counts=(n1, n2, n0)`` (known on the host from the
CPU batch) to avoid the one host sync.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch

from . import _lib as L
from . import ops
from .precision import compute_dtype

TEMPERATURE = 0.07
WEIGHT_POS_ALIGN = 1.0
WEIGHT_NEG_SUPPRESS = 0.3


# The N1 x N2 similarity matrix is never held whole (~0.5 GB per term at configs[4]): S is produced one
# [ROW_CHUNK x COL_CHUNK] block at a time into a reused workspace, reduced to the running row LSE / row sum, and RECOMPUTED block by
# block in the backward (one extra K = 128 GEMM per block).  Workspace: ROW_CHUNK x COL_CHUNK x (4 + 2) B = 100 MB at the defaults.
ROW_CHUNK = int(os.environ.get("AVAMD_CONTRAST_ROWS", "8192"))
COL_CHUNK = int(os.environ.get("AVAMD_CONTRAST_COLS", "2048"))


def _blocks(n_rows, n_cols):
    for r0 in range(0, n_rows, ROW_CHUNK):
        for c0 in range(0, n_cols, COL_CHUNK):
            yield r0, min(ROW_CHUNK, n_rows - r0), c0, min(COL_CHUNK, n_cols - c0)


def _sim_block(A_t, P_t, r0, nr, c0, nc, ws):
    """S[r0:r0+nr, c0:c0+nc] = A P^T / tau into the fp32 workspace (leading dimension = nc rounded up to 64)."""
    ld = (nc + 63) // 64 * 64
    sim = ws[: nr * ld].view(nr, ld)
    E = A_t.shape[1]
    ops.gemm(A_t, P_t, sim, M=nr, N=nc, K=E, lda=E, ldb=E, ldc=ld, alpha=1.0 / TEMPERATURE, a_off=r0 * E, b_off=c0 * E)
    return sim, ld


def _term_fwd(A_t, P_t, n_rows, n_cols):
    """returns (scalar [1] = mean_i lse_i - mean_ij s_ij, lse [n_rows])"""
    dev = A_t.device
    lse = torch.empty(n_rows, dtype=torch.float32, device=dev)
    rsum = torch.empty(n_rows, dtype=torch.float32, device=dev)
    ws = torch.empty(min(ROW_CHUNK, n_rows) * ((min(COL_CHUNK, n_cols) + 63) // 64 * 64), dtype=torch.float32, device=dev)
    for r0, nr, c0, nc in _blocks(n_rows, n_cols):
        sim, ld = _sim_block(A_t, P_t, r0, nr, c0, nc, ws)
        L.check(L.lib().av_lse_rows_chunk(ops.ptr(sim), ops.ptr(lse) + 4 * r0, ops.ptr(rsum) + 4 * r0, nr, nc, ld, int(c0 > 0), ops.stream()),
                "av_lse_rows_chunk")
    val = torch.empty(1, dtype=torch.float32, device=dev)
    L.check(L.lib().av_reduce_sum(ops.ptr(lse), n_rows, ops.ptr(val), 1.0 / n_rows, 0, ops.stream()), "av_reduce_sum")
    L.check(L.lib().av_reduce_sum(ops.ptr(rsum), n_rows, ops.ptr(val), -1.0 / (n_rows * n_cols), 1, ops.stream()), "av_reduce_sum")
    return val, lse


class _ContrastFn(torch.autograd.Function):
    @staticmethod
    def forward(fctx, mid, flat_mask, pw, pb, counts):
        dtype = compute_dtype()
        dev = mid.device
        B, T, D = mid.shape
        n1, n2, n0 = counts
        x = mid.contiguous().float().view(B * T, D)
        # class order 1 (anchors), 2 (positives), 0 (negatives), 3 (dropped): integer index plumbing only
        fm = flat_mask.contiguous()
        if fm.dtype != torch.long:
            fm = fm.long()
        order = torch.empty(B * T, dtype=torch.long, device=dev)
        L.check(L.lib().av_class_order(ops.ptr(fm), B * T, ops.ptr(order), ops.stream()), "av_class_order")
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
            val, lse = _term_fwd(f_t[:n1], f_t[n1:n1 + n2], n1, n2)
            ops.axpby(WEIGHT_POS_ALIGN, val, 1.0, total)
            terms.append((WEIGHT_POS_ALIGN, n1, n1 + n2, n2, lse))
        if n1 > 0 and n0 > 0:
            val, lse = _term_fwd(f_t[:n1], f_t[n1 + n2:n], n1, n0)
            ops.axpby(WEIGHT_NEG_SUPPRESS, val, 1.0, total)
            terms.append((WEIGHT_NEG_SUPPRESS, n1 + n2, n, n0, lse))
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
        for (wgt, k0, k1, ncols, lse) in s["terms"]:
            A_t, P_t = f_t[:n1], f_t[k0:]                                     # P_t runs into the zero rows behind the last class
            ws = torch.empty(min(ROW_CHUNK, n1) * ((min(COL_CHUNK, ncols) + 63) // 64 * 64), dtype=torch.float32, device=dev)
            ds_ws = torch.empty(ws.numel(), dtype=dtype, device=dev)
            for r0, nr, c0, nc in _blocks(n1, ncols):
                sim, ld = _sim_block(A_t, P_t, r0, nr, c0, nc, ws)           # recomputed, not stored
                dsim = ds_ws[: nr * ld].view(nr, ld)
                # d/dS of w*(mean lse - mean S) and the 1/temperature of S = A P^T / tau; columns >= nc of the block are written as zeros
                L.check(L.lib().av_contrastive_dsim_chunk(ops.ptr(sim), ops.ptr(lse) + 4 * r0, ops.ptr(dsim), ops.dt(dsim), nr, nc, ld,
                                                          wgt / (n1 * TEMPERATURE), ncols, ops.stream()), "av_contrastive_dsim_chunk")
                # dA[r0:r0+nr] += dS P_c  (ld >= nc rows of P are read: the surplus rows meet the zero columns of dS)
                ops.gemm(dsim, P_t, df, M=nr, N=E, K=ld, lda=ld, ldb=E, ldc=E, b_mode=L.B_KN, R=df, ldr=E, b_off=c0 * E, c_off=r0 * E)
                # dP_c (+)= dS^T A[r0:r0+nr]   (row chunks accumulate)
                ops.matmul_tn(dsim[:, :nc], A_t[r0:r0 + nr], out=df[k0 + c0:k0 + c0 + nc], accumulate=r0 > 0)
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
