"""Compute-dtype copies of parameters, kept coherent with the fp32 master weights.

``ParamCache.get`` returns a cached compute-dtype copy / re-layout of one or more parameters and rebuilds it when any of
them was updated in place (``Tensor._version``).  A bf16 entry whose storage is the plain concatenation of its
parameters' elements (``flat=True``: a cast, a ``cat`` along dim 0, a ``stack``) is also registered as the parameters'
*shadow*: ``AvAdam`` hands the shadow slices to the fused step kernel, which writes the updated weights there in bf16,
and then marks the entry fresh — so the perf path needs no cast pass over the trainable weights after an optimizer step.
"""
from __future__ import annotations

import weakref
from typing import Callable, List, Optional, Sequence

import torch

from ..precision import is_lp
from torch import Tensor

_SHADOW = {}          # id(parameter) -> (weakref to the parameter, flat bf16 view, entry); Tensor.__eq__ is elementwise, so no WeakKeyDictionary


def _versions(params: Sequence[Tensor], dtype):
    return tuple((p._version, p.data_ptr()) for p in params) + (dtype,)


class _Entry:
    __slots__ = ("ver", "val", "params", "dtype", "__weakref__")

    def __init__(self, ver, val, params, dtype):
        self.ver, self.val, self.dtype = ver, val, dtype
        self.params = [weakref.ref(p) for p in params]

    def current(self) -> bool:
        ps = [r() for r in self.params]
        return all(p is not None for p in ps) and self.ver == _versions(ps, self.dtype)

    def refresh(self):
        ps = [r() for r in self.params]
        if all(p is not None for p in ps):
            self.ver = _versions(ps, self.dtype)


class ParamCache:
    def __init__(self):
        self.d = {}

    def get(self, key, params: List[Tensor], dtype, fn: Callable[[], Tensor], flat: bool = False) -> Tensor:
        ver = _versions(params, dtype)
        hit = self.d.get(key)
        if hit is None or hit.ver != ver:
            with torch.no_grad():
                val = fn()
            hit = _Entry(ver, val, params, dtype)
            self.d[key] = hit
            if flat and is_lp(dtype) and val.is_contiguous() and val.numel() == sum(p.numel() for p in params):
                base, off = val.view(-1), 0
                for p in params:
                    pid = id(p)
                    _SHADOW[pid] = (weakref.ref(p, lambda _, pid=pid: _SHADOW.pop(pid, None)), base[off:off + p.numel()], hit)
                    off += p.numel()
        return hit.val


def lookup(p: Tensor) -> Optional[Tensor]:
    """bf16 shadow slice of parameter ``p`` if one is registered and coherent with the current master weights."""
    hit = _SHADOW.get(id(p))
    if hit is None or hit[0]() is not p:
        return None
    return hit[1] if hit[2].current() else None


def lookup_many(params: Sequence[Tensor]) -> List[Optional[Tensor]]:
    """``lookup`` for a list of parameters, validating every cache entry once."""
    ok, out = {}, []
    for p in params:
        hit = _SHADOW.get(id(p))
        if hit is None or hit[0]() is not p:
            out.append(None)
            continue
        e = hit[2]
        good = ok.get(id(e))
        if good is None:
            good = ok[id(e)] = e.current()
        out.append(hit[1] if good else None)
    return out


def mark_fresh(params: Sequence[Tensor]):
    """After the optimizer wrote the shadows of ``params`` (and bumped their versions): the entries are up to date."""
    seen, vals = set(), []
    for p in params:
        hit = _SHADOW.get(id(p))
        if hit is not None and hit[0]() is p and id(hit[2]) not in seen:
            seen.add(id(hit[2]))
            hit[2].refresh()
            vals.append(hit[2].val)
    if vals:
        # the kernel wrote the bf16 tensors through raw pointers: bump their version counters so that caches DERIVED from them
        # (ops.transpose_cached: W^T for the dX products) rebuild instead of serving last step's weights
        torch.autograd.graph.increment_version(vals)
