"""Data-parallel gradient exchange: one process per GPU, RCCL (torch.distributed backend "nccl") over xGMI.

The reference has no distributed code; the build adds exactly one collective: all-reduce of the
63.8 M trainable-and-used gradients.  Buckets follow backward order — decoder + fusion first (ready before the
audio backward starts), then one bucket per wav2vec2 layer 9 -> 6 — each packed into one flat buffer and
all-reduced on a SIDE stream while the next layer's backward runs on the main stream; only the last bucket is
exposed.  The 1/world_size is folded into the Adam kernel (``AvAdam.grad_scale``).  Works on CPU tensors with the
gloo backend (tests) as well.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.distributed as dist


class GradArena:
    """One gradient bucket as ONE persistent flat fp32 buffer: the backward kernels write their weight / bias gradients straight into
    views of it (``out(name, shape)``), the bucket is all-reduced in place - no packing copy (``torch.cat``) and no per-step allocation,
    and the optimizer's pointer table never changes.  The layout is discovered during the first backward (``out`` returns None there and
    the caller allocates as before); from the second step on the views are served - each name at most ONCE between two ``begin_step()``
    calls: a second backward through the same module within one step (two fusion calls, a module used outside the trainer) gets None and
    allocates, so gradients that autograd already holds as views of the buffer are never overwritten.
    Vector-shaped gradients (biases, LayerNorm gains: column sums that accumulate with atomics) live in a zone of their own at the front of
    the buffer which ``begin_step()`` clears with ONE fill; their views come back zeroed and are accumulated into (a separate clear per
    column sum was 38 fill launches per step)."""

    def __init__(self, align: int = 64):
        self._served = set()
        self.layout: Dict[str, tuple] = {}        # name -> (zone, offset inside the zone, shape); zone 0 = vectors (zeroed per step), 1 = matrices
        self.zone_size = [0, 0]
        self.flat: Optional[torch.Tensor] = None
        self.align = align                        # elements: every view starts 256-byte aligned (vector loads of the kernels that read it)
        self._device = None

    @staticmethod
    def _numel(shape) -> int:
        n = 1
        for x in shape:
            n *= x
        return n

    def out(self, name: str, shape, device, vec: bool = False) -> Optional[torch.Tensor]:
        """``vec``: the view lies in the per-step-zeroed zone: the caller ACCUMULATES into it."""
        shape = tuple(int(x) for x in shape)
        zone = 0 if vec else 1
        hit = self.layout.get(name)
        if hit is None or hit[2] != shape or hit[0] != zone:
            if self.flat is not None:             # a new tensor after the layout was frozen (freeze policy changed): start over
                self.flat, self.layout, self.zone_size = None, {}, [0, 0]
            self.layout[name] = (zone, self.zone_size[zone], shape)
            self.zone_size[zone] += (self._numel(shape) + self.align - 1) // self.align * self.align
            self._device = device
            return None
        if self.flat is None or name in self._served:
            return None
        self._served.add(name)
        off = hit[1] + (self.zone_size[0] if zone == 1 else 0)
        return self.flat[off:off + self._numel(shape)].view(shape)

    def begin_step(self) -> None:
        self._served.clear()
        if self.flat is not None and self.zone_size[0] > 0:
            self.flat[:self.zone_size[0]].zero_()

    def finalize(self) -> None:
        """End of a backward in which the layout was (re)discovered: allocate the buffer; the NEXT backward writes into it."""
        if self.flat is None and sum(self.zone_size) > 0:
            self.flat = torch.zeros(sum(self.zone_size), dtype=torch.float32, device=self._device)

    def owns(self, t: torch.Tensor) -> bool:
        if self.flat is None or t is None or t.device != self.flat.device or t.dtype != torch.float32:
            return False
        lo = self.flat.data_ptr()
        return lo <= t.data_ptr() and t.data_ptr() + t.numel() * 4 <= lo + self.flat.numel() * 4


class GradBucketReducer:
    def __init__(self, group=None, side_stream: bool = True, always_collective: bool = False):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # issue the collectives even in a group of one rank (rehearsal of the RCCL path on a one-GPU box: same calls, same streams)
        self.always = always_collective and dist.is_initialized()
        self.pending = []            # (work, flat, stream)
        self.stream: Optional[torch.cuda.Stream] = None
        self._use_side = side_stream
        # bookkeeping for the N > 1 bench line: buckets issued without / with a packing copy, their sizes, exposed wait on the GPU timeline
        self.flat_reduces = 0
        self.cat_reduces = 0
        self.bucket_bytes: List[int] = []         # of the current step (cleared by wait())
        self.last_bucket_bytes: List[int] = []
        self.timing = False
        self._wait_events = []                    # (before, after) event pairs around the join of the side stream

    def _side(self, dev):
        if self.stream is None:
            self.stream = torch.cuda.Stream(device=dev)
        return self.stream

    def _issue(self, flat: torch.Tensor, also_after=()) -> None:
        """``also_after``: further streams whose enqueued work writes ``flat`` (the second audio pass's stream): the collective waits for them
        too - the CURRENT stream does not have to."""
        self.bucket_bytes.append(flat.numel() * flat.element_size())
        if self.world > 1 or self.always:
            if flat.is_cuda and self._use_side:
                s = self._side(flat.device)
                s.wait_stream(torch.cuda.current_stream(flat.device))
                for x in also_after:
                    s.wait_stream(x)
                with torch.cuda.stream(s):
                    work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                flat.record_stream(s)
                self.pending.append((work, flat, s))
            else:
                work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                self.pending.append((work, flat, None))

    def reduce_flat(self, arena: GradArena, also_after=()) -> None:
        """All-reduce (SUM) of a bucket whose gradients already live in one flat buffer: in place, nothing is copied."""
        self.flat_reduces += 1
        if (self.world > 1 or self.always) and arena.flat.is_cuda and self._use_side:
            self._issue(arena.flat, also_after)
        else:                                                    # collective on the current stream: it has to see the other streams' work itself
            for x in also_after:
                torch.cuda.current_stream(arena.flat.device).wait_stream(x)
            self._issue(arena.flat)

    def reduce_async(self, tensors: List[torch.Tensor]) -> List[torch.Tensor]:
        """Pack ``tensors`` into one flat bucket, start its all-reduce (SUM), return views of the bucket that alias
        the reduced values once ``wait()`` has been called.  (Buckets whose layout is not known yet: the first step.)"""
        tensors = [t for t in tensors if t is not None]
        if not tensors:
            return []
        self.cat_reduces += 1
        flat = torch.cat([t.reshape(-1) for t in tensors])
        views, off = [], 0
        for t in tensors:
            views.append(flat[off:off + t.numel()].view(t.shape))
            off += t.numel()
        self._issue(flat)
        return views

    def wait(self) -> None:
        dev = None
        for work, flat, s in self.pending:
            if flat.is_cuda:
                dev = flat.device
        ev0 = ev1 = None
        if self.timing and dev is not None:
            ev0 = torch.cuda.Event(enable_timing=True); ev0.record(torch.cuda.current_stream(dev))
        for work, flat, s in self.pending:
            work.wait()
            if s is not None:
                torch.cuda.current_stream(flat.device).wait_stream(s)
        if ev0 is not None:
            ev1 = torch.cuda.Event(enable_timing=True); ev1.record(torch.cuda.current_stream(dev))
            self._wait_events.append((ev0, ev1))
        self.pending.clear()
        if self.bucket_bytes:
            self.last_bucket_bytes, self.bucket_bytes = self.bucket_bytes, []

    def exposed_ms(self) -> List[float]:
        """Per joined step: time the main stream spent waiting for the gradient exchange at the join (after a device synchronisation)."""
        out = [a.elapsed_time(b) for a, b in self._wait_events]
        self._wait_events = []
        return out


def shard_batch(batch: Dict[str, torch.Tensor], rank: int, world: int) -> Dict[str, torch.Tensor]:
    """Contiguous item shard of a collated batch (independent batch items)."""
    B = next(iter(batch.values())).shape[0]
    if B % world:
        raise ValueError(f"global batch {B} is not divisible by world size {world} (CTC mean needs equal shards)")
    per = B // world
    return {k: v[rank * per:(rank + 1) * per] for k, v in batch.items()}
