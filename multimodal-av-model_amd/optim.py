"""Adam with a fused HIP step kernel (torch.optim.Adam defaults: betas (0.9, 0.999), eps 1e-8, no weight decay —
model/trainer.py:34-39).  State keys (step / exp_avg / exp_avg_sq) match torch.optim.Adam, so the reference's
checkpoint dict (main.py:47-55) stays loadable.  ``grad_scale`` folds the data-parallel 1/world_size (or a loss
un-scale) into the step so that no extra pass over the gradients is needed."""
from __future__ import annotations

import torch

from . import _lib as L
from . import ops
from .utils import shadow


class AvGradScaler:
    """torch.amp.GradScaler semantics (model/trainer.py:40,121-123 of the reference; torch/amp/grad_scaler.py:126-129 defaults: init
    65536, growth 2x after 2000 clean steps, backoff 0.5x on overflow, overflowing steps skipped) kept ENTIRELY on the device: the
    state is five floats {scale, 1/scale, found_inf, growth tracker, optimizer steps taken}; ``step`` is three launches (non-finite
    check over every gradient, fused Adam that unscales / skips, scale update) and never synchronises with the host (torch's
    ``scaler.step`` calls ``found_inf.item()``).  ``update()`` exists for call-sequence compatibility; the update has already
    happened inside ``step``.  With bf16 MFMA operands (the package's perf mode) the scaling is not needed for range - bf16 has
    fp32's exponent - but the reference's training loop is fp16 autocast + GradScaler, so its step-skipping law is reproduced."""

    def __init__(self, init_scale: float = 65536.0, growth_factor: float = 2.0, backoff_factor: float = 0.5, growth_interval: int = 2000,
                 device="cuda", enabled: bool = True):
        self.growth_factor, self.backoff_factor, self.growth_interval = float(growth_factor), float(backoff_factor), int(growth_interval)
        self.enabled = enabled
        self.state = torch.tensor([init_scale, 1.0 / init_scale, 0.0, 0.0, 0.0], dtype=torch.float32, device=device)

    def scale(self, loss: torch.Tensor) -> torch.Tensor:
        return loss * self.state[0] if self.enabled else loss

    def step(self, optimizer: "AvAdam"):
        return optimizer.step(scaler=self if self.enabled else None)

    def update(self) -> None:            # folded into step(): the device already applied the growth / backoff law
        return None

    def get_scale(self) -> float:        # host synchronisation (diagnostics / checkpoints only)
        return float(self.state[0])

    def steps_taken(self) -> int:
        return int(self.state[4])

    def state_dict(self):
        st = self.state.tolist()
        return {"scale": st[0], "growth_factor": self.growth_factor, "backoff_factor": self.backoff_factor,
                "growth_interval": self.growth_interval, "_growth_tracker": int(st[3]), "_steps_taken": int(st[4])}

    def load_state_dict(self, sd):
        self.growth_factor, self.backoff_factor = float(sd["growth_factor"]), float(sd["backoff_factor"])
        self.growth_interval = int(sd["growth_interval"])
        self.state[0] = float(sd["scale"]); self.state[1] = 1.0 / float(sd["scale"]); self.state[3] = float(sd["_growth_tracker"])
        self.state[4] = float(sd.get("_steps_taken", 0))


class _Plan:
    """Static part of one multi-tensor launch (cached per set of (parameter, hyper-parameters)): sizes, {lr, beta1, beta2, eps} per tensor,
    chunk table, step-table slots - and the pointer table of its last launch (re-uploaded only when a pointer changed)."""
    __slots__ = ("sizes", "hyper", "ct", "cs", "slots", "nch", "n", "ptr_list", "ptr_host", "ptr_dev", "ptr_evt")


class AvAdam(torch.optim.Optimizer):
    """torch.optim.Adam semantics, one fused launch per step whatever the set of parameters that received a gradient: step counts are
    PER TENSOR (torch keeps ``state[p]['step']`` per parameter; LayerDrop leaves a whole layer without gradients now and then,
    hf:774-789) and live in a device table the kernel reads and a trailing one-block launch advances.  The host-side ``state[p]['step']``
    mirrors it exactly without loss scaling; under loss scaling the device decides which steps count (``sync_steps``)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self.grad_scale = 1.0
        self._plans = {}
        self._slot = None             # id(parameter) -> slot in the device step table
        self._steps_dev = None
        self._steps_seeded = False
        self.fused_launches = 0       # diagnostics / tests: every step is one av_adam_multi call

    CHUNK = 65536

    def _slots(self):
        if self._slot is None:
            self._slot, n = {}, 0
            for group in self.param_groups:
                for p in group["params"]:
                    self._slot[id(p)] = n
                    n += 1
        return self._slot

    def _step_table(self, dev):
        """Device int32 table of per-parameter step counts, seeded from the host state (fresh optimizer: zeros; after load_state_dict:
        the checkpoint's counts)."""
        slots = self._slots()
        if self._steps_dev is None or self._steps_dev.device != torch.device(dev):
            self._steps_dev = torch.zeros(max(1, len(slots)), dtype=torch.int32, device=dev)
            self._steps_seeded = False
        if not self._steps_seeded:
            import numpy as np
            host = np.zeros(max(1, len(slots)), dtype=np.int32)
            for group in self.param_groups:
                for p in group["params"]:
                    st = self.state.get(p)
                    if st and "step" in st:
                        host[slots[id(p)]] = int(st["step"])
            self._steps_dev.copy_(ops.h2d_async(host, dev))
            self._steps_seeded = True
        return self._steps_dev

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        for st in self.state.values():                              # torch >= 2 stores 'step' as a tensor in its own checkpoints
            if "step" in st and torch.is_tensor(st["step"]):
                st["step"] = int(st["step"])
        self._steps_seeded = False                                  # the device table is re-seeded from the loaded counts

    def reset_state(self) -> None:
        """Forget moments and step counts (a fresh optimizer over the same parameters)."""
        self.state.clear()
        self._steps_seeded = False
        self._keep = None

    def state_dict(self):
        """torch.optim.Adam's format.  Under loss scaling only the device table knows the per-parameter step counts (overflowing steps are
        skipped on the device): bring the host entries up to date first, so that a checkpoint taken through this method - not only through
        ``checkpoint_dict`` - resumes with the right bias corrections."""
        self.sync_steps()
        return super().state_dict()

    def add_param_group(self, param_group):
        if getattr(self, "_steps_dev", None) is not None:
            self.sync_steps()                                       # the device table is rebuilt below: seed it from CURRENT counts
        super().add_param_group(param_group)
        self._slot = None
        self._steps_dev = None
        self._steps_seeded = False
        self._plans = {}

    def _plan(self, plist, dev) -> _Plan:
        slots = self._slots()
        key = tuple((p.data_ptr(), p.numel(), hp, slots[id(p)]) for p, hp, _ in plist)
        pl = self._plans.get(key)
        if pl is None:
            import numpy as np
            if len(self._plans) >= 64:                              # LayerDrop patterns: a handful of distinct parameter sets
                self._plans.clear()
            pl = _Plan()
            ct, cs = [], []
            for t, (p, _, _) in enumerate(plist):
                for s0 in range(0, p.numel(), self.CHUNK):
                    ct.append(t); cs.append(s0)
            pl.sizes = ops.h2d_async(np.asarray([p.numel() for p, _, _ in plist], dtype=np.int64), dev)
            pl.hyper = ops.h2d_async(np.asarray([hp for _, hp, _ in plist], dtype=np.float32).reshape(-1, 4), dev)
            pl.ct = ops.h2d_async(np.asarray(ct, dtype=np.int32), dev)
            pl.cs = ops.h2d_async(np.asarray(cs, dtype=np.int64), dev)
            pl.slots = ops.h2d_async(np.asarray([slots[id(p)] for p, _, _ in plist], dtype=np.int32), dev)
            pl.nch, pl.n = len(ct), len(plist)
            pl.ptr_list = pl.ptr_evt = None
            pl.ptr_host = torch.empty(5 * len(plist), dtype=torch.long).pin_memory()
            pl.ptr_dev = torch.empty(5 * len(plist), dtype=torch.long, device=dev)
            self._plans[key] = pl
        return pl

    @torch.no_grad()
    def sync_steps(self, scaler: "AvGradScaler" = None) -> None:
        """Copy the device-side per-parameter step counts into the torch.optim.Adam-compatible ``state[p]['step']`` entries (one host
        sync; checkpoints).  Needed under loss scaling, where overflowing steps are skipped without telling the host."""
        if self._steps_dev is None or not self._steps_seeded:
            return
        host = self._steps_dev.tolist()
        slots = self._slots()
        for group in self.param_groups:
            for p in group["params"]:
                st = self.state.get(p)
                if st and "step" in st:
                    st["step"] = int(host[slots[id(p)]])

    def step(self, closure=None, scaler: "AvGradScaler" = None):
        """One fused multi-tensor launch for all parameters that have a gradient (per-tensor step counts, per-group lr / betas / eps)."""
        loss = closure() if closure is not None else None
        plist = []
        for group in self.param_groups:
            b1, b2 = group["betas"]
            hp = (float(group["lr"]), float(b1), float(b2), float(group["eps"]))
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32:
                    raise RuntimeError("AvAdam: parameters must be float32 CUDA tensors (no CPU fallback)")
                state = self.state[p]
                if len(state) == 0:
                    state["step"] = 0
                    state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                plist.append((p, hp, state))
        if not plist:
            return loss
        dev = plist[0][0].device
        steps = self._step_table(dev)                               # BEFORE the host counts move: a fresh table is seeded from them
        pl = self._plan(plist, dev)
        grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p, _, _ in plist]
        shadows = shadow.lookup_many([p for p, _, _ in plist])  # bf16 compute copies (perf path), written by the kernel
        flat = []
        for (p, _, st), g, sh in zip(plist, grads, shadows):
            flat += [p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), sh.data_ptr() if sh is not None else 0]
        if pl.ptr_list != flat:                                     # the caching allocator usually hands the gradients the same blocks
            if pl.ptr_evt is not None:
                pl.ptr_evt.synchronize()                            # the previous upload has left the pinned buffer
            pl.ptr_host.copy_(torch.tensor(flat, dtype=torch.long))
            pl.ptr_dev.copy_(pl.ptr_host, non_blocking=True)
            pl.ptr_evt = torch.cuda.Event(); pl.ptr_evt.record()
            pl.ptr_list = flat
        if scaler is not None:
            L.check(L.lib().av_adam_multi(ops.ptr(pl.ptr_dev), ops.ptr(pl.sizes), ops.ptr(pl.hyper), ops.ptr(pl.ct), ops.ptr(pl.cs), pl.nch,
                                          self.CHUNK, ops.ptr(steps), ops.ptr(pl.slots), pl.n, float(self.grad_scale), ops.ptr(scaler.state),
                                          scaler.growth_factor, scaler.backoff_factor, scaler.growth_interval, ops.stream()), "av_adam_multi")
        else:
            L.check(L.lib().av_adam_multi(ops.ptr(pl.ptr_dev), ops.ptr(pl.sizes), ops.ptr(pl.hyper), ops.ptr(pl.ct), ops.ptr(pl.cs), pl.nch,
                                          self.CHUNK, ops.ptr(steps), ops.ptr(pl.slots), pl.n, float(self.grad_scale), None, 1.0, 1.0, 1,
                                          ops.stream()), "av_adam_multi")
            for _, _, st in plist:                                  # exact host mirror (with a scaler the device decides: sync_steps)
                st["step"] = int(st["step"]) + 1
        self.fused_launches += 1
        self._keep = grads                                      # alive until the next step (stream-ordered use)
        params = [p for p, _, _ in plist]
        torch.autograd.graph.increment_version(params)          # other compute-dtype caches (re-layouts) rebuild
        shadow.mark_fresh([p for p, sh in zip(params, shadows) if sh is not None])      # ... the shadows were just written
        return loss
