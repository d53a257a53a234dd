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
                "growth_interval": self.growth_interval, "_growth_tracker": int(st[3])}

    def load_state_dict(self, sd):
        self.growth_factor, self.backoff_factor = float(sd["growth_factor"]), float(sd["backoff_factor"])
        self.growth_interval = int(sd["growth_interval"])
        self.state[0] = float(sd["scale"]); self.state[1] = 1.0 / float(sd["scale"]); self.state[3] = float(sd["_growth_tracker"])


class AvAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self.grad_scale = 1.0

    CHUNK = 65536

    def _plan(self, plist):
        """Static part of the multi-tensor launch: sizes, learning rates, chunk table (cached per parameter set)."""
        key = tuple((p.data_ptr(), p.numel(), lr) for p, lr, _ in plist)
        if getattr(self, "_plan_key", None) != key:
            dev = plist[0][0].device
            sizes = torch.tensor([p.numel() for p, _, _ in plist], dtype=torch.long, device=dev)
            lrs = torch.tensor([lr for _, lr, _ in plist], dtype=torch.float32, device=dev)
            ct, cs = [], []
            for t, (p, _, _) in enumerate(plist):
                for s0 in range(0, p.numel(), self.CHUNK):
                    ct.append(t); cs.append(s0)
            self._plan_key = key
            self._plan_data = (sizes, lrs, torch.tensor(ct, dtype=torch.int32, device=dev), torch.tensor(cs, dtype=torch.long, device=dev), len(ct))
        return self._plan_data

    @torch.no_grad()
    def sync_steps(self, scaler: "AvGradScaler") -> None:
        """Under loss scaling the number of steps actually taken lives on the device (overflowing steps are skipped without telling
        the host): copy it into the torch.optim.Adam-compatible ``state[p]['step']`` entries (one host sync; checkpoints)."""
        n = scaler.steps_taken()
        for st in self.state.values():
            if "step" in st:
                st["step"] = n

    def step(self, closure=None, scaler: "AvGradScaler" = None):
        """One fused multi-tensor launch for all parameters that have a gradient (same step count, per-group lr)."""
        loss = closure() if closure is not None else None
        plist = []
        for group in self.param_groups:
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
                state["step"] = int(state["step"]) + 1
                plist.append((p, float(group["lr"]), state))
        if not plist:
            return loss
        steps = {st["step"] for _, _, st in plist}
        b1, b2 = self.param_groups[0]["betas"]
        eps = float(self.param_groups[0]["eps"])
        if len(steps) != 1 or any(g["betas"] != (b1, b2) or g["eps"] != eps for g in self.param_groups):
            if scaler is not None:
                raise NotImplementedError("AvAdam: loss scaling needs one step count / betas / eps for all parameters (the fused path)")
            return self._step_per_tensor(plist, loss)
        sizes, lrs, ct, cs, nch = self._plan(plist)
        grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p, _, _ in plist]
        shadows = shadow.lookup_many([p for p, _, _ in plist])  # bf16 compute copies (perf path), written by the kernel
        flat = []
        for (p, _, st), g, sh in zip(plist, grads, shadows):
            flat += [p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), sh.data_ptr() if sh is not None else 0]
        if getattr(self, "_ptr_list", None) != flat:            # the caching allocator usually hands the gradients the same blocks
            dev = plist[0][0].device
            if getattr(self, "_ptr_host", None) is None or self._ptr_host.numel() != len(flat):
                self._ptr_host = torch.empty(len(flat), dtype=torch.long).pin_memory()
                self._ptr_dev = torch.empty(len(flat), dtype=torch.long, device=dev)
                self._ptr_evt = None
            if self._ptr_evt is not None:
                self._ptr_evt.synchronize()                     # the previous upload has left the pinned buffer
            self._ptr_host.copy_(torch.tensor(flat, dtype=torch.long))
            self._ptr_dev.copy_(self._ptr_host, non_blocking=True)
            self._ptr_evt = torch.cuda.Event(); self._ptr_evt.record()
            self._ptr_list = flat
        if scaler is not None:
            if getattr(self, "_scaler_seeded", None) is not scaler:       # resume: the device-side step count starts from the optimizer's
                scaler.state[4] = float(steps.pop() - 1)
                self._scaler_seeded = scaler
            L.check(L.lib().av_adam_multi_scaled(ops.ptr(self._ptr_dev), ops.ptr(sizes), ops.ptr(lrs), ops.ptr(ct), ops.ptr(cs), nch, self.CHUNK,
                                                 float(b1), float(b2), eps, float(self.grad_scale), ops.ptr(scaler.state), scaler.growth_factor,
                                                 scaler.backoff_factor, scaler.growth_interval, ops.stream()), "av_adam_multi_scaled")
        else:
            L.check(L.lib().av_adam_multi(ops.ptr(self._ptr_dev), ops.ptr(sizes), ops.ptr(lrs), ops.ptr(ct), ops.ptr(cs), nch, self.CHUNK, float(b1),
                                          float(b2), eps, steps.pop(), float(self.grad_scale), ops.stream()), "av_adam_multi")
        self._keep = grads                                      # alive until the next step (stream-ordered use)
        params = [p for p, _, _ in plist]
        torch.autograd.graph.increment_version(params)          # other compute-dtype caches (re-layouts) rebuild
        shadow.mark_fresh([p for p, sh in zip(params, shadows) if sh is not None])      # ... the shadows were just written
        return loss

    def _step_per_tensor(self, plist, loss):
        st_ = ops.stream()
        fn = L.lib().av_adam_step
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                state = self.state[p]
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                L.check(fn(ops.ptr(p), ops.ptr(g), ops.ptr(state["exp_avg"]), ops.ptr(state["exp_avg_sq"]), p.numel(), float(group["lr"]),
                           float(b1), float(b2), float(group["eps"]), state["step"], float(self.grad_scale), st_), "av_adam_step")
                torch.autograd.graph.increment_version(p)
        return loss
