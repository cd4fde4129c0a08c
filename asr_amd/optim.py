"""Optimizer step of the reference's training loop on flat device arenas.

Host-side mirror of what DistributedTrainer.train_epoch does after the backward
(src/training/distributed.py:207-234): global gradient norm, NaN check, clip_grad_norm_
(src/training/utils.py:653-665), torch.optim.AdamW.step (built at src/training/distributed.py:92-98)
and LearningRateScheduler.step (src/training/utils.py:224-263).  There the norm is a Python loop
with one `.item()` per parameter and AdamW runs per tensor; here parameters, gradients and both
moments live in flat fp32 arenas in one order, so a step is three kernel launches
(lmufft_grad_norm2_f32, lmufft_adamw_step_f32) and nothing returns to the host.

The gradient arena is the one asr_amd.parallel.GradBuckets exchanges over RCCL when `buckets` is
given (its layout is adopted), otherwise FlatAdamW makes its own and points every `p.grad` into it.
"""
from __future__ import annotations

import ctypes
from typing import Iterable, List, Optional

import torch

from . import _lib

_ALIGN = 64  # elements; the padding GradBuckets uses between parameters


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


class LearningRateScheduler:
    """Linear warm-up to `max_lr`, then `decay_rate ** ((step - warmup_steps) // decay_steps)`;
    same constructor, attributes and `step()` as src/training/utils.py:224-263."""

    def __init__(self, optimizer, warmup_steps: int = 1000, max_lr: float = 1e-3, decay_steps: int = 10000,
                 decay_rate: float = 0.96):
        self.optimizer = optimizer
        self.warmup_steps = warmup_steps
        self.max_lr = max_lr
        self.decay_steps = decay_steps
        self.decay_rate = decay_rate
        self.step_count = 0
        # an optimizer that keeps its clock on the device evaluates this rule there (FlatAdamW(device_clock=True))
        if getattr(optimizer, "device_clock", False):
            optimizer._attach_schedule(self)

    def step(self) -> float:
        self.step_count += 1
        if self.step_count <= self.warmup_steps:
            lr = self.max_lr * (self.step_count / self.warmup_steps)
        else:
            lr = self.max_lr * self.decay_rate ** ((self.step_count - self.warmup_steps) // self.decay_steps)
        for group in self.optimizer.param_groups:
            group["lr"] = lr
        return lr


class FlatAdamW:
    """AdamW with optional global-norm clipping over flat arenas.

    `params`: the model's parameters (fp32, one CUDA device), or None with `buckets=` a GradBuckets
    whose parameters and gradient arena are used.  The parameters' storage is moved into one flat
    tensor (`p.data` becomes a view of it), as is their `.grad` unless the buckets already own it.
    `max_grad_norm`: clip to this global L2 norm (None: no clipping); a NaN norm skips the update on
    the device -- unlike the reference, which then does not call optimizer.step(), the step counter
    still advances, since nothing is read back to the host.
    `param_groups[0]["lr"]` is read at every step, so LearningRateScheduler drives it unchanged.

    `device_clock=True`: the step count and the learning rate live in device memory and the kernels advance
    them themselves (lmufft_adamw_step_clock_f32), so that step() can be captured into a HIP graph and replayed:
    launch arguments are frozen at capture, these are not.  A LearningRateScheduler built on such an optimizer
    hands its rule to the device; its own step() then only keeps the host's copy of the count and the rate (for
    logging) in step, and `param_groups[0]["lr"]` is no longer read.  An update skipped for a NaN norm does not
    advance the optimizer's step count, as in the reference; `steps` on the host is then an upper bound and
    `device_steps()` reads the true value back.
    """

    def __init__(self, params: Optional[Iterable[torch.nn.Parameter]] = None, lr: float = 1e-3, betas=(0.9, 0.999),
                 eps: float = 1e-8, weight_decay: float = 1e-2, max_grad_norm: Optional[float] = None, buckets=None,
                 device_clock: bool = False):
        if buckets is not None:
            if buckets.arena is None:
                raise RuntimeError("FlatAdamW: the buckets do not share one arena (mixed dtypes or devices)")
            plist = [p for ps in buckets._params for p in ps]
            views = [v for vs in buckets._views for v in vs]
            self.grads = buckets.arena
            offsets = [v.storage_offset() - self.grads.storage_offset() for v in views]
            own_grads = False
        else:
            plist = [p for p in params]
            if not plist:
                raise ValueError("FlatAdamW: no parameters")
            offsets, off = [], 0
            for p in plist:
                offsets.append(off)
                off += -(-p.numel() // _ALIGN) * _ALIGN
            dev = plist[0].device
            self.grads = torch.zeros(off, dtype=torch.float32, device=dev)
            own_grads = True
        for p in plist:
            if p.dtype != torch.float32 or not p.is_cuda or p.device != self.grads.device:
                raise RuntimeError("FlatAdamW: parameters must be float32 on the GPU of the gradient arena "
                                   "(the optimizer runs only as HIP kernels; there is no CPU fallback)")
        if self.grads.dtype != torch.float32:
            raise RuntimeError("FlatAdamW: the gradient arena must be float32")
        n = self.grads.numel()
        dev = self.grads.device
        self.params: List[torch.nn.Parameter] = plist
        self._offsets = offsets
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, off in zip(plist, offsets):
                dst = self.flat[off:off + p.numel()].view_as(p)
                dst.copy_(p.data)
                p.data = dst
                if own_grads:
                    p.grad = self.grads[off:off + p.numel()].view_as(p)
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self._norm2 = torch.zeros(1, dtype=torch.float32, device=dev)
        self._ws = torch.empty(_lib.lib.lmufft_grad_norm_workspace_bytes(), dtype=torch.uint8, device=dev)
        self.max_grad_norm = max_grad_norm
        self.param_groups = [{"params": plist, "lr": lr, "betas": tuple(betas), "eps": eps,
                              "weight_decay": weight_decay}]
        self.steps = 0
        self.device_clock = bool(device_clock)
        self._clock = torch.zeros(8, dtype=torch.int32, device=dev) if device_clock else None
        self._schedule = _lib.LrSchedule(float(lr), float(lr), 1.0, 0, 1, 0)

    def _attach_schedule(self, sched) -> None:
        if self.steps:
            raise RuntimeError("FlatAdamW: attach the scheduler before the first step (the device clock has started)")
        self._schedule = _lib.LrSchedule(float(self.param_groups[0]["lr"]), float(sched.max_lr), float(sched.decay_rate),
                                         int(sched.warmup_steps), int(sched.decay_steps), 1)

    def device_steps(self):
        """(scheduler steps, optimizer steps) as the device counts them; synchronises."""
        if self._clock is None:
            return self.steps, self.steps
        c = self._clock[:2].tolist()
        return int(c[0]), int(c[1])

    def device_lr(self) -> float:
        """The learning rate the last device-clocked step used; synchronises."""
        return float(self._clock[2:3].view(torch.float32).item())

    # -- the step -----------------------------------------------------------------------------
    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.flat.device).cuda_stream)

    def grad_norm(self) -> torch.Tensor:
        """Global L2 norm of the gradient arena, as a 0-d tensor on the device (no synchronisation)."""
        with torch.cuda.device(self.flat.device):
            rc = _lib.lib.lmufft_grad_norm2_f32(_ptr(self.grads), self.grads.numel(), _ptr(self._norm2), _ptr(self._ws),
                                                self._ws.numel(), self._stream())
        _lib.check(rc, "lmufft_grad_norm2_f32")
        return self._norm2[0].sqrt()

    @torch.no_grad()
    def step(self) -> None:
        g = self.param_groups[0]
        self.steps += 1
        with torch.cuda.device(self.flat.device):
            clip = self.max_grad_norm is not None
            if clip:
                rc = _lib.lib.lmufft_grad_norm2_f32(_ptr(self.grads), self.grads.numel(), _ptr(self._norm2),
                                                    _ptr(self._ws), self._ws.numel(), self._stream())
                _lib.check(rc, "lmufft_grad_norm2_f32")
            if self.device_clock:
                rc = _lib.lib.lmufft_adamw_step_clock_f32(
                    _ptr(self.flat), _ptr(self.grads), _ptr(self.exp_avg), _ptr(self.exp_avg_sq), self.flat.numel(),
                    _ptr(self._clock), ctypes.byref(self._schedule), float(g["betas"][0]), float(g["betas"][1]),
                    float(g["eps"]), float(g["weight_decay"]), _ptr(self._norm2) if clip else None,
                    float(self.max_grad_norm) if clip else 0.0, self._stream())
            else:
                rc = _lib.lib.lmufft_adamw_step_f32(_ptr(self.flat), _ptr(self.grads), _ptr(self.exp_avg),
                                                    _ptr(self.exp_avg_sq), self.flat.numel(), float(g["lr"]),
                                                    float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]),
                                                    float(g["weight_decay"]), self.steps,
                                                    _ptr(self._norm2) if clip else None,
                                                    float(self.max_grad_norm) if clip else 0.0, self._stream())
        _lib.check(rc, "lmufft_adamw_step_f32")
        # the kernel wrote through raw pointers: tell autograd (and caches keyed on `_version`) so
        torch._C._increment_version(self.params)

    def zero_grad(self, set_to_none: bool = False) -> None:
        """Zeroes the arena (the gradients are views of it, so they are never set to None)."""
        self.grads.zero_()

    # -- checkpoints: the layout torch.optim.AdamW writes (src/training/utils.py:328-334 stores it) --
    def state_dict(self) -> dict:
        state = {}
        steps = self.device_steps()[1] if self.device_clock else self.steps
        if steps:
            for i, (p, off) in enumerate(zip(self.params, self._offsets)):
                sl = slice(off, off + p.numel())
                state[i] = {"step": torch.tensor(float(steps)),
                            "exp_avg": self.exp_avg[sl].view_as(p).clone(),
                            "exp_avg_sq": self.exp_avg_sq[sl].view_as(p).clone()}
        g = self.param_groups[0]
        group = {k: v for k, v in g.items() if k != "params"}
        group.update({"amsgrad": False, "maximize": False, "params": list(range(len(self.params)))})
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd: dict) -> None:
        group = sd["param_groups"][0]
        if len(group["params"]) != len(self.params):
            raise ValueError(f"FlatAdamW: the checkpoint holds {len(group['params'])} parameters, the model "
                             f"{len(self.params)}")
        for k in ("lr", "betas", "eps", "weight_decay"):
            if k in group:
                self.param_groups[0][k] = tuple(group[k]) if k == "betas" else group[k]
        steps = set()
        with torch.no_grad():
            for i, (p, off) in enumerate(zip(self.params, self._offsets)):
                st = sd["state"].get(i, sd["state"].get(str(i)))
                sl = slice(off, off + p.numel())
                if st is None:
                    self.exp_avg[sl].zero_()
                    self.exp_avg_sq[sl].zero_()
                    continue
                self.exp_avg[sl].view_as(p).copy_(st["exp_avg"])
                self.exp_avg_sq[sl].view_as(p).copy_(st["exp_avg_sq"])
                steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError(f"FlatAdamW: parameters at different step counts {sorted(steps)}; one flat step "
                             "cannot continue them")
        self.steps = steps.pop() if steps else 0
        if self._clock is not None:
            # the scheduler's count is not part of AdamW's state: the loop restores it (set_scheduler_steps)
            self._clock[1:2].fill_(self.steps)

    def set_scheduler_steps(self, n: int) -> None:
        """Device clock: continue the learning-rate rule from `n` scheduler steps (`LearningRateScheduler.step_count`,
        src/training/utils.py:246-250), e.g. after a checkpoint restore."""
        if self._clock is not None:
            self._clock[0:1].fill_(int(n))
