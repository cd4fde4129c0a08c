"""Data-parallel gradient exchange for stacks of LMUFFT layers.

The layer shards by batch: every rank runs the same kernels on its own
utterances and the only exchange step is the sum of parameter gradients
(reference: DistributedDataParallel in scripts/train_flexible.py:285-289).
Here each bucket (by default one per layer, ~0.9 MB) is a flat fp32 buffer
whose slices ARE the parameters' .grad tensors, so no pack/unpack copies are
needed; a bucket's all-reduce is launched as soon as the last of its
gradients has been accumulated, and overlaps the backward of earlier layers.
On MI355X `backend="nccl"` is RCCL over xGMI.  Constant buffers (A, B, H,
fft_H) are never communicated.
"""
from __future__ import annotations

from typing import Iterable, List, Sequence

import torch
import torch.distributed as dist


class GradBuckets:
    def __init__(self, param_groups: Sequence[Iterable[torch.nn.Parameter]], process_group=None,
                 average: bool = True):
        self.group = process_group
        self.average = average
        self.buckets: List[torch.Tensor] = []
        self._params: List[List[torch.nn.Parameter]] = []
        self._pending: List[int] = []
        self._handles = []
        self._hooks = []
        for gi, params in enumerate(param_groups):
            params = [p for p in params if p.requires_grad]
            if not params:
                continue
            n = sum(p.numel() for p in params)
            flat = torch.zeros(n, dtype=params[0].dtype, device=params[0].device)
            off = 0
            for p in params:
                p.grad = flat[off : off + p.numel()].view_as(p)
                off += p.numel()
            bi = len(self.buckets)
            self.buckets.append(flat)
            self._params.append(params)
            self._pending.append(len(params))
            for p in params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(bi)))

    def _make_hook(self, bi: int):
        def hook(_param):
            self._pending[bi] -= 1
            if self._pending[bi] == 0:
                self._launch(bi)

        return hook

    def _launch(self, bi: int) -> None:
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            self._handles.append(dist.all_reduce(self.buckets[bi], op=dist.ReduceOp.SUM, group=self.group,
                                                 async_op=True))

    def zero_grad(self) -> None:
        """Clear the flat buffers (gradients stay views of them) and re-arm the hooks."""
        for bi, flat in enumerate(self.buckets):
            flat.zero_()
            self._pending[bi] = len(self._params[bi])
        self._handles = []

    def finish(self) -> None:
        """Wait for the outstanding all-reduces and turn sums into means."""
        for h in self._handles:
            h.wait()
        self._handles = []
        if self.average and dist.is_available() and dist.is_initialized():
            ws = dist.get_world_size(self.group)
            if ws > 1:
                for flat in self.buckets:
                    flat.div_(ws)

    def remove(self) -> None:
        for h in self._hooks:
            h.remove()
        self._hooks = []
