"""Data-parallel gradient exchange for stacks of LMUFFT layers.

The layer shards by batch: every rank runs the same kernels on its own
utterances and the only exchange step is the sum of parameter gradients
(reference: DistributedDataParallel in scripts/train_flexible.py:285-289).
Here each bucket (by default one per layer, ~0.9 MB) is a flat fp32 buffer
whose slices ARE the parameters' .grad tensors, so no pack/unpack copies are
needed; a bucket's all-reduce is launched as soon as the last of its
gradients has been accumulated, and overlaps the backward of earlier layers.
The buckets are consecutive slices of one arena, so a graph-replayed step,
whose backward runs no Python, exchanges all of them in ONE all-reduce.
On MI355X `backend="nccl"` is RCCL over xGMI.  Constant buffers (A, B, H,
fft_H) are never communicated.
"""
from __future__ import annotations

from typing import Iterable, List, Sequence

import torch
import torch.distributed as dist


_ALIGN = 64  # every gradient starts on a 256-byte boundary of its bucket (vector stores in the kernels)


class GradBuckets:
    """groups: one entry per bucket, each an iterable of Parameters or an nn.Module.

    For an LMUFFT module with fp32 parameters the bucket also becomes the
    layer's gradient sink: its backward kernels write the gradients into the
    flat buffer directly.  Every other parameter reaches the buffer through
    autograd's accumulation into `.grad` (a view of the buffer).
    Every parameter of a bucket must receive a gradient in every step (as with
    DistributedDataParallel without find_unused_parameters).
    """

    def __init__(self, groups: Sequence, process_group=None, average: bool = True, overlap: bool = True):
        from .functional import GradSink
        from .lmu import LMUFFT

        self.group = process_group
        self.average = average
        # overlap=True: a bucket's all-reduce starts as soon as its gradients are complete, during the
        # backward.  overlap=False: finish() starts them all -- for steps replayed from a HIP graph,
        # whose backward runs no Python.
        self.overlap = overlap
        self.buckets: List[torch.Tensor] = []
        self._params: List[List[torch.nn.Parameter]] = []
        self._views: List[List[torch.Tensor]] = []
        self._sinks: List = []
        self._pending: List[int] = []
        self._handles = []
        self._hooks = []
        self._sunk_modules = []
        plan = []
        for grp in groups:
            module = grp if isinstance(grp, torch.nn.Module) else None
            params = [p for p in (module.parameters() if module is not None else grp) if p.requires_grad]
            if not params:
                continue
            offs, n = [], 0
            for p in params:
                offs.append(n)
                n += -(-p.numel() // _ALIGN) * _ALIGN
            plan.append((module, params, offs, n))
        # All buckets are slices of ONE allocation when they agree on dtype and device: a step whose
        # backward ran without Python (graph replay) then exchanges every gradient with a single
        # all-reduce -- one collective latency instead of one per layer.
        kinds = {(ps[0].dtype, ps[0].device) for _, ps, _, _ in plan}
        self.arena = None
        if len(kinds) == 1:
            (dtype, device), = kinds
            self.arena = torch.zeros(sum(n for _, _, _, n in plan), dtype=dtype, device=device)
        start = 0
        for module, params, offs, n in plan:
            if self.arena is not None:
                flat = self.arena[start : start + n]
                start += n
            else:
                flat = torch.zeros(n, dtype=params[0].dtype, device=params[0].device)
            views = [flat[o : o + p.numel()].view_as(p) for o, p in zip(offs, params)]
            bi = len(self.buckets)
            self.buckets.append(flat)
            self._params.append(params)
            self._views.append(views)
            self._pending.append(len(params))
            sink = None
            if (isinstance(module, LMUFFT) and len(params) == 4
                    and all(p.dtype == torch.float32 for p in params)):
                sink = GradSink(params, views, self._make_ready(bi))
                module.grad_sink = sink
                self._sunk_modules.append(module)
            self._sinks.append(sink)
            for p, v in zip(params, views):
                p.grad = None if sink is not None else v
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(bi, v)))

    def _make_ready(self, bi: int):
        def ready(count: int):
            self._pending[bi] -= count
            if self._pending[bi] == 0:
                self._launch(bi)

        return ready

    def _make_hook(self, bi: int, view: torch.Tensor):
        def hook(param):
            # autograd delivered this gradient itself; make sure it lives in the bucket
            g = param.grad
            if g is not None and g.data_ptr() != view.data_ptr():
                view.copy_(g)
                param.grad = view
            if self._pending[bi] > 0:
                self._pending[bi] -= 1
                if self._pending[bi] == 0:
                    self._launch(bi)

        return hook

    def _launch(self, bi: int, force: bool = False) -> None:
        if not (self.overlap or force):
            return
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            self._handles.append(dist.all_reduce(self.buckets[bi], op=dist.ReduceOp.SUM, group=self.group,
                                                 async_op=True))

    def zero_grad(self) -> None:
        """Start a step: re-arm the buckets.  Buckets fed by a gradient sink need no
        clearing (their first backward overwrites); the others are zeroed in place."""
        for bi, flat in enumerate(self.buckets):
            sink = self._sinks[bi]
            if sink is not None:
                sink.filled = False
                for p in self._params[bi]:
                    p.grad = None
            else:
                flat.zero_()
            self._pending[bi] = len(self._params[bi])
        self._handles = []

    def finish(self) -> None:
        """Wait for the outstanding all-reduces and turn sums into means."""
        distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1
        if not self.overlap and distributed:
            if self.arena is not None:
                self._handles.append(dist.all_reduce(self.arena, op=dist.ReduceOp.SUM, group=self.group,
                                                     async_op=True))
            else:
                for bi in range(len(self.buckets)):
                    self._launch(bi, force=True)
        for h in self._handles:
            h.wait()
        self._handles = []
        if self.average and distributed:
            ws = dist.get_world_size(self.group)
            for flat in ([self.arena] if self.arena is not None else self.buckets):
                flat.div_(ws)

    def capture_step(self, fn):
        """Capture fn() -- forward and backward of one step on static tensors -- into a HIP graph
        and return a callable that replays it and then exchanges the gradients.  The kernels of
        this library allocate nothing and launch on the capturing stream; gradients land in the
        flat buckets, whose addresses are fixed.  Call after a few eager warm-up steps."""
        self.overlap = False
        self.zero_grad()
        graph = torch.cuda.CUDAGraph()
        # thread_local: calls made meanwhile by other threads (e.g. the process group's watchdog)
        # neither break nor join the capture
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            fn()
        self._graph = graph  # keeps the graph's private memory pool alive

        def step():
            self._handles = []
            graph.replay()
            self.finish()

        return step

    def remove(self) -> None:
        for h in self._hooks:
            h.remove()
        self._hooks = []
        for m in self._sunk_modules:
            m.grad_sink = None
        self._sunk_modules = []
