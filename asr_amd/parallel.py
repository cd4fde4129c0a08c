"""Data-parallel gradient exchange for stacks of LMUFFT layers.

The layer shards by batch: every rank runs the same kernels on its own
utterances and the only exchange step is the sum of parameter gradients
(reference: DistributedDataParallel in scripts/train_flexible.py:285-289, the
per-GPU batch is total batch / world size, :198-203).  Here each bucket (by
default one per layer, ~0.9 MB; `bucket_mb` merges consecutive layers) is a
flat fp32 buffer whose slices ARE the parameters' .grad tensors, so no
pack/unpack copies are needed.  Three ways to run a step:

  eager, overlap=True   a bucket's all-reduce is launched as soon as the last
                        of its gradients has been written, and overlaps the
                        backward of the layers below;
  eager, deferred       (overlap=False, or inside `accumulate()`): nothing is
                        launched during the backward(s); finish() exchanges the
                        whole arena in ONE all-reduce.  This is the mode for
                        gradient accumulation (several backwards per step);
  HIP graph replay      capture_step (one graph, one all-reduce after it) or
                        capture_pipeline (the step cut into segments, one graph
                        each; the buckets completed by a segment are exchanged
                        while the next segment's graph runs).

On MI355X `backend="nccl"` is RCCL over xGMI.  Constant buffers (A, B, H,
fft_H) are never communicated.
"""
from __future__ import annotations

import contextlib
from typing import Callable, List, Sequence

import torch
import torch.distributed as dist


_ALIGN = 64  # every gradient starts on a 256-byte boundary of its bucket (vector stores in the kernels)


# Functions that complete gradients whose last reduction step was left for one common launch per step; a layer
# module registers its own (register_deferred_sums) and GradBuckets runs them all before it exchanges or hands out
# gradients.
_DEFERRED_SUMS = []


def register_deferred_sums(fn) -> None:
    if fn not in _DEFERRED_SUMS:
        _DEFERRED_SUMS.append(fn)


def run_deferred_sums() -> None:
    for fn in _DEFERRED_SUMS:
        fn()


def grad_slot(p):
    """The slice of a gradient bucket a backward may write `p`'s gradient into, or None: `p` belongs to a
    GradBuckets, is float32 and no backward has asked for the slice in this step.  Return `adopt(slot)` as the gradient:
    a fresh tensor on the slice's memory, which autograd takes over as `p.grad` without copying (it copies when
    other references to the tensor object exist, which is why the slice itself must not be returned)."""
    v = getattr(p, "_grad_view", None)
    if v is None or p.grad is not None or not getattr(p, "_grad_view_free", False) or v.dtype != torch.float32:
        return None
    # handed out once per step: autograd sums the gradients of a parameter used several times before it
    # touches `.grad`, so a second writer into the same slice would corrupt the first one's result
    p._grad_view_free = False
    return v


def adopt(slot: torch.Tensor) -> torch.Tensor:
    return slot.view(slot.shape)


def release_slot(p) -> None:
    """Hands back a slice obtained from grad_slot that was not written after all."""
    if hasattr(p, "_grad_view"):
        p._grad_view_free = True


def fused_slot(params):
    """(one tensor [sum of rows, ...] over the bucket slices of `params`, the slices) when all of them are free and
    lie back to back in the bucket in this order (the slices are then taken); else None (nothing is taken)."""
    slots = [grad_slot(p) for p in params]
    ok = all(s is not None for s in slots)
    if ok:
        for a, b in zip(slots, slots[1:]):
            ok = ok and b.data_ptr() == a.data_ptr() + a.numel() * a.element_size() and a.shape[1:] == b.shape[1:]
    if not ok:
        for p, s in zip(params, slots):
            if s is not None:
                release_slot(p)
        return None
    first = slots[0]
    shape = (sum(s.shape[0] for s in slots),) + tuple(first.shape[1:])
    return first.as_strided(shape, first.stride()), slots


def _group_params(grp):
    module = grp if isinstance(grp, torch.nn.Module) else None
    params = [p for p in (module.parameters() if module is not None else grp) if p.requires_grad]
    return module, params


class GradBuckets:
    """groups: modules and/or iterables of Parameters, in the order their gradients become ready
    (the last layer of a stack first).

    bucket_mb = None: one bucket per group.  Otherwise consecutive groups are merged into buckets
    of at most `bucket_mb` MB of gradients (a group larger than that is a bucket of its own;
    `float("inf")` puts everything into one bucket).

    For an LMUFFT module with fp32 parameters the bucket also becomes the layer's gradient sink:
    its backward kernels write the gradients into the flat buffer directly.  Every other parameter
    starts a step with `.grad = None` and carries its slice of the buffer as `p._grad_view`: a backward
    that knows of it (asr_amd.linear, asr_amd.layernorm) writes the gradient there and returns the
    view, which autograd adopts as `.grad` without a copy; any other gradient is copied into the
    slice when it arrives.  Further gradients for the same parameter in the same step are added in
    place by autograd.  No slice is cleared between steps: the first arrival overwrites.
    Every parameter of a bucket must receive a gradient in every step (as with
    DistributedDataParallel without find_unused_parameters): finish() raises otherwise.
    With overlap=True a step has exactly ONE backward: a gradient that arrives for a bucket whose
    all-reduce has already been launched raises.  Use `with buckets.accumulate():` (or
    overlap=False) for steps with several backwards.
    """

    def __init__(self, groups: Sequence, process_group=None, average: bool = True, overlap: bool = True,
                 bucket_mb=None, exchange_single_rank: bool = False):
        from .functional import GradSink
        from .lmu import LMUFFT

        self.group = process_group
        self.average = average
        self.overlap = overlap
        # run the collectives even in a one-rank group (they change nothing there; used to exercise the
        # RCCL stream hand-over on a single GPU)
        self.exchange_single_rank = exchange_single_rank
        # False: every step runs as on a single rank -- no collective is issued (used to time a step without its
        # exchange beside the same step with it; the gradients are then NOT synchronised)
        self.exchange_enabled = True
        self.buckets: List[torch.Tensor] = []
        self._params: List[List[torch.nn.Parameter]] = []
        self._views: List[List[torch.Tensor]] = []
        self._sinks: List[list] = []        # per bucket: the gradient sinks of its LMUFFT members
        self._pending: List[int] = []
        self._launched: List[bool] = []
        self._handles = []
        self._hooks = []
        self._sunk_modules = []
        self._deferred = 0       # depth of accumulate() contexts
        self._capturing = False  # inside a graph capture: record which buckets complete, launch nothing
        self._ready_log: List[int] = []
        self._replay = False     # steps come from graph replays: no Python runs in the backward

        members = []
        for grp in groups:
            module, params = _group_params(grp)
            if params:
                members.append((module, params, sum(-(-p.numel() // _ALIGN) * _ALIGN for p in params)))
        # bucket plan
        plan: List[list] = []
        cap = None if bucket_mb is None else float(bucket_mb) * (1 << 20)
        for mem in members:
            nbytes = mem[2] * mem[1][0].element_size()
            if (cap is not None and plan and plan[-1][1] + nbytes <= cap
                    and plan[-1][0][0][1][0].dtype == mem[1][0].dtype
                    and plan[-1][0][0][1][0].device == mem[1][0].device):
                plan[-1][0].append(mem)
                plan[-1][1] += nbytes
            else:
                plan.append([[mem], nbytes])
        # All buckets are slices of ONE allocation when they agree on dtype and device: a deferred
        # step then exchanges every gradient with a single all-reduce -- one collective latency
        # instead of one per bucket.
        kinds = {(mems[0][1][0].dtype, mems[0][1][0].device) for mems, _ in plan}
        self.arena = None
        if len(kinds) == 1:
            (dtype, device), = kinds
            self.arena = torch.zeros(sum(m[2] for mems, _ in plan for m in mems), dtype=dtype, device=device)
        start = 0
        for mems, _ in plan:
            n = sum(m[2] for m in mems)
            if self.arena is not None:
                flat = self.arena[start:start + n]
                start += n
            else:
                flat = torch.zeros(n, dtype=mems[0][1][0].dtype, device=mems[0][1][0].device)
            bi = len(self.buckets)
            self.buckets.append(flat)
            all_params, all_views, sinks = [], [], []
            off = 0
            for module, params, _ in mems:
                views = []
                for p in params:
                    views.append(flat[off:off + p.numel()].view_as(p))
                    off += -(-p.numel() // _ALIGN) * _ALIGN
                sink = None
                if (isinstance(module, LMUFFT) and len(params) == 4
                        and all(p.dtype == torch.float32 for p in params)):
                    sink = GradSink(params, views, self._make_ready(bi), self._make_late(bi))
                    module.grad_sink = sink
                    self._sunk_modules.append(module)
                    sinks.append(sink)
                for p, v in zip(params, views):
                    p.grad = None
                    if sink is None:
                        p._grad_view = v
                        p._grad_view_free = True
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(bi, v, sink)))
                all_params.extend(params)
                all_views.extend(views)
            self._params.append(all_params)
            self._views.append(all_views)
            self._sinks.append(sinks)
            self._pending.append(len(all_params))
            self._launched.append(False)

    # -- bookkeeping ------------------------------------------------------------------------
    def _distributed(self) -> bool:
        if not self.exchange_enabled or not (dist.is_available() and dist.is_initialized()):
            return False
        return dist.get_world_size(self.group) > 1 or self.exchange_single_rank

    def _immediate(self) -> bool:
        return self.overlap and self._deferred == 0

    def _check_open(self, bi: int) -> None:
        if self._launched[bi]:
            raise RuntimeError(
                f"GradBuckets: a gradient arrived for bucket {bi} after its all-reduce had been launched in this "
                "step (a second backward, or a layer used twice). Run such steps inside "
                "`with buckets.accumulate():` or construct GradBuckets(overlap=False); finish() then exchanges "
                "the accumulated gradients.")

    def _arrived(self, bi: int, count: int) -> None:
        self._check_open(bi)
        if self._pending[bi] > 0:
            self._pending[bi] = max(0, self._pending[bi] - count)
            if self._pending[bi] == 0:
                if self._capturing:
                    self._ready_log.append(bi)
                elif self._immediate():
                    self._launch(bi)

    def _make_ready(self, bi: int):
        def ready(count: int):
            self._arrived(bi, count)

        return ready

    def _make_late(self, bi: int):
        return lambda: self._check_open(bi)

    def _make_hook(self, bi: int, view: torch.Tensor, sink=None):
        def hook(param):
            # A layer whose sink took this step's gradients has been counted by the sink itself.
            # Autograd may still run this hook for the same backward (it does so although the
            # layer returned no gradient), and it runs it again when a further backward adds into
            # the view -- that case is refused, or allowed, where the layer's backward starts
            # (GradSink.on_late), before any kernel has touched the bucket.
            if sink is not None and sink.filled:
                return
            # autograd delivered this gradient itself; make sure it lives in the bucket
            g = param.grad
            if g is not None and g.data_ptr() != view.data_ptr():
                view.copy_(g)
                param.grad = view
            self._arrived(bi, 1)

        return hook

    def _all_reduce(self, flat: torch.Tensor) -> None:
        # RCCL averages inside the collective; gloo has no AVG, finish() divides there
        op = dist.ReduceOp.SUM
        if self.average and dist.get_backend(self.group) == "nccl":
            op = dist.ReduceOp.AVG
        self._handles.append((dist.all_reduce(flat, op=op, group=self.group, async_op=True), flat,
                              op == dist.ReduceOp.SUM and self.average))

    def _launch(self, bi: int) -> None:
        self._launched[bi] = True
        if self._distributed():
            self._join_side_streams()
            self._all_reduce(self.buckets[bi])

    @staticmethod
    def _join_side_streams() -> None:
        # weight gradients that layers with `async_weight_grad` run on a side stream write into the buckets
        from .functional import join_async_weight_grads

        run_deferred_sums()  # gradient sums that layers left for one common launch (asr_amd.linear, asr_amd.layernorm)
        join_async_weight_grads()

    # -- step protocol ----------------------------------------------------------------------
    @contextlib.contextmanager
    def accumulate(self):
        """Inside this context no all-reduce is launched during a backward, so a step may run
        several backwards (gradient accumulation; the reference declares accumulate_grad_batches,
        src/config/config.py:42).  finish() exchanges the sums."""
        self._deferred += 1
        try:
            yield self
        finally:
            self._deferred -= 1

    def zero_grad(self) -> None:
        """Start a step: re-arm the buckets.  Nothing is cleared: every parameter's first gradient of the
        step overwrites its slice (`.grad` is None until then)."""
        for bi in range(len(self.buckets)):
            for sink in self._sinks[bi]:
                sink.filled = False
            for p in self._params[bi]:
                p.grad = None
                if hasattr(p, "_grad_view"):
                    p._grad_view_free = True
            self._pending[bi] = len(self._params[bi])
            self._launched[bi] = False
        self._handles = []

    def finish(self) -> None:
        """Launch what has not been launched, wait for the all-reduces, turn sums into means."""
        self._join_side_streams()
        if not self._replay:
            missing = [bi for bi in range(len(self.buckets)) if self._pending[bi] > 0]
            if missing:
                raise RuntimeError(
                    f"GradBuckets.finish(): buckets {missing} are incomplete -- "
                    f"{[self._pending[bi] for bi in missing]} of their parameters received no gradient in this "
                    "step; every parameter of a bucket must get one (freeze unused parameters with "
                    "requires_grad_(False) before building the buckets)")
        todo = [bi for bi in range(len(self.buckets)) if not self._launched[bi]]
        if todo and self._distributed():
            if len(todo) == len(self.buckets) and self.arena is not None:
                self._all_reduce(self.arena)
            else:
                for bi in todo:
                    self._all_reduce(self.buckets[bi])
        for bi in todo:
            self._launched[bi] = True
        ws = dist.get_world_size(self.group) if self._distributed() else 1
        for h, flat, divide in self._handles:
            h.wait()
            if divide:
                flat.div_(ws)
        self._handles = []

    # -- HIP graph replay -------------------------------------------------------------------
    def _dropout_epoch_bump(self):
        """The first node of a captured step: increments the device's dropout epoch counter, so that every replay
        draws fresh dropout masks (asr_amd.dropout).  Returns the callable to run inside the capture; the counter
        itself is created here, outside it."""
        from .dropout import bump_dropout_epoch, dropout_epoch

        dev = self.buckets[0].device
        if dev.type != "cuda":
            return lambda: None
        dropout_epoch(dev)
        return lambda: bump_dropout_epoch(dev)

    def _capture(self, fn: Callable, pool=None, first: bool = True):
        graph = torch.cuda.CUDAGraph()
        self._ready_log = []
        self._capturing = True
        bump = self._dropout_epoch_bump() if first else (lambda: None)
        try:
            # thread_local: calls made meanwhile by other threads (e.g. the process group's
            # watchdog) neither break nor join the capture.  The segments share one memory pool:
            # a later segment reads tensors an earlier one produced.
            with torch.cuda.graph(graph, pool=pool, capture_error_mode="thread_local"):
                # every parameter enters the capture with `.grad = None` (capture_pipeline re-arms the
                # step first), so the captured kernels overwrite the slices and a replay needs no clearing
                bump()
                fn()
                self._join_side_streams()  # a segment graph ends with every stream it forked joined
        finally:
            self._capturing = False
        return graph, list(self._ready_log)

    def capture_step(self, fn: Callable):
        """Capture fn() -- forward and backward of one step on static tensors -- into a HIP graph
        and return a callable that replays it and then exchanges the gradients (one all-reduce
        over the arena).  The kernels of this library allocate nothing and launch on the capturing
        stream; gradients land in the flat buckets, whose addresses are fixed.  Call after a few
        eager warm-up steps."""
        return self.capture_pipeline([fn], overlap=False)

    def capture_pipeline(self, fns: Sequence[Callable], overlap: bool = True):
        """Capture a step given as consecutive segments fns[0], fns[1], ... (e.g. forward + the last
        layer's backward, then one backward segment per remaining layer), one HIP graph each.
        The returned step() replays them in order; with overlap=True the buckets a segment
        completed are all-reduced (on RCCL's stream) while the following segments run, otherwise
        one all-reduce follows the last segment.  Every bucket must be completed by some segment.
        Before calling, drop every reference to autograd graphs of earlier eager steps (outputs,
        losses, detached-boundary tensors): autograd would otherwise reuse their gradient-accumulation
        nodes, which belong to the eager stream, inside the capture."""
        self.zero_grad()
        graphs, ready = [], []
        for i, fn in enumerate(fns):
            g, r = self._capture(fn, pool=graphs[0].pool() if graphs else None, first=not graphs)
            graphs.append(g)
            ready.append(r)
        done = sorted(bi for r in ready for bi in r)
        if done != list(range(len(self.buckets))):
            raise RuntimeError(f"capture_pipeline: the captured segments completed buckets {done}, "
                               f"expected all of 0..{len(self.buckets) - 1}")
        self._graphs = graphs  # keeps the graphs' private memory pools alive
        self._replay = True

        def step():
            self._handles = []
            for bi in range(len(self.buckets)):
                self._launched[bi] = False
            for g, r in zip(graphs, ready):
                g.replay()
                if overlap:
                    for bi in r:
                        self._launch(bi)
            self.finish()

        return step

    def capture_graph_exchange(self, fns: Sequence[Callable], overlap: bool = True, after: Callable = None):
        """Capture the whole step -- the segments fns[0], fns[1], ... AND its all-reduces -- into ONE HIP graph;
        the returned step() is a single replay with no Python and no host-side launch in it.

        The collectives are issued during the capture exactly where an eager step would issue them (with
        overlap=True as soon as a bucket's last gradient has been written, else one all-reduce over the arena
        after the last segment), so inside the graph they are branches that fork off the backward and join at its
        end: a bucket's exchange runs beside the backward of the layers below it.  capture_pipeline launches
        the same collectives from the host between its segment graphs, which costs a short step more than the
        exchange can hide (three stacked layers, batch 64 x 400 frames, one rank with the RCCL path on: 0.392 ms
        without exchange, 0.474 ms with per-bucket host launches, 0.420 ms with one host launch per step).
        `after`, if given, is captured behind the exchange -- the optimizer step of a whole training step (one whose
        learning rate and step count live in device memory: FlatAdamW(device_clock=True)).
        Same preconditions as capture_pipeline.  Every rank must capture and replay the same sequence.
        Dropout inside the step: its seeds are frozen at capture; the graph's first node increments the device's
        dropout epoch counter, which the kernels add to the seeds when they run, so replays draw fresh masks."""
        self.zero_grad()
        graph = torch.cuda.CUDAGraph()
        saved = self.overlap
        self.overlap = overlap
        bump = self._dropout_epoch_bump()
        try:
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                bump()
                for fn in fns:
                    fn()
                self.finish()  # launches what is left and waits: the waits become the graph's joins
                if after is not None:
                    after()
        finally:
            self.overlap = saved
        self._graphs = [graph]
        self._replay = True
        return graph.replay

    def remove(self) -> None:
        for h in self._hooks:
            h.remove()
        self._hooks = []
        for m in self._sunk_modules:
            m.grad_sink = None
        self._sunk_modules = []
        for ps in self._params:
            for p in ps:
                if hasattr(p, "_grad_view"):
                    del p._grad_view, p._grad_view_free
