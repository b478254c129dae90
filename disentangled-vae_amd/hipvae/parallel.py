"""Data-parallel gradient synchronisation: one process per GPU, bucketed all-reduce (RCCL over xGMI
when the backend is 'nccl'), launched from gradient hooks so that early buckets travel while the rest
of backward still runs.

The reference has no data parallelism; the semantics fixed here are those of N independent replicas
that average gradients once per step:
  * parameters are broadcast from rank 0 at construction;
  * BatchNorm statistics and the independence term stay per rank (no cross-rank batch statistics);
  * gradients are summed over ranks and divided by the world size BEFORE clipping and Adam, so every
    rank clips the same global gradient and takes the same step;
  * buckets are filled in reverse parameter order (the decoder's gradients are produced first).

The all-reduce runs on a side stream; `finish()` makes the compute stream wait for it and points every
p.grad at its slice of the reduced bucket (no copy back).
"""
from typing import List, Optional

import torch
import torch.distributed as dist


class _Bucket:
    def __init__(self, params: List[torch.nn.Parameter], device, dtype):
        self.params = params
        self.numel = sum(p.numel() for p in params)
        self.flat = torch.zeros(self.numel, dtype=dtype, device=device)
        self.views = []
        off = 0
        for p in params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self.pending = len(params)
        self.got = [False] * len(params)
        self.work = None
        self.event = None
        self.streams = {}          # streams on which gradients of this bucket were produced this step



class GradSync:
    """Bucketed, overlapped gradient averaging for a replica of the model."""

    def __init__(self, module: torch.nn.Module, bucket_mb: float = 16.0, process_group=None,
                 broadcast_parameters: bool = True, overlap: bool = True):
        if not dist.is_initialized():
            raise RuntimeError("GradSync needs an initialised torch.distributed process group")
        self.group = process_group
        self.world = dist.get_world_size(self.group)
        # RCCL can average in the collective itself; gloo (CPU tests) sums and we divide afterwards
        self.use_avg = dist.get_backend(self.group) == 'nccl'
        self.module = module
        params = [p for p in module.parameters() if p.requires_grad]
        if not params:
            raise RuntimeError("module has no trainable parameters")
        self.device = params[0].device
        self.on_gpu = self.device.type == 'cuda'
        self.overlap = overlap and self.on_gpu
        self.comm_stream = torch.cuda.Stream(device=self.device) if self.on_gpu else None
        if broadcast_parameters:
            for t in list(module.parameters()) + list(module.buffers()):
                dist.broadcast(t.data, src=0, group=self.group)
        limit = int(bucket_mb * 1024 * 1024 / 4)
        self.buckets: List[_Bucket] = []
        cur, cur_n = [], 0
        for p in reversed(params):
            if cur and cur_n + p.numel() > limit:
                self.buckets.append(_Bucket(cur, self.device, p.dtype))
                cur, cur_n = [], 0
            cur.append(p)
            cur_n += p.numel()
        if cur:
            self.buckets.append(_Bucket(cur, self.device, params[0].dtype))
        self._where = {}
        for b in self.buckets:
            for i, p in enumerate(b.params):
                self._where[p] = (b, i)
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in params]

    # called by autograd once a parameter's gradient for this step is complete
    def _on_grad(self, p: torch.nn.Parameter):
        b, i = self._where[p]
        b.got[i] = True
        b.pending -= 1
        # autograd runs a node's backward on the stream its forward ran on: parts of a model that ran on a second
        # stream deliver their gradients there
        if self.device.type == 'cuda':
            cur = torch.cuda.current_stream(self.device)
            b.streams[cur.cuda_stream] = cur
        if b.pending == 0:
            self._fill(b)
            self._launch(b)

    def _fill(self, b: _Bucket):
        """Move the finished gradients of a bucket into its flat buffer with one multi-tensor copy
        (one launch per bucket instead of one per parameter) and drop the originals: the bucket
        slices become the gradients in finish().  The copy runs on the current stream, behind everything
        queued so far on the other streams that produced gradients of this bucket."""
        if b.streams:
            cur = torch.cuda.current_stream(self.device)
            for handle, other in b.streams.items():
                if handle != cur.cuda_stream:
                    cur.wait_stream(other)
            b.streams = {}
        idx = [i for i, got in enumerate(b.got) if got and b.params[i].grad is not None
               and b.params[i].grad.data_ptr() != b.views[i].data_ptr()]
        if idx:
            torch._foreach_copy_([b.views[i] for i in idx], [b.params[i].grad for i in idx])
        for i in idx:
            b.params[i].grad = None

    def _launch(self, b: _Bucket):
        if self.overlap:
            b.event = torch.cuda.Event()
            b.event.record(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(b.event)
                b.work = dist.all_reduce(b.flat, op=self._op(), group=self.group, async_op=True)
        else:
            b.work = dist.all_reduce(b.flat, op=self._op(), group=self.group, async_op=True)

    def _op(self):
        return dist.ReduceOp.AVG if self.use_avg else dist.ReduceOp.SUM

    def finish(self):
        """Wait for every bucket, divide by the world size and expose the result as p.grad."""
        for b in self.buckets:
            if b.pending != 0:
                # some parameter received no gradient this step: its slice counts as zero, and the
                # bucket is reduced anyway (every rank must join every collective)
                self._fill(b)
                for i, got in enumerate(b.got):
                    if not got:
                        b.views[i].zero_()
                self._launch(b)
            b.work.wait()
            if self.overlap:
                torch.cuda.current_stream(self.device).wait_stream(self.comm_stream)
            if not self.use_avg:
                b.flat.div_(self.world)
            for i, p in enumerate(b.params):
                p.grad = b.views[i]
            b.pending = len(b.params)
            b.got = [False] * len(b.params)
            b.work = None

    def zero_grad(self):
        """Start a step: drop the aliases into the buckets (the hooks refill them during backward)."""
        for b in self.buckets:
            for p in b.params:
                p.grad = None
            b.pending = len(b.params)
            b.got = [False] * len(b.params)

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


# =================================================================================================
# Backward in stages, so that the gradients of one stage can travel while the next computes.
# =================================================================================================
_BOUNDARY = None          # list of (level, tensors above the cut, detached leaves below it) while a staged step records


def stage_boundary(*tensors: torch.Tensor, level: int = 1):
    """Called by a model where a staged data-parallel step may cut its backward pass: level 1 between encoder(s)
    and decoder, level 2 inside an encoder (between its convolution stack and what follows).  Outside a staged
    step (the normal case) it returns its arguments.  Inside one it cuts the autograd graph there: the tensors are
    replaced by detached leaves, backward stops at them, and StagedGrads.stage(k) later continues from the cuts of
    level k-1 with the leaves' gradients.  Values are unchanged either way."""
    if _BOUNDARY is None or not torch.is_grad_enabled():
        return tensors
    leaves = tuple(t.detach().requires_grad_(True) if t.requires_grad else t for t in tensors)
    _BOUNDARY.append((level, tensors, leaves))
    return leaves


class StagedGrads:
    """One replica's gradients of a data-parallel step in ONE flat buffer made of runs, one per backward stage:
    stage 1 is backward from the loss to the model's level-1 boundary (decoder side), stage k continues from the
    level-(k-1) cuts to the next (encoders above their convolution stacks; the convolution stacks).  Each run is
    averaged over the ranks by one all-reduce, launched on a side stream as soon as the run is filled, so it travels
    while the next stage computes; only the last, smallest run is exposed.  p.grad of every parameter ends up as a
    view of the reduced buffer.  Which parameter belongs to which run is found during the first step (it is a
    property of the model) and fixed afterwards; parameters that never receive a gradient sit in the last run as
    zeros; a model without a boundary has everything in run 0.  The phases are separate methods so that a caller can
    capture each as its own graph and issue the collectives between the replays.

    A parameter used on both sides of a cut (tied weights) would be reduced before its gradient is complete: that
    is detected during the first step and refused."""

    def __init__(self, params: List[torch.nn.Parameter], process_group=None):
        if not dist.is_initialized():
            raise RuntimeError("StagedGrads needs an initialised torch.distributed process group")
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.use_avg = dist.get_backend(process_group) == 'nccl'
        self.params = [p for p in params if p.requires_grad]
        self.device = self.params[0].device
        self.on_gpu = self.device.type == 'cuda'
        self.comm_stream = torch.cuda.Stream(device=self.device) if self.on_gpu else None
        self.flat = None
        self.runs = []                # [(lo, hi)] of the runs in the flat buffer, one per stage
        self.order = []               # parameters in flat-buffer order
        self.views = {}
        self.nstages = None           # known once the first forward pass has met the model's boundaries
        self._run_of = {}             # id(parameter) -> run
        self._cuts = {}
        self._levels = []
        self._laid_out = False
        self._seen = {}               # first step only: id(parameter) -> (gradient tensor, its version) when placed
        self._copy_tables = {}        # run -> host-side pointer arrays of its multi-tensor copy

    # ---- layout -------------------------------------------------------------------------------
    def _place(self, run: int, last: bool):
        """first step: the parameters whose gradient appeared in this stage form the next run of the buffer"""
        if self.flat is None:
            n = sum((p.numel() + 3) // 4 * 4 for p in self.params)
            self.flat = torch.zeros(n, dtype=self.params[0].dtype, device=self.device)
        for p in self.params:                        # a gradient that grew after its run was closed
            seen = self._seen.get(id(p))
            if seen is not None and (p.grad is not seen[0] or p.grad._version != seen[1]):
                raise RuntimeError("a parameter receives gradient on both sides of a stage boundary (shared weights); "
                                   "StagedGrads would average it before it is complete: use GradSync for this model")
        lo = self.runs[-1][1] if self.runs else 0
        off = lo
        for p in self.params:
            if id(p) in self._run_of or (p.grad is None and not last):
                continue
            self._run_of[id(p)] = run
            self.order.append(p)
            self.views[id(p)] = self.flat[off:off + p.numel()].view_as(p)
            off += (p.numel() + 3) // 4 * 4          # every view starts on a 16-byte boundary (the gaps stay zero)
            if p.grad is not None:
                self._seen[id(p)] = (p.grad, p.grad._version)
        self.runs.append((lo, off))
        if last:
            self._laid_out = True
            self._seen = {}

    def _fill(self, run: int):
        """gradients of one run into the flat buffer (one multi-tensor copy); missing ones count as zero"""
        ps = [p for p in self.order if self._run_of[id(p)] == run]
        have = [p for p in ps if p.grad is not None and p.grad.data_ptr() != self.views[id(p)].data_ptr()]
        for p in ps:
            if p.grad is None:
                self.views[id(p)].zero_()
        if have and self.on_gpu and all(p.grad.dtype == torch.float32 and p.grad.is_contiguous() for p in have):
            from . import ops              # one launch of the library's multi-tensor copy (pointer table in its arguments)
            ops.copy_tensors([self.views[id(p)] for p in have], [p.grad for p in have],
                             self._copy_tables.setdefault(run, ops.TensorTable()))
        elif have:
            torch._foreach_copy_([self.views[id(p)] for p in have], [p.grad for p in have])

    def _after_stage(self, run: int):
        if not self._laid_out:
            self._place(run, last=run == self.nstages - 1)
        self._fill(run)

    # ---- the phases ---------------------------------------------------------------------------
    def stage1(self, total_loss_of):
        """total_loss_of() runs forward + loss and returns (whatever the caller wants back, total loss).
        Backward from the loss down to the first boundary; run 0 filled."""
        global _BOUNDARY
        for p in self.params:
            p.grad = None
        _BOUNDARY = []
        try:
            result, total = total_loss_of()
        finally:
            cuts, _BOUNDARY = _BOUNDARY, None
        self._cuts = {}
        for level, tensors, leaves in cuts:
            self._cuts.setdefault(level, []).append((tensors, leaves))
        levels = sorted(self._cuts)
        if self.nstages is None:
            self._levels, self.nstages = levels, 1 + len(levels)
        elif levels != self._levels:
            raise RuntimeError(f"the model's stage boundaries changed between steps ({self._levels} -> {levels})")
        total.backward()
        self._after_stage(0)
        return result

    def stage(self, k: int):
        """Stage k >= 2: backward from the cuts of the (k-1)-th boundary level down to the next; run k-1 filled."""
        above, grads = [], []
        for tensors, leaves in self._cuts.pop(self._levels[k - 2], []):
            for t, leaf in zip(tensors, leaves):
                if leaf is not t and leaf.grad is not None:
                    above.append(t)
                    grads.append(leaf.grad)
        if above:
            torch.autograd.backward(above, grads)
        self._after_stage(k - 1)

    def stage2(self):
        self.stage(2)

    def reduce(self, run: int):
        """Average one run over the ranks; on a GPU the collective is ordered after what the current stream
        has been given so far and runs on the side stream."""
        lo, hi = self.runs[run]
        if hi == lo:
            return
        buf = self.flat[lo:hi]
        op = dist.ReduceOp.AVG if self.use_avg else dist.ReduceOp.SUM
        if self.on_gpu:
            self.comm_stream.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self.comm_stream):
                self._all_reduce(buf, op)
                if not self.use_avg:
                    buf.div_(self.world)
        else:
            self._all_reduce(buf, op)
            if not self.use_avg:
                buf.div_(self.world)

    def _all_reduce(self, buf, op):
        dist.all_reduce(buf, op=op, group=self.group)

    def finish(self):
        """The compute stream waits for the collectives; every p.grad becomes its slice of the buffer."""
        if self.on_gpu:
            torch.cuda.current_stream(self.device).wait_stream(self.comm_stream)
        for p in self.order:
            p.grad = self.views[id(p)]
