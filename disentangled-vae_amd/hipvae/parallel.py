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
        if b.pending == 0:
            self._fill(b)
            self._launch(b)

    @staticmethod
    def _fill(b: _Bucket):
        """Move the finished gradients of a bucket into its flat buffer with one multi-tensor copy
        (one launch per bucket instead of one per parameter) and drop the originals: the bucket
        slices become the gradients in finish()."""
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
