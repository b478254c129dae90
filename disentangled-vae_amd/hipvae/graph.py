"""Whole-step hipGraph capture: zero_grad + forward + loss + backward + clip + Adam recorded once and
replayed per batch, which removes the per-launch host cost (several hundred kernel launches per step;
the batch-32 configurations are launch-bound without it).

Captured through torch.cuda.graph (HIP graphs on ROCm): every kernel of libdvae_hip.so is launched on
torch's current stream, so it is recorded like any torch op.  What changes from step to step lives in
device memory: the input clip (a static buffer the batch is copied into), the Adam step count and
learning rate (FusedAdam.enable_device_state), the KL weight when the loss is given a device scalar for it
(losses.*_loss(beta_device=...)) and the random draws (torch's graph-safe generator; or, when a run injects
prepared draws through hipvae.rng.inject_eps, static buffers that receive the next draws before each replay).
Values baked into the graph: shapes and the other loss weights.
"""
import contextlib
import gc
import os
from typing import Callable, Dict, Optional

import torch
import torch.distributed as dist

from . import rng


_CAPTURE_STREAMS = {}


@contextlib.contextmanager
def quiet_collector():
    """Around a capture: Python's cycle collector runs now and not again until the capture has ended.  A graph that
    an earlier step left behind in a reference cycle (closures over the step object are enough) is destroyed
    whenever the collector next runs, and destroying a graph is not allowed while a stream of the thread captures:
    the runtime refuses, the destructor throws, the process aborts.  torch.cuda.graph collects before capturing
    only when torch.compiler.config.force_cudagraph_gc is set."""
    gc.collect()
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was_enabled:
            gc.enable()


_KEPT = [] if os.environ.get('DVAE_KEEP_GRAPHS') == '1' else None      # see keep_graphs_alive


def keep_graphs_alive(on: bool = True):
    """From now on every graph captured here stays alive (with its memory pool) until the process ends or this is
    switched off.  For a process that captures many steps one after the other (a benchmark over several
    configurations): the runtime gives every instantiated graph a few streams of its own and takes them away when
    the graph is destroyed; hardware queues are handed to streams by use count, and it is the taking away that
    lets the counts drift apart until two streams of a new graph share the launch stream's queue, which
    hipGraphLaunch of ROCm 7.0 does not survive (it walks off the end of the graph's stream list)."""
    global _KEPT
    _KEPT = [] if on else None


def _keep(graph):
    if _KEPT is not None:
        _KEPT.append(graph)
    return graph


def capture_stream(device=None) -> torch.cuda.Stream:
    """The one stream per device on which every step is warmed up and captured.  Autograd keeps the
    gradient-accumulation node it made for a parameter, together with the stream it first ran on, for as long
    as anything refers to the graph; warming up or capturing a later step (another batch shape) on a different
    stream would put a cross-stream wait in front of every gradient (torch warns about exactly that)."""
    idx = torch.cuda.current_device() if device is None else torch.device(device).index
    s = _CAPTURE_STREAMS.get(idx)
    if s is None:
        s = _CAPTURE_STREAMS[idx] = torch.cuda.Stream(device=idx)
    return s


class GraphedTrainStep:

    def __init__(self, model: torch.nn.Module, optimizer, loss_of: Callable, example_video: torch.Tensor,
                 gradient_clip: float = 0.0, warmup: int = 2):
        """loss_of(model, video) -> LossDict (must contain 'total_loss').  The warm-up steps needed
        before capture run on the example batch and are undone afterwards: parameters, buffers and
        optimiser state are restored in place, so capture does not count as training and may be
        repeated mid-training (e.g. when beta moves to its next annealing value)."""
        self.model, self.opt, self.loss_of, self.clip = model, optimizer, loss_of, float(gradient_clip)
        self.static_video = example_video.detach().clone()
        tensors = list(model.parameters()) + list(model.buffers())
        snap = [t.detach().clone() for t in tensors]
        # optimiser state as it is now (empty before the first step; mid-training when re-capturing)
        optimizer._flush_replayed()
        opt_snap = {p: {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in st.items()}
                    for p, st in optimizer.state.items()}
        dyn = optimizer.enable_device_state()
        dyn_snap = dyn.detach().clone()
        # a run with injected normal draws: the captured step reads them from buffers of its own
        inject = rng.injecting()
        side = capture_stream(example_video.device)
        side.wait_stream(torch.cuda.current_stream())
        rehearsed = None
        # 'warmup' -> 'capture' -> 'ready': where an exception of this constructor came from (a data-parallel caller
        # may only fall back to another path by agreement when every rank got through the warm-up collectives)
        self.phase = 'warmup'
        try:
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    with (rng.draw_slots() if inject else contextlib.nullcontext()) as rehearsed:
                        self._step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self.phase = 'capture'
            self._capture_stream = side      # capture goes on the stream the warm-up ran on (see capture_stream)
            # the draw buffers exist before capture begins (see rng.draw_slots), one per draw of the rehearsed step
            self.eps_slots = [torch.zeros_like(b) for b in rehearsed] if inject else None
            with (rng.draw_slots(self.eps_slots) if inject else contextlib.nullcontext()):
                self._capture()
            self.phase = 'ready'
        except BaseException as e:
            e.dvae_phase = self.phase
            raise
        finally:
            # undo the warm-up steps, also when capture failed: whoever takes another path starts from the same state
            torch.cuda.synchronize()
            with torch.no_grad():
                for t, s in zip(tensors, snap):
                    t.copy_(s)
                for p, st in optimizer.state.items():
                    old = opt_snap.get(p, {})
                    for k in ('exp_avg', 'exp_avg_sq', 'step'):
                        if k in st:
                            if k in old:
                                st[k].copy_(old[k])
                            else:
                                st[k].zero_()
                optimizer._replayed = 0
                dyn.copy_(dyn_snap)
            torch.cuda.synchronize()

    def _capture(self):
        # thread-local capture mode: other threads of the process (a DataLoader's pinning thread, a checkpoint
        # writer, a process group's watchdog) may call the runtime while this thread captures
        self.graph = torch.cuda.CUDAGraph()
        with quiet_collector(), torch.cuda.graph(self.graph, stream=self._capture_stream,
                                                 capture_error_mode='thread_local'):
            self.losses = self._step()
        _keep(self.graph)

    def _before_replay(self, video: torch.Tensor):
        self.static_video.copy_(video, non_blocking=True)
        if self.eps_slots is not None:
            rng.fill_slots(self.eps_slots)
        elif rng.injecting():
            raise RuntimeError("this step was captured without injected draws; capture it inside inject_eps()")
        self.opt.sync_device_lr()

    def _step(self):
        self.opt.zero_grad(set_to_none=True)
        losses = self.loss_of(self.model, self.static_video)
        losses['total_loss'].backward()
        if self.clip > 0:
            self.opt.clip_grad_norm_(self.clip, fold_into_step=True)
        self.opt.step()
        return losses

    def __call__(self, video: torch.Tensor):
        """One training step on `video` (same shape as the example).  Returns the static LossDict whose
        device values are overwritten by every replay."""
        self._before_replay(video)
        self.graph.replay()
        self.opt.note_replayed_step()
        return self.losses


class GraphedForward:
    """A forward pass without gradients (reconstruction, sampling, metrics: the evaluation and serving loops of
    the reference, evaluation/metrics.py:60-122) captured once and replayed per batch.  The batch-32 forward is
    ≈ 110 launches of 2-20 us each: issued one by one it is bound by the host, replayed it is not.  The model's
    mode (train / eval) at construction is what the graph holds; outputs are static tensors that every replay
    overwrites (clone what must outlive the next call)."""

    def __init__(self, model: torch.nn.Module, example: torch.Tensor, fn: Optional[Callable] = None, warmup: int = 2):
        self.model = model
        self.fn = fn if fn is not None else (lambda m, v: m(v))
        self.static_in = example.detach().clone()
        side = capture_stream(example.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):
                self.fn(model, self.static_in)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with quiet_collector(), torch.cuda.graph(self.graph, stream=side, capture_error_mode='thread_local'), \
                torch.no_grad():
            self.out = self.fn(model, self.static_in)
        _keep(self.graph)

    def __call__(self, batch: torch.Tensor):
        self.static_in.copy_(batch, non_blocking=True)
        self.graph.replay()
        return self.out


def broadcast_flat(tensors, group=None, src: int = 0):
    """Rank `src`'s values of all tensors to every rank, one collective per dtype (tensors are packed into
    a flat buffer and unpacked in place)."""
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault(t.dtype, []).append(t.data)
    for ts in by_dtype.values():
        flat = torch.cat([t.reshape(-1) for t in ts])
        dist.broadcast(flat, src=src, group=group)
        off = 0
        for t in ts:
            t.copy_(flat[off:off + t.numel()].view_as(t))
            off += t.numel()


class GraphedDataParallelStep(GraphedTrainStep):
    """The captured step for one replica of a data-parallel job (one process per GPU).

    One graph per backward stage with the gradient all-reduces between them (hipvae.parallel.StagedGrads):
      graph 1   zero_grad + forward + loss + backward of the loss and the decoder, down to the model's first
                stage boundary, and one multi-tensor copy of those gradients into run 0 of the flat buffer;
      all-reduce of run 0 on the communication stream: it travels while
      graph 2   (backward of the encoders above their convolution stacks: heads, recurrences, the large Linear
                layers; gradients copied into run 1) runs on the compute stream;
      all-reduce of run 1, which travels while
      graph 3   (backward of the convolution stacks, run 2) runs;
      all-reduce of run 2 (the convolution filters: a few hundred KB, the only exposed one);
      last      clip + Adam reading the averaged gradients in place (p.grad are views of the flat buffer).
    A model with fewer boundaries has fewer graphs.  No collective is captured.  Compared with the eager path
    (GradSync: bucketed all-reduces launched from autograd hooks) the host issues a handful of replays and
    collectives per step instead of several hundred launches, which matters as soon as several ranks share a
    node's cores.  Semantics are GradSync's: parameters and buffers broadcast from rank 0, gradients averaged before
    clipping, BatchNorm statistics per rank."""

    def __init__(self, model, optimizer, loss_of, example_video, gradient_clip: float = 0.0, warmup: int = 2,
                 process_group=None, broadcast_parameters: bool = True):
        if not dist.is_initialized():
            raise RuntimeError("GraphedDataParallelStep needs an initialised torch.distributed process group")
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        if broadcast_parameters:
            broadcast_flat(list(model.parameters()) + list(model.buffers()), process_group)
        self.staged = self._make_staged([p for p in model.parameters() if p.requires_grad], process_group)
        self.params = self.staged.params
        super().__init__(model, optimizer, loss_of, example_video, gradient_clip, max(warmup, 1))

    def _make_staged(self, params, process_group):
        from .parallel import StagedGrads
        return StagedGrads(params, process_group)

    @property
    def flat(self):
        return self.staged.flat

    @property
    def views(self):
        return [self.staged.views[id(p)] for p in self.params]

    def _stage1(self):
        def run():
            losses = self.loss_of(self.model, self.static_video)
            return losses, losses['total_loss']
        return self.staged.stage1(run)

    def _update_part(self):
        self.staged.finish()
        if self.clip > 0:
            self.opt.clip_grad_norm_(self.clip, fold_into_step=True)
        self.opt.step()

    def _step(self):              # the eager warm-up steps: the same sequence, collectives included
        losses = self._stage1()
        self.staged.reduce(0)
        for k in range(2, self.staged.nstages + 1):
            self.staged.stage(k)
            self.staged.reduce(k - 1)
        self._update_part()
        return losses

    def _capture(self):
        # thread-local capture mode: the process group's watchdog thread polls events of finished
        # collectives while this thread captures, which the default (global) mode may refuse.  The
        # graphs share one memory pool and are always replayed in this order.
        torch.cuda.current_stream().wait_stream(self.staged.comm_stream)
        cs = self._capture_stream
        self.graph = torch.cuda.CUDAGraph()
        with quiet_collector(), torch.cuda.graph(self.graph, stream=cs, capture_error_mode='thread_local'):
            self.losses = self._stage1()
        self.stage_graphs = []                     # stages 2.. in order
        for k in range(2, self.staged.nstages + 1):
            g = torch.cuda.CUDAGraph()
            with quiet_collector(), torch.cuda.graph(g, pool=self.graph.pool(), stream=cs,
                                                     capture_error_mode='thread_local'):
                self.staged.stage(k)
            self.stage_graphs.append(g)
        self.graph_update = torch.cuda.CUDAGraph()
        with quiet_collector(), torch.cuda.graph(self.graph_update, pool=self.graph.pool(), stream=cs,
                                                 capture_error_mode='thread_local'):
            # (finish() makes the capturing stream wait for the communication stream: outside capture)
            for p in self.staged.order:
                p.grad = self.staged.views[id(p)]
            if self.clip > 0:
                self.opt.clip_grad_norm_(self.clip, fold_into_step=True)
            self.opt.step()
        for g in [self.graph, self.graph_update] + self.stage_graphs:
            _keep(g)

    def __call__(self, video: torch.Tensor):
        self._before_replay(video)
        self.graph.replay()
        self.staged.reduce(0)
        for k, g in enumerate(self.stage_graphs, start=1):
            g.replay()
            self.staged.reduce(k)
        torch.cuda.current_stream().wait_stream(self.staged.comm_stream)
        self.graph_update.replay()
        self.opt.note_replayed_step()
        return self.losses
