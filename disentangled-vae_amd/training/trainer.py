"""Trainer with the reference's surface (training/trainer.py:26-474): same constructor, epoch loop,
beta schedule, loss dispatch, clip + Adam step, LR schedule rule, history, TensorBoard tags and
checkpoint layout, driving the MI355X kernels.

What differs underneath:
  * forward, loss, backward, gradient norm/clip and Adam are HIP kernels (hipvae.*); there is no CPU
    path, so a configuration that does not resolve to a GPU is an error instead of a silent fallback;
  * the training step (zero_grad ... Adam) is captured into a hipGraph per batch shape and replayed, and so is
    the validation forward + loss; what changes between steps (the clip, Adam's step count and learning rate,
    the annealed KL weight) is read from device memory.  Where a step cannot be captured the Trainer says why,
    once, and issues the same kernels one by one (use_graph=False / DVAE_HIPGRAPH=0 selects that from the start);
  * loss values stay on the device during an epoch: one packed add per step, one read-back per
    logging interval / epoch, instead of >= 9 .item() synchronisations per step;
  * with torch.distributed initialised (one process per GPU) gradients are averaged across ranks by
    bucketed all-reduce overlapped with backward (hipvae.parallel.GradSync); rank 0 logs and saves.
"""
import copy
import json
import math
import os
import sys
import threading
import time
from typing import Dict, Optional, Tuple

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch
import torch.distributed as dist
import torch.nn as nn
import torch.optim as optim

from hipvae import ops as hip_ops
from hipvae.losses import baseline_loss, disentangled_loss, lstm_loss
from hipvae.optim import FusedAdam
from training.config import ExperimentConfig

try:
    from torch.utils.tensorboard import SummaryWriter
except Exception:                                       # tensorboard is optional: logging only
    class SummaryWriter:                                # type: ignore
        def __init__(self, *a, **k): pass
        def add_scalar(self, *a, **k): pass
        def add_image(self, *a, **k): pass
        def close(self): pass

try:
    from tqdm import tqdm
except Exception:
    def tqdm(it, **k):                                  # type: ignore
        return it


def beta_for_epoch(t, epoch: int) -> float:
    """KL weight schedule, per epoch (reference: Trainer._get_beta, training/trainer.py:115-149)."""
    if epoch < t.kl_start_epoch:
        return 0.0
    if not t.use_kl_annealing or t.warmup_epochs <= 0:
        return t.beta_vae
    since = epoch - t.kl_start_epoch
    if t.annealing_type == 'linear':
        return t.beta_vae * (since / t.warmup_epochs) if since < t.warmup_epochs else t.beta_vae
    if t.annealing_type == 'cosine':
        if since >= t.warmup_epochs:
            return t.beta_vae
        return t.beta_vae * (1 - math.cos(since / t.warmup_epochs * math.pi)) / 2
    if t.annealing_type == 'cyclical':
        pos = since % t.warmup_epochs
        return t.beta_vae * 0.5 * (1 - math.cos(pos / t.warmup_epochs * math.pi))
    return t.beta_vae


def _map_tensors(o, fn):
    """o with fn applied to every tensor inside dicts, lists and tuples; containers keep their type and a
    state_dict keeps its _metadata attribute."""
    if torch.is_tensor(o):
        return fn(o)
    if isinstance(o, dict):
        out = type(o)((k, _map_tensors(v, fn)) for k, v in o.items())
        if hasattr(o, '_metadata'):
            out._metadata = copy.deepcopy(o._metadata)
        return out
    if isinstance(o, (list, tuple)):
        return type(o)(_map_tensors(v, fn) for v in o)
    return o


class DevicePrefetcher:
    """Iterates a loader of (video, metadata) host batches one batch ahead: batch i+1 is copied to the
    device on a side stream (asynchronously when the loader pins its memory) while step i computes, so
    the PCIe transfer is off the step's critical path.  Batches already on the device pass through."""

    def __init__(self, loader, device: torch.device):
        self.loader, self.device = loader, torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device)

    def __len__(self):
        return len(self.loader)

    def _upload(self, batch):
        video, meta = batch
        if video.is_cuda:
            return video, meta, None
        with torch.cuda.stream(self.stream):
            dv = video.to(self.device, non_blocking=True)
            done = torch.cuda.Event()
            done.record(self.stream)
        return dv, meta, done

    def __iter__(self):
        pending = None
        for batch in self.loader:
            nxt = self._upload(batch)
            if pending is not None:
                yield self._hand_over(pending)
            pending = nxt
        if pending is not None:
            yield self._hand_over(pending)

    def _hand_over(self, item):
        dv, meta, done = item
        if done is not None:
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(done)
            dv.record_stream(cur)           # allocated on the copy stream, consumed on the compute stream
        return dv, meta


class Trainer:

    def __init__(self, model: nn.Module, config: ExperimentConfig, train_loader, val_loader,
                 device: Optional[torch.device] = None, use_graph: Optional[bool] = None):
        """use_graph: replay the training step and the validation forward from captured hipGraphs (with several
        ranks: three graphs with the gradient all-reduces between them); default True, the environment variable
        DVAE_HIPGRAPH=0 turns it off.  A step that cannot be captured falls back to eager launches and the
        reason is printed (see graph_fallback_reason)."""
        self.config = config
        self.train_loader = train_loader
        self.val_loader = val_loader
        self.distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        self.rank = dist.get_rank() if self.distributed else 0
        if device is None:
            if config.training.device == 'cuda' and torch.cuda.is_available():
                device = torch.device('cuda', torch.cuda.current_device())
            else:
                raise RuntimeError(
                    f"training.device={config.training.device!r} does not resolve to a GPU here "
                    f"(torch.cuda.is_available()={torch.cuda.is_available()}). This build runs on MI355X "
                    "only and has no CPU fallback: launch with --device cuda on a ROCm machine.")
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError("this build has no CPU path; pass a cuda device")
        if self.rank == 0:
            print(f"Training on device: {self.device}")
        self.model = model.to(self.device)
        self.model_type = getattr(model, 'model_type', 'unknown')

        t = config.training
        self.optimizer = FusedAdam(self.model.parameters(), lr=t.learning_rate, betas=(t.beta1, t.beta2),
                                   weight_decay=t.weight_decay)
        self.scheduler = self._setup_scheduler()
        self.loss_fn = self._get_loss_fn()
        self.grad_sync = None
        if self.distributed:
            from hipvae.parallel import GradSync
            self.grad_sync = GradSync(self.model)
        if use_graph is None:
            use_graph = os.environ.get('DVAE_HIPGRAPH', '1') != '0'
        self.use_graph = bool(use_graph)
        self._graphs = {}                 # batch shape -> GraphedTrainStep
        self._val_graphs = {}             # batch shape -> GraphedForward of the validation forward + loss
        self.graph_fallback_reason = None  # why a step ran eagerly although use_graph is set (last reason)
        self._fallback_said = set()
        # the KL weight of the current epoch as a device scalar: the captured loss kernels read it, so one
        # capture per batch shape serves every value of the annealing schedule
        self._beta_dev = torch.zeros(1, dtype=torch.float32, device=self.device)
        self._beta_host = 0.0
        self.async_checkpoint = os.environ.get('DVAE_ASYNC_CKPT', '1') != '0'
        self._ckpt_thread, self._ckpt_stream, self._ckpt_keep, self._ckpt_error = None, None, None, None

        self.checkpoint_dir = os.path.join(t.checkpoint_dir, config.name)
        self.results_dir = os.path.join(t.results_dir, config.name)
        if self.rank == 0:
            for d in (self.checkpoint_dir, self.results_dir, os.path.join(self.results_dir, 'figures'),
                      os.path.join(self.results_dir, 'videos')):
                os.makedirs(d, exist_ok=True)
            config.save(os.path.join(self.checkpoint_dir, 'config.json'))
            self.writer = SummaryWriter(os.path.join(self.results_dir, 'logs'))
        else:
            self.writer = None

        self.current_epoch = 0
        self.global_step = 0
        self.best_val_loss = float('inf')
        self.history = {k: [] for k in ('train_loss', 'val_loss', 'train_recon', 'val_recon', 'train_kl',
                                        'val_kl', 'learning_rate')}

    # ------------------------------------------------------------------ set-up helpers
    def _setup_scheduler(self):
        t = self.config.training
        if t.lr_scheduler == 'exponential':
            return optim.lr_scheduler.ExponentialLR(self.optimizer, gamma=t.lr_decay)
        if t.lr_scheduler == 'cosine':
            return optim.lr_scheduler.CosineAnnealingLR(self.optimizer, T_max=t.num_epochs, eta_min=t.lr_min)
        return None

    def _get_loss_fn(self):
        return {'baseline_vae': baseline_loss, 'lstm_vae': lstm_loss,
                'disentangled_vae': disentangled_loss}.get(self.model_type, baseline_loss)

    def _get_beta(self, epoch: int) -> float:
        return beta_for_epoch(self.config.training, epoch)

    def _compute_loss(self, video: torch.Tensor, beta: float, beta_device: Optional[torch.Tensor] = None
                      ) -> Tuple[Dict[str, torch.Tensor], torch.Tensor]:
        """beta_device: the device scalar holding beta, for a step that is being captured (the kernels then read
        beta when they run instead of taking it as a launch constant)."""
        t = self.config.training
        dyn = {} if beta_device is None else {'beta_device': beta_device}
        if self.model_type == 'disentangled_vae':
            output = self.model(video)
            losses = self.loss_fn(output, video, beta=beta, beta_content=t.beta_content, beta_motion=t.beta_motion,
                                  recon_loss_type=t.recon_loss_type, temporal_weight=t.temporal_weight,
                                  independence_weight=t.independence_weight, **dyn)
            return losses, output['recon']
        recon, z, mu, logvar = self.model(video)
        # temporal_weight is not forwarded on this path (the reference does not either, :171-177)
        losses = self.loss_fn(recon, video, mu, logvar, beta=beta, recon_loss_type=t.recon_loss_type,
                              free_bits=t.free_bits, use_free_bits=t.use_free_bits, **dyn)
        return losses, recon

    def _set_device_beta(self, beta: float):
        if float(beta) != self._beta_host:
            self._beta_dev.fill_(float(beta))
            self._beta_host = float(beta)

    def _capturable(self) -> Optional[str]:
        """None when a step may be captured now, else the reason why not."""
        if self.loss_fn not in (baseline_loss, lstm_loss, disentangled_loss):
            return "a user-supplied loss function may hold beta as a launch constant"
        return None

    def _fall_back(self, what: str, why: str):
        self.graph_fallback_reason = why
        if (what, why) not in self._fallback_said:
            self._fallback_said.add((what, why))
            if self.rank == 0:
                print(f"[trainer] {what} runs as eager launches, not as a hipGraph replay: {why}", flush=True)

    # ------------------------------------------------------------------ one optimisation step
    def train_step(self, video: torch.Tensor, beta: float):
        """zero_grad, forward, loss, backward, (all-reduce), clip, Adam (training/trainer.py:192-204)."""
        if self.use_graph:
            losses = self._graphed_step(video, beta)
            if losses is not None:
                return losses
        if self.grad_sync is None and self.distributed:
            from hipvae.parallel import GradSync            # the captured step had taken the hooks away
            self.grad_sync = GradSync(self.model)
        if self.grad_sync is not None:
            self.grad_sync.zero_grad()
        else:
            self.optimizer.zero_grad(set_to_none=True)
        losses, _ = self._compute_loss(video, beta)
        losses['total_loss'].backward()
        if self.grad_sync is not None:
            self.grad_sync.finish()
        if self.config.training.gradient_clip > 0:
            # the clip coefficient is folded into Adam's gradient read: one pass less over the gradients
            self.optimizer.clip_grad_norm_(self.config.training.gradient_clip, fold_into_step=True)
        self.optimizer.step()
        return losses

    def _graphed_step(self, video: torch.Tensor, beta: float):
        """The same step replayed from a hipGraph captured for this batch shape (the full and the ragged last
        batch get one each).  Beta is read from device memory by the captured kernels.  Returns None when the
        step cannot be captured (the caller then runs it eagerly)."""
        from hipvae.graph import GraphedDataParallelStep, GraphedTrainStep
        why = self._capturable()
        if why is not None:
            self._fall_back('the training step', why)
            return None
        self._set_device_beta(beta)
        from hipvae import rng
        key = tuple(video.shape) + (rng.injecting(),)          # a run with injected draws captures its own form
        g = self._graphs.get(key)
        if g is False:                                         # capture failed for this shape before
            return None
        if g is None:
            while len(self._graphs) >= 4:                      # oldest first; two shapes per epoch in practice
                self._graphs.pop(next(iter(self._graphs)))
            loss_of = lambda m, v: self._compute_loss(v, beta, self._beta_dev)[0]
            clip = self.config.training.gradient_clip
            try:
                if self.distributed:
                    # several ranks: graphs around the all-reduces of the flat gradient buffer.  The hooks of
                    # the eager path must not fire under capture; the parameters were broadcast when the
                    # Trainer was built and every rank has taken the same steps since.
                    if self.grad_sync is not None:
                        self.grad_sync.remove()
                        self.grad_sync = None
                    g = GraphedDataParallelStep(self.model, self.optimizer, loss_of, video, gradient_clip=clip,
                                                broadcast_parameters=False)
                else:
                    g = GraphedTrainStep(self.model, self.optimizer, loss_of, video, gradient_clip=clip)
            except Exception as e:                              # noqa: BLE001
                if self.distributed:
                    raise           # ranks that disagree about the path would hang in each other's collectives
                self._graphs[key] = False
                self._fall_back('the training step', f'capture failed ({type(e).__name__}: {e})')
                return None
            self._graphs[key] = g
        losses = g(video)
        losses['beta'] = torch.tensor(beta)                     # the static dictionary reports this step's beta
        return losses

    @staticmethod
    def _unpack(losses, packed_cpu) -> Dict[str, float]:
        out = {k: float(packed_cpu[i]) for k, i in losses.layout.items()}
        out['beta'] = float(losses['beta'])
        return out

    def train_epoch(self) -> Dict[str, float]:
        self.model.train()
        beta = self._get_beta(self.current_epoch)
        t = self.config.training
        # a distributed sampler derives its shuffle from (seed, epoch): without this call every epoch
        # would walk the clips in the same order
        sampler = getattr(self.train_loader, 'sampler', None)
        if hasattr(sampler, 'set_epoch'):
            sampler.set_epoch(self.current_epoch)
        acc = None
        last = None
        n = 0
        batches = DevicePrefetcher(self.train_loader, self.device)
        bar = tqdm(batches, desc=f'Epoch {self.current_epoch}', dynamic_ncols=True, leave=True,
                   mininterval=0.5, disable=self.rank != 0) if self.rank == 0 else batches
        for video, _meta in bar:
            last = self.train_step(video, beta)
            packed = last.packed.detach()
            acc = packed.clone() if acc is None else acc.add_(packed)
            n += 1
            if self.global_step % t.log_every == 0 and (self.rank == 0 or self.distributed):
                # the only per-step synchronisation; with several ranks the logged values are the mean
                # over ranks (every rank takes part in the collective, rank 0 writes)
                cur = self._unpack(last, self._rank_mean(packed.clone()).cpu())
                hip_ops.check_lstm_seq_timeouts()      # the device is idle here anyway
            if self.rank == 0 and self.global_step % t.log_every == 0:
                for k, v in cur.items():
                    self.writer.add_scalar(f'train/{k}', v, self.global_step)
                self.writer.add_scalar('train/beta', beta, self.global_step)
                self.writer.add_scalar('train/lr', self.optimizer.param_groups[0]['lr'], self.global_step)
                if hasattr(bar, 'set_postfix'):
                    post = {'loss': f"{cur['total_loss']:.4f}", 'recon': f"{cur['recon_loss']:.4f}",
                            'kl': f"{cur['kl_loss']:.4f}", 'β': f'{beta:.3f}'}
                    if 'recon_loss_per_pixel' in cur:
                        post['pix'] = f"{cur['recon_loss_per_pixel']:.6f}"
                    if 'kl_loss_raw' in cur:
                        post['kl_raw'] = f"{cur['kl_loss_raw']:.4f}"
                    bar.set_postfix(post, refresh=False)
            self.global_step += 1
        if n == 0:
            return {}
        hip_ops.check_lstm_seq_timeouts()      # (the read-back below synchronises anyway)
        out = self._unpack(last, self._rank_mean(acc / n).cpu())
        out['beta'] = beta
        return out

    def _rank_mean(self, packed: torch.Tensor) -> torch.Tensor:
        """Mean of a packed loss vector over the ranks (for logging and the history only; the identity on
        one rank).  Every rank must call it at the same points."""
        if not self.distributed:
            return packed
        dist.all_reduce(packed, op=dist.ReduceOp.SUM)
        return packed / dist.get_world_size()

    @torch.no_grad()
    def validate(self) -> Dict[str, float]:
        self.model.eval()
        beta = self.config.training.beta_vae          # the final beta, not the annealed one (reference :243)
        acc, last, n = None, None, 0
        for video, _meta in DevicePrefetcher(self.val_loader, self.device):
            last = self._validation_losses(video, beta)
            acc = last.packed.clone() if acc is None else acc.add_(last.packed)
            n += 1
        if n == 0:
            return {}
        hip_ops.check_lstm_seq_timeouts()      # (the read-back below synchronises anyway)
        out = self._unpack(last, self._rank_mean(acc / n).cpu())
        out['beta'] = beta
        return out

    def _validation_losses(self, video: torch.Tensor, beta: float):
        """Forward + loss of one validation batch (eval mode, no gradients): replayed from a hipGraph per batch
        shape, or launched eagerly when capture is off or not possible."""
        why = self._capturable() if self.use_graph else 'off'
        if why is None:
            key = tuple(video.shape)
            g = self._val_graphs.get(key)
            if g is None:
                from hipvae.graph import GraphedForward
                self._set_device_beta(beta)
                try:
                    g = GraphedForward(self.model, video, lambda m, v: self._compute_loss(v, beta, self._beta_dev)[0])
                except Exception as e:                          # noqa: BLE001
                    g = False
                    self._fall_back('validation', f'capture failed ({type(e).__name__}: {e})')
                while len(self._val_graphs) >= 4:
                    self._val_graphs.pop(next(iter(self._val_graphs)))
                self._val_graphs[key] = g
            if g is not False:
                self._set_device_beta(beta)
                return g(video)
        elif self.use_graph:
            self._fall_back('validation', why)
        return self._compute_loss(video, beta)[0]

    # ------------------------------------------------------------------ checkpoints
    def save_checkpoint(self, filename: str = 'checkpoint.pt'):
        """The reference's checkpoint dictionary (training/trainer.py:261-276), written off the training
        path: device tensors are cloned on the compute stream (a fraction of a millisecond), the clones
        travel to pinned host memory on a side stream, and a writer thread serialises the file once they
        have arrived, while training goes on.  One write is in flight at a time; load_checkpoint() and the
        end of train() wait for it.  async_checkpoint=False writes before returning."""
        if self.rank != 0:
            return
        self.wait_for_checkpoint()
        ckpt = {
            'epoch': self.current_epoch,
            'global_step': self.global_step,
            'model_state_dict': self.model.state_dict(),
            'optimizer_state_dict': self.optimizer.state_dict(),
            'best_val_loss': self.best_val_loss,
            'history': copy.deepcopy(self.history),
            'config': self.config.to_dict(),
        }
        if self.scheduler is not None:
            ckpt['scheduler_state_dict'] = copy.deepcopy(self.scheduler.state_dict())
        path = os.path.join(self.checkpoint_dir, filename)
        if not self.async_checkpoint:
            torch.save(ckpt, path)
            print(f"Saved checkpoint to {path}")
            return
        ckpt, arrived = self._to_host_async(ckpt)

        def write():
            try:
                arrived.synchronize()
                tmp = path + '.tmp'
                torch.save(ckpt, tmp)
                os.replace(tmp, path)
                print(f"Saved checkpoint to {path}")
            except BaseException as e:                      # surfaced by wait_for_checkpoint()
                self._ckpt_error = e
        self._ckpt_thread = threading.Thread(target=write, name='checkpoint-writer')
        self._ckpt_thread.start()

    def _to_host_async(self, obj):
        """obj with every GPU tensor replaced by a pinned host copy that is still arriving, and the event
        that marks the arrival of all of them."""
        cur = torch.cuda.current_stream(self.device)
        if self._ckpt_stream is None:
            self._ckpt_stream = torch.cuda.Stream(device=self.device)
        side = self._ckpt_stream
        clones = []

        def snap(t):                                         # compute stream: a consistent copy, now
            c = t.detach().clone()
            if c.is_cuda:
                clones.append(c)
            return c
        snapped = _map_tensors(obj, snap)
        side.wait_stream(cur)
        hosts = {}
        with torch.cuda.stream(side):
            for c in clones:
                h = torch.empty(c.shape, dtype=c.dtype, pin_memory=True)
                h.copy_(c, non_blocking=True)
                hosts[id(c)] = h
        arrived = torch.cuda.Event()
        arrived.record(side)

        out = _map_tensors(snapped, lambda t: hosts.get(id(t), t))
        self._ckpt_keep = clones                             # alive until the copies have been waited for
        return out, arrived

    def wait_for_checkpoint(self):
        """Block until the checkpoint being written (if any) is on disk; re-raises a failed write."""
        th = self._ckpt_thread
        if th is not None:
            th.join()
            self._ckpt_thread = None
            self._ckpt_keep = None
        if self._ckpt_error is not None:
            e, self._ckpt_error = self._ckpt_error, None
            raise RuntimeError("writing the previous checkpoint failed") from e

    def load_checkpoint(self, filename: str = 'checkpoint.pt'):
        self.wait_for_checkpoint()
        path = os.path.join(self.checkpoint_dir, filename)
        if not os.path.exists(path):
            print(f"No checkpoint found at {path}")
            return False
        ckpt = torch.load(path, map_location=self.device, weights_only=False)
        self.model.load_state_dict(ckpt['model_state_dict'])
        self.optimizer.load_state_dict(ckpt['optimizer_state_dict'])
        # graphs captured before the load are dropped: they are re-captured against the loaded state on
        # the next step (parameters and Adam moments keep their storage, but a capture also bakes in
        # anything the optimiser created lazily at its first step)
        self._graphs.clear()
        self._val_graphs.clear()
        self.current_epoch = ckpt['epoch']            # resumes AT the saved epoch, as the reference does
        self.global_step = ckpt['global_step']
        self.best_val_loss = ckpt['best_val_loss']
        self.history = ckpt['history']
        if self.scheduler is not None and 'scheduler_state_dict' in ckpt:
            self.scheduler.load_state_dict(ckpt['scheduler_state_dict'])
        print(f"Loaded checkpoint from {path} (epoch {self.current_epoch})")
        return True

    # ------------------------------------------------------------------ TensorBoard images
    @torch.no_grad()
    def visualize_reconstructions(self, num_samples: int = 4):
        if self.rank != 0:
            return
        self.model.eval()
        video, _ = next(iter(self.val_loader))
        video = video[:num_samples].to(self.device)
        out = self.model(video)
        recon = out['recon'] if self.model_type == 'disentangled_vae' else out[0]
        for i in range(min(num_samples, video.size(0))):
            for tag, src in (('original', video), ('reconstructed', recon)):
                self.writer.add_image(f'recon/{tag}_{i}_first', src[i, 0], self.current_epoch)
                self.writer.add_image(f'recon/{tag}_{i}_last', src[i, -1], self.current_epoch)

    @torch.no_grad()
    def visualize_samples(self, num_samples: int = 4):
        if self.rank != 0:
            return
        self.model.eval()
        samples = self.model.sample(num_samples=num_samples, device=self.device)
        for i in range(samples.size(0)):
            self.writer.add_image(f'samples/sample_{i}_first', samples[i, 0], self.current_epoch)
            self.writer.add_image(f'samples/sample_{i}_last', samples[i, -1], self.current_epoch)

    # ------------------------------------------------------------------ the epoch driver
    def train(self, resume: bool = False):
        if resume:
            self.load_checkpoint()                    # looks for 'checkpoint.pt', like the reference
        t = self.config.training
        start_epoch = self.current_epoch
        if self.rank == 0:
            print(f"\nStarting training from epoch {start_epoch}")
            print(f"Model type: {self.model_type}")
            print(f"Device: {self.device}")
            print(f"Number of parameters: {sum(p.numel() for p in self.model.parameters()):,}\n")
        t0 = time.time()
        for epoch in range(start_epoch, t.num_epochs):
            self.current_epoch = epoch
            tr = self.train_epoch()
            va = self.validate()
            if self.scheduler is not None and self.optimizer.param_groups[0]['lr'] > t.lr_min:
                self.scheduler.step()                 # stops once lr <= lr_min (reference :373-375)
            lr = self.optimizer.param_groups[0]['lr']
            for key, src, name in (('train_loss', tr, 'total_loss'), ('val_loss', va, 'total_loss'),
                                   ('train_recon', tr, 'recon_loss'), ('val_recon', va, 'recon_loss'),
                                   ('train_kl', tr, 'kl_loss'), ('val_kl', va, 'kl_loss')):
                self.history[key].append(src[name])
            self.history['learning_rate'].append(lr)
            if self.rank == 0:
                self.writer.add_scalar('epoch/train_loss', tr['total_loss'], epoch)
                self.writer.add_scalar('epoch/val_loss', va['total_loss'], epoch)
                self.writer.add_scalar('epoch/learning_rate', lr, epoch)
                print(f"\nEpoch {epoch}:")
                for label, d in (('Train', tr), ('Val  ', va)):
                    pix = d.get('recon_loss_per_pixel', d['recon_loss'])
                    print(f"  {label} - Total: {d['total_loss']:.2f}, Recon: {d['recon_loss']:.2f} "
                          f"(pix: {pix:.6f}), KL: {d['kl_loss']:.2f}")
                print(f"  LR: {lr:.6f}, β: {self._get_beta(epoch):.3f}")
            if va['total_loss'] < self.best_val_loss:
                self.best_val_loss = va['total_loss']
                self.save_checkpoint('best_model.pt')
                if self.rank == 0:
                    print(f"  New best model! Val loss: {self.best_val_loss:.4f}")
            if (epoch + 1) % t.save_every == 0:
                self.save_checkpoint(f'checkpoint_epoch_{epoch}.pt')
            if (epoch + 1) % t.visualize_every == 0:
                self.visualize_reconstructions()
                self.visualize_samples()
        self.save_checkpoint('final_model.pt')
        self.wait_for_checkpoint()
        if self.rank == 0:
            with open(os.path.join(self.results_dir, 'training_history.json'), 'w') as f:
                json.dump(self.history, f, indent=2)
            print(f"\nTraining completed in {(time.time() - t0) / 60:.1f} minutes")
            print(f"Best validation loss: {self.best_val_loss:.4f}")
            self.writer.close()
        return self.history


def model_kwargs_from_config(config: ExperimentConfig) -> Dict:
    """Config -> constructor arguments (reference: create_trainer, training/trainer.py:435-470;
    note content_channels <- encoder_channels and both feature dims <- feature_dim)."""
    m = config.model
    kw = {'in_channels': m.in_channels, 'sequence_length': m.sequence_length}
    if m.model_type == 'baseline':
        kw.update(latent_dim=m.latent_dim, encoder_channels=m.encoder_channels,
                  decoder_channels=m.decoder_channels, feature_dim=m.feature_dim)
    elif m.model_type == 'lstm':
        kw.update(latent_dim=m.latent_dim, encoder_channels=m.encoder_channels,
                  decoder_channels=m.decoder_channels, feature_dim=m.feature_dim, lstm_hidden=m.lstm_hidden,
                  lstm_layers=m.lstm_layers, bidirectional=m.bidirectional)
    elif m.model_type == 'disentangled':
        kw.update(content_dim=m.content_dim, motion_dim=m.motion_dim, content_channels=m.encoder_channels,
                  motion_channels=m.motion_channels, decoder_channels=m.decoder_channels,
                  content_feature_dim=m.feature_dim, motion_feature_dim=m.feature_dim, lstm_hidden=m.lstm_hidden,
                  lstm_layers=m.lstm_layers, bidirectional=m.bidirectional,
                  content_aggregation=m.content_aggregation)
    return kw


def create_trainer(config: ExperimentConfig, train_loader, val_loader) -> Trainer:
    from models import get_model
    model = get_model(config.model.model_type, **model_kwargs_from_config(config))
    return Trainer(model, config, train_loader, val_loader)
