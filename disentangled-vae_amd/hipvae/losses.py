"""Fused loss nodes: reconstruction (MSE/BCE) + temporal stencil + KL (+ free bits) + independence.

One autograd node per loss call.  Forward is two launches (the pass over the clips, then one workgroup that finishes every
term; the independence term adds its own) and leaves every term as a device scalar; backward launches the gradient kernels, each reading the upstream gradient of
``total_loss`` from device memory, so nothing synchronises with the host.  Only ``total_loss``
carries gradient (the other entries are reporting values, as the reference's Trainer uses them).
"""
from typing import Dict, Optional

import torch

from . import ops
from .functional import _backward_pass_id, sigmoid_tail, tail_usable

Tensor = torch.Tensor
_KIND = {'mse': 0, 'bce': 1}


def _recon_grad(tail, recon, target, gs, kind, tw):
    """Gradient of the reconstruction terms: in logit form, handed to the decoder that produced `recon`, when that
    decoder left a hand-off record and nobody has used it yet in this backward pass (see functional.SigmoidTail)."""
    if tail_usable(tail, recon):
        d, db = ops.recon_loss_bwd_logit(recon, target, gs, kind, tw)
        tail.dlogit, tail.dbias, tail.task = d, db, _backward_pass_id()
        return d
    return ops.recon_loss_bwd(recon, target, gs, kind, tw)


class LossDict(dict):
    """The reference's loss dictionary (name -> 0-dim tensor).  Every device entry is a view of one
    packed tensor, exposed as `.packed` with `.layout` (name -> index), so a caller can accumulate a
    whole step's values with one device add and read them back once per logging interval instead of
    calling .item() per entry per step."""

    def __init__(self, packed: Tensor, layout: Dict[str, int], beta: float, total: Optional[Tensor] = None):
        super().__init__({k: packed[i] for k, i in layout.items()})
        if total is not None:
            # the entry that carries the gradient is a tensor of its own (same value as packed[0]): its backward
            # then starts from the loss node directly instead of going through an indexing node's zero-fill + copy
            self['total_loss'] = total
        self['beta'] = torch.tensor(beta)          # a CPU tensor, as in the reference
        self.packed = packed
        self.layout = layout


def _kind(name: str) -> int:
    if name not in _KIND:
        raise ValueError(f"Unknown recon_loss_type: {name}")
    return _KIND[name]


class _DisentangledLoss(torch.autograd.Function):
    # values layout: 0 total, 1 kl, 2 recon_per_pixel, 3 recon_scaled, 4 temporal, 5..7 content KL
    # (mean, raw, per-dim), 8..10 motion KL, 11 independence.  Two outputs: the packed values (reporting only) and
    # the total as a tensor of its own, which is what carries the gradient.
    @staticmethod
    def forward(ctx, recon, target, mu_c, lv_c, mu_m, lv_m, z_c, z_m, beta_c, beta_m, kind, tw, iw, tail, beta_dev):
        recon, target = recon.contiguous(), target.contiguous()
        mu_c, lv_c, mu_m, lv_m = [t.contiguous() for t in (mu_c, lv_c, mu_m, lv_m)]
        vals = torch.empty(12, dtype=torch.float32, device=recon.device)
        total = torch.empty(1, dtype=torch.float32, device=recon.device)
        ws = None
        if iw > 0:
            z_c, z_m = z_c.contiguous(), z_m.contiguous()
            ws = ops.indep_fwd(z_c, z_m, vals[11:12])
        # the pass over the clips, then one workgroup for everything small (both KL terms, weights, totals)
        ops.vae_loss_fwd(recon, target, kind, tw > 0, [(mu_c, lv_c, beta_c), (mu_m, lv_m, beta_m)], 0.0, tw,
                         vals[11:12] if iw > 0 else None, iw, vals, total, beta_dev)
        ctx.save_for_backward(recon, target, mu_c, lv_c, mu_m, lv_m, z_c if iw > 0 else None,
                              z_m if iw > 0 else None, ws)
        ctx.cfg = (beta_c, beta_m, kind, tw, iw)
        ctx.beta_dev = beta_dev
        ctx.tail = tail
        ctx.mark_non_differentiable(vals)
        return vals, total.view(())

    @staticmethod
    def backward(ctx, _gvals, g):
        recon, target, mu_c, lv_c, mu_m, lv_m, z_c, z_m, ws = ctx.saved_tensors
        beta_c, beta_m, kind, tw, iw = ctx.cfg
        gs = g.reshape(1).contiguous()                             # d/d total_loss, a device scalar
        need = ctx.needs_input_grad
        drecon = _recon_grad(ctx.tail, recon, target, gs, kind, tw) if need[0] else None
        dmu_c = dlv_c = dmu_m = dlv_m = dz_c = dz_m = None
        if need[2] or need[3] or need[4] or need[5]:
            (dmu_c, dlv_c), (dmu_m, dlv_m) = ops.kl_bwd2([(mu_c, lv_c, beta_c), (mu_m, lv_m, beta_m)], gs, 0.0,
                                                         ctx.beta_dev)
        if iw > 0 and (need[6] or need[7]):
            dz_c, dz_m = ops.indep_bwd(z_c, z_m, ws, gs, iw)
        return (drecon, None, dmu_c, dlv_c, dmu_m, dlv_m, dz_c, dz_m, None, None, None, None, None, None, None)


def _beta_arg(beta, beta_device):
    """The KL weight as the kernels take it: a float baked into the launch, or None = read the device scalar
    beta_device when the kernel runs (a captured step then serves every value of an annealing schedule)."""
    if beta_device is None:
        return float(beta)
    if not (torch.is_tensor(beta_device) and beta_device.is_cuda and beta_device.dtype == torch.float32
            and beta_device.numel() == 1):
        raise ValueError("beta_device must be a one-element fp32 tensor on the GPU")
    return None


def disentangled_loss(output: Dict[str, Tensor], target: Tensor, beta: float = 1.0,
                      beta_content: Optional[float] = None, beta_motion: Optional[float] = None,
                      recon_loss_type: str = 'mse', temporal_weight: float = 0.0,
                      independence_weight: float = 0.0, beta_device: Optional[Tensor] = None) -> Dict[str, Tensor]:
    """Same signature and keys as the reference's compute_loss (models/disentangled_vae.py:222-292).
    beta_device (not in the reference): a device scalar holding `beta`; the kernels then read it instead of the
    launch constant (`beta` itself is still what the 'beta' entry reports)."""
    kind = _kind(recon_loss_type)
    bc = float(beta_content) if beta_content is not None else _beta_arg(beta, beta_device)
    bm = float(beta_motion) if beta_motion is not None else _beta_arg(beta, beta_device)
    v, total = _DisentangledLoss.apply(output['recon'], target, output['mu_content'], output['logvar_content'],
                                       output['mu_motion'], output['logvar_motion'], output['z_content'],
                                       output['z_motion'], bc, bm, kind, float(temporal_weight),
                                       float(independence_weight), sigmoid_tail(output['recon']),
                                       beta_device if (bc is None or bm is None) else None)
    return LossDict(v, {'total_loss': 0, 'recon_loss': 3, 'recon_loss_per_pixel': 2, 'kl_loss': 1,
                        'kl_content': 5, 'kl_motion': 8, 'temporal_loss': 4, 'independence_loss': 11}, beta, total)


class _TupleLoss(torch.autograd.Function):
    # values layout: 0 total, 1 recon_per_pixel, 2 recon_scaled, 3 temporal, 4 kl, 5 kl_raw, 6 kl_per_dim
    @staticmethod
    def forward(ctx, recon, target, mu, logvar, beta, kind, tw, free_bits, tail, beta_dev):
        recon, target, mu, logvar = [t.contiguous() for t in (recon, target, mu, logvar)]
        vals = torch.empty(7, dtype=torch.float32, device=recon.device)
        total = torch.empty(1, dtype=torch.float32, device=recon.device)
        ops.vae_loss_fwd(recon, target, kind, tw > 0, [(mu, logvar, beta)], free_bits, tw, None, 0.0, vals, total,
                         beta_dev)
        ctx.save_for_backward(recon, target, mu, logvar)
        ctx.cfg = (beta, kind, tw, free_bits)
        ctx.beta_dev = beta_dev
        ctx.tail = tail
        ctx.mark_non_differentiable(vals)
        return vals, total.view(())

    @staticmethod
    def backward(ctx, _gvals, g):
        recon, target, mu, logvar = ctx.saved_tensors
        beta, kind, tw, free_bits = ctx.cfg
        gs = g.reshape(1).contiguous()
        need = ctx.needs_input_grad
        drecon = _recon_grad(ctx.tail, recon, target, gs, kind, tw) if need[0] else None
        dmu = dlv = None
        if need[2] or need[3]:
            dmu, dlv = ops.kl_bwd(mu, logvar, gs, beta, free_bits, ctx.beta_dev)
        return drecon, None, dmu, dlv, None, None, None, None, None, None


def _tuple_loss(recon, target, mu, logvar, beta, recon_loss_type, temporal_weight, free_bits, use_free_bits,
                beta_device=None):
    kind = _kind(recon_loss_type)
    fb = float(free_bits) if (use_free_bits and free_bits > 0) else 0.0
    return _TupleLoss.apply(recon, target, mu, logvar, _beta_arg(beta, beta_device), kind, float(temporal_weight),
                            fb, sigmoid_tail(recon), beta_device)


def lstm_loss(recon, target, mu, logvar, beta: float = 1.0, recon_loss_type: str = 'mse',
              temporal_weight: float = 0.0, free_bits: float = 0.0, use_free_bits: bool = False,
              beta_device: Optional[Tensor] = None):
    """Same signature and keys as models/lstm_vae.py:167-220; beta_device as in disentangled_loss."""
    v, total = _tuple_loss(recon, target, mu, logvar, beta, recon_loss_type, temporal_weight, free_bits, use_free_bits,
                           beta_device)
    return LossDict(v, {'total_loss': 0, 'recon_loss': 2, 'recon_loss_per_pixel': 1, 'kl_loss': 4,
                        'kl_loss_raw': 5, 'kl_per_dim': 6, 'temporal_loss': 3}, beta, total)


def baseline_loss(recon, target, mu, logvar, beta: float = 1.0, recon_loss_type: str = 'mse',
                  free_bits: float = 0.0, use_free_bits: bool = False, beta_device: Optional[Tensor] = None):
    """Same signature and keys as models/baseline_vae.py:208-252; beta_device as in disentangled_loss."""
    v, total = _tuple_loss(recon, target, mu, logvar, beta, recon_loss_type, 0.0, free_bits, use_free_bits,
                           beta_device)
    return LossDict(v, {'total_loss': 0, 'recon_loss': 2, 'recon_loss_per_pixel': 1, 'kl_loss': 4,
                        'kl_loss_raw': 5, 'kl_per_dim': 6}, beta, total)
