"""Fused multi-tensor gradient clipping and Adam.

FusedAdam subclasses torch.optim.Adam so that param_groups, state layout ('step' as an fp32 scalar
tensor, 'exp_avg', 'exp_avg_sq'), state_dict()/load_state_dict() and the LR schedulers behave exactly
as with the optimiser the reference builds (training/trainer.py:56-61); only step() is replaced: one
kernel chain over all tensors instead of per-tensor elementwise ops.  L2 weight decay (g += wd*p),
not AdamW.  Nothing here synchronises with the host.
"""
from typing import Optional

import torch

from . import ops


class FusedAdam(torch.optim.Adam):

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        self._tables = {}
        self._norm_out = None          # [norm, clip coefficient] on the device
        self._pending_coef = None
        self._dyn = None               # device {step count, learning rate} for graph replay
        self._dyn_lr = None
        self._replayed = 0             # steps taken by graph replays, not yet written to state['step']

    # ---- device-resident step state (needed when step() is captured into a hipGraph) -----------
    def enable_device_state(self):
        """Keep the step count and learning rate in device memory and let the Adam kernel derive its
        bias corrections from them, so that a captured graph stays correct as they change."""
        if len(self.param_groups) != 1:
            raise RuntimeError("device-resident step state supports a single parameter group")
        p0 = self.param_groups[0]['params'][0]
        steps = 0.0
        for p in self.param_groups[0]['params']:
            if p in self.state and len(self.state[p]):
                steps = float(self.state[p]['step'])
                break
        self._dyn = torch.tensor([steps, float(self.param_groups[0]['lr'])], dtype=torch.float32, device=p0.device)
        self._dyn_lr = float(self.param_groups[0]['lr'])
        return self._dyn

    def sync_device_lr(self):
        """Push a learning rate changed by a scheduler to the device copy (no-op when unchanged)."""
        if self._dyn is not None:
            lr = float(self.param_groups[0]['lr'])
            if lr != self._dyn_lr:
                self._dyn[1:2].fill_(lr)
                self._dyn_lr = lr

    def note_replayed_step(self):
        self._replayed += 1

    def _flush_replayed(self):
        if self._replayed:
            for st in self.state.values():
                if 'step' in st:
                    st['step'] += float(self._replayed)
            self._replayed = 0

    def state_dict(self):
        self._flush_replayed()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        self._replayed = 0
        super().load_state_dict(state_dict)
        if self._dyn is not None:
            self.enable_device_state()

    # ---- gradient clipping -------------------------------------------------------------------
    def _grads(self):
        return [p.grad for g in self.param_groups for p in g['params'] if p.grad is not None]

    def grad_norm(self, max_norm: float = 0.0) -> torch.Tensor:
        """Total 2-norm of all gradients (device scalar); also prepares the clip coefficient
        min(1, max_norm/(norm+1e-6)) that the next step() folds into the gradient read."""
        grads = self._grads()
        if not grads:
            raise RuntimeError("no gradients")
        if self._norm_out is None or self._norm_out.device != grads[0].device:
            self._norm_out = torch.zeros(2, dtype=torch.float32, device=grads[0].device)
        ops.grad_norm(grads, max_norm, self._norm_out, self._tables.setdefault('norm', ops.TensorTable()))
        return self._norm_out[0]

    def clip_grad_norm_(self, max_norm: float, fold_into_step: bool = False) -> torch.Tensor:
        """nn.utils.clip_grad_norm_ semantics (training/trainer.py:198-202): returns the total norm and
        scales the gradients by the clip coefficient, in place, or (fold_into_step) leaves them
        untouched and applies the coefficient inside the next step(), which saves a pass."""
        norm = self.grad_norm(max_norm)
        if fold_into_step:
            self._pending_coef = self._norm_out[1:2]
        else:
            ops.scale_tensors(self._grads(), self._norm_out[1:2], self._tables.setdefault('scale', ops.TensorTable()))
        return norm

    # ---- the step ----------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        coef, self._pending_coef = self._pending_coef, None
        self._flush_replayed()
        if self._dyn is not None:
            ops.adam_advance(self._dyn)
        for gi, group in enumerate(self.param_groups):
            params, grads, m, v = [], [], [], []
            step_val = None
            for p in group['params']:
                if p.grad is None:
                    continue
                st = self.state[p]
                if len(st) == 0:
                    st['step'] = torch.tensor(0.0, dtype=torch.float32)
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st['step'] += 1
                s = float(st['step'])
                if step_val is None:
                    step_val = s
                elif s != step_val:
                    raise RuntimeError("FusedAdam needs every tensor of a group at the same step count")
                params.append(p)
                grads.append(p.grad)
                m.append(st['exp_avg'])
                v.append(st['exp_avg_sq'])
            if not params:
                continue
            b1, b2 = group['betas']
            lr = group['lr']
            lr = float(lr) if not isinstance(lr, torch.Tensor) else float(lr.item())
            ops.adam_step(params, grads, m, v, lr, b1, b2, group['eps'], group['weight_decay'],
                          1.0 - b1 ** step_val, 1.0 - b2 ** step_val, coef,
                          self._tables.setdefault(('adam', gi), ops.TensorTable()), dyn=self._dyn)
        return loss
