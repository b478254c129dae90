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
        self._dyn = None               # device {step count, learning rate, bias corrections} for graph replay
        self._dyn_lr = None
        self._replayed = 0             # steps taken by graph replays, not yet written to state['step']

    # ---- device-resident step state (needed when step() is captured into a hipGraph) -----------
    def enable_device_state(self):
        """Keep the step count and learning rate in device memory and let the Adam kernel derive its
        bias corrections from them, so that a captured graph stays correct as they change.  The device
        pair is allocated once per optimiser and only refreshed in place afterwards: every graph
        captured from this optimiser holds its raw address."""
        if len(self.param_groups) != 1:
            raise RuntimeError("device-resident step state supports a single parameter group")
        p0 = self.param_groups[0]['params'][0]
        self._flush_replayed()
        steps = 0.0
        for p in self.param_groups[0]['params']:
            if p in self.state and len(self.state[p]):
                steps = float(self.state[p]['step'])
                break
        lr = float(self.param_groups[0]['lr'])
        # slots 2 and 3 (the bias corrections) are rewritten by every adam_advance before they are read
        host = torch.tensor([steps, lr, 1.0, 1.0], dtype=torch.float32)
        if self._dyn is None:
            self._dyn = host.to(p0.device)
        else:
            self._dyn.copy_(host)
        self._dyn_lr = lr
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
        """torch.optim.Adam.load_state_dict, except that moment tensors which already exist keep their
        storage (the loaded values are copied into them): captured graphs hold their addresses."""
        self._replayed = 0
        old = {p: dict(st) for p, st in self.state.items()}
        super().load_state_dict(state_dict)
        for p, st in self.state.items():
            # the step count is read on the host at every step: keep it there (a checkpoint loaded with
            # map_location=<gpu> would otherwise turn that read into a device synchronisation, which a
            # stream under graph capture refuses)
            if torch.is_tensor(st.get('step')) and st['step'].is_cuda:
                st['step'] = st['step'].cpu()
            prev = old.get(p)
            if not prev:
                continue
            for k in ('exp_avg', 'exp_avg_sq'):
                a, b = prev.get(k), st.get(k)
                if a is not None and b is not None and a is not b and a.shape == b.shape and a.device == b.device:
                    a.copy_(b)
                    st[k] = a
        self._tables = {}
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
            ops.adam_advance(self._dyn, *self.param_groups[0]['betas'])
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
