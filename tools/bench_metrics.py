"""Rate of the one-pass evaluation-metrics kernel on a batch of 256 clips of 20x64x64.
usage: python tools/bench_metrics.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'disentangled-vae_amd'))
from hipvae import ops  # noqa: E402

B, T, S = 256, 20, 64
r = torch.rand(B, T, 1, S, S, device='cuda')
x = torch.rand(B, T, 1, S, S, device='cuda')
out = torch.empty(3, device='cuda')
for _ in range(3):
    ops.recon_metrics(r, x, out=out)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(50):
    ops.recon_metrics(r, x, out=out)
e.record()
torch.cuda.synchronize()
us = s.elapsed_time(e) / 50 * 1e3
print(f'recon_metrics: {us:.1f} us per batch of {B} clips, {2 * r.numel() * 4 / us / 1e3:.0f} GB/s, '
      f'{B / us * 1e6:.0f} clips/s')
# the reference's three separate passes with torch ops on the same device, for scale
s.record()
for _ in range(50):
    a = torch.nn.functional.mse_loss(r, x).item()
    b = torch.nn.functional.mse_loss(r, x).item()
    c = (r[:, 1:] - r[:, :-1]).abs().mean().item()
e.record()
torch.cuda.synchronize()
print(f'torch ops as the reference calls them (3 passes, 3 read-backs): {s.elapsed_time(e) / 50 * 1e3:.1f} us')
