"""eager training steps on a side stream, with and without the library's launch trace (debugging aid)"""
import faulthandler
import os
import sys

import torch

faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'disentangled-vae_amd'))
import bench  # noqa: E402
import models  # noqa: E402
from hipvae import losses as L, ops  # noqa: E402
from hipvae.graph import capture_stream, GraphedTrainStep  # noqa: E402
from hipvae.optim import FusedAdam  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device('cuda', 0)
cfg = bench.load_config('disentangled_simple.json')
kind, t = cfg['model']['model_type'], cfg['training']
m = models.get_model(kind, **bench.model_kwargs(cfg)).to(dev).train()
opt = FusedAdam(m.parameters(), lr=1e-4)
v = bench.synthetic_clips(B, 20, 64, 1).to(dev)


def step():
    opt.zero_grad(set_to_none=True)
    losses = bench.loss_call(kind, L, m(v), v, t, 4.0)
    losses['total_loss'].backward()
    opt.clip_grad_norm_(1.0, fold_into_step=True)
    opt.step()
    return losses


if '--graph-first' in sys.argv:
    g = GraphedTrainStep(m, opt, lambda mm, x: bench.loss_call(kind, L, mm(x), x, t, 4.0), v, gradient_clip=1.0)
    g(v)
    print('graph ok', flush=True)
side = capture_stream(dev)
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    step()
    torch.cuda.synchronize()
    print('side eager ok', flush=True)
    rows = bench.trace_steps(lambda i: step(), 2, 157.3)
    print('side traced ok', len(rows), flush=True)
torch.cuda.synchronize()
print('ok')
