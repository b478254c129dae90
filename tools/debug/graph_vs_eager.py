"""Diagnostic: parameter differences between graph-replayed and eager training in the tiny Trainer setup."""
import json
import os
import sys
import tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'disentangled-vae_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import pathlib
import torch
from hipvae import rng
from training.trainer import create_trainer
from test_trainer_gpu import _tiny_trainer_config, synthetic_clips

ragged = int(sys.argv[1]) if len(sys.argv) > 1 else 1
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
train = [(synthetic_clips(4, 6, 20), [{}] * 4), (synthetic_clips(4, 6, 21), [{}] * 4),
         (synthetic_clips(2 if ragged else 4, 6, 22), [{}] * (2 if ragged else 4))]
val = [(synthetic_clips(4, 6, 40), [{}] * 4)]
rng.draw_like = lambda t: torch.zeros_like(t)
tmp = pathlib.Path(tempfile.mkdtemp())
runs = []
for tag, use_graph in (('eager', False), ('graph', True), ('eager2', False)):
    cfg = _tiny_trainer_config(tmp, tag, epochs)
    cfg.training.use_kl_annealing = False
    cfg.training.kl_start_epoch = 0
    cfg.training.lr_scheduler = 'exponential'
    cfg.training.lr_decay = 0.5
    cfg.training.lr_min = 1e-9
    cfg.training.learning_rate = 2e-3
    torch.manual_seed(5)
    tr = create_trainer(cfg, train, val)
    tr.use_graph = use_graph
    hist = tr.train()
    runs.append((hist, {k: v.detach().clone() for k, v in tr.model.state_dict().items()}))
(he, pe), (hg, pg), (h2, p2) = runs
print('history eager', he['train_loss'], 'graph', hg['train_loss'])
for name, other in (('graph', pg), ('eager2', p2)):
    worst = []
    for k in pe:
        if pe[k].is_floating_point():
            d = (pe[k] - other[k]).abs()
            worst.append((d.max().item(), d.mean().item(), (d > 1e-5).float().mean().item(), k))
    worst.sort(reverse=True)
    print(name, 'vs eager: (max, mean, frac>1e-5)')
    for w in worst[:6]:
        print('   ', w)
