"""Per-step losses of the tiny reference Trainer run with injected draws: eager launches against graph replay."""
import json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'disentangled-vae_amd')); sys.path.insert(0, ROOT)
from hipvae import rng
from training.config import DataConfig, ExperimentConfig, ModelConfig, TrainingConfig
from training.trainer import create_trainer
G = os.path.join(ROOT, 'tests', 'golden')
info = json.load(open(os.path.join(G, 'trainer_run.json'))); arrays = np.load(os.path.join(G, 'trainer_run.npz'))
def clips(B, T, seed):
    g = torch.Generator().manual_seed(seed)
    u1 = torch.rand(B, T, 1, 64, 64, generator=g); u2 = torch.rand(B, T, 1, 64, 64, generator=g)
    return (u1 > 0.9).float() * u2
raw = info['config']
for use_graph in (False, True):
    cfg = ExperimentConfig(name=raw['name'])
    cfg.data, cfg.model, cfg.training = DataConfig(**raw['data']), ModelConfig(**raw['model']), TrainingConfig(**raw['training'])
    cfg.training.device = 'cuda'; cfg.training.checkpoint_dir = f'/tmp/p{int(use_graph)}/ck'; cfg.training.results_dir = f'/tmp/p{int(use_graph)}/res'
    train = [(clips(4, 6, 20 + i), [{}] * 4) for i in range(3)]
    torch.manual_seed(5)
    tr = create_trainer(cfg, train, train)
    tr.use_graph = use_graph
    eps = [torch.from_numpy(arrays[f'eps{k}']) for k in range(9)]
    tr.model.train()
    import contextlib
    if os.environ.get('ZERO_EPS') == '1':
        rng.draw_like = lambda t: torch.zeros_like(t)
    with (contextlib.nullcontext() if os.environ.get('ZERO_EPS') == '1' else rng.inject_eps(eps)):
        for i in range(6):
            l = tr.train_step(train[i % 3][0].cuda(), 0.25)
            print(use_graph, i, [round(float(x), 4) for x in l.packed.tolist()], flush=True)
print('ref', info['history']['train_loss'])
