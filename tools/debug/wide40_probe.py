"""128x128x40 full-width step: per-tensor gradient error, the fp32 restatement's own error and the kink sensitivity."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'disentangled-vae_amd')); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import models
from hipvae import rng
from test_models_gpu import FULL, SETTINGS, loss_of, rel
from test_bench_size_gpu import _oracle_step
name, B, T, size = 'disentangled_128_wide40', int(os.environ.get('PB', 2)), int(os.environ.get('PT', 40)), 128
torch.manual_seed(0)
cfgm = dict(FULL[name], sequence_length=T)
m = models.get_model('disentangled', in_channels=1, **cfgm)
sd = {k: v.clone() for k, v in m.state_dict().items()}
gen = torch.Generator().manual_seed(1)
video = (torch.rand(B, T, 1, size, size, generator=gen) > 0.9).float() * torch.rand(B, T, 1, size, size, generator=gen)
eps = [torch.randn(B, 128, generator=torch.Generator().manual_seed(100 + k)) for k in range(2)]
tcfg = SETTINGS['tcfg']['disentangled']
keys, oout, oloss, og, _ = _oracle_step('disentangled', sd, video, eps, tcfg, 0.4, 1, torch.float64)
_, _, _, og32, _ = _oracle_step('disentangled', sd, video, eps, tcfg, 0.4, 1, torch.float32)
_, _, _, lo, _ = _oracle_step('disentangled', sd, video, eps, tcfg, 0.4, 1, torch.float64, kink=-1e-5)
_, _, _, hi, _ = _oracle_step('disentangled', sd, video, eps, tcfg, 0.4, 1, torch.float64, kink=1e-5)
m = m.cuda().train()
with rng.inject_eps(eps):
    out = m(video.cuda())
loss_of('disentangled', out, video.cuda(), tcfg, 0.4)['total_loss'].backward()
named = dict(m.named_parameters())
for k, a, b, c, d in zip(keys, og, og32, lo, hi):
    r, own, kk = rel(named[k].grad.cpu().numpy(), a.numpy()), rel(b.numpy(), a.numpy()), rel(d.numpy(), c.numpy())
    flag = '' if r < max(1e-4, 2 * own) else ('  <-- above max(1e-4, 2 own)' + (' but within kink' if r <= kk else ' AND above kink'))
    print(f'{k:60s} err {r:.2e} own {own:.2e} kink {kk:.2e}{flag}', flush=True)
