"""Which lines of this package launch torch's own kernels in one eager training step: the aten operators that ran a
device kernel, with the innermost frame inside the package, counted per step."""
import importlib, os, sys, collections
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'disentangled-vae_amd'))
import bench
import models
from hipvae import losses as L
from hipvae.optim import FusedAdam

cfg = bench.load_config(sys.argv[1] if len(sys.argv) > 1 else 'disentangled_simple.json')
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
kind = cfg['model']['model_type']
t = cfg['training']
dev = torch.device('cuda:0')
model = models.get_model(kind, **bench.model_kwargs(cfg)).to(dev).train()
opt = FusedAdam(model.parameters(), lr=t['learning_rate'], betas=(t['beta1'], t['beta2']), weight_decay=t['weight_decay'])
T, size = cfg['data']['sequence_length'], cfg['data']['frame_size']
video = bench.synthetic_clips(B, T, size, 0).to(dev)


def step():
    opt.zero_grad(set_to_none=True)
    out = model(video)
    losses = bench.loss_call(kind, L, out, video, t, 4.0)
    losses['total_loss'].backward()
    opt.clip_grad_norm_(t['gradient_clip'], fold_into_step=True)
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
count = collections.Counter()
import threading
main_thread = min(ev.thread for ev in prof.events())
for ev in prof.events():
    if not ev.name.startswith('aten::') or ev.device_time_total <= 0 or not ev.kernels:
        continue
    where = 'autograd engine (gradient accumulation)'
    for fr in ev.stack or []:
        if 'disentangled-vae_amd' in fr or 'bench.py' in fr:
            where = fr.strip()
            break
    where = where[-110:] + '  ' + str(ev.input_shapes)[:90] + ('  [backward]' if ev.is_async or getattr(ev, 'thread', 0) != main_thread else '')
    count[(ev.name, where, tuple(k.name[:50] for k in ev.kernels))] += 1
for (name, where, ks), n in sorted(count.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    print(f'{n:3d} {name:22s} {where}  {ks[0]}')
