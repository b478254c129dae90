"""Generate the golden vectors under tests/golden/ by running the reference itself on CPU.

Run in the build container only (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference tree never travels; only the arrays written here do.  Two optional third-party
modules the reference imports but this image lacks (tensorboard's SummaryWriter, torchvision's
MNIST loader) are replaced by inert doubles so that the reference's *unchanged* Trainer and
MovingMNIST classes run; MNIST itself is replaced by a synthetic seeded digit bank.
"""
import hashlib
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))


# ---------------------------------------------------------------------------- test doubles
def synthetic_digit_bank(n: int = 64, seed: int = 1234) -> np.ndarray:
    """[n,28,28] float32 = uint8/255, blob-like, seeded; stands in for MNIST digits."""
    rng = np.random.RandomState(seed)
    raw = rng.randint(0, 256, size=(n, 28, 28)).astype(np.uint8)
    mask = rng.rand(n, 28, 28) < 0.35
    raw = np.where(mask, raw, 0).astype(np.uint8)
    raw[:, :3, :] = 0
    raw[:, -3:, :] = 0
    return raw.astype(np.float32) / 255.0


def install_doubles(bank: np.ndarray):
    tb = types.ModuleType('torch.utils.tensorboard')

    class SummaryWriter:                                   # inert
        def __init__(self, *a, **k): pass
        def add_scalar(self, *a, **k): pass
        def add_image(self, *a, **k): pass
        def close(self): pass
    tb.SummaryWriter = SummaryWriter
    sys.modules['torch.utils.tensorboard'] = tb

    tv = types.ModuleType('torchvision')
    ds = types.ModuleType('torchvision.datasets')
    tr = types.ModuleType('torchvision.transforms')

    class MNIST:
        def __init__(self, root=None, train=True, download=False, transform=None):
            self.images = bank
        def __len__(self): return len(self.images)
        def __getitem__(self, i): return torch.from_numpy(self.images[i])[None], int(i % 10)
        def __iter__(self):
            for i in range(len(self)):
                yield self[i]

    class ToTensor:
        def __call__(self, x): return x

    class Compose:
        def __init__(self, ts): self.ts = ts
    ds.MNIST = MNIST
    tr.ToTensor = ToTensor
    tr.Compose = Compose
    tv.datasets = ds
    tv.transforms = tr
    sys.modules['torchvision'] = tv
    sys.modules['torchvision.datasets'] = ds
    sys.modules['torchvision.transforms'] = tr


# ---------------------------------------------------------------------------- helpers
def synthetic_clips(B, T, seed):
    g = torch.Generator().manual_seed(seed)
    u1 = torch.rand(B, T, 1, 64, 64, generator=g)
    u2 = torch.rand(B, T, 1, 64, 64, generator=g)
    return (u1 > 0.9).float() * u2


class EpsInjector:
    """Replaces torch.randn_like by draws from a list of prepared tensors (content first, then motion)."""
    def __init__(self, eps_list):
        self.eps = list(eps_list)
        self.i = 0
        self.orig = torch.randn_like
    def __enter__(self):
        def fake(t, **kw):
            e = self.eps[self.i].to(t.dtype)
            self.i += 1
            assert e.shape == t.shape
            return e
        torch.randn_like = fake
        return self
    def __exit__(self, *a):
        torch.randn_like = self.orig


SMALL = {
    'disentangled': dict(content_dim=16, motion_dim=16, content_channels=[4, 8, 16, 32],
                         motion_channels=[4, 8], decoder_channels=[32, 16, 8, 4],
                         content_feature_dim=32, motion_feature_dim=32, lstm_hidden=32,
                         lstm_layers=1, bidirectional=True, sequence_length=6),
    'lstm': dict(latent_dim=16, encoder_channels=[4, 8, 16, 32], decoder_channels=[32, 16, 8, 4],
                 feature_dim=32, lstm_hidden=32, lstm_layers=2, bidirectional=True, sequence_length=6),
    'baseline': dict(latent_dim=16, encoder_channels=[4, 8, 16, 32], decoder_channels=[32, 16, 8, 4],
                     feature_dim=32, sequence_length=6),
}
TCFG = {
    'disentangled': dict(beta_content=0.4, beta_motion=0.4, recon_loss_type='bce', temporal_weight=0.05,
                         independence_weight=0.02, gradient_clip=5.0, weight_decay=1e-5,
                         beta1=0.9, beta2=0.999),
    'lstm': dict(recon_loss_type='bce', free_bits=0.25, use_free_bits=True, gradient_clip=5.0,
                 weight_decay=1e-5, beta1=0.9, beta2=0.999),
    'baseline': dict(recon_loss_type='mse', free_bits=0.3, use_free_bits=True, gradient_clip=5.0,
                     weight_decay=1e-5, beta1=0.9, beta2=0.999),
}
BETA = 0.4
LR = 3e-4


def ref_loss(kind, models, out, video, tcfg):
    if kind == 'disentangled':
        return models.disentangled_loss(out, video, beta=BETA, beta_content=tcfg['beta_content'],
                                        beta_motion=tcfg['beta_motion'],
                                        recon_loss_type=tcfg['recon_loss_type'],
                                        temporal_weight=tcfg['temporal_weight'],
                                        independence_weight=tcfg['independence_weight'])
    fn = models.lstm_loss if kind == 'lstm' else models.baseline_loss
    recon, z, mu, logvar = out
    return fn(recon, video, mu, logvar, beta=BETA, recon_loss_type=tcfg['recon_loss_type'],
              free_bits=tcfg['free_bits'], use_free_bits=tcfg['use_free_bits'])


def out_dict(kind, out):
    if kind == 'disentangled':
        return dict(out)
    recon, z, mu, logvar = out
    return {'recon': recon, 'z': z, 'mu': mu, 'logvar': logvar}


def model_vectors(kind, models):
    """One training step of the reference on a shrunken model; fp32 and fp64."""
    B, T = 4, 6
    torch.manual_seed(7)
    model = models.get_model(kind, in_channels=1, **SMALL[kind])
    # make BatchNorm affine and running buffers non-trivial so the test exercises them
    g = torch.Generator().manual_seed(11)
    for name, buf in model.state_dict().items():
        if name.endswith('running_mean'):
            buf.copy_(0.1 * torch.randn(buf.shape, generator=g))
        elif name.endswith('running_var'):
            buf.copy_(1.0 + 0.2 * torch.rand(buf.shape, generator=g))
    for name, p in model.named_parameters():
        if p.dim() == 1 and ('conv.' in name or 'deconv.' in name):
            with torch.no_grad():
                if name.endswith('weight'):
                    p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
                else:
                    p.copy_(0.1 * torch.randn(p.shape, generator=g))
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    video = synthetic_clips(B, T, 3)
    n_eps = 2 if kind == 'disentangled' else 1
    dim = SMALL[kind].get('content_dim', SMALL[kind].get('latent_dim'))
    eps = [torch.randn(B, dim, generator=torch.Generator().manual_seed(100 + k)) for k in range(n_eps)]
    tcfg = TCFG[kind]
    res = {'video': video.numpy()}
    for k, e in enumerate(eps):
        res[f'eps{k}'] = e.numpy()
    for k, v in sd0.items():
        res['sd/' + k] = v.numpy()

    # eval-mode forward (no_grad: z = mu, running statistics)
    model.eval()
    with torch.no_grad():
        out = out_dict(kind, model(video))
        for k, v in out.items():
            res['eval/' + k] = v.numpy()
        for k, v in ref_loss(kind, models, model(video), video, tcfg).items():
            res['eval_loss/' + k] = v.numpy()

    for dtype, tag in ((torch.float32, 'f32'), (torch.float64, 'f64')):
        m = models.get_model(kind, in_channels=1, **SMALL[kind])
        m.load_state_dict(sd0)
        m = m.to(dtype)
        m.train()
        v = video.to(dtype)
        opt = torch.optim.Adam(m.parameters(), lr=LR, betas=(tcfg['beta1'], tcfg['beta2']),
                               weight_decay=tcfg['weight_decay'])
        opt.zero_grad()
        with EpsInjector(eps):
            out = m(v)
        losses = ref_loss(kind, models, out, v, tcfg)
        losses['total_loss'].backward()
        for k, t in out_dict(kind, out).items():
            res[f'train_{tag}/{k}'] = t.detach().float().numpy()
        for k, t in losses.items():
            res[f'loss_{tag}/{k}'] = np.float64(t.item())
        for name, p in m.named_parameters():
            res[f'grad_{tag}/{name}'] = p.grad.detach().float().numpy().copy()
        if tag == 'f32':
            total = torch.nn.utils.clip_grad_norm_(m.parameters(), tcfg['gradient_clip'])
            res['grad_norm_f32'] = np.float64(total.item())
            opt.step()
            for name, t in m.state_dict().items():
                res['post/' + name] = t.detach().numpy()
    return res


def schedules(cfgmod):
    """beta per epoch (Trainer._get_beta) and LR per epoch as Trainer.train steps the scheduler."""
    import training.trainer as trainer_mod
    out = {}
    for fn in sorted(os.listdir(os.path.join(REF, 'configs'))):
        cfg = cfgmod.ExperimentConfig.load(os.path.join(REF, 'configs', fn))
        t = trainer_mod.Trainer.__new__(trainer_mod.Trainer)
        t.config = cfg
        betas = [t._get_beta(e) for e in range(cfg.training.num_epochs)]
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.Adam([p], lr=cfg.training.learning_rate)
        t.optimizer = opt
        sched = t._setup_scheduler()
        lrs = []
        for e in range(cfg.training.num_epochs):
            opt.step()
            if sched is not None and opt.param_groups[0]['lr'] > cfg.training.lr_min:
                sched.step()
            lrs.append(opt.param_groups[0]['lr'])
        out[fn] = {'beta': betas, 'lr': lrs}
    # the other annealing types, on a synthetic config
    cfg = cfgmod.ExperimentConfig()
    for kind in ('cosine', 'cyclical'):
        cfg.training.annealing_type = kind
        cfg.training.warmup_epochs = 7
        cfg.training.kl_start_epoch = 3
        cfg.training.beta_vae = 2.0
        t = trainer_mod.Trainer.__new__(trainer_mod.Trainer)
        t.config = cfg
        out['synthetic_' + kind] = {'beta': [t._get_beta(e) for e in range(30)],
                                    'cfg': dict(beta_vae=2.0, kl_start_epoch=3, warmup_epochs=7,
                                                annealing_type=kind, use_kl_annealing=True)}
    return out


def manifests(models, trainer_mod, cfgmod):
    out = {}
    for fn in sorted(os.listdir(os.path.join(REF, 'configs'))):
        cfg = cfgmod.ExperimentConfig.load(os.path.join(REF, 'configs', fn))
        cfg.training.checkpoint_dir = tempfile.mkdtemp()
        cfg.training.results_dir = tempfile.mkdtemp()
        cfg.training.device = 'cpu'
        tr = trainer_mod.create_trainer(cfg, [], [])
        sd = tr.model.state_dict()
        out[fn] = {'n_params': sum(p.numel() for p in tr.model.parameters() if p.requires_grad),
                   'n_trainable_tensors': len(list(tr.model.parameters())),
                   'model_type': tr.model.model_type,
                   'keys': {k: list(v.shape) for k, v in sd.items()}}
    m = models.DisentangledVAE(content_channels=[64, 128, 256, 512, 512], motion_channels=[32, 64, 64],
                               decoder_channels=[512, 512, 256, 128, 64], sequence_length=40)
    out['config5_128x128x40'] = {'n_params': sum(p.numel() for p in m.parameters()),
                                 'keys': {k: list(v.shape) for k, v in m.state_dict().items()}}
    return out


def generator_vectors(bank):
    from data.moving_mnist import MovingMNIST
    out = {'bank_sha1': hashlib.sha1(bank.tobytes()).hexdigest(), 'items': []}
    arrays = {}
    for seed, idxs, kw in ((42, (0, 1, 17), {}), (1042, (0, 5), {}), (0, (3,), {}),
                           (7, (2,), dict(sequence_length=8, num_digits=3))):
        ds = MovingMNIST(root='.', train=True, num_videos=32, seed=seed, deterministic=True, **kw)
        for i in idxs:
            video, meta = ds[i]
            v = video.numpy()
            assert v.dtype == np.float32
            key = f'video_s{seed}_i{i}'
            q = np.round(v * 255).astype(np.uint8)
            assert np.array_equal(q.astype(np.float32) / 255.0, v)
            arrays[key] = q
            out['items'].append({
                'seed': seed, 'index': i, 'kwargs': kw, 'shape': list(v.shape),
                'sha1': hashlib.sha1(v.tobytes()).hexdigest(),
                'labels': [int(x) for x in meta['labels']],
                'initial_positions': [p.tolist() for p in meta['initial_positions']],
                'initial_velocities': [p.tolist() for p in meta['initial_velocities']],
                'last_positions': [p.tolist() for p in meta['position_history'][-1]],
                'last_velocities': [p.tolist() for p in meta['velocity_history'][-1]],
            })
    return out, arrays


def trainer_run(trainer_mod, cfgmod):
    """Two epochs of the reference's unchanged Trainer on fixed synthetic batches (host-logic pin)."""
    cfg = cfgmod.ExperimentConfig(name='golden_run')
    cfg.model.model_type = 'baseline'
    cfg.model.latent_dim = 16
    cfg.model.encoder_channels = [4, 8, 16, 32]
    cfg.model.decoder_channels = [32, 16, 8, 4]
    cfg.model.feature_dim = 32
    cfg.model.sequence_length = 6
    cfg.training.num_epochs = 3
    cfg.training.warmup_epochs = 2
    cfg.training.kl_start_epoch = 1
    cfg.training.beta_vae = 0.5
    cfg.training.free_bits = 0.3
    cfg.training.recon_loss_type = 'bce'
    cfg.training.learning_rate = 5e-4
    cfg.training.lr_scheduler = 'cosine'
    cfg.training.gradient_clip = 5.0
    cfg.training.save_every = 2
    cfg.training.visualize_every = 100
    cfg.training.device = 'cpu'
    tmp = tempfile.mkdtemp()
    cfg.training.checkpoint_dir = os.path.join(tmp, 'ck')
    cfg.training.results_dir = os.path.join(tmp, 'res')
    train = [(synthetic_clips(4, 6, 20 + i), [{}] * 4) for i in range(3)]
    val = [(synthetic_clips(4, 6, 40 + i), [{}] * 4) for i in range(2)]
    torch.manual_seed(5)
    tr = trainer_mod.create_trainer(cfg, train, val)
    sd0 = {k: v.clone() for k, v in tr.model.state_dict().items()}
    eps = [torch.randn(4, 16, generator=torch.Generator().manual_seed(500 + k)) for k in range(9)]
    with EpsInjector(eps):
        hist = tr.train()
    files = sorted(os.listdir(os.path.join(cfg.training.checkpoint_dir, cfg.name)))
    ck = torch.load(os.path.join(cfg.training.checkpoint_dir, cfg.name, 'final_model.pt'),
                    weights_only=False)
    info = {
        'history': hist, 'files': files, 'checkpoint_keys': sorted(ck.keys()),
        'epoch': ck['epoch'], 'global_step': ck['global_step'],
        'optimizer_param_group_keys': sorted(ck['optimizer_state_dict']['param_groups'][0].keys()),
        'optimizer_state_keys': sorted(ck['optimizer_state_dict']['state'][0].keys()),
        'optimizer_step_dtype': str(ck['optimizer_state_dict']['state'][0]['step'].dtype),
        'optimizer_step_value': float(ck['optimizer_state_dict']['state'][0]['step']),
        'config': ck['config'],
        'results_files': sorted(os.listdir(os.path.join(cfg.training.results_dir, cfg.name))),
    }
    arrays = {'sd0/' + k: v.numpy() for k, v in sd0.items()}
    arrays.update({'final/' + k: v.numpy() for k, v in ck['model_state_dict'].items()})
    for k in range(9):
        arrays[f'eps{k}'] = eps[k].numpy()
    return info, arrays


def main():
    bank = synthetic_digit_bank()
    install_doubles(bank)
    sys.path.insert(0, REF)
    import models
    import training.config as cfgmod
    import training.trainer as trainer_mod

    for kind in ('disentangled', 'lstm', 'baseline'):
        res = model_vectors(kind, models)
        np.savez_compressed(os.path.join(HERE, f'step_{kind}.npz'), **res)
        print(kind, 'loss_f32', res['loss_f32/total_loss'], 'loss_f64', res['loss_f64/total_loss'])

    with open(os.path.join(HERE, 'schedules.json'), 'w') as f:
        json.dump(schedules(cfgmod), f)
    with open(os.path.join(HERE, 'manifests.json'), 'w') as f:
        json.dump(manifests(models, trainer_mod, cfgmod), f)
    gen, arrays = generator_vectors(bank)
    with open(os.path.join(HERE, 'generator.json'), 'w') as f:
        json.dump(gen, f)
    np.savez_compressed(os.path.join(HERE, 'generator_videos.npz'), **arrays)
    info, arrays = trainer_run(trainer_mod, cfgmod)
    with open(os.path.join(HERE, 'trainer_run.json'), 'w') as f:
        json.dump(info, f)
    np.savez_compressed(os.path.join(HERE, 'trainer_run.npz'), **arrays)
    with open(os.path.join(HERE, 'settings.json'), 'w') as f:
        json.dump({'small': SMALL, 'tcfg': TCFG, 'beta': BETA, 'lr': LR}, f)
    print('done')


if __name__ == '__main__':
    main()
