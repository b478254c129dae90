"""Golden vectors for the model methods the reference's evaluation code and notebooks call besides
forward(): sample, swap_content / swap_motion, interpolate_content / interpolate_motion, get_latents,
the tuple-API wrapper, content_aggregation='middle', and the baseline / LSTM models' sample, interpolate,
get_latent, get_frame_features.  Produced by running the reference itself on CPU (build container only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_surface.py

Writes tests/golden/surface.npz.  Only arrays travel.
"""
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import REF, SMALL, EpsInjector, synthetic_clips  # noqa: E402

sys.path.insert(0, REF)
import models  # noqa: E402  (the reference's package)


def randomise_bn(model, seed):
    g = torch.Generator().manual_seed(seed)
    for name, buf in model.state_dict().items():
        if name.endswith('running_mean'):
            buf.copy_(0.1 * torch.randn(buf.shape, generator=g))
        elif name.endswith('running_var'):
            buf.copy_(1.0 + 0.2 * torch.rand(buf.shape, generator=g))
    with torch.no_grad():
        for name, p in model.named_parameters():
            if p.dim() == 1 and ('conv.' in name or 'deconv.' in name):
                if name.endswith('weight'):
                    p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
                else:
                    p.copy_(0.1 * torch.randn(p.shape, generator=g))


def put(res, prefix, **arrays):
    """Clip-shaped outputs [.., T, 1, 64, 64] are stored as every third pixel in both directions (a test
    subsamples its own output the same way): the fixture stays small and still pins every frame."""
    for k, v in arrays.items():
        a = v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)
        if a.ndim == 5 and a.shape[-1] == 64:
            a = a[..., ::3, ::3]
        res[f'{prefix}/{k}'] = np.ascontiguousarray(a)


def put_sd(res, prefix, model):
    for k, v in model.state_dict().items():
        res[f'{prefix}/sd/{k}'] = v.detach().numpy().copy()


def disentangled(res):
    T = 6
    v1, v2 = synthetic_clips(3, T, 61), synthetic_clips(3, T, 62)
    g = torch.Generator().manual_seed(63)
    zc, zm = torch.randn(3, 16, generator=g), torch.randn(3, 16, generator=g)
    for agg in ('mean', 'middle'):
        torch.manual_seed(17)
        m = models.DisentangledVAE(**SMALL['disentangled'], content_aggregation=agg)
        randomise_bn(m, 19)
        tag = f'dis_{agg}'
        put_sd(res, tag, m)
        put(res, tag, zc=zc, zm=zm)              # v1, v2 = synthetic_clips(3, 6, 61 / 62)
        m.eval()
        put(res, tag, sample=m.sample(z_content=zc, z_motion=zm), sample_T4=m.sample(sequence_length=4, z_content=zc,
                                                                                     z_motion=zm),
            swap=m.swap_content(v1, v2), swap_motion=m.swap_motion(v1, v2),
            ic_first=m.interpolate_content(v1[:1], v2[:1], num_steps=5, use_motion_from='first'),
            ic_second=m.interpolate_content(v1[:1], v2[:1], num_steps=4, use_motion_from='second'),
            im_first=m.interpolate_motion(v1[:1], v2[:1], num_steps=5, use_content_from='first'),
            im_second=m.interpolate_motion(v1[:1], v2[:1], num_steps=3, use_content_from='second'),
            lat_c=m.get_latents(v1)[0], lat_m=m.get_latents(v1)[1])
        with torch.no_grad():
            out = m(v1)
        put(res, tag, **{'eval_' + k: v for k, v in out.items()})
        w = models.DisentangledVAEWrapper(m)
        with torch.no_grad():
            recon, z, mu, lv = w(v1)
            ez, emu, elv = w.encode(v2)
            dec = w.decode(torch.cat([zc, zm], 1), 5)
        put(res, tag, w_recon=recon, w_z=z, w_mu=mu, w_logvar=lv, w_enc_z=ez, w_enc_mu=emu, w_enc_logvar=elv, w_dec=dec)
        # the same calls on a model left in training mode: BatchNorm takes batch statistics and moves its
        # running buffers even under no_grad, reparameterize still returns mu
        m.train()
        put(res, tag, train_swap=m.swap_content(v1, v2))
        for k, v in m.state_dict().items():
            if 'running' in k or 'num_batches' in k:
                res[f'{tag}/after_train_swap/{k}'] = v.detach().numpy().copy()
        # training-mode forward with injected draws (covers 'middle' in the training path)
        m2 = models.DisentangledVAE(**SMALL['disentangled'], content_aggregation=agg)
        m2.load_state_dict({k[len(tag) + 4:]: torch.from_numpy(v) for k, v in res.items() if k.startswith(tag + '/sd/')})
        m2.train()
        eps = [torch.randn(3, 16, generator=torch.Generator().manual_seed(200 + k)) for k in range(2)]
        with EpsInjector(eps):
            out = m2(v1)
        loss = models.disentangled_loss(out, v1, beta=0.4, recon_loss_type='bce', temporal_weight=0.05,
                                        independence_weight=0.02)
        loss['total_loss'].backward()
        put(res, tag, eps0=eps[0], eps1=eps[1], train_recon=out['recon'], train_total_loss=loss['total_loss'],
            **{'grad/' + n: p.grad for n, p in m2.named_parameters() if n.startswith('content_encoder')})


def tuple_models(res):
    T = 6
    v1, v2 = synthetic_clips(3, T, 71), synthetic_clips(3, T, 72)
    z = torch.randn(3, 16, generator=torch.Generator().manual_seed(73))
    for kind in ('baseline', 'lstm'):
        torch.manual_seed(23)
        m = models.get_model(kind, in_channels=1, **SMALL[kind])
        randomise_bn(m, 29)
        put_sd(res, kind, m)
        m.eval()
        with torch.no_grad():
            dec = m.decode(z)
            dec4 = m.decode(z, sequence_length=4)
            enc = m.encode(v1)
        put(res, kind, z=z, decode=dec,              # v1, v2 = synthetic_clips(3, 6, 71 / 72)
            decode_T4=dec4, enc_z=enc[0], enc_mu=enc[1], enc_logvar=enc[2],
            interp=m.interpolate(v1[:1], v2[:1], num_steps=5), latent=m.get_latent(v1))
        if kind == 'lstm':
            put(res, kind, frame_features=m.get_frame_features(v1))
        else:
            with torch.no_grad():
                _, _, _, frame_mu, frame_lv = m.encode(v1, return_all_frames=True)
            put(res, kind, frame_mu=frame_mu, frame_logvar=frame_lv)


if __name__ == '__main__':
    res = {}
    disentangled(res)
    tuple_models(res)
    np.savez_compressed(os.path.join(HERE, 'surface.npz'), **res)
    print(len(res), 'arrays', os.path.getsize(os.path.join(HERE, 'surface.npz')) // 1024, 'KiB')
