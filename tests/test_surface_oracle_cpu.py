"""The oracle's restatement of the model methods used downstream of training (sample, swap, interpolate,
get_latents, 'middle' aggregation) against vectors the reference itself produced
(tests/golden/surface.npz, written by tests/golden/make_golden_surface.py)."""
import os

import numpy as np
import torch

from oracle import vae_oracle as O

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')


def synthetic_clips(B, T, seed):
    g = torch.Generator().manual_seed(seed)
    u1 = torch.rand(B, T, 1, 64, 64, generator=g)
    u2 = torch.rand(B, T, 1, 64, 64, generator=g)
    return (u1 > 0.9).float() * u2


def load(tag):
    z = np.load(os.path.join(GOLDEN, 'surface.npz'))
    pre = tag + '/'
    flat = {k[len(pre):]: z[k] for k in z.files if k.startswith(pre)}
    sd = {k[3:]: torch.from_numpy(v) for k, v in flat.items() if k.startswith('sd/')}
    return sd, {k: v for k, v in flat.items() if not k.startswith('sd/')}


def sub(t):
    """fixtures hold every third pixel of clip-shaped outputs"""
    a = t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)
    return a[..., ::3, ::3] if (a.ndim == 5 and a.shape[-1] == 64) else a


def close(got, ref, tol=2e-5):
    got, ref = sub(got).astype(np.float64), np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    err = np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-30)
    assert err < tol, err


def test_disentangled_surface_matches_reference():
    v1, v2 = synthetic_clips(3, 6, 61), synthetic_clips(3, 6, 62)
    for agg in ('mean', 'middle'):
        sd, g = load(f'dis_{agg}')
        zc, zm = torch.from_numpy(g['zc']), torch.from_numpy(g['zm'])
        kw = dict(aggregation=agg)
        close(O.disentangled_decode(sd, zc, zm, 6, False), g['sample'])
        close(O.disentangled_decode(sd, zc, zm, 4, False), g['sample_T4'])
        close(O.swap_content(sd, v1, v2, **kw), g['swap'])
        close(O.swap_content(sd, v1, v2, **kw), g['swap_motion'])          # alias, same argument order
        close(O.interpolate_content(sd, v1[:1], v2[:1], 5, 'first', **kw), g['ic_first'])
        close(O.interpolate_content(sd, v1[:1], v2[:1], 4, 'second', **kw), g['ic_second'])
        close(O.interpolate_motion(sd, v1[:1], v2[:1], 5, 'first', **kw), g['im_first'])
        close(O.interpolate_motion(sd, v1[:1], v2[:1], 3, 'second', **kw), g['im_second'])
        mu_c, mu_m = O.disentangled_encode(sd, v1, False, **kw)
        close(mu_c, g['lat_c'])
        close(mu_m, g['lat_m'])
        out = O.disentangled_forward(sd, v1, None, None, False, aggregation=agg)
        for k, v in out.items():
            close(v, g['eval_' + k])
        # tuple-API wrapper = concatenations of the same tensors
        close(torch.cat([out['mu_content'], out['mu_motion']], 1), g['w_mu'])
        close(torch.cat([out['z_content'], out['z_motion']], 1), g['w_z'])
        close(out['recon'], g['w_recon'])
        close(O.disentangled_decode(sd, zc, zm, 5, False), g['w_dec'])
        # a model left in training mode: batch statistics, buffers move on all three passes
        ns = {}
        close(O.swap_content(sd, v1, v2, training=True, new_stats=ns, **kw), g['train_swap'])
        for k, v in ns.items():
            np.testing.assert_allclose(v.numpy(), g['after_train_swap/' + k], rtol=2e-5, atol=1e-6, err_msg=k)
        # training forward with injected draws ('middle' inside the training path), content-encoder gradients
        leaves = {k: sd[k].clone().requires_grad_(True) for k in O.trainable_keys(sd)}
        work = dict(sd)
        work.update(leaves)
        out = O.disentangled_forward(work, v1, torch.from_numpy(g['eps0']), torch.from_numpy(g['eps1']), True,
                                     aggregation=agg, new_stats={})
        close(out['recon'], g['train_recon'])
        loss = O.disentangled_loss(out, v1, beta=0.4, recon_loss_type='bce', temporal_weight=0.05,
                                   independence_weight=0.02)
        assert abs(float(loss['total_loss']) - float(g['train_total_loss'])) <= 1e-5 * abs(float(g['train_total_loss']))
        loss['total_loss'].backward()
        for k in leaves:
            if k.startswith('content_encoder'):
                close(leaves[k].grad, g['grad/' + k], tol=2e-4)


def test_tuple_model_surface_matches_reference():
    v1, v2 = synthetic_clips(3, 6, 71), synthetic_clips(3, 6, 72)
    for kind, layers in (('baseline', 1), ('lstm', 2)):
        sd, g = load(kind)
        z = torch.from_numpy(g['z'])
        dec = (lambda zz, T: O.baseline_decode(sd, zz, T, False)) if kind == 'baseline' else \
            (lambda zz, T: O.lstm_vae_decode(sd, zz, T, False, layers))
        close(dec(z, 6), g['decode'])
        close(dec(z, 4), g['decode_T4'])
        fwd = O.baseline_forward(sd, v1, None, False) if kind == 'baseline' else \
            O.lstm_vae_forward(sd, v1, None, False, layers)
        close(fwd[1], g['enc_z'])
        close(fwd[2], g['enc_mu'])
        close(fwd[3], g['enc_logvar'])
        close(fwd[2], g['latent'])
        close(O.tuple_interpolate(kind, sd, v1[:1], v2[:1], 5, layers), g['interp'])
        feats = O.cnn_encoder(v1.reshape(18, 1, 64, 64), sd, 'cnn_encoder', False, None).reshape(3, 6, -1)
        if kind == 'lstm':
            close(feats, g['frame_features'])
        else:
            fmu, flv = O.vae_head(feats, sd, 'vae_head')
            close(fmu, g['frame_mu'])
            close(flv, g['frame_logvar'])
