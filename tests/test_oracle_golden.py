"""Pins the CPU oracle (oracle/vae_oracle.py) to vectors produced by the reference itself."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import vae_oracle as O

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')
SETTINGS = json.load(open(os.path.join(GOLDEN, 'settings.json')))


def load_step(kind):
    z = np.load(os.path.join(GOLDEN, f'step_{kind}.npz'))
    groups = {}
    for k in z.files:
        if '/' in k:
            g, name = k.split('/', 1)
            groups.setdefault(g, {})[name] = z[k]
        else:
            groups[k] = z[k]
    return groups


def sd_of(g, dtype):
    sd = {}
    for k, v in g['sd'].items():
        t = torch.from_numpy(v)
        sd[k] = t.to(dtype) if t.is_floating_point() else t
    return sd


def eps_of(g, dtype):
    return [torch.from_numpy(g[f'eps{k}']).to(dtype) for k in range(2) if f'eps{k}' in g]


def fwd(kind, sd, video, eps, training, new_stats=None):
    layers = SETTINGS['small'][kind].get('lstm_layers', 1)
    return O.model_forward(kind, sd, video, eps, training, lstm_layers=layers, new_stats=new_stats)


def as_dict(kind, out):
    if kind == 'disentangled':
        return out
    return dict(zip(('recon', 'z', 'mu', 'logvar'), out))


@pytest.mark.parametrize('kind', ['disentangled', 'lstm', 'baseline'])
def test_eval_forward_matches_reference(kind):
    g = load_step(kind)
    sd = sd_of(g, torch.float32)
    video = torch.from_numpy(g['video'])
    with torch.no_grad():
        out = fwd(kind, sd, video, [None, None], False)
        losses = O.model_loss(kind, out, video, SETTINGS['beta'], SETTINGS['tcfg'][kind])
    for k, ref in g['eval'].items():
        np.testing.assert_allclose(as_dict(kind, out)[k].numpy(), ref, rtol=2e-5, atol=2e-6, err_msg=k)
    for k, ref in g['eval_loss'].items():
        np.testing.assert_allclose(losses[k].numpy(), ref, rtol=2e-5, atol=1e-6, err_msg=k)


@pytest.mark.parametrize('kind', ['disentangled', 'lstm', 'baseline'])
@pytest.mark.parametrize('tag', ['f32', 'f64'])
def test_train_step_matches_reference(kind, tag):
    g = load_step(kind)
    dtype = torch.float32 if tag == 'f32' else torch.float64
    sd = sd_of(g, dtype)
    video = torch.from_numpy(g['video']).to(dtype)
    eps = eps_of(g, dtype)
    tcfg = SETTINGS['tcfg'][kind]
    keys = O.trainable_keys(sd)
    leaves = {k: sd[k].clone().requires_grad_(True) for k in keys}
    work = dict(sd)
    work.update(leaves)
    ns = {}
    out = fwd(kind, work, video, eps + [None], True, ns)
    losses = O.model_loss(kind, out, video, SETTINGS['beta'], tcfg)
    grads = torch.autograd.grad(losses['total_loss'], [leaves[k] for k in keys])
    # fp64 golden values were stored rounded to fp32
    rtol = 3e-4 if tag == 'f32' else 2e-6
    for k, ref in g[f'train_{tag}'].items():
        np.testing.assert_allclose(as_dict(kind, out)[k].detach().float().numpy(), ref, rtol=rtol,
                                   atol=1e-6, err_msg=k)
    for k, ref in g[f'loss_{tag}'].items():
        np.testing.assert_allclose(float(losses[k]), float(ref), rtol=1e-5 if tag == 'f32' else 1e-10,
                                   atol=1e-7, err_msg=k)
    for k, gr in zip(keys, grads):
        ref = g[f'grad_{tag}'][k]
        err = np.linalg.norm(gr.float().numpy() - ref) / max(np.linalg.norm(ref), 1e-12)
        assert err < (5e-4 if tag == 'f32' else 1e-6), (k, err)
    if tag == 'f32':
        new_sd, new_opt, _, _ = O.train_step(kind, sd, {}, video, eps + [None], SETTINGS['beta'], tcfg,
                                             SETTINGS['lr'],
                                             lstm_layers=SETTINGS['small'][kind].get('lstm_layers', 1))
        for k, ref in g['post'].items():
            got = new_sd[k].numpy()
            if ref.dtype.kind == 'f':
                np.testing.assert_allclose(got, ref, rtol=1e-4, atol=3e-6, err_msg=k)
            else:
                assert np.array_equal(got, ref), k


def test_schedules_match_reference():
    sched = json.load(open(os.path.join(GOLDEN, 'schedules.json')))
    cfg_dir = os.path.join(os.path.dirname(GOLDEN), 'golden')
    for name, rec in sched.items():
        if name.startswith('synthetic_'):
            c = rec['cfg']
            got = [O.beta_schedule(e, c['beta_vae'], c['kl_start_epoch'], c['use_kl_annealing'],
                                   c['warmup_epochs'], c['annealing_type']) for e in range(len(rec['beta']))]
            np.testing.assert_allclose(got, rec['beta'], rtol=1e-12, atol=1e-15)
    # the JSON configs: values pinned from a run of the reference itself
    d = sched['disentangled_simple.json']
    got = [O.beta_schedule(e, 0.4, 10, True, 40, 'linear') for e in range(100)]
    np.testing.assert_allclose(got, d['beta'], rtol=1e-12, atol=1e-15)
    lr = [O.cosine_lr(e + 1, 3e-4, 100, 1e-6) for e in range(100)]
    # the reference stops stepping the scheduler once lr <= lr_min (training/trainer.py:373-375)
    np.testing.assert_allclose(lr[:99], d['lr'][:99], rtol=1e-6)


def test_generator_matches_reference():
    from tests.golden.make_golden import synthetic_digit_bank
    import hashlib
    info = json.load(open(os.path.join(GOLDEN, 'generator.json')))
    vids = np.load(os.path.join(GOLDEN, 'generator_videos.npz'))
    bank = synthetic_digit_bank()
    assert hashlib.sha1(bank.tobytes()).hexdigest() == info['bank_sha1']
    for item in info['items']:
        video, idx, pos, vel = O.moving_mnist_item(item['seed'], item['index'], bank, **item['kwargs'])
        assert list(video.shape) == item['shape'] and video.dtype == np.float32
        assert hashlib.sha1(video.tobytes()).hexdigest() == item['sha1']
        ref = vids[f"video_s{item['seed']}_i{item['index']}"].astype(np.float32) / 255.0
        assert np.array_equal(video, ref)
        assert [i % 10 for i in idx] == item['labels']
        np.testing.assert_array_equal(pos[0], np.array(item['initial_positions']))
        np.testing.assert_array_equal(vel[-1], np.array(item['last_velocities']))
        np.testing.assert_array_equal(pos[-1], np.array(item['last_positions']))


METRIC_CASES = ['small', 'one_frame', 'clip', 'identical', 'half_range']


def _close_or_same_special(got, want, rtol):
    if np.isnan(want):
        return np.isnan(got)
    if np.isinf(want):
        return got == want
    return abs(got - want) <= rtol * max(abs(want), 1e-30)


@pytest.mark.parametrize('case', METRIC_CASES)
def test_recon_metrics_match_reference(case):
    """mse / psnr / temporal consistency against the values the reference's functions returned
    (tests/golden/make_golden_metrics.py), the special values included (inf at zero error, NaN for a
    single frame)."""
    z = np.load(os.path.join(GOLDEN, "metrics.npz"))
    recon, target = torch.from_numpy(z[f'{case}/recon']), torch.from_numpy(z[f'{case}/target'])
    got = O.recon_metrics(recon, target, max_val=0.5 if case == 'half_range' else 1.0)
    for g, w in zip(got, z[f'{case}/expect']):
        assert _close_or_same_special(g, float(w), 1e-12), (case, got, z[f'{case}/expect'])
