"""The model methods used downstream of training (evaluate.py, the notebooks, evaluation/metrics.py) on the
HIP path against vectors the reference itself produced (tests/golden/surface.npz): sample, swap_content /
swap_motion, interpolate_content / interpolate_motion, get_latents, the tuple-API wrapper,
content_aggregation='middle', and the baseline / LSTM models' decode, interpolate, get_latent,
get_frame_features; then the timing / diversity helpers of the evaluation package.
Tolerance: 1e-4 relative L2 (the fp32 contract)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from test_surface_oracle_cpu import load, sub, synthetic_clips


def close(got, ref, tol=1e-4):
    got, ref = sub(got).astype(np.float64), np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    err = np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-30)
    assert err < tol, err


def _small(kind):
    import json
    import os
    settings = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'settings.json')))
    return dict(settings['small'][kind])


@pytest.mark.parametrize('agg', ['mean', 'middle'])
def test_disentangled_methods_match_reference(agg):
    import models
    from hipvae import losses, rng
    sd, g = load(f'dis_{agg}')
    m = models.DisentangledVAE(**_small('disentangled'), content_aggregation=agg)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    v1, v2 = synthetic_clips(3, 6, 61).cuda(), synthetic_clips(3, 6, 62).cuda()
    zc, zm = torch.from_numpy(g['zc']).cuda(), torch.from_numpy(g['zm']).cuda()
    close(m.sample(z_content=zc, z_motion=zm), g['sample'])
    close(m.sample(sequence_length=4, z_content=zc, z_motion=zm), g['sample_T4'])
    assert m.sample(num_samples=2).shape == (2, 6, 1, 64, 64)
    close(m.swap_content(v1, v2), g['swap'])
    close(m.swap_motion(v1, v2), g['swap_motion'])
    close(m.interpolate_content(v1[:1], v2[:1], num_steps=5, use_motion_from='first'), g['ic_first'])
    close(m.interpolate_content(v1[:1], v2[:1], num_steps=4, use_motion_from='second'), g['ic_second'])
    close(m.interpolate_motion(v1[:1], v2[:1], num_steps=5, use_content_from='first'), g['im_first'])
    close(m.interpolate_motion(v1[:1], v2[:1], num_steps=3, use_content_from='second'), g['im_second'])
    lat_c, lat_m = m.get_latents(v1)
    close(lat_c, g['lat_c'])
    close(lat_m, g['lat_m'])
    assert not lat_c.requires_grad and not lat_m.requires_grad
    with torch.no_grad():
        out = m(v1)
    for k, v in out.items():
        close(v, g['eval_' + k])
    w = models.DisentangledVAEWrapper(m)
    assert (w.latent_dim, w.content_dim, w.motion_dim, w.model_type) == (32, 16, 16, 'disentangled_vae')
    with torch.no_grad():
        recon, z, mu, lv = w(v1)
        ez, emu, elv = w.encode(v2)
        dec = w.decode(torch.cat([zc, zm], 1), 5)
    for got, key in ((recon, 'w_recon'), (z, 'w_z'), (mu, 'w_mu'), (lv, 'w_logvar'), (ez, 'w_enc_z'), (emu, 'w_enc_mu'),
                     (elv, 'w_enc_logvar'), (dec, 'w_dec')):
        close(got, g[key])
    assert w.sample(num_samples=2, sequence_length=3).shape == (2, 3, 1, 64, 64)
    # buffers untouched so far (eval mode) ...
    for k, v in m.state_dict().items():
        if 'running' in k or 'num_batches' in k:
            assert np.array_equal(v.cpu().numpy(), sd[k].numpy()), k
    # ... and moved by the same call on a model left in training mode (batch statistics under no_grad)
    m.train()
    close(m.swap_content(v1, v2), g['train_swap'])
    for k, v in m.state_dict().items():
        if 'running' in k:
            np.testing.assert_allclose(v.cpu().numpy(), g['after_train_swap/' + k], rtol=1e-4, atol=1e-6, err_msg=k)
        elif 'num_batches' in k:
            assert int(v) == int(g['after_train_swap/' + k]), k
    # training forward + backward with injected draws: 'middle' aggregation inside the training path
    m2 = models.DisentangledVAE(**_small('disentangled'), content_aggregation=agg)
    m2.load_state_dict(sd, strict=True)
    m2 = m2.cuda().train()
    with rng.inject_eps([torch.from_numpy(g['eps0']), torch.from_numpy(g['eps1'])]):
        out = m2(v1)
    close(out['recon'], g['train_recon'])
    loss = losses.disentangled_loss(out, v1, beta=0.4, recon_loss_type='bce', temporal_weight=0.05,
                                    independence_weight=0.02)
    ref = float(g['train_total_loss'])
    assert abs(loss['total_loss'].item() - ref) <= 1e-4 * abs(ref)
    loss['total_loss'].backward()
    for n, p in m2.named_parameters():
        if n.startswith('content_encoder'):
            close(p.grad, g['grad/' + n], tol=5e-4)          # fp32 reference gradients (their own error ~1e-4)


def test_unknown_aggregation_is_rejected_like_the_reference():
    import models
    m = models.DisentangledVAE(**_small('disentangled'), content_aggregation='max').cuda().eval()
    with pytest.raises(ValueError, match='Unknown aggregation'):
        with torch.no_grad():
            m(synthetic_clips(2, 6, 1).cuda())


@pytest.mark.parametrize('kind', ['baseline', 'lstm'])
def test_tuple_model_methods_match_reference(kind):
    import models
    sd, g = load(kind)
    m = models.get_model(kind, in_channels=1, **_small(kind))
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    v1, v2 = synthetic_clips(3, 6, 71).cuda(), synthetic_clips(3, 6, 72).cuda()
    z = torch.from_numpy(g['z']).cuda()
    with torch.no_grad():
        close(m.decode(z), g['decode'])
        close(m.decode(z, sequence_length=4), g['decode_T4'])
        ez, emu, elv = m.encode(v1)
    close(ez, g['enc_z'])
    close(emu, g['enc_mu'])
    close(elv, g['enc_logvar'])
    close(m.get_latent(v1), g['latent'])
    close(m.interpolate(v1[:1], v2[:1], num_steps=5), g['interp'])
    assert m.sample(num_samples=2, sequence_length=3).shape == (2, 3, 1, 64, 64)
    if kind == 'lstm':
        close(m.get_frame_features(v1), g['frame_features'])
    else:
        with torch.no_grad():
            _, _, _, fmu, flv = m.encode(v1, return_all_frames=True)
        close(fmu, g['frame_mu'])
        close(flv, g['frame_logvar'])


def test_efficiency_and_diversity_helpers():
    """evaluation.metrics.computational_efficiency / generation_diversity / compare_models: the reference's
    keys, parameter counts of the reference's models, positive finite times; diversity statistics equal a
    direct computation on the same prior draws."""
    import models
    from evaluation import metrics as M
    m = models.DisentangledVAE(**_small('disentangled')).cuda()
    video = synthetic_clips(4, 6, 5).cuda()
    eff = M.computational_efficiency(m, video, num_runs=5)
    assert sorted(eff) == sorted(['total_parameters', 'trainable_parameters', 'encode_time_ms', 'encode_time_std_ms',
                                  'forward_time_ms', 'forward_time_std_ms', 'generation_time_ms',
                                  'generation_time_std_ms'])
    assert eff['total_parameters'] == eff['trainable_parameters'] == sum(p.numel() for p in m.parameters())
    for k in ('encode_time_ms', 'forward_time_ms', 'generation_time_ms'):
        assert np.isfinite(eff[k]) and eff[k] > 0
    assert eff['encode_time_ms'] < eff['forward_time_ms'] * 1.5
    assert not m.training
    torch.manual_seed(3)
    div = M.generation_diversity(m, num_samples=6)
    torch.manual_seed(3)
    s = m.sample(num_samples=6).double().cpu()
    flat = s.reshape(6, -1)
    d = torch.cdist(flat, flat)
    ref = [s.var(dim=0).mean().item(), d[torch.triu(torch.ones(6, 6), diagonal=1).bool()].mean().item(),
           s.var(dim=1).mean().item()]
    for k, r in zip(('sample_variance', 'mean_pairwise_distance', 'temporal_variance'), ref):
        assert abs(div[k] - r) <= 1e-4 * abs(r) + 1e-9, (k, div[k], r)
    loader = [(synthetic_clips(4, 6, 8), [{}] * 4), (synthetic_clips(4, 6, 9), [{}] * 4)]
    res = M.compare_models({'dis': m, 'base': models.get_model('baseline', in_channels=1, **_small('baseline'))},
                           loader, torch.device('cuda'), max_batches=2)
    assert sorted(res) == ['base', 'dis']
    for r in res.values():
        assert {'mse', 'psnr', 'temporal_consistency', 'sample_variance', 'forward_time_ms',
                'trainable_parameters'} <= set(r)


@pytest.mark.parametrize('H', [16, 128])                       # per-step kernels / one launch per recurrence
@pytest.mark.parametrize('layers,bidirectional', [(1, True), (2, True), (2, False)])
def test_lstm_encoder_returns_what_nn_lstm_returns(H, layers, bidirectional):
    """LSTMEncoder.forward(x, return_sequence) of the reference (models/encoders.py:156-173): the output sequence or
    the final state, and (h_n, c_n) beside either; gradients from the sequence, the final state and h_n.  Against
    torch's own nn.LSTM in fp64 on the CPU with the same weights."""
    import copy
    from models.encoders import LSTMEncoder
    torch.manual_seed(3)
    B, T, I = 3, 5, 12
    enc = LSTMEncoder(input_dim=I, hidden_dim=H, num_layers=layers, bidirectional=bidirectional).cuda()
    ref = copy.deepcopy(enc.lstm).double().cpu()
    x = torch.randn(B, T, I)
    xg, xr = x.cuda().requires_grad_(True), x.double().requires_grad_(True)
    D = 2 if bidirectional else 1
    w_seq, w_fin, w_h = torch.randn(B, T, D * H), torch.randn(B, D * H), torch.randn(layers * D, B, H)

    out, (h_n, c_n) = enc(xg, return_sequence=True)
    fin, (h_n2, c_n2) = enc(xg)
    r_out, (r_h, r_c) = ref(xr)
    r_fin = torch.cat([r_h[-2], r_h[-1]], dim=1) if bidirectional else r_h[-1]
    for got, want in ((out, r_out), (h_n, r_h), (c_n, r_c), (fin, r_fin), (h_n2, r_h), (c_n2, r_c)):
        assert got.shape == want.shape
        torch.testing.assert_close(got.cpu().double(), want.detach(), rtol=2e-5, atol=2e-6)
    ((out * w_seq.cuda()).sum() + (fin * w_fin.cuda()).sum() + (h_n * w_h.cuda()).sum()).backward()
    ((r_out * w_seq.double()).sum() + (r_fin * w_fin.double()).sum() + (r_h * w_h.double()).sum()).backward()
    torch.testing.assert_close(xg.grad.cpu().double(), xr.grad, rtol=1e-4, atol=1e-5)
    for (n, p), (_, q) in zip(enc.lstm.named_parameters(), ref.named_parameters()):
        torch.testing.assert_close(p.grad.cpu().double(), q.grad, rtol=1e-4, atol=1e-5, msg=n)
    # the economy switch the models use: states not gathered
    enc.return_states = False
    fin3, states = enc(xg)
    assert states == (None, None) and torch.equal(fin3, fin)


def test_recurrence_under_the_several_ranks_limit_is_one_launch_per_direction(monkeypatch):
    """With several ranks the one-launch recurrence kernels may hold one workgroup per compute unit (the collectives'
    kernels need room), which the two directions of the motion encoder at batch 256 exceed together: they then run
    as one launch each instead of falling back to one launch per step.  Same bits either way."""
    from hipvae import functional as HF, ops
    from models.encoders import LSTMEncoder
    torch.manual_seed(5)
    B, T, I, H = 256, 19, 64, 256
    enc = LSTMEncoder(input_dim=I, hidden_dim=H, num_layers=1, bidirectional=True).cuda()
    x = torch.randn(B, T, I, device='cuda')
    w = torch.randn(B, 2 * H, device='cuda')

    def run():
        calls = []
        plain_f, plain_b = ops.lstm_seq_fwd, ops.lstm_seq_bwd
        monkeypatch.setattr(ops, 'lstm_seq_fwd', lambda chains, *a: (calls.append(('f', len(chains))), plain_f(chains, *a))[1])
        monkeypatch.setattr(ops, 'lstm_seq_bwd', lambda chains, *a: (calls.append(('b', len(chains))), plain_b(chains, *a))[1])
        enc.zero_grad(set_to_none=True)
        xg = x.clone().requires_grad_(True)
        fin, _ = enc(xg)
        (fin * w).sum().backward()
        monkeypatch.setattr(ops, 'lstm_seq_fwd', plain_f)
        monkeypatch.setattr(ops, 'lstm_seq_bwd', plain_b)
        return calls, fin.detach().clone(), xg.grad.clone(), [p.grad.clone() for p in enc.parameters()]

    assert HF._seq_groups(2, T, B, H) == [[0, 1]]
    calls1, fin1, dx1, g1 = run()
    assert calls1 == [('f', 2), ('b', 2)]
    monkeypatch.setattr(HF, '_several_ranks', lambda: True)
    assert HF._seq_groups(2, T, B, H) == [[0], [1]]
    calls2, fin2, dx2, g2 = run()
    assert calls2 == [('f', 1), ('f', 1), ('b', 1), ('b', 1)]
    assert torch.equal(fin1, fin2) and torch.equal(dx1, dx2) and all(torch.equal(a, b) for a, b in zip(g1, g2))
    ops.check_lstm_seq_timeouts()
