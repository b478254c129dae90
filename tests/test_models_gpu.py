"""Model-level parity of the HIP path against the vectors produced by the reference itself
(tests/golden/step_*.npz) and against the CPU oracle at the reference's full layer sizes.

Tolerance contract (fp32 path): outputs, per-step loss and gradients within 1e-4 relative of the
reference.  The reference's own fp32 gradients deviate from an fp64 run of the same code by up to a
few 1e-4 (per-tensor relative L2), so gradients are judged against the fp64 vectors with the bar
max(1e-4, 2 x the reference's own fp32-vs-fp64 error for that tensor).
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

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


def build(kind, g, cfg=None):
    import models
    cfg = dict(SETTINGS['small'][kind]) if cfg is None else cfg
    m = models.get_model(kind, in_channels=1, **cfg)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in g['sd'].items()}, strict=True)
    return m.cuda()


def rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def as_dict(kind, out):
    return out if kind == 'disentangled' else dict(zip(('recon', 'z', 'mu', 'logvar'), out))


def loss_of(kind, out, video, tcfg, beta):
    from hipvae import losses
    if kind == 'disentangled':
        return losses.disentangled_loss(out, video, beta=beta, beta_content=tcfg['beta_content'],
                                        beta_motion=tcfg['beta_motion'], recon_loss_type=tcfg['recon_loss_type'],
                                        temporal_weight=tcfg['temporal_weight'],
                                        independence_weight=tcfg['independence_weight'])
    fn = losses.lstm_loss if kind == 'lstm' else losses.baseline_loss
    recon, z, mu, logvar = out
    return fn(recon, video, mu, logvar, beta=beta, recon_loss_type=tcfg['recon_loss_type'],
              free_bits=tcfg['free_bits'], use_free_bits=tcfg['use_free_bits'])


@pytest.mark.parametrize('kind', ['disentangled', 'lstm', 'baseline'])
def test_eval_forward_matches_reference_vectors(kind):
    g = load_step(kind)
    m = build(kind, g).eval()
    video = torch.from_numpy(g['video']).cuda()
    with torch.no_grad():
        out = m(video)
        losses = loss_of(kind, out, video, SETTINGS['tcfg'][kind], SETTINGS['beta'])
    for k, ref in g['eval'].items():
        got = as_dict(kind, out)[k].cpu().numpy()
        assert got.shape == ref.shape
        assert rel(got, ref) < 1e-4, (k, rel(got, ref))
    for k, ref in g['eval_loss'].items():
        assert abs(losses[k].item() - float(ref)) <= 1e-4 * abs(float(ref)) + 1e-6, k
    # buffers untouched in eval mode
    for k, v in m.state_dict().items():
        if 'running' in k or 'num_batches' in k:
            assert np.array_equal(v.cpu().numpy(), g['sd'][k]), k


@pytest.mark.parametrize('kind', ['disentangled', 'lstm', 'baseline'])
def test_train_step_matches_reference_vectors(kind):
    from hipvae import rng
    from hipvae.optim import FusedAdam
    g = load_step(kind)
    tcfg = SETTINGS['tcfg'][kind]
    m = build(kind, g).train()
    video = torch.from_numpy(g['video']).cuda()
    eps = [torch.from_numpy(g[f'eps{k}']) for k in range(2) if f'eps{k}' in g]
    opt = FusedAdam(m.parameters(), lr=SETTINGS['lr'], betas=(tcfg['beta1'], tcfg['beta2']),
                    weight_decay=tcfg['weight_decay'])
    opt.zero_grad()
    with rng.inject_eps(eps):
        out = m(video)
    losses = loss_of(kind, out, video, tcfg, SETTINGS['beta'])
    losses['total_loss'].backward()
    for k, ref in g['train_f32'].items():
        got = as_dict(kind, out)[k].detach().cpu().numpy()
        assert rel(got, ref) < 1e-4, (k, rel(got, ref))
    for k, ref in g['loss_f32'].items():
        assert abs(losses[k].item() - float(ref)) <= 1e-4 * abs(float(ref)) + 1e-6, (k, losses[k].item(), float(ref))
    worst = 0.0
    for name, p in m.named_parameters():
        g64, g32 = g['grad_f64'][name], g['grad_f32'][name]
        assert p.grad is not None, name
        own = rel(g32, g64)
        err = rel(p.grad.cpu().numpy(), g64)
        bar = max(1e-4, 2 * own)
        assert err <= bar, (name, err, own)
        worst = max(worst, err)
    # clip + Adam, then every parameter and buffer against the reference's post-step state
    total = opt.clip_grad_norm_(tcfg['gradient_clip'])
    assert abs(total.item() - float(g['grad_norm_f32'])) <= 2e-4 * float(g['grad_norm_f32'])
    opt.step()
    for k, v in m.state_dict().items():
        ref = g['post'][k]
        if ref.dtype.kind == 'f':
            np.testing.assert_allclose(v.cpu().numpy(), ref, rtol=2e-4, atol=5e-6, err_msg=k)
        else:
            assert np.array_equal(v.cpu().numpy(), ref), k


def _grads_of_step(kind, extra=None, fuse=True):
    """Parameter gradients of one training step on the golden inputs; `extra(recon)` adds a second consumer of the
    reconstruction to the objective."""
    from hipvae import functional as HF, rng
    g = load_step(kind)
    m = build(kind, g).train()
    video = torch.from_numpy(g['video']).cuda()
    eps = [torch.from_numpy(g[f'eps{k}']) for k in range(2) if f'eps{k}' in g]
    from hipvae import ops
    calls = []
    plain = ops.sigmoid_bwd
    old, HF.FUSE_TAIL = HF.FUSE_TAIL, fuse
    ops.sigmoid_bwd = lambda *a, **k: (calls.append(1), plain(*a, **k))[1]
    try:
        with rng.inject_eps(eps):
            out = m(video)
        total = loss_of(kind, out, video, SETTINGS['tcfg'][kind], SETTINGS['beta'])['total_loss']
        if extra is not None:
            total = total + extra(as_dict(kind, out)['recon'])
        total.backward()
    finally:
        HF.FUSE_TAIL = old
        ops.sigmoid_bwd = plain
    # the separate sigmoid-backward pass runs exactly when the one-pass path does not cover the whole gradient
    # (the baseline model repeats one decoded frame T times: its loss does not see the decoder's own tensor)
    assert len(calls) == (0 if fuse and extra is None and kind != 'baseline' else 1)
    return {n: p.grad.clone() for n, p in m.named_parameters()}


@pytest.mark.parametrize('kind', ['disentangled', 'lstm', 'baseline'])
def test_logit_form_loss_gradient_gives_the_gradients_of_the_two_pass_path(kind):
    # the loss node writes d/d(logit) straight into the decoder's hands (one pass) or d/d(recon) followed by the
    # decoder's own sigmoid backward (two passes): the same float operations, so the same bits, except for the
    # final layer's bias gradient (a sum taken in a different order)
    one, two = _grads_of_step(kind, fuse=True), _grads_of_step(kind, fuse=False)
    for n in one:
        if n.endswith('final.bias'):
            assert rel(one[n].cpu().numpy(), two[n].cpu().numpy()) < 1e-5, n
        else:
            assert torch.equal(one[n], two[n]), n


def test_a_second_consumer_of_the_reconstruction_is_added_correctly_to_the_logit_form_gradient():
    # autograd then hands the decoder (logit-form gradient of the loss) + (plain gradient of the other term)
    w = torch.linspace(-1.0, 1.0, 64, device='cuda').view(1, 1, 1, 1, 64)
    extra = lambda recon: 37.0 * (recon * w).sum() + 5.0 * (recon * recon).sum()   # noqa: E731
    one, two = _grads_of_step('disentangled', extra, fuse=True), _grads_of_step('disentangled', extra, fuse=False)
    for n in one:
        assert rel(one[n].cpu().numpy(), two[n].cpu().numpy()) < 2e-5, (n, rel(one[n].cpu().numpy(), two[n].cpu().numpy()))


def _step_for_tail(kind='disentangled'):
    from hipvae import rng
    g = load_step(kind)
    m = build(kind, g).train()
    video = torch.from_numpy(g['video']).cuda()
    eps = [torch.from_numpy(g[f'eps{k}']) for k in range(2) if f'eps{k}' in g]
    with rng.inject_eps(eps):
        out = m(video)
    return m, video, out


def _total(kind, out, video):
    return loss_of(kind, out, video, SETTINGS['tcfg'][kind], SETTINGS['beta'])['total_loss']


def test_whoever_observes_the_gradient_of_the_reconstruction_sees_the_true_one():
    """the logit-form hand-off is only taken when nobody can see or rewrite the tensor that travels as "the gradient
    of recon": autograd.grad(total, recon), retain_grad() and a tensor hook all get d(total)/d(recon), and the
    parameter gradients stay those of the two-pass path"""
    from hipvae import functional as HF
    kind = 'disentangled'
    want = _grads_of_step(kind, fuse=False)
    old, HF.FUSE_TAIL = HF.FUSE_TAIL, False
    try:
        m, video, out = _step_for_tail(kind)
        recon = as_dict(kind, out)['recon']
        (true_grad,) = torch.autograd.grad(_total(kind, out, video), recon)
    finally:
        HF.FUSE_TAIL = old
    assert HF.FUSE_TAIL
    # asked for directly: the decoder's node does not run in that backward pass
    m, video, out = _step_for_tail(kind)
    recon = as_dict(kind, out)['recon']
    total = _total(kind, out, video)
    (got,) = torch.autograd.grad(total, recon, retain_graph=True)
    assert torch.equal(got, true_grad)
    # ... and a full backward pass over the same graph afterwards is still right
    total.backward()
    for n, p in m.named_parameters():
        assert rel(p.grad.cpu().numpy(), want[n].cpu().numpy()) < 1e-5, n
    # retained
    m, video, out = _step_for_tail(kind)
    recon = as_dict(kind, out)['recon']
    recon.retain_grad()
    _total(kind, out, video).backward()
    assert torch.equal(recon.grad, true_grad)
    for n, p in m.named_parameters():
        assert rel(p.grad.cpu().numpy(), want[n].cpu().numpy()) < 1e-5, n
    # a hook that rewrites it, registered after the loss was formed: the decoder must see the rewritten TRUE gradient
    m, video, out = _step_for_tail(kind)
    recon = as_dict(kind, out)['recon']
    total = _total(kind, out, video)
    seen = []
    recon.register_hook(lambda g: (seen.append(g.clone()), 0.5 * g)[1])
    total.backward()
    assert torch.equal(seen[0], true_grad)
    half = {n: p.grad.clone() for n, p in m.named_parameters()}
    m2, video, out = _step_for_tail(kind)
    old, HF.FUSE_TAIL = HF.FUSE_TAIL, False
    try:
        recon = as_dict(kind, out)['recon']
        total = _total(kind, out, video)
        recon.register_hook(lambda g: 0.5 * g)
        total.backward()
    finally:
        HF.FUSE_TAIL = old
    for n, p in m2.named_parameters():
        assert rel(half[n].cpu().numpy(), p.grad.cpu().numpy()) < 1e-5, n


def test_two_backward_passes_over_one_graph_and_a_record_left_behind():
    from hipvae import functional as HF
    kind = 'disentangled'
    want = _grads_of_step(kind, fuse=True)
    m, video, out = _step_for_tail(kind)
    total = _total(kind, out, video)
    total.backward(retain_graph=True)
    first = {n: p.grad.clone() for n, p in m.named_parameters()}
    m.zero_grad(set_to_none=True)
    total.backward()
    for n, p in m.named_parameters():
        assert torch.equal(first[n], want[n]) and torch.equal(p.grad, want[n]), n
    # a backward pass that wrote the record and never reached the decoder (here: faked) must not leak into the next
    m, video, out = _step_for_tail(kind)
    tail = as_dict(kind, out)['recon']._dvae_tail
    total = _total(kind, out, video)
    tail.dlogit, tail.dbias, tail.task = torch.full_like(as_dict(kind, out)['recon'], 7.0), torch.ones(1, device='cuda'), -12345
    total.backward()
    for n, p in m.named_parameters():
        assert rel(p.grad.cpu().numpy(), want[n].cpu().numpy()) < 1e-5, n


def test_second_stream_changes_no_bit_of_the_gradients():
    from hipvae import functional as HF
    two = _grads_of_step('disentangled')
    old, HF.USE_SIDE_ENCODER = HF.USE_SIDE_ENCODER, False
    try:
        one = _grads_of_step('disentangled')
    finally:
        HF.USE_SIDE_ENCODER = old
    for n in one:
        assert torch.equal(one[n], two[n]), n


def test_content_encoder_runs_on_its_own_stream_with_its_own_scratch_forward_and_backward():
    # every scratch request made while the second stream is current (forward under the stream context, backward
    # wherever autograd replays it) must get the second stream's buffer, and both passes must make such requests
    from hipvae import functional as HF, ops, rng
    assert HF.USE_SIDE_ENCODER
    g = load_step('disentangled')
    m = build('disentangled', g).train()
    video = torch.from_numpy(g['video']).cuda()
    eps = [torch.from_numpy(g[f'eps{k}']) for k in range(2)]
    seen = {'fwd': [], 'bwd': []}
    phase = ['fwd']
    plain = ops.workspace

    def spy(nfloats, device, slot=None):
        buf = plain(nfloats, device, slot)
        side = HF._SIDE_STREAMS.get(torch.device(device).index)
        on_side = side is not None and torch.cuda.current_stream(device).cuda_stream == side.cuda_stream
        if slot is None:
            mine = ops._WS.get((torch.device(device).index, 'side_branch'))
            seen[phase[0]].append((on_side, mine is not None and buf.data_ptr() == mine.data_ptr()))
        return buf
    ops.workspace = spy
    try:
        with rng.inject_eps(eps):
            out = m(video)
        total = loss_of('disentangled', out, video, SETTINGS['tcfg']['disentangled'], SETTINGS['beta'])['total_loss']
        phase[0] = 'bwd'
        total.backward()
        torch.cuda.synchronize()
    finally:
        ops.workspace = plain
    for ph in ('fwd', 'bwd'):
        assert any(on_side for on_side, _ in seen[ph]), ph
        assert any(not on_side for on_side, _ in seen[ph]), ph
        assert all(on_side == got_side_buffer for on_side, got_side_buffer in seen[ph]), ph


FULL = {
    'disentangled': dict(content_dim=128, motion_dim=128, content_channels=[32, 64, 128, 256],
                         motion_channels=[16, 32], decoder_channels=[256, 128, 64, 32], content_feature_dim=512,
                         motion_feature_dim=512, lstm_hidden=256, lstm_layers=1, bidirectional=True,
                         sequence_length=20),
    'lstm': dict(latent_dim=256, encoder_channels=[32, 64, 128, 256], decoder_channels=[256, 128, 64, 32],
                 feature_dim=512, lstm_hidden=256, lstm_layers=2, bidirectional=True, sequence_length=20),
    'baseline': dict(latent_dim=128, encoder_channels=[16, 32, 64, 128], decoder_channels=[128, 64, 32, 16],
                     feature_dim=256, sequence_length=20),
}


# 128x128 frames need one more stride-2 level everywhere (five encoder / decoder levels, three motion
# levels); the widened channel lists are those of configs/disentangled_128x128x40.json
FULL['disentangled_128_narrow'] = dict(content_dim=16, motion_dim=16, content_channels=[8, 16, 32, 64, 64],
                                       motion_channels=[8, 16, 16], decoder_channels=[64, 64, 32, 16, 8],
                                       content_feature_dim=64, motion_feature_dim=64, lstm_hidden=32, lstm_layers=1,
                                       bidirectional=True, sequence_length=6)
FULL['disentangled_128_wide'] = dict(content_dim=128, motion_dim=128, content_channels=[64, 128, 256, 512, 512],
                                     motion_channels=[32, 64, 64], decoder_channels=[512, 512, 256, 128, 64],
                                     content_feature_dim=512, motion_feature_dim=512, lstm_hidden=256, lstm_layers=1,
                                     bidirectional=True, sequence_length=3)
# ... and at the configuration's own clip length: 39 / 40-step recurrences in one launch each (run by
# test_bench_size_gpu.py, whose bar knows about LeakyReLU's kink)
FULL['disentangled_128_wide40'] = dict(FULL['disentangled_128_wide'], sequence_length=40)
CASES = [('disentangled', 4, 20, 64), ('lstm', 4, 20, 64), ('baseline', 4, 20, 64),
         ('disentangled_128_narrow', 2, 6, 128), ('disentangled_128_wide', 4, 3, 128),
         # ragged: odd batch with few frames, a single clip of two frames, single-frame clips
         ('disentangled', 3, 5, 64), ('lstm', 1, 2, 64), ('baseline', 5, 1, 64)]


def test_full_size_step_in_fp32_split_mode_meets_the_fp32_bars():
    """The opt-in mode that forms the Linear layers' fp32 products from bf16 MFMA products is held to the
    same bars as the default fp32 path: whole training step of the DisentangledVAE against the fp64 oracle."""
    from hipvae import ops
    ops.set_precision('fp32_split')
    try:
        test_full_size_step_against_oracle('disentangled', 4, 20, 64)
    finally:
        ops.set_precision('fp32')


@pytest.mark.parametrize('name,B,T,size', CASES)
def test_full_size_step_against_oracle(name, B, T, size):
    """The three JSON configurations at their real layer sizes (B=4), and the 128x128 five-level
    configuration (narrow and at its full widths, short clips): HIP path vs the fp64 oracle."""
    import models
    from hipvae import rng
    kind = name.split('_')[0]
    torch.manual_seed(0)
    m = models.get_model(kind, in_channels=1, **FULL[name])
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    gen = torch.Generator().manual_seed(1)
    video = (torch.rand(B, T, 1, size, size, generator=gen) > 0.9).float() * torch.rand(B, T, 1, size, size,
                                                                                         generator=gen)
    FULLK = FULL[name]
    dim = FULLK.get('content_dim', FULLK.get('latent_dim'))
    eps = [torch.randn(B, dim, generator=torch.Generator().manual_seed(100 + k)) for k in range(2)]
    tcfg = SETTINGS['tcfg'][kind]
    layers = FULLK.get('lstm_layers', 1)
    # oracle, fp64
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    keys = O.trainable_keys(sd64)
    leaves = {k: sd64[k].clone().requires_grad_(True) for k in keys}
    work = dict(sd64)
    work.update(leaves)
    ns = {}
    oout = O.model_forward(kind, work, video.double(), [e.double() for e in eps], True, lstm_layers=layers,
                           new_stats=ns)
    oloss = O.model_loss(kind, oout, video.double(), 0.4, tcfg)
    ograds = torch.autograd.grad(oloss['total_loss'], [leaves[k] for k in keys])
    # the same restatement in fp32: how far fp32 arithmetic itself is from fp64 on this problem
    leaves32 = {k: sd[k].clone().requires_grad_(True) for k in keys}
    work32 = dict(sd)
    work32.update(leaves32)
    oout32 = O.model_forward(kind, work32, video, eps, True, lstm_layers=layers, new_stats={})
    ograds32 = torch.autograd.grad(O.model_loss(kind, oout32, video, 0.4, tcfg)['total_loss'],
                                   [leaves32[k] for k in keys])
    # HIP path
    m = m.cuda().train()
    with rng.inject_eps(eps):
        out = m(video.cuda())
    losses = loss_of(kind, out, video.cuda(), tcfg, 0.4)
    losses['total_loss'].backward()
    for k, v in as_dict(kind, out).items():
        r = rel(v.detach().cpu().numpy(), as_dict(kind, oout)[k].detach().numpy())
        assert r < 1e-4, (k, r)
    for k in ('total_loss', 'recon_loss', 'kl_loss'):
        assert abs(losses[k].item() - oloss[k].item()) <= 1e-4 * abs(oloss[k].item()) + 1e-6, k
    named = dict(m.named_parameters())
    for k, og, og32 in zip(keys, ograds, ograds32):
        r = rel(named[k].grad.cpu().numpy(), og.numpy())
        own = rel(og32.numpy(), og.numpy())
        # the contract's bar: 1e-4, or twice what fp32 arithmetic itself (the CPU restatement in fp32)
        # strays from fp64 on this tensor -- with a handful of frames per BatchNorm batch that is more
        assert r < max(1e-4, 2 * own), (k, r, own)
    for k, v in ns.items():
        got = m.state_dict()[k].cpu()
        if v.is_floating_point():
            assert rel(got.numpy(), v.numpy()) < 1e-4, k
        else:
            assert int(got) == int(v), k
