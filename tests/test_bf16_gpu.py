"""The bf16-MFMA variant of the GEMM / convolution kernels (hipvae.set_precision('bf16')).

Contract: tensors stay fp32 in memory; the two operands of every product are rounded to bf16
(round-to-nearest-even) when staged into LDS, products are exact and accumulate in fp32.
  * operands that are already bf16-representable give the fp32 path's result (rel. L2 <= 2e-6 vs
    fp64) -- this pins indexing, layouts and rounding mode;
  * arbitrary fp32 operands: rel. L2 error of a kernel output <= 6e-3 (bf16 has 8 significand bits,
    2^-9 relative rounding error per operand);
  * a whole training step of the shrunken models (B=4, random init): loss within 1e-3 relative of
    the fp32 reference, reconstructions and latents within 1e-2 rel. L2, per-tensor gradients within
    1.5e-1 rel. L2 of the fp64 gradients (measured on MI355X: median 6e-2, worst 1.3e-1 -- every
    activation and every back-propagated gradient is rounded to 8 bits on its way into each layer).
"""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import vae_oracle as O

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')


@pytest.fixture()
def ops():
    from hipvae import ops as _ops
    assert torch.cuda.is_available()
    _ops.set_precision('bf16')
    assert _ops.get_precision() == 'bf16'
    yield _ops
    _ops.set_precision('fp32')


def dev(t):
    return t.to('cuda', torch.float32).contiguous()


def rel_err(got, ref):
    got = got.detach().double().cpu()
    ref = ref.detach().double().cpu()
    return float((got - ref).norm() / max(float(ref.norm()), 1e-30))


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float64) * scale


def q(t):
    """nearest bf16-representable values, as float64"""
    return t.float().bfloat16().double()


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


EXACT, LOOSE = 2e-6, 6e-3
GRAD_TOL = 1.5e-1
GEMM_SHAPES = [(32, 128, 512), (4, 32, 2048), (24, 64, 32), (608, 512, 2048), (130, 260, 36), (256, 128, 4096),
               (1, 4, 4), (640, 1024, 640)]


def test_unknown_precision_rejected():
    from hipvae import ops as _ops
    with pytest.raises(ValueError):
        _ops.set_precision('fp8')
    assert _ops.get_precision() == 'fp32'


@pytest.mark.parametrize('M,N,K', GEMM_SHAPES)
def test_gemm_bf16(ops, M, N, K):
    a, b, bias = rnd(M, K, seed=10), rnd(N, K, seed=11), rnd(N, seed=12)
    got = ops.gemm_nt(dev(q(a)), dev(q(b)), dev(bias), ops.ACT_LRELU)
    assert rel_err(got, O.leaky_relu(q(a) @ q(b).t() + bias)) < EXACT
    got = ops.gemm_nt(dev(a), dev(b), None, ops.ACT_NONE)
    assert rel_err(got, q(a) @ q(b).t()) < EXACT            # rounding mode: nearest even, like torch's
    assert rel_err(got, a @ b.t()) < LOOSE
    bn = rnd(K, N, seed=14)
    assert rel_err(ops.gemm_nn(dev(q(a)), dev(q(bn))), q(a) @ q(bn)) < EXACT
    assert rel_err(ops.gemm_nn(dev(a), dev(bn)), a @ bn) < LOOSE
    if M % 4 == 0:
        at = rnd(K, M, seed=16)
        assert rel_err(ops.gemm_tn(dev(q(at)), dev(q(bn))), q(at).t() @ q(bn)) < EXACT
        base = rnd(M, N, seed=15)
        out = dev(base)
        ops.gemm_tn(dev(at), dev(bn), out=out, accumulate=True)
        assert rel_err(out, base + at.t() @ bn) < LOOSE


CONV_CASES = [(5, 32, 16, 32), (3, 16, 32, 64), (4, 8, 64, 128), (9, 4, 8, 4), (2, 8, 128, 256), (130, 4, 4, 8),
              (3, 64, 1, 16)]


@pytest.mark.parametrize('N,H,Cin,Cout', CONV_CASES)
def test_conv_bf16(ops, N, H, Cin, Cout):
    x, w = rnd(N, Cin, H, H, seed=20), rnd(Cout, Cin, 4, 4, seed=21, scale=0.2)
    wp = ops.pack_conv_w(dev(q(w)))
    y, stats, _ = ops.conv4s2_fwd(dev(nhwc(q(x))), None, None, ops.ACT_NONE, wp, True)
    ref = F.conv2d(q(x), q(w), None, stride=2, padding=1)
    assert rel_err(nchw(y), ref) < EXACT
    assert rel_err(stats.double().sum(0).cpu()[0], ref.sum(dim=(0, 2, 3))) < 1e-5
    if Cin > 1:
        sc, sh = 1 + 0.3 * rnd(Cin, seed=22), 0.2 * rnd(Cin, seed=23)
        xin = O.leaky_relu(x * sc[None, :, None, None] + sh[None, :, None, None])
        y, _, _ = ops.conv4s2_fwd(dev(nhwc(x)), dev(sc), dev(sh), ops.ACT_LRELU, ops.pack_conv_w(dev(w)), False)
        assert rel_err(nchw(y), F.conv2d(xin, w, None, stride=2, padding=1)) < LOOSE


DECONV_CASES = [(3, 4, 64, 32), (2, 8, 32, 16), (3, 16, 16, 8), (2, 32, 8, 4), (6, 4, 256, 128), (70, 4, 4, 36)]


@pytest.mark.parametrize('N,H,Cin,Cout', DECONV_CASES)
def test_deconv_bf16(ops, N, H, Cin, Cout):
    x, w = rnd(N, Cin, H, H, seed=24), rnd(Cin, Cout, 4, 4, seed=25, scale=0.2)
    y, stats, _ = ops.deconv4s2_fwd(dev(nhwc(q(x))), None, None, ops.ACT_NONE, ops.pack_deconv_w(dev(q(w))), True)
    ref = F.conv_transpose2d(q(x), q(w), None, stride=2, padding=1)
    assert rel_err(nchw(y), ref) < EXACT
    assert rel_err(stats.double().sum(0).cpu()[1], (ref ** 2).sum(dim=(0, 2, 3))) < 1e-5
    sc, sh = 1 + 0.3 * rnd(Cin, seed=26), 0.2 * rnd(Cin, seed=27)
    xin = O.leaky_relu(x * sc[None, :, None, None] + sh[None, :, None, None])
    y, _, _ = ops.deconv4s2_fwd(dev(nhwc(x)), dev(sc), dev(sh), ops.ACT_LRELU, ops.pack_deconv_w(dev(w)), False)
    assert rel_err(nchw(y), F.conv_transpose2d(xin, w, None, stride=2, padding=1)) < LOOSE


WGRAD_CASES = [(4, 16, 32, 16), (3, 8, 64, 32), (2, 4, 128, 64), (2, 4, 8, 4), (40, 16, 16, 8), (3, 2, 256, 128),
               (3, 32, 16, 1)]


@pytest.mark.parametrize('N,Hs,C1,C2', WGRAD_CASES)
def test_wgrad_bf16(ops, N, Hs, C1, C2):
    x, dy = rnd(N, C2, 2 * Hs, 2 * Hs, seed=31), rnd(N, C1, Hs, Hs, seed=32)

    def ref_dw(xx, dd):
        w = torch.zeros(C1, C2, 4, 4, dtype=torch.float64, requires_grad=True)
        (dw,) = torch.autograd.grad(F.conv2d(xx, w, None, stride=2, padding=1), w, dd)
        return dw

    got = ops.conv4s2_wgrad(dev(nhwc(q(dy))), None, None, ops.ACT_NONE, dev(nhwc(q(x))), None, None, ops.ACT_NONE)
    assert rel_err(got, ref_dw(q(x), q(dy))) < 5e-6
    if C2 > 1:
        sc, sh = 1 + 0.3 * rnd(C2, seed=33), 0.2 * rnd(C2, seed=34)
        xin = O.leaky_relu(x * sc[None, :, None, None] + sh[None, :, None, None])
        got = ops.conv4s2_wgrad(dev(nhwc(dy)), None, None, ops.ACT_NONE, dev(nhwc(x)), dev(sc), dev(sh),
                                ops.ACT_LRELU)
        assert rel_err(got, ref_dw(xin, dy)) < LOOSE


@pytest.mark.parametrize('kind', ['disentangled', 'lstm', 'baseline'])
def test_train_step_bf16_against_reference_vectors(ops, kind):
    """One training step of the shrunken models in bf16-MFMA mode against the reference's fp32/fp64
    vectors, at the looser tolerance stated in the module docstring."""
    from test_models_gpu import SETTINGS, as_dict, build, load_step, loss_of, rel
    from hipvae import rng
    g = load_step(kind)
    tcfg = SETTINGS['tcfg'][kind]
    m = build(kind, g).train()
    video = torch.from_numpy(g['video']).cuda()
    eps = [torch.from_numpy(g[f'eps{k}']) for k in range(2) if f'eps{k}' in g]
    with rng.inject_eps(eps):
        out = m(video)
    losses = loss_of(kind, out, video, tcfg, SETTINGS['beta'])
    losses['total_loss'].backward()
    for k, ref in g['train_f32'].items():
        r = rel(as_dict(kind, out)[k].detach().cpu().numpy(), ref)
        assert r < 1e-2, (k, r)
    ref = float(g['loss_f32']['total_loss'])
    assert abs(losses['total_loss'].item() - ref) <= 1e-3 * abs(ref), (losses['total_loss'].item(), ref)
    errs = {name: rel(p.grad.cpu().numpy(), g['grad_f64'][name]) for name, p in m.named_parameters()}
    worst = max(errs, key=errs.get)
    print(f'bf16 {kind}: gradient rel-L2 error median {np.median(list(errs.values())):.3e}, '
          f'worst {errs[worst]:.3e} ({worst})')
    assert errs[worst] < GRAD_TOL, (worst, errs[worst])
