"""Per-kernel parity of the HIP library (through its C ABI) against fp64 CPU restatements.

Tolerances: fp32 kernels vs an fp64 reference of the same operation; relative L2 error <= 2e-5 for
GEMM-shaped kernels (K up to a few thousand), element tolerances written per test.
"""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import vae_oracle as O


@pytest.fixture(scope='module')
def ops():
    from hipvae import ops as _ops
    assert torch.cuda.is_available()
    return _ops


def dev(t):
    return t.to('cuda', torch.float32).contiguous()


def rel_err(got, ref):
    got = got.detach().double().cpu()
    ref = ref.detach().double().cpu()
    return float((got - ref).norm() / max(float(ref.norm()), 1e-30))


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float64) * scale


# ------------------------------------------------------------------ split / losses
@pytest.mark.parametrize('B,T', [(3, 6), (2, 20), (1, 1), (5, 2)])
def test_split_mean_diff(ops, B, T):
    v = torch.rand(B, T, 1, 64, 64, generator=torch.Generator().manual_seed(1))
    mean, diff = ops.split_mean_diff(dev(v))
    rm, rd = O.split_mean_diff(v)
    assert mean.shape == rm.shape and diff.shape == rd.shape
    torch.testing.assert_close(mean.cpu(), rm, rtol=1e-6, atol=1e-7)
    assert torch.equal(diff.cpu(), rd)          # differences of fp32 values are exact in both


@pytest.mark.parametrize('kind', ['mse', 'bce'])
@pytest.mark.parametrize('tw', [0.0, 0.05])
def test_recon_loss(ops, kind, tw):
    B, T = 3, 5
    g = torch.Generator().manual_seed(2)
    target = (torch.rand(B, T, 1, 64, 64, generator=g) > 0.9).float() * torch.rand(B, T, 1, 64, 64, generator=g)
    recon = torch.sigmoid(torch.randn(B, T, 1, 64, 64, generator=g) * 3)
    recon[0, 0, 0, 0, :4] = torch.tensor([0.0, 1.0, 1e-30, 1 - 1e-7])   # exercises the -100 clamp
    r64 = recon.double().requires_grad_(True)
    t64 = target.double()
    rl = O.recon_term(r64, t64, kind)
    tl = O.temporal_term(r64, t64)
    out = torch.zeros(4, device='cuda')
    ops.recon_loss_fwd(dev(recon), dev(target), 0 if kind == 'mse' else 1, tw > 0, out)
    o = out.cpu().double()
    assert abs(o[0] - rl.item()) <= 1e-6 * abs(rl.item())
    assert abs(o[1] - rl.item() * T * 4096) <= 1e-6 * abs(rl.item() * T * 4096)
    if tw > 0:
        assert abs(o[2] - tl.item()) <= 1e-5 * abs(tl.item())
    else:
        assert o[2] == 0
    # backward, scaled by a device scalar
    gs = torch.tensor([0.7], device='cuda')
    d = ops.recon_loss_bwd(dev(recon), dev(target), gs, 0 if kind == 'mse' else 1, tw)
    if kind == 'mse':
        total = 0.7 * (rl * T * 4096 + tw * tl)
        (gref,) = torch.autograd.grad(total, r64)
    else:
        # torch's BCE backward formula (r-x)/max((1-r)r,1e-12), not the derivative of the clamped log
        gref = 0.7 * ((r64 - t64) / torch.clamp((1 - r64) * r64, min=1e-12) / B).detach()
        if tw > 0:
            (gt,) = torch.autograd.grad(0.7 * tw * tl, r64)
            gref = gref + gt
    mask = torch.ones_like(recon, dtype=torch.bool)
    mask[0, 0, 0, 0, :4] = False
    assert rel_err(d.cpu()[mask], gref[mask]) < 2e-6
    assert torch.isfinite(d).all()


def test_reparam_and_kl(ops):
    B, D = 7, 24
    mu, lv, eps = rnd(B, D, seed=3), rnd(B, D, seed=4) * 0.5, rnd(B, D, seed=5)
    z = ops.reparam_fwd(dev(mu), dev(lv), dev(eps))
    assert rel_err(z, O.reparameterize(mu, lv, eps)) < 1e-6
    dz = rnd(B, D, seed=6)
    dmu, dlv = ops.reparam_bwd(dev(dz), dev(lv), dev(eps))
    assert rel_err(dmu, dz) < 1e-7
    assert rel_err(dlv, dz * eps * 0.5 * torch.exp(0.5 * lv)) < 1e-6
    for fb in (0.0, 0.25):
        out = torch.zeros(3, device='cuda')
        ops.kl_fwd(dev(mu), dev(lv), fb, out)
        kpd = O.kl_per_dim(mu, lv)
        floored = torch.maximum(kpd, torch.tensor(fb, dtype=torch.float64)) if fb > 0 else kpd
        ref = torch.stack([floored.sum(1).mean(), kpd.sum(1).mean(), floored.mean()])
        torch.testing.assert_close(out.cpu().double(), ref, rtol=2e-6, atol=1e-6)
        m64, l64 = mu.clone().requires_grad_(True), lv.clone().requires_grad_(True)
        kp = O.kl_per_dim(m64, l64)
        if fb > 0:
            kp = torch.maximum(kp, torch.tensor(fb, dtype=torch.float64))
        gm, gl = torch.autograd.grad(0.4 * 1.3 * kp.sum(1).mean(), [m64, l64])
        dmu, dlv = ops.kl_bwd(dev(mu), dev(lv), torch.tensor([1.3], device='cuda'), 0.4, fb)
        assert rel_err(dmu, gm) < 2e-6 and rel_err(dlv, gl) < 2e-6


@pytest.mark.parametrize('B,Dc,Dm', [(4, 16, 16), (32, 128, 128), (9, 8, 20), (256, 128, 128), (70, 250, 37), (3, 5, 256)])
def test_independence(ops, B, Dc, Dm):
    zc, zm = rnd(B, Dc, seed=7), rnd(B, Dm, seed=8) + 0.3 * rnd(B, 1, seed=9)
    a, b = zc.clone().requires_grad_(True), zm.clone().requires_grad_(True)
    an = (a - a.mean(0)) / (a.std(0) + 1e-8)
    bn = (b - b.mean(0)) / (b.std(0) + 1e-8)
    loss = ((an.t() @ bn / B) ** 2).mean()
    ga, gb = torch.autograd.grad(1.7 * 0.02 * loss, [a, b])
    out = torch.zeros(1, device='cuda')
    ws = ops.indep_fwd(dev(zc), dev(zm), out)
    assert abs(out.item() - loss.item()) < 2e-5 * abs(loss.item())
    dzc, dzm = ops.indep_bwd(dev(zc), dev(zm), ws, torch.tensor([1.7], device='cuda'), 0.02)
    assert rel_err(dzc, ga) < 5e-5 and rel_err(dzm, gb) < 5e-5


# ------------------------------------------------------------------ GEMMs
GEMM_SHAPES = [(32, 128, 512), (4, 32, 2048), (24, 64, 32), (608, 512, 8192 // 4), (130, 260, 36),
               (640, 4096 // 4, 640), (256, 128, 4096), (1, 4, 4),
               # 304 tiles of 128 x 128: between one and two per CU, computed with 128 x 64 tiles
               (4864, 1024, 64), (4837, 960, 36)]


@pytest.mark.parametrize('M,N,K', GEMM_SHAPES)
def test_gemm_nt(ops, M, N, K):
    a, b, bias = rnd(M, K, seed=10), rnd(N, K, seed=11), rnd(N, seed=12)
    for act, with_bias in ((ops.ACT_NONE, False), (ops.ACT_LRELU, True)):
        ref = a @ b.t()
        if with_bias:
            ref = ref + bias
        if act == ops.ACT_LRELU:
            ref = O.leaky_relu(ref)
        got = ops.gemm_nt(dev(a), dev(b), dev(bias) if with_bias else None, act)
        assert got.shape == (M, N)
        assert rel_err(got, ref) < 2e-6, (M, N, K, act)


@pytest.mark.parametrize('M,N,K', GEMM_SHAPES)
def test_gemm_nn_tn(ops, M, N, K):
    a, b = rnd(M, K, seed=13), rnd(K, N, seed=14)
    got = ops.gemm_nn(dev(a), dev(b))
    assert rel_err(got, a @ b) < 2e-6
    base = rnd(M, N, seed=15)
    out = dev(base)
    ops.gemm_nn(dev(a), dev(b), out=out, accumulate=True)
    assert rel_err(out, base + a @ b) < 2e-6
    if M % 4 == 0:
        at = rnd(K, M, seed=16)
        got = ops.gemm_tn(dev(at), dev(b))
        assert rel_err(got, at.t() @ b) < 2e-6
        out = dev(base)
        ops.gemm_tn(dev(at), dev(b), out=out, accumulate=True)
        assert rel_err(out, base + at.t() @ b) < 2e-6


@pytest.mark.parametrize('M,N,K', GEMM_SHAPES + [(512, 384, 1024)])
def test_gemm_split_precision_is_fp32_accurate(ops, M, N, K):
    """'fp32_split' forms every fp32 product from six bf16 x bf16 products of a three-term split: its error
    against an fp64 product stays at the fp32-MFMA kernels' level (same bound as their tests, and no more
    than 1.5x what the fp32 mode gives on the same operands), wide dynamic range included."""
    g = torch.Generator().manual_seed(21)
    scale = torch.exp2(torch.randint(-20, 20, (M, 1), generator=g).float())     # rows 2^-20 .. 2^20 apart
    a = (torch.randn(M, K, generator=g) * scale).float()
    b, bias = rnd(N, K, seed=23).float(), rnd(N, seed=24).float()
    ref = O.leaky_relu(a.double() @ b.double().t() + bias.double())
    e32 = rel_err(ops.gemm_nt(dev(a), dev(b), dev(bias), ops.ACT_LRELU), ref)
    ops.set_precision('fp32_split')
    try:
        assert ops.get_precision() == 'fp32_split'
        es = rel_err(ops.gemm_nt(dev(a), dev(b), dev(bias), ops.ACT_LRELU), ref)
        bt = rnd(K, N, seed=25).float()
        e_nn = rel_err(ops.gemm_nn(dev(a), dev(bt)), a.double() @ bt.double())
        c = rnd(M, N, seed=26).float()
        e_tn = rel_err(ops.gemm_tn(dev(a), dev(c)), a.double().t() @ c.double()) if K % 4 == 0 and N % 4 == 0 else 0.0
    finally:
        ops.set_precision('fp32')
    assert es < 2e-6 and es <= 1.5 * e32 + 1e-8, (es, e32)
    assert e_nn < 2e-6 and e_tn < 2e-6, (e_nn, e_tn)


@pytest.mark.parametrize('M,N,K', [(2560, 256, 128), (2500, 384, 256), (1300, 640, 96)])
def test_gemm_with_column_tiles_grouped_per_xcd(ops, M, N, K):
    """A tall product with two to five column tiles per row of tiles takes the tile numbering that puts the column
    tiles of one row of tiles on one XCD (20 groups of tiles: sixteen renumbered, four as launched; 11: eight and
    three): every tile is still computed exactly once, in all three operand layouts and in bf16 mode, whose
    kernels hold three workgroups per CU and send the output tile through LDS in two runs of rows."""
    a, b, bias = rnd(M, K, seed=31), rnd(N, K, seed=32), rnd(N, seed=33)
    ref = O.leaky_relu(a @ b.t() + bias)
    assert rel_err(ops.gemm_nt(dev(a), dev(b), dev(bias), ops.ACT_LRELU), ref) < 2e-6
    dy, w = rnd(M, N, seed=34), rnd(N, K, seed=35)                      # dx = dy w: K plays N here
    assert rel_err(ops.gemm_nn(dev(dy), dev(w)), dy @ w) < 2e-6
    if M % 4 == 0:
        at, bt = rnd(K * 8, M, seed=36), rnd(K * 8, N, seed=37)        # a long K over few tiles: split-K as well
        assert rel_err(ops.gemm_tn(dev(at), dev(bt)), at.t() @ bt) < 2e-6
    ops.set_precision('bf16')
    try:
        got = ops.gemm_nt(dev(a), dev(b), dev(bias), ops.ACT_LRELU)
    finally:
        ops.set_precision('fp32')
    a16, b16 = a.bfloat16().float(), b.bfloat16().float()
    assert rel_err(got, O.leaky_relu(a16 @ b16.t() + bias)) < 2e-6


def test_gemm_strided_views(ops):
    # operands that are column slices of wider matrices (row stride != width)
    big = dev(rnd(40, 96, seed=17))
    a = big[:, 32:64]
    w = dev(rnd(48, 32, seed=18))
    out_big = torch.zeros(40, 100, device='cuda')
    ops.gemm_nt(a, w, None, ops.ACT_NONE, out=out_big[:, 4:52])
    assert rel_err(out_big[:, 4:52], a.double().cpu() @ w.double().cpu().t()) < 2e-6
    assert out_big[:, :4].abs().sum() == 0 and out_big[:, 52:].abs().sum() == 0


def test_gemm_rejects_bad_shapes(ops):
    from hipvae._lib import DvaeError
    with pytest.raises(DvaeError):
        ops.gemm_nt(dev(rnd(4, 6)), dev(rnd(8, 6)))          # K % 4 != 0
    with pytest.raises(RuntimeError):
        ops.gemm_nt(rnd(4, 8).float(), rnd(8, 8).float())     # CPU tensors: no fallback


# ------------------------------------------------------------------ conv family
def nhwc(t):   # NCHW -> NHWC
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


CONV_CASES = [  # N, H, Cin, Cout
    (3, 64, 1, 16), (2, 64, 1, 32), (5, 32, 16, 32), (3, 16, 32, 64), (4, 8, 64, 128), (9, 4, 8, 4),
    (2, 8, 128, 256), (130, 4, 4, 8),
    # (32,64) and (16,32) on frames at least 32 wide take the row-ring kernel: one and two groups of
    # sixteen output pixels per row, more workgroups than share out the rows (N=300)
    (4, 32, 32, 64), (3, 64, 16, 32), (2, 64, 32, 64), (300, 32, 32, 64), (37, 32, 16, 32),
    # 128-wide tiles on a quarter to half of the chip's workgroup slots: the K loop is shared out over several
    # workgroups per tile and the partial outputs are summed by a second launch (three uneven shares at the
    # batch-32 shape of the last encoder layer; 512 channels)
    (640, 8, 128, 256), (320, 8, 256, 512),
]


@pytest.mark.parametrize('N,H,Cin,Cout', CONV_CASES)
def test_conv4s2_fwd(ops, N, H, Cin, Cout):
    x = rnd(N, Cin, H, H, seed=20)
    w = rnd(Cout, Cin, 4, 4, seed=21, scale=0.2)
    use_affine = Cin > 1
    sc, sh = 1 + 0.3 * rnd(Cin, seed=22), 0.2 * rnd(Cin, seed=23)
    xin = O.leaky_relu(x * sc[None, :, None, None] + sh[None, :, None, None]) if use_affine else x
    ref = F.conv2d(xin, w, None, stride=2, padding=1)
    wp = ops.pack_conv_w(dev(w))
    y, stats, nparts = ops.conv4s2_fwd(dev(nhwc(x)), dev(sc) if use_affine else None,
                                       dev(sh) if use_affine else None,
                                       ops.ACT_LRELU if use_affine else ops.ACT_NONE, wp, True)
    assert rel_err(nchw(y), ref) < 2e-6
    s = stats.double().sum(0).cpu()
    assert rel_err(s[0], ref.sum(dim=(0, 2, 3))) < 1e-5
    assert rel_err(s[1], (ref ** 2).sum(dim=(0, 2, 3))) < 1e-5


@pytest.mark.parametrize('N,H,Cin,Cout', [(4, 8, 64, 128), (2, 8, 128, 256), (9, 4, 8, 4), (640, 8, 128, 256)])
def test_conv4s2_fwd_in_split_precision_is_fp32_accurate(ops, N, H, Cin, Cout):
    """the tiled kernels in 'fp32_split' mode (six bf16 MFMA products of a three-term split per fp32 product) keep
    the fp32 bar of the plain kernels"""
    x = rnd(N, Cin, H, H, seed=28)
    w = rnd(Cout, Cin, 4, 4, seed=29, scale=0.2)
    ref = F.conv2d(x, w, None, stride=2, padding=1)
    wp = ops.pack_conv_w(dev(w))
    ops.set_precision('fp32_split')
    try:
        y, stats, _ = ops.conv4s2_fwd(dev(nhwc(x)), None, None, ops.ACT_NONE, wp, True)
        wt = rnd(Cout, Cin, 4, 4, seed=30, scale=0.2)
        yt, _, _ = ops.deconv4s2_fwd(dev(nhwc(ref)), None, None, ops.ACT_NONE, ops.pack_deconv_w(dev(wt)), False)
    finally:
        ops.set_precision('fp32')
    assert rel_err(nchw(y), ref) < 2e-6
    assert rel_err(stats.double().sum(0).cpu()[1], (ref ** 2).sum(dim=(0, 2, 3))) < 1e-5
    assert rel_err(nchw(yt), F.conv_transpose2d(ref, wt, None, stride=2, padding=1)) < 2e-6


# (64,32) and (32,16) on frames at least 16 wide take the row-ring kernel: one and two groups of
# sixteen pixels per row (H=16, 32), a single row per image never (H >= 16), and more workgroups than
# share out the rows (N=70 at 64->32); the same channels on smaller frames take the tiled kernel
DECONV_CASES = [(3, 4, 64, 32), (2, 8, 32, 16), (3, 16, 16, 8), (2, 32, 8, 4), (6, 4, 256, 128), (70, 4, 4, 36),
                (3, 2, 64, 32), (5, 2, 32, 16), (100, 32, 32, 16), (40, 32, 64, 32), (3, 16, 32, 16), (2, 16, 64, 32),
                (300, 16, 32, 16), (70, 32, 64, 32),
                # 128 -> 64 on 8-wide frames: one phase per workgroup, two rows per group of sixteen pixels
                (3, 8, 128, 64), (300, 8, 128, 64),
                # K shared out over two workgroups per tile (152 tiles of 128 x 128)
                (300, 4, 256, 128)]


@pytest.mark.parametrize('N,H,Cin,Cout', DECONV_CASES)
def test_deconv4s2_fwd(ops, N, H, Cin, Cout):
    x = rnd(N, Cin, H, H, seed=24)
    w = rnd(Cin, Cout, 4, 4, seed=25, scale=0.2)
    sc, sh = 1 + 0.3 * rnd(Cin, seed=26), 0.2 * rnd(Cin, seed=27)
    xin = O.leaky_relu(x * sc[None, :, None, None] + sh[None, :, None, None])
    ref = F.conv_transpose2d(xin, w, None, stride=2, padding=1)
    wp = ops.pack_deconv_w(dev(w))
    y, stats, nparts = ops.deconv4s2_fwd(dev(nhwc(x)), dev(sc), dev(sh), ops.ACT_LRELU, wp, True)
    assert rel_err(nchw(y), ref) < 2e-6
    s = stats.double().sum(0).cpu()
    assert rel_err(s[0], ref.sum(dim=(0, 2, 3))) < 1e-5
    assert rel_err(s[1], (ref ** 2).sum(dim=(0, 2, 3))) < 1e-5
    # identity load transform
    y2, _, _ = ops.deconv4s2_fwd(dev(nhwc(x)), None, None, ops.ACT_NONE, wp, False)
    assert rel_err(nchw(y2), F.conv_transpose2d(x, w, None, stride=2, padding=1)) < 2e-6


def test_deconv_is_conv_data_gradient(ops):
    # the transposed-conv kernel with a Conv2d weight gives that conv's input gradient
    N, H, Cin, Cout = 3, 16, 8, 16
    x = rnd(N, Cin, H, H, seed=28).requires_grad_(True)
    w = rnd(Cout, Cin, 4, 4, seed=29, scale=0.2)
    y = F.conv2d(x, w, None, stride=2, padding=1)
    dy = rnd(*y.shape, seed=30)
    (dx,) = torch.autograd.grad(y, x, dy)
    wp = ops.pack_deconv_w(dev(w))      # [Kc=Cout][Nc=Cin]
    got, _, _ = ops.deconv4s2_fwd(dev(nhwc(dy)), None, None, ops.ACT_NONE, wp, False)
    assert rel_err(nchw(got), dx) < 2e-6


# (64,32) and (32,16) with Ws 16 or 32 take the row-ring kernel (several rows per workgroup at N=300)
WGRAD_CASES = [(3, 32, 16, 1), (4, 16, 32, 16), (3, 8, 64, 32), (2, 4, 128, 64), (5, 32, 32, 1), (2, 4, 8, 4),
               (40, 16, 16, 8), (3, 2, 256, 128), (3, 16, 64, 32), (2, 32, 32, 16), (2, 32, 64, 32), (300, 16, 64, 32),
               (70, 16, 32, 16),
               # 128 x 64 on 8-wide P: one filter row per workgroup, pairs of rows
               (3, 8, 128, 64), (40, 8, 128, 64),
               # 256 x 128 on 4x4 P with at least 32 frames: one filter tap per workgroup, a frame per step
               (40, 4, 256, 128), (67, 4, 256, 128)]


@pytest.mark.parametrize('N,Hs,C1,C2', WGRAD_CASES)
def test_conv4s2_wgrad(ops, N, Hs, C1, C2):
    # as a Conv2d weight gradient: P = dY [N,Hs,Hs,C1], Q = layer input [N,2Hs,2Hs,C2]
    x = rnd(N, C2, 2 * Hs, 2 * Hs, seed=31)
    dy = rnd(N, C1, Hs, Hs, seed=32)
    use_affine = C2 > 1
    sc, sh = 1 + 0.3 * rnd(C2, seed=33), 0.2 * rnd(C2, seed=34)
    xin = O.leaky_relu(x * sc[None, :, None, None] + sh[None, :, None, None]) if use_affine else x
    w = torch.zeros(C1, C2, 4, 4, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(xin, w, None, stride=2, padding=1)
    (dw,) = torch.autograd.grad(y, w, dy)
    got = ops.conv4s2_wgrad(dev(nhwc(dy)), None, None, ops.ACT_NONE, dev(nhwc(x)),
                            dev(sc) if use_affine else None, dev(sh) if use_affine else None,
                            ops.ACT_LRELU if use_affine else ops.ACT_NONE)
    assert rel_err(got, dw) < 5e-6


@pytest.mark.parametrize('N,Hs,C1,C2', [(3, 8, 64, 32), (3, 2, 256, 128), (9, 8, 256, 256), (40, 16, 16, 8),
                                        # the row kernels whose operands meet in LDS: 64 x 32 and 32 x 16 (16 and 32
                                        # pixels wide), 128 x 64 on 8-wide P, 256 x 128 on 4 x 4 P
                                        (3, 16, 64, 32), (70, 32, 64, 32), (2, 32, 32, 16), (70, 16, 32, 16),
                                        (3, 8, 128, 64), (40, 8, 128, 64), (40, 4, 256, 128), (67, 4, 256, 128)])
def test_conv4s2_wgrad_in_split_precision_is_fp32_accurate(ops, N, Hs, C1, C2, monkeypatch):
    """the weight-gradient kernels in 'fp32_split' mode keep the fp32 bar (the row kernels take that mode on
    request only: they are slower in it)"""
    monkeypatch.setenv('DVAE_SPLIT_ROW_WGRAD', '1')
    x = rnd(N, C2, 2 * Hs, 2 * Hs, seed=35)
    dy = rnd(N, C1, Hs, Hs, seed=36)
    w = torch.zeros(C1, C2, 4, 4, dtype=torch.float64, requires_grad=True)
    (dw,) = torch.autograd.grad(F.conv2d(x, w, None, stride=2, padding=1), w, dy)
    ops.set_precision('fp32_split')
    try:
        got = ops.conv4s2_wgrad(dev(nhwc(dy)), None, None, ops.ACT_NONE, dev(nhwc(x)), None, None, ops.ACT_NONE)
    finally:
        ops.set_precision('fp32')
    assert rel_err(got, dw) < 5e-6


@pytest.mark.parametrize('N,H,Cin,Cout', [(3, 16, 64, 32), (2, 16, 32, 16), (5, 8, 128, 64), (40, 4, 256, 128)])
def test_wgrad_for_transposed_conv_in_split_precision(ops, N, H, Cin, Cout, monkeypatch):
    """the row kernels with a load transform on P, operands spread over 2^-12 .. 2^12, in 'fp32_split' mode"""
    monkeypatch.setenv('DVAE_SPLIT_ROW_WGRAD', '1')
    x = rnd(N, Cin, H, H, seed=35) * torch.exp2(torch.randint(-12, 13, (N, Cin, 1, 1), generator=torch.Generator().manual_seed(1)).double())
    sc, sh = 1 + 0.3 * rnd(Cin, seed=36), 0.2 * rnd(Cin, seed=37)
    xin = O.leaky_relu(x * sc[None, :, None, None] + sh[None, :, None, None])
    w = torch.zeros(Cin, Cout, 4, 4, dtype=torch.float64, requires_grad=True)
    y = F.conv_transpose2d(xin, w, None, stride=2, padding=1)
    dy = rnd(*y.shape, seed=38)
    (dw,) = torch.autograd.grad(y, w, dy)
    ops.set_precision('fp32_split')
    try:
        got = ops.conv4s2_wgrad(dev(nhwc(x)), dev(sc), dev(sh), ops.ACT_LRELU, dev(nhwc(dy)), None, None, ops.ACT_NONE)
    finally:
        ops.set_precision('fp32')
    plain = ops.conv4s2_wgrad(dev(nhwc(x)), dev(sc), dev(sh), ops.ACT_LRELU, dev(nhwc(dy)), None, None, ops.ACT_NONE)
    assert rel_err(got, dw) < 5e-6
    assert rel_err(got, dw) < 2 * rel_err(plain, dw) + 1e-7          # no worse than the fp32 MFMA on the same operands


@pytest.mark.parametrize('N,H,Cin,Cout', [(3, 8, 16, 8), (3, 16, 64, 32), (2, 16, 32, 16), (5, 8, 128, 64), (40, 4, 256, 128)])
def test_wgrad_for_transposed_conv(ops, N, H, Cin, Cout):
    # ConvTranspose2d weight gradient: P = layer input (with load transform), Q = dY
    x = rnd(N, Cin, H, H, seed=35)
    sc, sh = 1 + 0.3 * rnd(Cin, seed=36), 0.2 * rnd(Cin, seed=37)
    xin = O.leaky_relu(x * sc[None, :, None, None] + sh[None, :, None, None])
    w = torch.zeros(Cin, Cout, 4, 4, dtype=torch.float64, requires_grad=True)
    y = F.conv_transpose2d(xin, w, None, stride=2, padding=1)
    dy = rnd(*y.shape, seed=38)
    (dw,) = torch.autograd.grad(y, w, dy)
    got = ops.conv4s2_wgrad(dev(nhwc(x)), dev(sc), dev(sh), ops.ACT_LRELU, dev(nhwc(dy)), None, None, ops.ACT_NONE)
    assert rel_err(got, dw) < 5e-6


@pytest.mark.parametrize('N,H,Cin', [(3, 32, 32), (2, 8, 4), (5, 16, 16)])
def test_deconv_final(ops, N, H, Cin):
    x = rnd(N, Cin, H, H, seed=39)
    w = rnd(Cin, 1, 4, 4, seed=40, scale=0.3)
    b = rnd(1, seed=41)
    sc, sh = 1 + 0.3 * rnd(Cin, seed=42), 0.2 * rnd(Cin, seed=43)
    xin = O.leaky_relu(x * sc[None, :, None, None] + sh[None, :, None, None])
    ref = torch.sigmoid(F.conv_transpose2d(xin, w, b, stride=2, padding=1))[:, 0]
    got = ops.deconv4s2_final_fwd(dev(nhwc(x)), dev(sc), dev(sh), ops.ACT_LRELU, dev(w), dev(b), ops.ACT_SIGMOID)
    torch.testing.assert_close(got.cpu().double(), ref, rtol=1e-5, atol=1e-6)


def test_pack_many_is_the_single_packs_in_one_launch(ops):
    """20 weights (more than one table holds), both layouts, sizes from 1 to 256 channels: bit-equal to the
    per-weight packs, which the convolution tests pin against torch."""
    shapes = [(32, 1), (64, 32), (128, 64), (256, 128), (5, 3), (1, 1), (7, 130), (256, 256), (16, 8), (2, 67)]
    items = []
    for i, (a, b) in enumerate(shapes):
        w = dev(rnd(a, b, 4, 4, seed=900 + i).float())
        items += [(w, False), (w, True)]
    got = ops.pack_many(items)
    assert len(got) == len(items)
    for (w, tr), g in zip(items, got):
        want = ops.pack_deconv_w(w) if tr else ops.pack_conv_w(w)
        assert g.shape == want.shape and torch.equal(g, want)
    assert ops.pack_many([]) == []


# ------------------------------------------------------------------ BatchNorm glue
@pytest.mark.parametrize('M,C', [(4 * 32 * 32, 16), (640 * 16, 128), (37, 4), (2000, 512), (4099, 8), (5003, 12),
                                 (70001, 256)])
def test_bn_forward_backward(ops, M, C):
    y = rnd(M, C, seed=44) * (1 + rnd(C, seed=45).abs()) + rnd(C, seed=46)
    gamma, beta = 1 + 0.2 * rnd(C, seed=47), 0.1 * rnd(C, seed=48)
    rm, rv = 0.1 * rnd(C, seed=49), 1 + 0.2 * rnd(C, seed=50).abs()
    # statistics as the conv kernels would emit them: a few partial rows
    parts = 5
    chunks = y.chunk(parts, dim=0)
    partials = torch.stack([torch.stack([c.sum(0), (c ** 2).sum(0)]) for c in chunks])
    d_rm, d_rv = dev(rm), dev(rv)
    nbt = torch.tensor(3, dtype=torch.int64, device='cuda')
    scale, shift, mean, invstd = ops.bn_finalize(dev(partials), len(chunks), M, dev(gamma), dev(beta), d_rm, d_rv,
                                                 nbt, 0.1, 1e-5)
    mu = y.mean(0)
    var = y.var(0, unbiased=False)
    inv = 1 / torch.sqrt(var + 1e-5)
    assert rel_err(mean, mu) < 1e-5 and rel_err(invstd, inv) < 1e-5
    assert rel_err(scale, gamma * inv) < 1e-5 and rel_err(shift, beta - mu * gamma * inv) < 2e-5
    assert rel_err(d_rm, 0.9 * rm + 0.1 * mu) < 1e-5
    assert rel_err(d_rv, 0.9 * rv + 0.1 * y.var(0, unbiased=True)) < 1e-5
    assert nbt.item() == 4
    # eval affine
    es, eh = ops.bn_eval_affine(dev(gamma), dev(beta), dev(rm), dev(rv), 1e-5)
    assert rel_err(es, gamma / torch.sqrt(rv + 1e-5)) < 1e-6
    assert rel_err(eh, beta - rm * gamma / torch.sqrt(rv + 1e-5)) < 1e-5
    # materialised activation
    act = ops.affine_act(dev(y), scale, shift, ops.ACT_LRELU)
    ref_act = O.leaky_relu((y - mu) * inv * gamma + beta)
    assert rel_err(act, ref_act) < 1e-5
    # backward
    y64 = y.clone().requires_grad_(True)
    g64, b64 = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    u = O.leaky_relu((y64 - y64.mean(0)) / torch.sqrt(y64.var(0, unbiased=False) + 1e-5) * g64 + b64)
    du = rnd(M, C, seed=51)
    dy, dg, db = torch.autograd.grad(u, [y64, g64, b64], du)
    d_du = dev(du)
    dgamma, dbeta = ops.bn_bwd(d_du, dev(y), scale, shift, mean, invstd, dev(gamma))
    assert rel_err(dgamma, dg) < 2e-5 and rel_err(dbeta, db) < 2e-5
    assert rel_err(d_du, dy) < 2e-5


@pytest.mark.parametrize('M,N', [(32, 512), (640, 4096), (7, 4), (100, 36)])
def test_bias_act_bwd(ops, M, N):
    out = O.leaky_relu(rnd(M, N, seed=52))
    dout = rnd(M, N, seed=53)
    ref = torch.where(out > 0, dout, 0.2 * dout)
    d = dev(dout)
    db = ops.bias_act_bwd(d, dev(out), ops.ACT_LRELU, True)
    assert rel_err(d, ref) < 1e-7 and rel_err(db, ref.sum(0)) < 1e-5
    d = dev(dout)
    db = ops.bias_act_bwd(d, None, ops.ACT_NONE, True)
    assert rel_err(d, dout) < 1e-7 and rel_err(db, dout.sum(0)) < 1e-5
    # into a buffer of its own: the same values, the incoming gradient untouched
    for act, want in ((ops.ACT_LRELU, ref), (ops.ACT_NONE, dout)):
        d, dst = dev(dout), torch.full((M, N), float('nan'), device='cuda')
        db = ops.bias_act_bwd(d, dev(out) if act == ops.ACT_LRELU else None, act, True, dst=dst)
        assert torch.equal(d.cpu().double(), dout.float().double())
        assert rel_err(dst, want) < 1e-7 and rel_err(db, want.sum(0)) < 1e-5


@pytest.mark.parametrize('M,N', [(5120, 512), (4864, 1024), (256, 256), (5120, 4096), (1000, 36)])
def test_column_sums_in_one_launch(ops, M, N, monkeypatch):
    """Bias gradient and BatchNorm-backward sums as ONE launch each: the workgroup of a column group that finishes
    last adds the group's partial rows inside the launch (hand-off through write-through stores and an arrival
    counter).  Against fp64 sums; the same bits on every call (fixed order of the additions, whichever workgroup is
    last) with other work keeping the chip unevenly busy in between; the counters are back at zero afterwards; and
    the two-launch form (no counters handed in) agrees to rounding."""
    from hipvae._lib import check, lib
    monkeypatch.setattr(ops, 'USE_TICKETS', True)           # (not the default path: measured slower inside the step)
    out = O.leaky_relu(rnd(M, N, seed=152))
    dout = rnd(M, N, seed=153)
    ref = torch.where(out > 0, dout, 0.2 * dout)
    noise = torch.randn(4096, 4096, device='cuda')
    side = torch.cuda.Stream()
    results = []
    for rep in range(4):
        d = dev(dout)
        torch.cuda.synchronize()
        if rep % 2:
            with torch.cuda.stream(side):                   # something else on the chip while the kernel runs
                noise @ noise
        db = ops.bias_act_bwd(d, dev(out), ops.ACT_LRELU, True)
        results.append(db.clone())
        assert rel_err(d, ref) < 1e-7
    assert rel_err(results[0], ref.sum(0)) < 1e-5
    assert all(torch.equal(results[0], r) for r in results[1:])
    assert int(ops.tickets(torch.device('cuda', 0)).abs().sum()) == 0
    # the two-launch form of the same entry point
    d, db2 = dev(dout), torch.empty(N, device='cuda')
    ws = ops.workspace(lib.dvae_colsum_ws_floats(M, N), d.device)
    check(lib.dvae_bias_act_bwd_to_f32(d.data_ptr(), d.data_ptr(), dev(out).data_ptr(), ops.ACT_LRELU, db2.data_ptr(),
                                       ws.data_ptr(), None, M, N, torch.cuda.current_stream().cuda_stream))
    assert rel_err(db2, ref.sum(0)) < 1e-5 and rel_err(db2, results[0].double().cpu()) < 2e-6
    if N % 4 == 0 and N <= 1024:
        # BatchNorm backward: sums + coefficient vectors in one launch against the two-launch pair
        y, gamma, beta = rnd(M, N, seed=154), rnd(N, seed=155) + 1.5, rnd(N, seed=156)
        mu, inv = y.mean(0), 1 / torch.sqrt(y.var(0, unbiased=False) + 1e-5)
        aff = [dev(t) for t in (gamma * inv, beta - mu * gamma * inv, mu, inv)]
        du = dev(rnd(M, N, seed=157))
        one = [t.clone() for t in ops.bn_bwd_reduce_coeffs(du, dev(y), aff)]
        again = ops.bn_bwd_reduce_coeffs(du, dev(y), aff)
        two = ops.bn_bwd_coeffs(ops.bn_bwd_reduce(du, dev(y), aff), aff, M)
        for a, b, c in zip(one, again, two):
            assert torch.equal(a, b) and rel_err(a, c.double().cpu()) < 5e-6
        assert int(ops.tickets(torch.device('cuda', 0)).abs().sum()) == 0


def test_sigmoid_bwd(ops):
    r = torch.sigmoid(rnd(3, 5, 64, 64, seed=54))
    dr = rnd(3, 5, 64, 64, seed=55)
    dl, db = ops.sigmoid_bwd(dev(dr), dev(r))
    ref = dr * r * (1 - r)
    assert rel_err(dl, ref) < 1e-6 and abs(db.item() - ref.sum().item()) < 1e-4 * abs(ref.sum().item()) + 1e-4


# ------------------------------------------------------------------ LSTM steps
@pytest.mark.parametrize('B,H,I,mode', [(4, 32, 32, 'xproj'), (32, 256, 512, 'xproj'), (5, 32, 32, 'x'),
                                        (33, 64, 64, 'x'), (32, 256, 256, 'x'), (4, 32, 32, 'first')])
def test_lstm_step_fwd(ops, B, H, I, mode):
    w_ih, w_hh = rnd(4 * H, I, seed=60, scale=0.1), rnd(4 * H, H, seed=61, scale=0.1)
    b_ih, b_hh = rnd(4 * H, seed=62, scale=0.1), rnd(4 * H, seed=63, scale=0.1)
    x, h, c = rnd(B, I, seed=64), rnd(B, H, seed=65), rnd(B, H, seed=66)
    if mode == 'first':
        h = torch.zeros_like(h)
        c = torch.zeros_like(c)
    proj = x @ w_ih.t() + b_ih + b_hh
    h2, c2 = O.lstm_cell(proj, h, c, w_hh)
    gates = torch.empty(B, 4 * H, device='cuda')
    # outputs written into column slices of wider buffers (row strides exercised)
    hbuf = torch.zeros(B, 2 * H, device='cuda')
    cbuf = torch.zeros(B, H, device='cuda')
    if mode == 'x':
        ops.lstm_step_fwd(None, dev(x), dev(w_ih), dev(h), dev(c), dev(w_hh), dev(b_ih), dev(b_hh), gates,
                          hbuf[:, H:], cbuf)
    elif mode == 'xproj':
        ops.lstm_step_fwd(dev(proj), None, None, dev(h), dev(c), dev(w_hh), None, None, gates, hbuf[:, H:], cbuf)
    else:
        ops.lstm_step_fwd(dev(proj), None, None, None, None, dev(w_hh), None, None, gates, hbuf[:, H:], cbuf)
    assert rel_err(hbuf[:, H:], h2) < 5e-6 and rel_err(cbuf, c2) < 5e-6
    assert hbuf[:, :H].abs().sum() == 0
    pre = proj + h @ w_hh.t()
    i, f, g, o = pre.chunk(4, 1)
    ref_g = torch.cat([torch.sigmoid(i), torch.sigmoid(f), torch.tanh(g), torch.sigmoid(o)], 1)
    assert rel_err(gates, ref_g) < 5e-6


def test_lstm_step_bwd(ops):
    B, H = 6, 32
    w_hh, w_ih = rnd(4 * H, H, seed=67, scale=0.2), rnd(4 * H, H, seed=68, scale=0.2)
    proj = rnd(B, 4 * H, seed=69)
    h = rnd(B, H, seed=70).requires_grad_(True)
    c = rnd(B, H, seed=71).requires_grad_(True)
    p = proj.clone().requires_grad_(True)
    h2, c2 = O.lstm_cell(p, h, c, w_hh)
    dh2, dc2 = rnd(B, H, seed=72), rnd(B, H, seed=73)
    dp, dh, dc = torch.autograd.grad([h2, c2], [p, h, c], [dh2, dc2])
    pre = proj + h.detach() @ w_hh.t()
    i, f, g, o = pre.chunk(4, 1)
    gates = torch.cat([torch.sigmoid(i), torch.sigmoid(f), torch.tanh(g), torch.sigmoid(o)], 1)
    dgates = torch.empty(B, 4 * H, device='cuda')
    dcp = torch.empty(B, H, device='cuda')
    ops.lstm_step_bwd_pointwise(dev(dh2), dev(dc2), dev(gates), dev(c.detach()), dev(c2.detach()), dgates, dcp)
    assert rel_err(dgates, dp) < 5e-6 and rel_err(dcp, dc) < 5e-6
    out = torch.empty(B, H, device='cuda')
    ops.skinny_gemm_nt(dgates, dev(w_hh.t()), out, False)
    assert rel_err(out, dh) < 5e-6
    ops.skinny_gemm_nt(dgates, dev((w_hh + w_ih).t()), out, True)
    assert rel_err(out, dh + dp @ w_hh + dp @ w_ih) < 5e-6


@pytest.mark.parametrize('B,H', [(6, 32), (37, 24), (256, 256)])
def test_lstm_bwd_step_is_product_then_pointwise(ops, B, H):
    """The one-launch backward step against the two kernels it fuses."""
    dg_t, w = dev(rnd(B, 4 * H, seed=76)), dev(rnd(4 * H, H, seed=77, scale=0.2))
    wt = w.t().contiguous()
    dh0, dc = rnd(B, H, seed=78), dev(rnd(B, H, seed=79))
    gates = dev(torch.sigmoid(rnd(B, 4 * H, seed=80)))
    c_prev, c = dev(rnd(B, H, seed=81)), dev(rnd(B, H, seed=82))
    for cp, dcin in ((c_prev, dc), (None, None)):
        ref_dh = dev(dh0)
        ops.skinny_gemm_nt(dg_t, wt, ref_dh, True)
        ref_dg, ref_dc = torch.empty(B, 4 * H, device='cuda'), torch.empty(B, H, device='cuda')
        ops.lstm_step_bwd_pointwise(ref_dh, dcin, gates, cp, c, ref_dg, ref_dc)
        got_dh = dev(dh0)
        got_dg, got_dc = torch.empty(B, 4 * H, device='cuda'), torch.empty(B, H, device='cuda')
        ops.lstm_bwd_step(dg_t, wt, got_dh, dcin, gates, cp, c, got_dg, got_dc)
        assert torch.equal(got_dh, ref_dh) and torch.equal(got_dg, ref_dg) and torch.equal(got_dc, ref_dc)


@pytest.mark.parametrize('B,H,second', [(6, 32, True), (37, 24, True), (32, 256, True), (32, 256, False)])
def test_lstm_bwd_step2_adds_both_products(ops, B, H, second):
    """The stacked decoder's backward cell: two products onto one hidden-state gradient, then the pointwise
    backward, against the separate kernels."""
    dg_a, dg_b = dev(rnd(B, 4 * H, seed=176)), dev(rnd(B, 4 * H, seed=177))
    wt_a = dev(rnd(4 * H, H, seed=178, scale=0.2)).t().contiguous()
    wt_b = dev(rnd(4 * H, H, seed=179, scale=0.2)).t().contiguous()
    dh0, dc = rnd(B, H, seed=180), dev(rnd(B, H, seed=181))
    gates = dev(torch.sigmoid(rnd(B, 4 * H, seed=182)))
    c_prev, c = dev(rnd(B, H, seed=183)), dev(rnd(B, H, seed=184))
    ref_dh = dev(dh0)
    ops.skinny_gemm_nt(dg_a, wt_a, ref_dh, True)
    if second:
        ops.skinny_gemm_nt(dg_b, wt_b, ref_dh, True)
    ref_dg, ref_dc = torch.empty(B, 4 * H, device='cuda'), torch.empty(B, H, device='cuda')
    ops.lstm_step_bwd_pointwise(ref_dh, dc, gates, c_prev, c, ref_dg, ref_dc)
    got_dh = dev(dh0)
    got_dg, got_dc = torch.empty(B, 4 * H, device='cuda'), torch.empty(B, H, device='cuda')
    ops.lstm_bwd_step2(dg_a, wt_a, dg_b if second else None, wt_b if second else None, got_dh, dc, gates, c_prev, c,
                       got_dg, got_dc)
    # one sum over both products instead of two roundings onto dh: equal to rounding, not bit for bit
    assert rel_err(got_dh, ref_dh) < 2e-6 and rel_err(got_dg, ref_dg) < 2e-6 and rel_err(got_dc, ref_dc) < 2e-6


@pytest.mark.parametrize('B,N,K', [(32, 256, 1024), (256, 256, 1024), (70, 36, 132), (1, 4, 4), (17, 3, 2052)])
def test_skinny_gemm(ops, B, N, K):
    a, w = rnd(B, K, seed=74), rnd(K, N, seed=75)
    base = rnd(B, N, seed=76)
    # strided rows on all three operands: column slices of wider buffers
    abuf, wbuf, obuf = dev(torch.cat([a, a], 1)), dev(torch.cat([w.t(), w.t()], 1)), dev(torch.cat([base, base], 1))
    ops.skinny_gemm_nt(abuf[:, K:], wbuf[:, :K], obuf[:, N:], False)
    assert rel_err(obuf[:, N:], a @ w) < 2e-6
    assert torch.equal(obuf[:, :N].cpu().double(), base.float().double())
    ops.skinny_gemm_nt(abuf[:, :K], wbuf[:, K:], obuf[:, :N], True)
    assert rel_err(obuf[:, :N], base + a @ w) < 2e-6


@pytest.mark.parametrize('T,B,H,chains', [(19, 32, 256, 2), (5, 70, 128, 2), (20, 256, 256, 1), (7, 3, 256, 2),
                                          (4, 33, 512, 1), (1, 32, 256, 2), (6, 256, 256, 2), (3, 100, 512, 2)])
def test_lstm_seq_fwd_whole_recurrence_in_one_launch(ops, T, B, H, chains):
    """dvae_lstm_seq_fwd_f32 (weights in registers, hidden state handed between workgroups inside the
    launch) against the fp64 recurrence: encoder-style chains (input projections, zero initial state, the
    second one walking backwards) writing into column slices of one [T,B,chains*H] buffer."""
    assert ops.lstm_seq_supported(chains, T, B, H)
    D = chains
    hseq = torch.zeros(T, B, D * H, device='cuda')
    specs, refs = [], []
    for d in range(D):
        xproj = rnd(T, B, 4 * H, seed=90 + d)
        w = rnd(4 * H, H, seed=92 + d, scale=0.08)
        b_a, b_b = rnd(4 * H, seed=94 + d, scale=0.1), rnd(4 * H, seed=96 + d, scale=0.1)
        h, c = torch.zeros(B, H, dtype=torch.float64), torch.zeros(B, H, dtype=torch.float64)
        rg, rh, rc = [None] * T, [None] * T, [None] * T
        for t in (range(T - 1, -1, -1) if d == 1 else range(T)):
            pre = xproj[t] + b_a + b_b + h @ w.t()
            i, f, g, o = pre.chunk(4, 1)
            rg[t] = torch.cat([torch.sigmoid(i), torch.sigmoid(f), torch.tanh(g), torch.sigmoid(o)], 1)
            h, c = O.lstm_cell(xproj[t] + b_a + b_b, h, c, w)
            rh[t], rc[t] = h, c
        refs.append((torch.stack(rg), torch.stack(rh), torch.stack(rc)))
        specs.append(dict(xproj=dev(xproj), w_rec=dev(w), b_a=dev(b_a), b_b=dev(b_b),
                          gates=torch.empty(T, B, 4 * H, device='cuda'), hseq=hseq[:, :, d * H:(d + 1) * H],
                          cseq=torch.empty(T, B, H, device='cuda'), reverse=(d == 1)))
    ws = ops.lstm_seq_fwd(specs, T, B, H)
    torch.cuda.synchronize()
    assert int(ws[:1].view(torch.int32).item()) == 0, 'a wait inside the launch ran into its time bound'
    for d in range(D):
        rg, rh, rc = refs[d]
        assert rel_err(specs[d]['gates'], rg) < 5e-6
        assert rel_err(specs[d]['cseq'], rc) < 5e-6
        assert rel_err(hseq[:, :, d * H:(d + 1) * H], rh) < 5e-6
    # and twice in a row gives the same bits (counters and exchange buffers are re-initialised per launch)
    first = hseq.clone()
    ops.lstm_seq_fwd(specs, T, B, H)
    assert torch.equal(first, hseq)


def test_lstm_seq_launch_that_is_not_fully_resident_is_reported(ops, monkeypatch):
    """A one-launch recurrence whose workgroups are not all on the chip: the others run into the bound of their
    first wait (200 ms), the launch drains, the missing part of the outputs is NaN, the per-GPU counter is raised
    and the check the Trainer makes turns it into an error -- once: the counter is cleared by the report.  The
    missing workgroups are forced by a test knob (one slice of every group leaves at once)."""
    T, B, H = 4, 32, 128
    assert ops.lstm_seq_timeouts() == 0
    # the residency the library admits comes from the device, not from a constant
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    assert ops.lstm_resident_workgroups(0, H, 1) == cus and ops.lstm_resident_workgroups(1, H, 1) == cus
    assert ops.lstm_resident_workgroups(0, 256, 0) in (cus, 2 * cus)
    assert not ops.lstm_seq_supported(2, T, 32 * (2 * cus // 32 + 1), 256)        # one group too many
    spec = dict(xproj=dev(rnd(T, B, 4 * H, seed=1)), w_rec=dev(rnd(4 * H, H, seed=2, scale=0.08)),
                gates=torch.empty(T, B, 4 * H, device='cuda'), hseq=torch.zeros(T, B, H, device='cuda'),
                cseq=torch.zeros(T, B, H, device='cuda'))
    monkeypatch.setenv('DVAE_LSTM_SEQ_DEBUG_DROP_SLICE', '3')
    ws = ops.lstm_seq_fwd([spec], T, B, H)
    torch.cuda.synchronize()
    monkeypatch.delenv('DVAE_LSTM_SEQ_DEBUG_DROP_SLICE')
    assert int(ws[:1].view(torch.int32).item()) != 0
    h = spec['hseq']
    assert torch.isfinite(h[0, :, :24]).all() and torch.isfinite(h[0, :, 32:]).all()      # step 0 needs no hand-off
    assert torch.isnan(h[1:, :, :24]).all() and torch.isnan(h[1:, :, 32:]).all()          # nothing after it arrived
    assert ops.lstm_seq_timeouts() >= 1
    with pytest.raises(RuntimeError, match='not fully resident'):
        ops.check_lstm_seq_timeouts()
    assert ops.lstm_seq_timeouts() == 0
    ops.check_lstm_seq_timeouts()
    # and the same launch without the knob is healthy again
    ws = ops.lstm_seq_fwd([spec], T, B, H)
    torch.cuda.synchronize()
    assert int(ws[:1].view(torch.int32).item()) == 0 and torch.isfinite(spec['hseq']).all()
    assert ops.lstm_seq_timeouts() == 0


@pytest.mark.parametrize('T,B,H', [(19, 32, 256), (6, 40, 128)])
def test_lstm_seq_fwd_from_an_initial_state_without_input(ops, T, B, H):
    """the decoder's form: no input projection, strided initial state, combined weights"""
    w = rnd(4 * H, H, seed=101, scale=0.08)
    b_a, b_b = rnd(4 * H, seed=102, scale=0.1), rnd(4 * H, seed=103, scale=0.1)
    init = rnd(B, 2 * H, seed=104)                       # h0 | c0 side by side: row stride 2H
    h, c = init[:, :H].clone(), init[:, H:].clone()
    rh = []
    for t in range(T):
        h, c = O.lstm_cell((b_a + b_b).expand(B, -1), h, c, w)
        rh.append(h)
    dinit = dev(init)
    spec = dict(w_rec=dev(w), b_a=dev(b_a), b_b=dev(b_b), h_init=dinit[:, :H], c_init=dinit[:, H:],
                gates=torch.empty(T, B, 4 * H, device='cuda'), hseq=torch.empty(T, B, H, device='cuda'),
                cseq=torch.empty(T, B, H, device='cuda'))
    ws = ops.lstm_seq_fwd([spec], T, B, H)
    assert int(ws[:1].view(torch.int32).item()) == 0
    assert rel_err(spec['hseq'], torch.stack(rh)) < 5e-6
    assert rel_err(spec['cseq'][-1], c) < 5e-6


def _lstm_chain_autograd(T, B, H, reverse, with_init, seed):
    """fp64 autograd of one chain: returns the kernel's inputs (activated gates, cell states, dh_ext) and the
    reference gradients wrt gate pre-activations and the initial state"""
    xproj = rnd(T, B, 4 * H, seed=seed).requires_grad_(True)
    w = rnd(4 * H, H, seed=seed + 1, scale=0.08)
    ext = rnd(T, B, H, seed=seed + 2)
    h = rnd(B, H, seed=seed + 3).requires_grad_(True) if with_init else torch.zeros(B, H, dtype=torch.float64)
    c = rnd(B, H, seed=seed + 4).requires_grad_(True) if with_init else torch.zeros(B, H, dtype=torch.float64)
    h_init, c_init = h, c
    gates, cs, loss = [None] * T, [None] * T, 0.0
    for t in (range(T - 1, -1, -1) if reverse else range(T)):
        pre = xproj[t] + h @ w.t()
        i, f, g, o = pre.chunk(4, 1)
        i, f, g, o = torch.sigmoid(i), torch.sigmoid(f), torch.tanh(g), torch.sigmoid(o)
        c = f * c + i * g
        h = o * torch.tanh(c)
        gates[t], cs[t] = torch.cat([i, f, g, o], 1).detach(), c.detach()
        loss = loss + (h * ext[t]).sum()
    loss.backward()
    return dict(w=w, ext=ext, gates=torch.stack(gates), cs=torch.stack(cs), dgates=xproj.grad,
                h_init=h_init, c_init=c_init)


@pytest.mark.parametrize('T,B,H,chains', [(19, 32, 256, 2), (5, 70, 128, 2), (20, 256, 256, 1), (7, 3, 256, 2),
                                          (4, 33, 512, 1), (1, 32, 256, 2), (6, 256, 256, 2), (3, 100, 512, 2)])
def test_lstm_seq_bwd_whole_recurrence_in_one_launch(ops, T, B, H, chains):
    """dvae_lstm_seq_bwd_f32 against fp64 autograd: encoder-style chains from a zero state, the second one
    reversed, the outside gradient read from column slices of one [T,B,chains*H] buffer."""
    D = chains
    refs = [_lstm_chain_autograd(T, B, H, d == 1, False, 200 + 10 * d) for d in range(D)]
    dh = torch.zeros(T, B, D * H, device='cuda')
    specs = []
    for d, r in enumerate(refs):
        dh[:, :, d * H:(d + 1) * H] = dev(r['ext'])
        specs.append(dict(w_rec=dev(r['w']), gates=dev(r['gates']), cseq=dev(r['cs']),
                          dh_ext=dh[:, :, d * H:(d + 1) * H], dgates=torch.empty(T, B, 4 * H, device='cuda'),
                          reverse=(d == 1)))
    ws = ops.lstm_seq_bwd(specs, T, B, H)
    torch.cuda.synchronize()
    assert int(ws[:1].view(torch.int32).item()) == 0, 'a wait inside the launch ran into its time bound'
    for d, r in enumerate(refs):
        assert rel_err(specs[d]['dgates'], r['dgates']) < 1e-5
    first = [s['dgates'].clone() for s in specs]
    ops.lstm_seq_bwd(specs, T, B, H)
    assert all(torch.equal(a, s['dgates']) for a, s in zip(first, specs))


@pytest.mark.parametrize('T,B,H', [(19, 32, 256), (6, 40, 128), (3, 256, 256)])
def test_lstm_seq_bwd_hands_on_the_initial_state_gradients(ops, T, B, H):
    """the decoder's form: an initial state whose gradients are wanted, dh_init accumulated onto what is there"""
    r = _lstm_chain_autograd(T, B, H, False, True, 300)
    base = rnd(B, H, seed=310)
    dh_init = dev(base)
    dc_init = torch.empty(B, 2 * H, device='cuda')[:, H:]          # strided output
    spec = dict(w_rec=dev(r['w']), gates=dev(r['gates']), cseq=dev(r['cs']), c_init=dev(r['c_init'].detach()),
                dh_ext=dev(r['ext']), dgates=torch.empty(T, B, 4 * H, device='cuda'), dh_init=dh_init, dc_init=dc_init)
    ws = ops.lstm_seq_bwd([spec], T, B, H, accumulate_dh_init=True)
    assert int(ws[:1].view(torch.int32).item()) == 0
    assert rel_err(spec['dgates'], r['dgates']) < 1e-5
    assert rel_err(dh_init, base + r['h_init'].grad) < 1e-5
    assert rel_err(dc_init, r['c_init'].grad) < 1e-5
    ops.lstm_seq_bwd([spec], T, B, H, accumulate_dh_init=False)
    assert rel_err(dh_init, r['h_init'].grad) < 1e-5


@pytest.mark.parametrize('T,B,H', [(20, 32, 256), (5, 70, 128), (3, 256, 256), (1, 7, 128)])
def test_lstm_dec2_fwd_stacked_decoder_in_one_launch(ops, T, B, H):
    """dvae_lstm_dec2_fwd_f32 against the fp64 loop of two stacked cells fed with the top layer's previous output"""
    assert ops.lstm_dec2_supported(T, B, H)
    ws_ = [rnd(4 * H, H, seed=400 + k, scale=0.08) for k in range(4)]          # w_ih0, w_hh0, w_ih1, w_hh1
    bs_ = [rnd(4 * H, seed=410 + k, scale=0.1) for k in range(4)]              # b_ih0, b_hh0, b_ih1, b_hh1
    h0, c0, init = rnd(B, 2 * H, seed=420), rnd(B, 2 * H, seed=421), rnd(H, seed=422)
    h = [h0[:, :H], h0[:, H:]]
    c = [c0[:, :H], c0[:, H:]]
    x = init.expand(B, H)
    ref_h = torch.zeros(2, T, B, H, dtype=torch.float64)
    ref_c = torch.zeros(2, T, B, H, dtype=torch.float64)
    for t in range(T):
        for l in range(2):
            xin = x if l == 0 else h[0]
            proj = xin @ ws_[2 * l].t() + bs_[2 * l] + bs_[2 * l + 1]
            h[l], c[l] = O.lstm_cell(proj, h[l], c[l], ws_[2 * l + 1])
            ref_h[l, t], ref_c[l, t] = h[l], c[l]
        x = h[1]
    weights = [dev(ws_[0]), dev(ws_[1]), dev(bs_[0]), dev(bs_[1]), dev(ws_[2]), dev(ws_[3]), dev(bs_[2]), dev(bs_[3])]
    gates = torch.empty(2, T, B, 4 * H, device='cuda')
    hs, cs = torch.empty(2, T, B, H, device='cuda'), torch.empty(2, T, B, H, device='cuda')
    ws = ops.lstm_dec2_fwd(weights, dev(h0), dev(c0), dev(init), gates, hs, cs)
    torch.cuda.synchronize()
    assert int(ws[:1].view(torch.int32).item()) == 0, 'a wait inside the launch ran into its time bound'
    assert rel_err(hs, ref_h) < 1e-5 and rel_err(cs, ref_c) < 1e-5
    # the activated output gate reproduces h from c
    o = gates[:, :, :, 3 * H:]
    assert rel_err(o * torch.tanh(cs), ref_h) < 1e-5
    first = hs.clone()
    ops.lstm_dec2_fwd(weights, dev(h0), dev(c0), dev(init), gates, hs, cs)
    assert torch.equal(first, hs)


@pytest.mark.parametrize('T,B,H', [(20, 32, 256), (5, 70, 128), (3, 256, 256), (1, 7, 128)])
def test_lstm_dec2_bwd_stacked_decoder_in_one_launch(ops, T, B, H):
    """dvae_lstm_dec2_bwd_f32 against fp64 autograd of the stacked decoder loop: gate gradients of every cell,
    gradients of both initial states and of the first input."""
    ws_ = [rnd(4 * H, H, seed=430 + k, scale=0.08) for k in range(4)]          # w_ih0, w_hh0, w_ih1, w_hh1
    h0 = rnd(B, 2 * H, seed=440).requires_grad_(True)
    c0 = rnd(B, 2 * H, seed=441).requires_grad_(True)
    x0 = rnd(B, H, seed=442).requires_grad_(True)          # per-row first input (the model broadcasts one vector)
    ext = rnd(B, T, H, seed=443)
    pres = [[None] * T for _ in range(2)]
    h = [h0[:, :H], h0[:, H:]]
    c = [c0[:, :H], c0[:, H:]]
    x = x0
    gates = torch.zeros(2, T, B, 4 * H, dtype=torch.float64)
    cs = torch.zeros(2, T, B, H, dtype=torch.float64)
    loss = 0.0
    for t in range(T):
        for l in range(2):
            xin = x if l == 0 else h[0]
            pre = xin @ ws_[2 * l].t() + h[l] @ ws_[2 * l + 1].t()
            pre.retain_grad()
            pres[l][t] = pre
            i, f, g, o = pre.chunk(4, 1)
            i, f, g, o = torch.sigmoid(i), torch.sigmoid(f), torch.tanh(g), torch.sigmoid(o)
            c[l] = f * c[l] + i * g
            h[l] = o * torch.tanh(c[l])
            gates[l, t], cs[l, t] = torch.cat([i, f, g, o], 1).detach(), c[l].detach()
        x = h[1]
        loss = loss + (h[1] * ext[:, t]).sum()
    loss.backward()
    ref_dg = torch.stack([torch.stack([pres[l][t].grad for t in range(T)]) for l in range(2)])
    weights = [dev(ws_[0]), dev(ws_[1]), None, None, dev(ws_[2]), dev(ws_[3]), None, None]
    dg = torch.empty(2, T, B, 4 * H, device='cuda')
    dh0, dc0 = torch.empty(B, 2 * H, device='cuda'), torch.empty(B, 2 * H, device='cuda')
    dx0 = torch.empty(B, H, device='cuda')
    ws = ops.lstm_dec2_bwd(weights, dev(gates), dev(cs), dev(c0.detach()), dev(ext), dg, dh0, dc0, dx0)
    torch.cuda.synchronize()
    assert int(ws[:1].view(torch.int32).item()) == 0, 'a wait inside the launch ran into its time bound'
    assert rel_err(dg, ref_dg) < 1e-5
    assert rel_err(dh0, h0.grad) < 1e-5 and rel_err(dc0, c0.grad) < 1e-5 and rel_err(dx0, x0.grad) < 1e-5
    first = dg.clone()
    ops.lstm_dec2_bwd(weights, dev(gates), dev(cs), dev(c0.detach()), dev(ext), dg, dh0, dc0, dx0)
    assert torch.equal(first, dg)


def test_lstm_seq_kernels_never_ran_into_a_wait_bound(ops):
    """the per-GPU counter that outlives the launches (this module's LSTM tests have run before this one)"""
    assert ops.lstm_seq_timeouts() == 0


def test_lstm_seq_rejects_what_cannot_be_resident(ops):
    assert not ops.lstm_seq_supported(2, 19, 512, 256)           # 1024 workgroups
    assert not ops.lstm_seq_supported(1, 19, 32, 32)             # hidden size without a kernel
    assert ops.lstm_seq_supported(2, 19, 256, 256) and ops.lstm_seq_supported(1, 19, 256, 512)


@pytest.mark.parametrize('kind,tw,free_bits', [('bce', 0.05, 0.0), ('mse', 0.0, 0.0), ('mse', 0.1, 0.5)])
def test_vae_loss_in_two_launches(ops, kind, tw, free_bits):
    """dvae_vae_loss_fwd_f32 / dvae_kl_bwd2_f32 against the fp64 terms of the oracle: one latent (with the
    free-bits floor) and two latents (with an independence value handed in)."""
    B, T, D0, D1 = 5, 4, 16, 24
    g = torch.Generator().manual_seed(12)
    target = (torch.rand(B, T, 1, 64, 64, generator=g) > 0.9).float() * torch.rand(B, T, 1, 64, 64, generator=g)
    recon = torch.sigmoid(torch.randn(B, T, 1, 64, 64, generator=g) * 3)
    mu0, lv0 = rnd(B, D0, seed=13), rnd(B, D0, seed=14, scale=0.5)
    mu1, lv1 = rnd(B, D1, seed=15), rnd(B, D1, seed=16, scale=0.5)
    r64, t64 = recon.double(), target.double()
    rec_pp = O.recon_term(r64, t64, kind)
    rec = rec_pp * (T * 4096)
    tmp = O.temporal_term(r64, t64)

    def kl_terms(mu, lv, fb):
        per = -0.5 * (1 + lv - mu ** 2 - lv.exp())
        floored = torch.clamp(per, min=fb) if fb > 0 else per
        return floored.sum(1).mean(), per.sum(1).mean(), floored.mean()

    knd = 0 if kind == 'mse' else 1
    # one latent
    out, total = torch.full((7,), -1.0, device='cuda'), torch.full((1,), -1.0, device='cuda')
    ops.vae_loss_fwd(dev(recon), dev(target), knd, tw > 0, [(dev(mu0), dev(lv0), 0.4)], free_bits, tw, None, 0.0,
                     out, total)
    k, kraw, kdim = kl_terms(mu0, lv0, free_bits)
    ref = [rec + 0.4 * k + tw * (tmp if tw > 0 else 0.0), rec_pp, rec, tmp if tw > 0 else torch.tensor(0.0), k, kraw, kdim]
    o = out.cpu().double()
    for i, rv in enumerate(ref):
        assert abs(o[i] - float(rv)) <= 2e-6 * max(abs(float(rv)), 1e-3), (i, float(o[i]), float(rv))
    assert total.item() == out[0].item()
    # two latents, independence value handed in
    ind = torch.tensor([0.37], device='cuda')
    out, total = torch.full((12,), -1.0, device='cuda'), torch.full((1,), -1.0, device='cuda')
    ops.vae_loss_fwd(dev(recon), dev(target), knd, tw > 0, [(dev(mu0), dev(lv0), 0.4), (dev(mu1), dev(lv1), 4.0)], 0.0,
                     tw, ind, 0.02, out, total)
    k0, k0raw, k0dim = kl_terms(mu0, lv0, 0.0)
    k1, k1raw, k1dim = kl_terms(mu1, lv1, 0.0)
    klsum = 0.4 * k0 + 4.0 * k1
    tt = tmp if tw > 0 else torch.tensor(0.0)
    ref = [rec + klsum + tw * tt + 0.02 * 0.37, klsum, rec_pp, rec, tt, k0, k0raw, k0dim, k1, k1raw, k1dim, 0.37]
    o = out.cpu().double()
    for i, rv in enumerate(ref):
        assert abs(o[i] - float(rv)) <= 2e-6 * max(abs(float(rv)), 1e-3), (i, float(o[i]), float(rv))
    assert total.item() == out[0].item()
    # both KL gradients in one launch against the single-latent kernel
    gs = torch.tensor([0.7], device='cuda')
    (a0, b0), (a1, b1) = ops.kl_bwd2([(dev(mu0), dev(lv0), 0.4), (dev(mu1), dev(lv1), 4.0)], gs, free_bits)
    r0 = ops.kl_bwd(dev(mu0), dev(lv0), gs, 0.4, free_bits)
    r1 = ops.kl_bwd(dev(mu1), dev(lv1), gs, 4.0, free_bits)
    assert torch.equal(a0, r0[0]) and torch.equal(b0, r0[1]) and torch.equal(a1, r1[0]) and torch.equal(b1, r1[1])
    # the KL weight read from a device scalar (what a captured step uses while beta is annealed): same bits as
    # the launch constant, for either latent alone and for both, and it follows the scalar between launches
    bd = torch.tensor([4.0], device='cuda')
    for lat, dyn in (([(dev(mu0), dev(lv0), 0.4), (dev(mu1), dev(lv1), None)], bd),
                     ([(dev(mu0), dev(lv0), None), (dev(mu1), dev(lv1), None)], torch.tensor([0.4], device='cuda'))):
        out2, total2 = torch.full((12,), -1.0, device='cuda'), torch.full((1,), -1.0, device='cuda')
        ops.vae_loss_fwd(dev(recon), dev(target), knd, tw > 0, lat, 0.0, tw, ind, 0.02, out2, total2, dyn)
        if lat[0][2] is not None:
            assert torch.equal(out2, out) and torch.equal(total2, total)
            (c0, d0), (c1, d1) = ops.kl_bwd2(lat, gs, free_bits, dyn)
            assert torch.equal(c0, a0) and torch.equal(d0, b0) and torch.equal(c1, a1) and torch.equal(d1, b1)
        else:
            k01 = 0.4 * k0 + 0.4 * k1
            assert abs(out2[1].item() - float(k01)) <= 2e-6 * abs(float(k01))
            (c0, d0), (c1, d1) = ops.kl_bwd2(lat, gs, free_bits, dyn)
            r1b = ops.kl_bwd(dev(mu1), dev(lv1), gs, 0.4, free_bits)
            assert torch.equal(c0, a0) and torch.equal(c1, r1b[0]) and torch.equal(d1, r1b[1])
    bd.fill_(0.4)
    r1c = ops.kl_bwd(dev(mu1), dev(lv1), gs, None, free_bits, bd)
    r1d = ops.kl_bwd(dev(mu1), dev(lv1), gs, 0.4, free_bits)
    assert torch.equal(r1c[0], r1d[0]) and torch.equal(r1c[1], r1d[1])
    with pytest.raises(RuntimeError):
        ops.kl_bwd(dev(mu1), dev(lv1), gs, None, free_bits)


# ------------------------------------------------------------------ optimiser
def test_grad_norm_clip_adam(ops):
    g = torch.Generator().manual_seed(80)
    shapes = [(17,), (64, 33), (1,), (256, 128, 4, 4), (5, 3)] * 12      # 60 tensors: more than one launch
    params = [torch.randn(s, generator=g) for s in shapes]
    grads = [torch.randn(s, generator=g) * 3 for s in shapes]
    m = [torch.randn(s, generator=g) * 0.1 for s in shapes]
    v = [torch.rand(s, generator=g) * 0.1 for s in shapes]
    total, clipped = O.clip_grad_norm([x.double() for x in grads], 5.0)
    out = torch.zeros(2, device='cuda')
    dg = [dev(x) for x in grads]
    ops.grad_norm(dg, 5.0, out)
    assert abs(out[0].item() - total.item()) < 1e-5 * total.item()
    coef = min(1.0, 5.0 / (total.item() + 1e-6))
    assert abs(out[1].item() - coef) < 1e-5 * coef
    dp, dm, dv = [dev(x) for x in params], [dev(x) for x in m], [dev(x) for x in v]
    step, lr, b1, b2, eps, wd = 7, 3e-4, 0.9, 0.999, 1e-8, 1e-5
    ops.adam_step(dp, dg, dm, dv, lr, b1, b2, eps, wd, 1 - b1 ** step, 1 - b2 ** step, out[1:2])
    for i in range(len(shapes)):
        p2, m2, v2 = O.adam_step(params[i].double(), clipped[i], m[i].double(), v[i].double(), step, lr, b1, b2,
                                 eps, wd)
        assert rel_err(dp[i], p2) < 1e-6 and rel_err(dm[i], m2) < 1e-6 and rel_err(dv[i], v2) < 1e-6
    # in-place scaling of gradients by the device coefficient
    ops.scale_tensors(dg, out[1:2])
    for i in range(len(shapes)):
        assert rel_err(dg[i], clipped[i]) < 1e-6
    # no clipping when max_norm <= 0
    ops.grad_norm(dg, 0.0, out)
    assert out[1].item() == 1.0


# ------------------------------------------------------------------ Moving-MNIST renderer
@pytest.mark.parametrize('kwargs,seed', [(dict(sequence_length=20, frame_size=64, num_digits=2), 42),
                                         (dict(sequence_length=5, frame_size=128, num_digits=3), 0),
                                         (dict(sequence_length=40, frame_size=32, num_digits=1), 7)])
def test_device_rendered_batches_are_bit_identical(ops, kwargs, seed):
    """DeviceClipLoader (trajectories on the host, frames rendered by the kernel) against the host
    DataLoader over the same dataset, shuffled with the same torch seed: identical index order,
    bit-identical clips, identical metadata."""
    from torch.utils.data import DataLoader
    from data.moving_mnist import DeviceClipLoader, MovingMNIST, _collate_fn
    from tests.golden.make_golden import synthetic_digit_bank
    bank = synthetic_digit_bank()
    ds = MovingMNIST(num_videos=23, seed=seed, deterministic=True, digit_bank=bank, **kwargs)
    torch.manual_seed(3)
    host = list(DataLoader(ds, batch_size=5, shuffle=True, num_workers=0, collate_fn=_collate_fn))
    torch.manual_seed(3)
    dev_batches = list(DeviceClipLoader(ds, batch_size=5, shuffle=True))
    assert len(host) == len(dev_batches) == 5
    for (hv, hm), (dv, dm) in zip(host, dev_batches):
        assert dv.is_cuda and dv.dtype == torch.float32 and dv.shape == hv.shape
        assert torch.equal(dv.cpu(), hv)
        for a, b in zip(hm, dm):
            assert [int(x) for x in a['labels']] == [int(x) for x in b['labels']]
            assert all(np.array_equal(x, y) for x, y in zip(a['initial_velocities'], b['initial_velocities']))


# ------------------------------------------------------------------ BatchNorm backward folded into the consumer's loads
@pytest.mark.parametrize('N,H', [(3, 32), (70, 32)])
def test_dgrad_that_applies_the_bn_backward_of_its_input(ops, N, H):
    """conv4s2_dgrad_bnstat_bnin against the three-step path it replaces (bn_bwd from partials, then
    conv4s2_dgrad_bnstat): same dgamma/dbeta, the finished dL/dy it hands on, its own output and the
    statistics of the layer below."""
    Cin, Cout = 32, 64
    du = rnd(N, H, H, Cin, seed=90)                    # NHWC throughout
    y = rnd(N, H, H, Cin, seed=91)
    gamma, beta = 1 + 0.3 * rnd(Cin, seed=92), 0.2 * rnd(Cin, seed=93)
    mean, var = y.mean(dim=(0, 1, 2)), y.var(dim=(0, 1, 2), unbiased=False)
    invstd = 1 / torch.sqrt(var + 1e-5)
    sc, sh = gamma * invstd, beta - mean * gamma * invstd
    aff = (dev(sc), dev(sh), dev(mean), dev(invstd))
    z = y * sc + sh
    dz = du * torch.where(z > 0, 1.0, 0.2)
    xhat = (y - mean) * invstd
    M = N * H * H
    # partial rows as a producer would leave them (any split of the sums will do)
    parts = torch.stack([torch.stack([dz.sum(dim=(0, 1, 2)) * f, (dz * xhat).sum(dim=(0, 1, 2)) * f])
                         for f in (0.25, 0.75)])
    w = rnd(Cout, Cin, 4, 4, seed=94, scale=0.2)
    wp = ops.pack_conv_w(dev(w))
    y_below = rnd(N, H // 2, H // 2, Cout, seed=95)
    gb, bb = 1 + 0.3 * rnd(Cout, seed=96), 0.2 * rnd(Cout, seed=97)
    mb, vb = y_below.mean(dim=(0, 1, 2)), y_below.var(dim=(0, 1, 2), unbiased=False)
    ib = 1 / torch.sqrt(vb + 1e-5)
    aff_below = (dev(gb * ib), dev(bb - mb * gb * ib), dev(mb), dev(ib))
    assert ops.conv4s2_bnin_supported(dev(du), Cout)
    dg, db, ca, cb = ops.bn_bwd_coeffs(dev(parts), aff, M)
    got_du, got_parts, got_dx = ops.conv4s2_dgrad_bnstat_bnin(dev(du), dev(y), aff, ca, cb, wp, dev(y_below), aff_below)
    # the path it replaces
    ref_dx = dev(du)
    rg, rb = ops.bn_bwd(ref_dx, dev(y), *aff, dev(gamma), dev(parts))
    ref_du, ref_parts = ops.conv4s2_dgrad_bnstat(ref_dx, wp, dev(y_below), aff_below)
    assert rel_err(dg, rg) < 1e-6 and rel_err(db, rb) < 1e-6
    dx64 = sc * (dz - dz.sum(dim=(0, 1, 2)) / M - xhat * (dz * xhat).sum(dim=(0, 1, 2)) / M)
    assert rel_err(got_dx, dx64) < 2e-6 and rel_err(ref_dx, dx64) < 2e-6
    assert rel_err(got_du, ref_du) < 2e-6
    assert rel_err(got_parts.double().sum(0), ref_parts.double().sum(0)) < 2e-5


def _bn_case(shape, C, seed):
    """(du, y, gamma, aff on device, dz, xhat, sc) for a BatchNorm + LeakyReLU layer with NHWC tensors of `shape`"""
    du, y = rnd(*shape, seed=seed), rnd(*shape, seed=seed + 1)
    gamma, beta = 1 + 0.3 * rnd(C, seed=seed + 2), 0.2 * rnd(C, seed=seed + 3)
    mean, var = y.mean(dim=(0, 1, 2)), y.var(dim=(0, 1, 2), unbiased=False)
    invstd = 1 / torch.sqrt(var + 1e-5)
    sc, sh = gamma * invstd, beta - mean * gamma * invstd
    dz = du * torch.where(y * sc + sh > 0, 1.0, 0.2)
    return du, y, gamma, (dev(sc), dev(sh), dev(mean), dev(invstd)), dz, (y - mean) * invstd, sc


@pytest.mark.parametrize('N,H', [(3, 16), (70, 16), (5, 32)])
def test_transposed_dgrad_that_applies_the_bn_backward_of_its_input(ops, N, H):
    """deconv4s2_dgrad_bnstat_bnin (and bn_bwd_reduce feeding it) against bn_bwd + deconv4s2_dgrad_bnstat."""
    Cin, Cout = 32, 16
    du, y, gamma, aff, dz, xhat, sc = _bn_case((N, H, H, Cin), Cin, 100)
    M = N * H * H
    w = rnd(Cin, Cout, 4, 4, seed=104, scale=0.2)
    wp = ops.pack_deconv_w(dev(w))
    _, yb, _, aff_below, _, _, _ = _bn_case((N, 2 * H, 2 * H, Cout), Cout, 110)
    assert ops.deconv4s2_bnin_supported(dev(du), Cout)
    parts = ops.bn_bwd_reduce(dev(du), dev(y), aff)
    assert rel_err(parts.double().sum(0)[0], dz.sum(dim=(0, 1, 2))) < 2e-5
    assert rel_err(parts.double().sum(0)[1], (dz * xhat).sum(dim=(0, 1, 2))) < 2e-5
    dg, db, ca, cb = ops.bn_bwd_coeffs(parts, aff, M)
    got_du, got_parts, got_dx = ops.deconv4s2_dgrad_bnstat_bnin(dev(du), dev(y), aff, ca, cb, wp, dev(yb), aff_below)
    ref_dx = dev(du)
    rg, rb = ops.bn_bwd(ref_dx, dev(y), *aff, dev(gamma))
    ref_du, ref_parts = ops.deconv4s2_dgrad_bnstat(ref_dx, wp, dev(yb), aff_below)
    assert rel_err(dg, rg) < 1e-5 and rel_err(db, rb) < 1e-5
    dx64 = sc * (dz - dz.sum(dim=(0, 1, 2)) / M - xhat * (dz * xhat).sum(dim=(0, 1, 2)) / M)
    assert rel_err(got_dx, dx64) < 2e-6
    assert rel_err(got_du, ref_du) < 2e-6
    assert rel_err(got_parts.double().sum(0), ref_parts.double().sum(0)) < 2e-5


@pytest.mark.parametrize('N,Hs,C1', [(3, 32, 16), (5, 32, 32), (2, 16, 64), (40, 32, 16)])
def test_first_layer_wgrad_that_applies_the_bn_backward_of_its_gradient(ops, N, Hs, C1):
    """conv4s2_wgrad_bnp against bn_bwd followed by conv4s2_wgrad with a single-channel Q."""
    du, y, gamma, aff, dz, xhat, sc = _bn_case((N, Hs, Hs, C1), C1, 120)
    M = N * Hs * Hs
    frames = rnd(N, 2 * Hs, 2 * Hs, 1, seed=125)
    parts = ops.bn_bwd_reduce(dev(du), dev(y), aff)
    dg, db, ca, cb = ops.bn_bwd_coeffs(parts, aff, M)
    got = ops.conv4s2_wgrad_bnp(dev(du), dev(y), aff, ca, cb, dev(frames))
    ref_dx = dev(du)
    ops.bn_bwd(ref_dx, dev(y), *aff, dev(gamma))
    ref = ops.conv4s2_wgrad(ref_dx, None, None, ops.ACT_NONE, dev(frames), None, None, ops.ACT_NONE)
    dx64 = sc * (dz - dz.sum(dim=(0, 1, 2)) / M - xhat * (dz * xhat).sum(dim=(0, 1, 2)) / M)
    wz = torch.zeros(C1, 1, 4, 4, dtype=torch.float64, requires_grad=True)
    (dw64,) = torch.autograd.grad(F.conv2d(nchw(frames), wz, None, stride=2, padding=1), wz, nchw(dx64))
    assert rel_err(got, dw64) < 5e-6 and rel_err(ref, dw64) < 5e-6


@pytest.mark.parametrize('N', [3, 40])
def test_wgrad_that_applies_the_bn_backward_of_its_gradient_operand(ops, N):
    """conv4s2_wgrad_bnq (128 x 64 on 8-wide P) against bn_bwd followed by conv4s2_wgrad."""
    Hs, C1, C2 = 8, 128, 64
    du, y, gamma, aff, dz, xhat, sc = _bn_case((N, 2 * Hs, 2 * Hs, C2), C2, 130)
    M = N * 4 * Hs * Hs
    Pt = rnd(N, Hs, Hs, C1, seed=135)
    psc, psh = dev(1 + 0.3 * rnd(C1, seed=136)), dev(0.2 * rnd(C1, seed=137))
    assert ops.conv4s2_wgrad_bnq_supported(dev(Pt), C2)
    parts = ops.bn_bwd_reduce(dev(du), dev(y), aff)
    dg, db, ca, cb = ops.bn_bwd_coeffs(parts, aff, M)
    got_dw, got_dx = ops.conv4s2_wgrad_bnq(dev(Pt), psc, psh, ops.ACT_LRELU, dev(du), dev(y), aff, ca, cb)
    ref_dx = dev(du)
    ops.bn_bwd(ref_dx, dev(y), *aff, dev(gamma))
    ref_dw = ops.conv4s2_wgrad(dev(Pt), psc, psh, ops.ACT_LRELU, ref_dx, None, None, ops.ACT_NONE)
    dx64 = sc * (dz - dz.sum(dim=(0, 1, 2)) / M - xhat * (dz * xhat).sum(dim=(0, 1, 2)) / M)
    assert rel_err(got_dx, dx64) < 2e-6 and rel_err(ref_dx, dx64) < 2e-6
    assert rel_err(got_dw, ref_dw) < 5e-6


# ------------------------------------------------------------------ data gradient + BatchNorm-backward statistics
@pytest.mark.parametrize('kind,N,H,Cin,Cout', [('conv', 5, 32, 16, 32), ('conv', 3, 64, 1, 16), ('conv', 3, 64, 1, 32),
                                               ('conv', 4, 32, 32, 64), ('conv', 2, 64, 32, 64),
                                               ('conv', 2, 8, 128, 256), ('conv', 70, 4, 4, 36),
                                               ('deconv', 3, 4, 64, 32), ('deconv', 2, 8, 32, 16),
                                               ('deconv', 6, 4, 256, 128), ('deconv', 70, 4, 4, 36),
                                               ('deconv', 5, 2, 32, 16), ('deconv', 100, 32, 32, 16),
                                               ('deconv', 40, 32, 64, 32), ('deconv', 3, 16, 32, 16),
                                               ('deconv', 2, 16, 64, 32), ('deconv', 5, 8, 128, 64)])
def test_dgrad_with_fused_bn_backward_statistics(ops, kind, N, H, Cin, Cout):
    """The data-gradient kernels with the BatchNorm-backward reduction folded into their epilogue: the
    gradient they write is the plain kernel's, the partial rows sum to sum(dz) and sum(dz*xhat), and
    bn_bwd fed with them gives what the two-pass bn_bwd gives."""
    x = rnd(N, Cin, H, H, seed=50)
    if kind == 'conv':
        w = rnd(Cout, Cin, 4, 4, seed=51, scale=0.2)
        wp = ops.pack_conv_w(dev(w))
        plain, _, _ = ops.conv4s2_fwd(dev(nhwc(x)), None, None, ops.ACT_NONE, wp, False)
    else:
        w = rnd(Cin, Cout, 4, 4, seed=51, scale=0.2)
        wp = ops.pack_deconv_w(dev(w))
        plain, _, _ = ops.deconv4s2_fwd(dev(nhwc(x)), None, None, ops.ACT_NONE, wp, False)
    y = rnd(*plain.shape, seed=52)                      # raw output of the layer below, NHWC
    gamma, beta = 1 + 0.3 * rnd(Cout, seed=53), 0.2 * rnd(Cout, seed=54)
    mean = y.mean(dim=(0, 1, 2))
    var = y.var(dim=(0, 1, 2), unbiased=False)
    invstd = 1 / torch.sqrt(var + 1e-5)
    sc, sh = gamma * invstd, beta - mean * gamma * invstd
    aff = (dev(sc), dev(sh), dev(mean), dev(invstd))
    fn = ops.conv4s2_dgrad_bnstat if kind == 'conv' else ops.deconv4s2_dgrad_bnstat
    du, parts = fn(dev(nhwc(x)), wp, dev(y), aff)
    # the plain entry point may share a long K loop out over several workgroups (another order of additions)
    assert rel_err(du, plain) < 1e-6
    du64 = du.double().cpu()
    z = y * sc + sh
    dz = du64 * torch.where(z > 0, 1.0, 0.2)
    xhat = (y - mean) * invstd
    s = parts.double().sum(0).cpu()
    assert rel_err(s[0], dz.sum(dim=(0, 1, 2))) < 2e-5
    assert rel_err(s[1], (dz * xhat).sum(dim=(0, 1, 2))) < 2e-5
    if Cout % 4 == 0:
        a = du.clone()
        dg1, db1 = ops.bn_bwd(a, dev(y), *aff, dev(gamma), parts)
        b = plain.clone()
        dg2, db2 = ops.bn_bwd(b, dev(y), *aff, dev(gamma))
        assert rel_err(a, b) < 2e-6 and rel_err(dg1, dg2) < 2e-5 and rel_err(db1, db2) < 2e-5


# ------------------------------------------------------------------------------- layout helper
@pytest.mark.parametrize('dims', [(512, 256, 16), (7, 33, 65), (3, 1, 40), (5, 40, 1), (16, 20, 512), (2, 3, 6)])
@pytest.mark.parametrize('perm', [(0, 1, 2), (0, 2, 1), (1, 0, 2), (1, 2, 0), (2, 0, 1), (2, 1, 0)])
def test_permute3_is_exact(ops, dims, perm):
    """Every permutation of three axes (the tiled-transpose and row-swap paths and the generic one) is a
    pure data movement: bit-equal to torch's permute."""
    x = torch.randn(*dims, generator=torch.Generator().manual_seed(5)).cuda()
    got = ops.permute3(x, dims, perm)
    assert got.shape == tuple(dims[p] for p in perm)
    assert torch.equal(got, x.permute(*perm).contiguous())


# ------------------------------------------------------------------------------- evaluation metrics
@pytest.mark.parametrize('case', ['small', 'one_frame', 'clip', 'identical', 'half_range'])
def test_recon_metrics_kernel_matches_reference_values(ops, case):
    """One-pass mse / psnr / temporal consistency against the values the reference's own functions gave
    for the same tensors (fixture) and against the fp64 oracle; 1e-6 relative (fp32 inputs, double sums)."""
    z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'metrics.npz'))
    recon, target = torch.from_numpy(z[f'{case}/recon']), torch.from_numpy(z[f'{case}/target'])
    mv = 0.5 if case == 'half_range' else 1.0
    got = ops.recon_metrics(dev(recon), dev(target), max_val=mv).cpu().double().numpy()
    want64 = O.recon_metrics(recon.double(), target.double(), max_val=mv)
    for g, w, w64 in zip(got, z[f'{case}/expect'], want64):
        for ref_v in (float(w), float(w64)):
            if np.isnan(ref_v):
                assert np.isnan(g)
            elif np.isinf(ref_v):
                assert g == ref_v
            else:
                assert abs(g - ref_v) <= 1e-6 * abs(ref_v), (case, got, z[f'{case}/expect'], want64)


def test_evaluation_metrics_module(ops):
    """evaluation.metrics keeps the reference's function names and dictionary keys; values agree with the
    oracle on a two-batch loader of an untrained model."""
    import models
    from evaluation import metrics as M
    torch.manual_seed(3)
    model = models.get_model('baseline', in_channels=1, latent_dim=16, encoder_channels=[8, 16, 16, 32],
                             decoder_channels=[32, 16, 16, 8], feature_dim=32, sequence_length=4).cuda()
    g = torch.Generator().manual_seed(4)
    batches = [(torch.rand(2, 4, 1, 64, 64, generator=g), {}) for _ in range(2)]
    res = M.reconstruction_metrics(model, batches, torch.device('cuda'))
    assert set(res) == {'mse', 'psnr', 'ssim', 'temporal_consistency', 'mse_std', 'psnr_std', 'ssim_std'}
    model.eval()
    rows = []
    with torch.no_grad():
        for v, _ in batches:
            recon = model(v.cuda())[0].cpu()
            rows.append(O.recon_metrics(recon.double(), v.double()))
    rows = np.asarray(rows)
    assert abs(res['mse'] - rows[:, 0].mean()) <= 1e-6 * rows[:, 0].mean()
    assert abs(res['psnr'] - rows[:, 1].mean()) <= 1e-6 * abs(rows[:, 1].mean())
    assert abs(res['temporal_consistency'] - rows[:, 2].mean()) <= 1e-6 * rows[:, 2].mean()
    assert abs(res['mse_std'] - rows[:, 0].std()) <= 1e-5 * rows[:, 0].mean()
    v = batches[0][0]
    assert abs(M.temporal_consistency(v.cuda()) - O.recon_metrics(v.double(), v.double())[2]) <= 1e-6
    assert M.psnr(v.cuda(), v.cuda()) == float('inf')


@pytest.mark.parametrize('kind,tw', [(0, 0.0), (0, 0.5), (1, 0.0), (1, 0.25)])
@pytest.mark.parametrize('B,T,P', [(3, 5, 64 * 64), (2, 1, 128), (5, 7, 36)])
def test_recon_gradient_in_logit_form_is_the_two_pass_result(ops, kind, tw, B, T, P):
    torch.manual_seed(11)
    recon = torch.sigmoid(3 * torch.randn(B, T, P, device='cuda'))
    target = torch.rand(B, T, P, device='cuda')
    gs = torch.tensor([0.37], device='cuda')
    d = ops.recon_loss_bwd(recon, target, gs, kind, tw)
    want, want_b = ops.sigmoid_bwd(d, recon)
    got, got_b = ops.recon_loss_bwd_logit(recon, target, gs, kind, tw)
    assert torch.equal(got, want)
    ref_b = want.double().sum().item()
    assert abs(got_b.item() - ref_b) <= 1e-5 * max(abs(ref_b), want.double().abs().sum().item() * 1e-2)
    assert abs(want_b.item() - ref_b) <= 1e-5 * max(abs(ref_b), want.double().abs().sum().item() * 1e-2)
