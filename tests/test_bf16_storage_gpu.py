"""bf16 storage of the decoder's largest tensors (bf16-MFMA mode): every kernel that reads or writes a tensor as
bfloat16 against the same kernel on fp32 storage holding the same (bf16-representable) values.

What a kernel READS as bf16 it converts to the bf16 MFMA operand it would have made of the fp32 value anyway, so
results on bf16-representable inputs are required to be EQUAL; what it WRITES as bf16 is the round-to-nearest-even
of the fp32 value it would have written (checked with torch's own conversion), and the BatchNorm statistics it
reduces beside the store come from the fp32 accumulators, so they are required to be equal too.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ops():
    from hipvae import ops as _ops
    assert torch.cuda.is_available()
    _ops.set_precision('bf16')
    yield _ops
    _ops.set_precision('fp32')


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).cuda()


def aff_of(C, seed):
    gamma, beta = 1 + 0.3 * rnd(C, seed=seed), 0.2 * rnd(C, seed=seed + 1)
    mean, invstd = 0.1 * rnd(C, seed=seed + 2), 1 + 0.2 * rnd(C, seed=seed + 3).abs()
    return (gamma * invstd).contiguous(), (beta - mean * gamma * invstd).contiguous(), mean, invstd


@pytest.mark.parametrize('N', [3, 70])
def test_last_block_writes_and_reads_bf16(ops, N):
    x = rnd(N, 16, 16, 64, seed=1)
    sc, sh, _, _ = aff_of(64, 2)
    wp = ops.pack_deconv_w(rnd(64, 32, 4, 4, seed=6, scale=0.2))
    y32, s32, n32 = ops.deconv4s2_fwd(x, sc, sh, ops.ACT_LRELU, wp, True)
    y16, s16, n16 = ops.deconv4s2_fwd(x, sc, sh, ops.ACT_LRELU, wp, True, out_bf16=True)
    assert y16.dtype == torch.bfloat16 and y16.shape == y32.shape and n16 == n32
    assert torch.equal(y16, y32.bfloat16())
    assert torch.equal(s16, s32)


@pytest.mark.parametrize('N,Cin', [(3, 32), (70, 32)])
def test_final_layer_reads_bf16(ops, N, Cin):
    H = 32
    x = rnd(N, H, H, Cin, seed=7).bfloat16()
    sc, sh, _, _ = aff_of(Cin, 8)
    w, b = rnd(Cin, 1, 4, 4, seed=12, scale=0.3), rnd(1, seed=13)
    got = ops.deconv4s2_final_fwd(x, sc, sh, ops.ACT_LRELU, w, b, ops.ACT_SIGMOID)
    ref = ops.deconv4s2_final_fwd(x.float(), sc, sh, ops.ACT_LRELU, w, b, ops.ACT_SIGMOID)
    assert torch.equal(got, ref)


@pytest.mark.parametrize('N', [3, 70])
def test_final_layer_backward_on_bf16(ops, N):
    """weight gradient of the one-channel layer with its input stored as bf16, and its data gradient written as bf16
    beside the BatchNorm-backward sums of the block below"""
    y6 = rnd(N, 32, 32, 32, seed=14).bfloat16()
    aff = aff_of(32, 15)
    dlogit = rnd(N, 64, 64, 1, seed=19, scale=0.1)
    w = rnd(32, 1, 4, 4, seed=20, scale=0.3)
    dw16 = ops.conv4s2_wgrad(y6, aff[0], aff[1], ops.ACT_LRELU, dlogit, None, None, ops.ACT_NONE)
    dw32 = ops.conv4s2_wgrad(y6.float(), aff[0], aff[1], ops.ACT_LRELU, dlogit, None, None, ops.ACT_NONE)
    assert torch.equal(dw16, dw32)
    du16, p16 = ops.conv4s2_dgrad_bnstat(dlogit, w, y6, aff, out_bf16=True)
    du32, p32 = ops.conv4s2_dgrad_bnstat(dlogit, w, y6.float(), aff)
    assert du16.dtype == torch.bfloat16
    assert torch.equal(du16, du32.bfloat16())
    assert torch.equal(p16, p32)


@pytest.mark.parametrize('N', [3, 70])
def test_last_block_backward_on_bf16(ops, N):
    """the data-gradient kernel that takes du through the block's BatchNorm backward, on bf16 du and y, handing a
    bf16 dL/dy to the weight-gradient kernel"""
    du = rnd(N, 32, 32, 32, seed=21).bfloat16()
    y6 = rnd(N, 32, 32, 32, seed=22).bfloat16()
    aff = aff_of(32, 23)
    ca, cb = 0.01 * rnd(32, seed=27), 0.01 * rnd(32, seed=28)
    wp = ops.pack_conv_w(rnd(64, 32, 4, 4, seed=29, scale=0.2))
    y3 = rnd(N, 16, 16, 64, seed=30)
    aff3 = aff_of(64, 31)
    assert ops.conv4s2_bnin_supported(du, 64)
    g_du, g_parts, g_dx = ops.conv4s2_dgrad_bnstat_bnin(du, y6, aff, ca, cb, wp, y3, aff3)
    r_du, r_parts, r_dx = ops.conv4s2_dgrad_bnstat_bnin(du.float(), y6.float(), aff, ca, cb, wp, y3, aff3)
    assert g_dx.dtype == torch.bfloat16 and g_du.dtype == torch.float32
    assert torch.equal(g_dx, r_dx.bfloat16())
    assert torch.equal(g_du, r_du) and torch.equal(g_parts, r_parts)
    dw16 = ops.conv4s2_wgrad(y3, aff3[0], aff3[1], ops.ACT_LRELU, g_dx, None, None, ops.ACT_NONE)
    dw32 = ops.conv4s2_wgrad(y3, aff3[0], aff3[1], ops.ACT_LRELU, g_dx.float(), None, None, ops.ACT_NONE)
    assert torch.equal(dw16, dw32)


@pytest.mark.parametrize('N', [3, 70])
def test_decoder_64_channel_block_on_bf16(ops, N):
    """128 -> 64 on 8 x 8 written as bf16; read by the 64 -> 32 block (forward, weight gradient, BatchNorm-backward sums
    of its data gradient, which then writes a bf16 du) and by the 128 x 64 weight gradient that folds the BatchNorm
    backward of that du in"""
    x = rnd(N, 8, 8, 128, seed=80)
    sc, sh, _, _ = aff_of(128, 81)
    wp = ops.pack_deconv_w(rnd(128, 64, 4, 4, seed=85, scale=0.1))
    y32, s32, n32 = ops.deconv4s2_fwd(x, sc, sh, ops.ACT_LRELU, wp, True)
    y16, s16, n16 = ops.deconv4s2_fwd(x, sc, sh, ops.ACT_LRELU, wp, True, out_bf16=True)
    assert y16.dtype == torch.bfloat16 and n16 == n32
    assert torch.equal(y16, y32.bfloat16()) and torch.equal(s16, s32)
    # forward of the next block: bf16 in, bf16 out
    aff3 = aff_of(64, 86)
    wp2 = ops.pack_deconv_w(rnd(64, 32, 4, 4, seed=90, scale=0.2))
    a, sa, _ = ops.deconv4s2_fwd(y16, aff3[0], aff3[1], ops.ACT_LRELU, wp2, True, out_bf16=True)
    b, sb, _ = ops.deconv4s2_fwd(y16.float(), aff3[0], aff3[1], ops.ACT_LRELU, wp2, True, out_bf16=True)
    assert torch.equal(a, b) and torch.equal(sa, sb)
    # its weight gradient: bf16 P and bf16 Q
    dy6 = rnd(N, 32, 32, 32, seed=91).bfloat16()
    dw16 = ops.conv4s2_wgrad(y16, aff3[0], aff3[1], ops.ACT_LRELU, dy6, None, None, ops.ACT_NONE)
    dw32 = ops.conv4s2_wgrad(y16.float(), aff3[0], aff3[1], ops.ACT_LRELU, dy6, None, None, ops.ACT_NONE)
    assert torch.equal(dw16, dw32)
    # its data gradient: everything bf16
    du6 = rnd(N, 32, 32, 32, seed=92).bfloat16()
    y6 = rnd(N, 32, 32, 32, seed=93).bfloat16()
    aff6 = aff_of(32, 94)
    ca, cb = 0.01 * rnd(32, seed=98), 0.01 * rnd(32, seed=99)
    wpc = ops.pack_conv_w(rnd(64, 32, 4, 4, seed=100, scale=0.2))
    g_du, g_parts, g_dx = ops.conv4s2_dgrad_bnstat_bnin(du6, y6, aff6, ca, cb, wpc, y16, aff3)
    r_du, r_parts, r_dx = ops.conv4s2_dgrad_bnstat_bnin(du6, y6, aff6, ca, cb, wpc, y16.float(), aff3)
    assert g_du.dtype == torch.bfloat16 and r_du.dtype == torch.float32
    assert torch.equal(g_du, r_du.bfloat16()) and torch.equal(g_parts, r_parts) and torch.equal(g_dx, r_dx)
    # weight gradient of the 128 -> 64 block on that du and y
    assert ops.conv4s2_wgrad_bnq_supported(x, 64)
    ca3, cb3 = 0.01 * rnd(64, seed=101), 0.01 * rnd(64, seed=102)
    dwa, dxa = ops.conv4s2_wgrad_bnq(x, sc, sh, ops.ACT_LRELU, g_du, y16, aff3, ca3, cb3)
    dwb, dxb = ops.conv4s2_wgrad_bnq(x, sc, sh, ops.ACT_LRELU, g_du.float(), y16.float(), aff3, ca3, cb3)
    assert dxa.dtype == torch.float32
    assert torch.equal(dwa, dwb) and torch.equal(dxa, dxb)


@pytest.mark.parametrize('N', [3, 70])
def test_few_channel_encoder_forward_on_bf16(ops, N):
    """1 -> 16 channels written as bf16, 16 -> 32 reading it"""
    x = rnd(N, 64, 64, 1, seed=50).abs()
    wp0 = ops.pack_conv_w(rnd(16, 1, 4, 4, seed=51, scale=0.3))
    y32, s32, n32 = ops.conv4s2_fwd(x, None, None, ops.ACT_NONE, wp0, True)
    y16, s16, n16 = ops.conv4s2_fwd(x, None, None, ops.ACT_NONE, wp0, True, out_bf16=True)
    assert y16.dtype == torch.bfloat16 and n16 == n32
    assert torch.equal(y16, y32.bfloat16()) and torch.equal(s16, s32)
    sc, sh, _, _ = aff_of(16, 52)
    wp1 = ops.pack_conv_w(rnd(32, 16, 4, 4, seed=56, scale=0.2))
    a, sa, _ = ops.conv4s2_fwd(y16, sc, sh, ops.ACT_LRELU, wp1, True)
    b, sb, _ = ops.conv4s2_fwd(y16.float(), sc, sh, ops.ACT_LRELU, wp1, True)
    assert a.dtype == torch.float32 and torch.equal(a, b) and torch.equal(sa, sb)


@pytest.mark.parametrize('N', [3, 70])
def test_few_channel_encoder_backward_on_bf16(ops, N):
    """the 32 -> 16 data gradient (BatchNorm backward of its input folded in) reading the first block's bf16 output
    and writing a bf16 du; the 32 x 16 weight gradient with a bf16 Q; the first layer's weight gradient on bf16 du, y"""
    du2 = rnd(N, 16, 16, 32, seed=60)
    y2 = rnd(N, 16, 16, 32, seed=61)
    aff2 = aff_of(32, 62)
    ca, cb = 0.01 * rnd(32, seed=66), 0.01 * rnd(32, seed=67)
    wpd = ops.pack_deconv_w(rnd(32, 16, 4, 4, seed=68, scale=0.2))
    y1 = rnd(N, 32, 32, 16, seed=69).bfloat16()
    aff1 = aff_of(16, 70)
    assert ops.deconv4s2_bnin_supported(du2, 16)
    g_du, g_parts, g_dy = ops.deconv4s2_dgrad_bnstat_bnin(du2, y2, aff2, ca, cb, wpd, y1, aff1)
    r_du, r_parts, r_dy = ops.deconv4s2_dgrad_bnstat_bnin(du2, y2, aff2, ca, cb, wpd, y1.float(), aff1)
    assert g_du.dtype == torch.bfloat16 and g_dy.dtype == torch.float32
    assert torch.equal(g_du, r_du.bfloat16()) and torch.equal(g_parts, r_parts) and torch.equal(g_dy, r_dy)
    dw16 = ops.conv4s2_wgrad(g_dy, None, None, ops.ACT_NONE, y1, aff1[0], aff1[1], ops.ACT_LRELU)
    dw32 = ops.conv4s2_wgrad(g_dy, None, None, ops.ACT_NONE, y1.float(), aff1[0], aff1[1], ops.ACT_LRELU)
    assert torch.equal(dw16, dw32)
    frames = rnd(N, 64, 64, 1, seed=74).abs()
    ca1, cb1 = 0.01 * rnd(16, seed=75), 0.01 * rnd(16, seed=76)
    a = ops.conv4s2_wgrad_bnp(g_du, y1, aff1, ca1, cb1, frames)
    b = ops.conv4s2_wgrad_bnp(g_du.float(), y1.float(), aff1, ca1, cb1, frames)
    assert torch.equal(a, b)


def test_bf16_tensors_are_refused_where_no_kernel_takes_them(ops):
    x = rnd(3, 4, 4, 256, seed=40).bfloat16()
    wp = ops.pack_deconv_w(rnd(256, 128, 4, 4, seed=41, scale=0.1))
    with pytest.raises(RuntimeError):
        ops.deconv4s2_fwd(x, None, None, ops.ACT_NONE, wp, False)
    # the 64 -> 32 row kernel reads bf16 only where it also writes it, and the 64-channel final layer not at all
    x = rnd(3, 16, 16, 64, seed=44).bfloat16()
    wp = ops.pack_deconv_w(rnd(64, 32, 4, 4, seed=45, scale=0.1))
    with pytest.raises(RuntimeError):
        ops.deconv4s2_fwd(x, None, None, ops.ACT_NONE, wp, False)
    with pytest.raises(RuntimeError):
        ops.deconv4s2_final_fwd(x, None, None, ops.ACT_NONE, rnd(64, 1, 4, 4, seed=46), rnd(1, seed=47), ops.ACT_SIGMOID)
    ops.set_precision('fp32')
    x = rnd(3, 16, 16, 64, seed=42)
    wp = ops.pack_deconv_w(rnd(64, 32, 4, 4, seed=43, scale=0.1))
    with pytest.raises(RuntimeError):
        ops.deconv4s2_fwd(x, None, None, ops.ACT_NONE, wp, False, out_bf16=True)


def test_step_on_bf16_storage_follows_the_fp32_storage_step(ops, monkeypatch):
    """DisentangledVAE training step in bf16 mode with and without bf16 storage: same loss to 1e-3 relative, every
    gradient within 1e-1 relative L2 of the other -- the scale of the bf16 mode's own error at this batch size (a
    handful of clips); what the mode is held to is the fp64 oracle at batch 256, tests/test_bench_size_gpu.py."""
    import models
    from hipvae import losses

    def run(storage):
        monkeypatch.setattr(ops, 'USE_BF16_STORAGE', storage)
        torch.manual_seed(5)
        model = models.DisentangledVAE().cuda().train()
        v = torch.rand(4, 20, 1, 64, 64, generator=torch.Generator().manual_seed(9)).cuda()
        torch.manual_seed(11)                           # (the reparameterisation draws)
        out = model(v)
        loss = losses.disentangled_loss(out, v, beta=2.0, beta_content=2.0, beta_motion=1.0)['total_loss']
        loss.backward()
        return float(loss.detach()), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}

    l0, g0 = run(False)
    l1, g1 = run(True)
    assert abs(l1 - l0) <= 1e-3 * abs(l0)
    assert g0.keys() == g1.keys()
    assert any(not torch.equal(g0[n], g1[n]) for n in g0)        # (the second run did take the bf16 route)
    errs = {n: float((g1[n] - g0[n]).norm() / g0[n].norm().clamp_min(1e-30)) for n in g0}
    worst = max(errs, key=errs.get)
    print('bf16 storage on vs off, largest gradient differences:', sorted(errs.items(), key=lambda kv: -kv[1])[:5])
    assert errs[worst] < 1e-1, (worst, errs[worst])
