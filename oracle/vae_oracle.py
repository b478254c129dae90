"""CPU oracle for the video-VAE training / inference hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package may import this
module; it is used by tests/, by __graft_entry__.smoke() and by bench.py's
cpu_baseline leg as the checker the HIP path is compared against.

Every function is a functional restatement (weights come in as a flat
``state_dict``-style mapping, nothing is an nn.Module) of one piece of the
reference algorithm and cites the reference file:line it follows.  All
arithmetic runs in the dtype of the tensors handed in, so the same code gives
the fp32 restatement and the fp64 arbiter.

Parity pin: tests/test_oracle_golden.py checks these functions against vectors
produced by importing the reference itself (tests/golden/make_golden.py).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]

LRELU_SLOPE = 0.2      # models/encoders.py:36, models/decoders.py:45
BN_EPS = 1e-5          # nn.BatchNorm2d default used at models/encoders.py:35
BN_MOMENTUM = 0.1


# ---------------------------------------------------------------------------
# building blocks
# ---------------------------------------------------------------------------
def leaky_relu(x: Tensor) -> Tensor:
    return torch.where(x > 0, x, x * LRELU_SLOPE)


def batch_norm(x: Tensor, sd: SD, prefix: str, training: bool,
               new_stats: Optional[SD]) -> Tensor:
    """BatchNorm2d as used at models/encoders.py:35,79 and models/decoders.py:44.

    Training: biased batch variance normalises, the running variance is updated
    with the unbiased one, momentum 0.1.  Eval: running statistics.
    ``new_stats`` (if given) receives the post-update buffers; ``sd`` is never
    modified.
    """
    gamma, beta = sd[prefix + '.weight'], sd[prefix + '.bias']
    if training:
        n = x.numel() // x.shape[1]
        mean = x.mean(dim=(0, 2, 3))
        var_b = ((x - mean[None, :, None, None]) ** 2).mean(dim=(0, 2, 3))
        if new_stats is not None:
            var_u = var_b * (n / max(n - 1, 1))
            rm = sd[prefix + '.running_mean'].to(x.dtype)
            rv = sd[prefix + '.running_var'].to(x.dtype)
            new_stats[prefix + '.running_mean'] = ((1 - BN_MOMENTUM) * rm + BN_MOMENTUM * mean).detach()
            new_stats[prefix + '.running_var'] = ((1 - BN_MOMENTUM) * rv + BN_MOMENTUM * var_u).detach()
            new_stats[prefix + '.num_batches_tracked'] = sd[prefix + '.num_batches_tracked'] + 1
    else:
        mean = sd[prefix + '.running_mean'].to(x.dtype)
        var_b = sd[prefix + '.running_var'].to(x.dtype)
    inv = torch.rsqrt(var_b + BN_EPS)
    return (x - mean[None, :, None, None]) * (inv * gamma)[None, :, None, None] + beta[None, :, None, None]


def _conv_indices(sd: SD, prefix: str) -> List[int]:
    """Sequential indices of the conv layers: 0,3,6,... (conv, bn, act triples)."""
    idx = []
    i = 0
    while f'{prefix}.{i}.weight' in sd:
        idx.append(i)
        i += 3
    return idx


def conv_stack(x: Tensor, sd: SD, prefix: str, training: bool, new_stats: Optional[SD]) -> Tensor:
    """[Conv2d(k4,s2,p1,bias=False) -> BN -> LeakyReLU(0.2)]* ; models/encoders.py:28-39,72-83."""
    h = x
    for i in _conv_indices(sd, prefix):
        h = F.conv2d(h, sd[f'{prefix}.{i}.weight'], None, stride=2, padding=1)
        h = batch_norm(h, sd, f'{prefix}.{i + 1}', training, new_stats)
        h = leaky_relu(h)
    return h


def cnn_encoder(x: Tensor, sd: SD, prefix: str, training: bool, new_stats: Optional[SD]) -> Tensor:
    """CNNEncoder / LightCNNEncoder.forward; models/encoders.py:48-53,92-97.

    Frames (N,C,64,64) -> conv stack -> NCHW flatten -> Linear -> LeakyReLU.
    """
    h = conv_stack(x, sd, prefix + '.conv', training, new_stats)
    h = h.reshape(h.shape[0], -1)
    return leaky_relu(F.linear(h, sd[prefix + '.fc.0.weight'], sd[prefix + '.fc.0.bias']))


def lstm_cell(x_proj: Tensor, h: Tensor, c: Tensor, w_hh: Tensor) -> Tuple[Tensor, Tensor]:
    """One nn.LSTM cell step, gates packed [i,f,g,o]; x_proj already holds W_ih x + b_ih + b_hh."""
    gates = x_proj + h @ w_hh.t()
    i, f, g, o = gates.chunk(4, dim=1)
    i, f, g, o = torch.sigmoid(i), torch.sigmoid(f), torch.tanh(g), torch.sigmoid(o)
    c2 = f * c + i * g
    h2 = o * torch.tanh(c2)
    return h2, c2


def lstm_direction(x: Tensor, sd: SD, prefix: str, layer: int, reverse: bool) -> Tuple[Tensor, Tensor, Tensor]:
    """One direction of one nn.LSTM layer from zero state over x (B,T,I) -> (out (B,T,H), h_T, c_T)."""
    sfx = f'_l{layer}' + ('_reverse' if reverse else '')
    w_ih, w_hh = sd[prefix + '.weight_ih' + sfx], sd[prefix + '.weight_hh' + sfx]
    bias = sd[prefix + '.bias_ih' + sfx] + sd[prefix + '.bias_hh' + sfx]
    B, T, _ = x.shape
    H = w_hh.shape[1]
    h = x.new_zeros(B, H)
    c = x.new_zeros(B, H)
    proj = x @ w_ih.t() + bias
    outs: List[Optional[Tensor]] = [None] * T
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        h, c = lstm_cell(proj[:, t], h, c, w_hh)
        outs[t] = h
    return torch.stack(outs, dim=1), h, c


def lstm_encoder(x: Tensor, sd: SD, prefix: str, num_layers: int, bidirectional: bool) -> Tensor:
    """LSTMEncoder.forward with return_sequence=False; models/encoders.py:156-173.

    Returns cat(h_n[-2], h_n[-1]) for bidirectional, h_n[-1] otherwise.
    """
    inp = x
    finals = []
    for layer in range(num_layers):
        out_f, h_f, _ = lstm_direction(inp, sd, prefix, layer, False)
        if bidirectional:
            out_b, h_b, _ = lstm_direction(inp, sd, prefix, layer, True)
            inp = torch.cat([out_f, out_b], dim=2)
            finals = [h_f, h_b]
        else:
            inp = out_f
            finals = [h_f]
    return torch.cat(finals, dim=1)


def vae_head(h: Tensor, sd: SD, prefix: str) -> Tuple[Tensor, Tensor]:
    """VAEEncoder.forward with hidden_dim=None; models/encoders.py:200-204."""
    mu = F.linear(h, sd[prefix + '.fc_mu.weight'], sd[prefix + '.fc_mu.bias'])
    logvar = F.linear(h, sd[prefix + '.fc_logvar.weight'], sd[prefix + '.fc_logvar.bias'])
    return mu, logvar


def reparameterize(mu: Tensor, logvar: Tensor, eps: Optional[Tensor]) -> Tensor:
    """models/encoders.py:207-214 with the normal draw made explicit; eps=None is the no-grad branch."""
    if eps is None:
        return mu
    return mu + eps * torch.exp(0.5 * logvar)


def lstm_decoder_states(z: Tensor, sd: SD, prefix: str, num_layers: int, T: int) -> Tensor:
    """Autoregressive decoder recurrence; models/decoders.py:118-134 and :253-269.

    h0,c0 = view(B,L,H) of two Linear outputs; x_0 = initial_input, x_t = top-layer h_{t-1}.
    Returns the stacked top-layer outputs (B,T,H).
    """
    B = z.shape[0]
    h0 = F.linear(z, sd[prefix + '.fc_hidden.weight'], sd[prefix + '.fc_hidden.bias'])
    c0 = F.linear(z, sd[prefix + '.fc_cell.weight'], sd[prefix + '.fc_cell.bias'])
    H = h0.shape[1] // num_layers
    hs = [h0[:, l * H:(l + 1) * H] for l in range(num_layers)]
    cs = [c0[:, l * H:(l + 1) * H] for l in range(num_layers)]
    x = sd[prefix + '.initial_input'].expand(B, -1)
    outs = []
    for _ in range(T):
        inp = x
        for l in range(num_layers):
            w_ih, w_hh = sd[f'{prefix}.lstm.weight_ih_l{l}'], sd[f'{prefix}.lstm.weight_hh_l{l}']
            bias = sd[f'{prefix}.lstm.bias_ih_l{l}'] + sd[f'{prefix}.lstm.bias_hh_l{l}']
            hs[l], cs[l] = lstm_cell(inp @ w_ih.t() + bias, hs[l], cs[l], w_hh)
            inp = hs[l]
        outs.append(inp)
        x = inp
    return torch.stack(outs, dim=1)


def cnn_decoder(feat: Tensor, sd: SD, prefix: str, training: bool, new_stats: Optional[SD]) -> Tensor:
    """CNNDecoder.forward; models/decoders.py:59-70.

    Linear+LeakyReLU -> view(N,C0,4,4) -> [ConvT(k4,s2,p1)+BN+LeakyReLU]* -> ConvT(bias) -> sigmoid.
    """
    h = leaky_relu(F.linear(feat, sd[prefix + '.fc.0.weight'], sd[prefix + '.fc.0.bias']))
    c0 = sd[prefix + '.deconv.0.weight'].shape[0] if (prefix + '.deconv.0.weight') in sd \
        else sd[prefix + '.final.weight'].shape[0]
    h = h.reshape(h.shape[0], c0, 4, 4)
    for i in _conv_indices(sd, prefix + '.deconv'):
        h = F.conv_transpose2d(h, sd[f'{prefix}.deconv.{i}.weight'], None, stride=2, padding=1)
        h = batch_norm(h, sd, f'{prefix}.deconv.{i + 1}', training, new_stats)
        h = leaky_relu(h)
    h = F.conv_transpose2d(h, sd[prefix + '.final.weight'], sd[prefix + '.final.bias'], stride=2, padding=1)
    return torch.sigmoid(h)


# ---------------------------------------------------------------------------
# the three models
# ---------------------------------------------------------------------------
def split_mean_diff(video: Tensor) -> Tuple[Tensor, Tensor]:
    """Temporal mean (models/encoders.py:251) and frame differences (models/encoders.py:303)."""
    return video.mean(dim=1), video[:, 1:] - video[:, :-1]


def disentangled_forward(sd: SD, video: Tensor, eps_c: Optional[Tensor], eps_m: Optional[Tensor],
                         training: bool, lstm_layers: int = 1, bidirectional: bool = True,
                         aggregation: str = 'mean',
                         new_stats: Optional[SD] = None) -> Dict[str, Tensor]:
    """DisentangledVAE.forward; models/disentangled_vae.py:82-116 (content draw first, then motion)."""
    B, T = video.shape[:2]
    mean, diff = split_mean_diff(video)
    if aggregation == 'mean':
        frame = mean
    elif aggregation == 'middle':                                   # models/encoders.py:252-253
        frame = video[:, T // 2]
    else:
        raise ValueError(f"Unknown aggregation: {aggregation}")
    feat_c = cnn_encoder(frame, sd, 'content_encoder.cnn_encoder', training, new_stats)
    mu_c, lv_c = vae_head(feat_c, sd, 'content_encoder.vae_head')
    z_c = reparameterize(mu_c, lv_c, eps_c)

    d = diff.reshape(B * (T - 1), *diff.shape[2:])
    feat_m = cnn_encoder(d, sd, 'motion_encoder.cnn_encoder', training, new_stats).reshape(B, T - 1, -1)
    h_m = lstm_encoder(feat_m, sd, 'motion_encoder.lstm.lstm', lstm_layers, bidirectional)
    mu_m, lv_m = vae_head(h_m, sd, 'motion_encoder.vae_head')
    z_m = reparameterize(mu_m, lv_m, eps_m)

    recon = disentangled_decode(sd, z_c, z_m, T, training, lstm_layers, new_stats)
    return {'recon': recon, 'z_content': z_c, 'z_motion': z_m, 'mu_content': mu_c,
            'logvar_content': lv_c, 'mu_motion': mu_m, 'logvar_motion': lv_m}


def disentangled_decode(sd: SD, z_c: Tensor, z_m: Tensor, T: int, training: bool,
                        lstm_layers: int = 1, new_stats: Optional[SD] = None) -> Tensor:
    """DisentangledDecoder.forward; models/decoders.py:242-281."""
    B = z_c.shape[0]
    states = lstm_decoder_states(torch.cat([z_c, z_m], dim=-1), sd, 'decoder', lstm_layers, T)
    ff = leaky_relu(F.linear(states, sd['decoder.fc_frame.0.weight'], sd['decoder.fc_frame.0.bias']))
    ff = torch.cat([ff, z_c[:, None, :].expand(-1, T, -1)], dim=-1).reshape(B * T, -1)
    frames = cnn_decoder(ff, sd, 'decoder.cnn_decoder', training, new_stats)
    return frames.reshape(B, T, *frames.shape[1:])


def lstm_vae_forward(sd: SD, video: Tensor, eps: Optional[Tensor], training: bool,
                     lstm_layers: int = 1, bidirectional: bool = True,
                     new_stats: Optional[SD] = None) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """LSTMVAE.forward; models/lstm_vae.py:70-112."""
    B, T = video.shape[:2]
    feats = cnn_encoder(video.reshape(B * T, *video.shape[2:]), sd, 'cnn_encoder', training, new_stats)
    h = lstm_encoder(feats.reshape(B, T, -1), sd, 'lstm_encoder.lstm', lstm_layers, bidirectional)
    mu, logvar = vae_head(h, sd, 'vae_head')
    z = reparameterize(mu, logvar, eps)
    recon = lstm_vae_decode(sd, z, T, training, lstm_layers, new_stats)
    return recon, z, mu, logvar


def lstm_vae_decode(sd: SD, z: Tensor, T: int, training: bool, lstm_layers: int = 1,
                    new_stats: Optional[SD] = None) -> Tensor:
    """LSTMVAE.decode; models/lstm_vae.py:88-103 with LSTMDecoder.forward models/decoders.py:110-137."""
    B = z.shape[0]
    states = lstm_decoder_states(z, sd, 'lstm_decoder', lstm_layers, T)
    ff = leaky_relu(F.linear(states, sd['lstm_decoder.fc_out.0.weight'], sd['lstm_decoder.fc_out.0.bias']))
    frames = cnn_decoder(ff.reshape(B * T, -1), sd, 'cnn_decoder', training, new_stats)
    return frames.reshape(B, T, *frames.shape[1:])


def baseline_forward(sd: SD, video: Tensor, eps: Optional[Tensor], training: bool,
                     new_stats: Optional[SD] = None) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """BaselineVAE.forward; models/baseline_vae.py:60-142 (per-frame mu/logvar averaged over T)."""
    B, T = video.shape[:2]
    feats = cnn_encoder(video.reshape(B * T, *video.shape[2:]), sd, 'cnn_encoder', training, new_stats)
    fmu, flv = vae_head(feats, sd, 'vae_head')
    mu = fmu.reshape(B, T, -1).mean(dim=1)
    logvar = flv.reshape(B, T, -1).mean(dim=1)
    z = reparameterize(mu, logvar, eps)
    recon = baseline_decode(sd, z, T, training, new_stats)
    return recon, z, mu, logvar


def baseline_decode(sd: SD, z: Tensor, T: int, training: bool, new_stats: Optional[SD] = None) -> Tensor:
    """BaselineVAE.decode; models/baseline_vae.py:96-121 (the same z for every frame)."""
    B = z.shape[0]
    zf = z[:, None, :].expand(-1, T, -1).reshape(B * T, -1)
    frames = cnn_decoder(zf, sd, 'cnn_decoder', training, new_stats)
    return frames.reshape(B, T, *frames.shape[1:])


# ---------------------------------------------------------------------------
# the methods evaluation code and notebooks call besides forward()
# (all under no_grad in the reference: reparameterize returns mu; BatchNorm follows `training`)
# ---------------------------------------------------------------------------
def disentangled_encode(sd: SD, video: Tensor, training: bool, lstm_layers: int = 1, bidirectional: bool = True,
                        aggregation: str = 'mean', new_stats: Optional[SD] = None):
    """DisentangledVAE.encode under no_grad; models/disentangled_vae.py:82-91 -> (mu_content, mu_motion)."""
    B, T = video.shape[:2]
    mean, diff = split_mean_diff(video)
    frame = mean if aggregation == 'mean' else video[:, T // 2]
    mu_c, _ = vae_head(cnn_encoder(frame, sd, 'content_encoder.cnn_encoder', training, new_stats), sd,
                       'content_encoder.vae_head')
    feat_m = cnn_encoder(diff.reshape(B * (T - 1), *diff.shape[2:]), sd, 'motion_encoder.cnn_encoder', training,
                         new_stats).reshape(B, T - 1, -1)
    mu_m, _ = vae_head(lstm_encoder(feat_m, sd, 'motion_encoder.lstm.lstm', lstm_layers, bidirectional), sd,
                       'motion_encoder.vae_head')
    return mu_c, mu_m


def _running(sd: SD, new_stats: Optional[SD]) -> SD:
    """sd with the BatchNorm buffers a previous training-mode call has produced."""
    if not new_stats:
        return sd
    out = dict(sd)
    out.update(new_stats)
    return out


def swap_content(sd: SD, video_content: Tensor, video_motion: Tensor, training: bool = False,
                 new_stats: Optional[SD] = None, **kw) -> Tensor:
    """DisentangledVAE.swap_content; models/disentangled_vae.py:139-149: content of the first clip, motion of
    the second.  Each encode() runs BOTH encoders, so in training mode the BatchNorm buffers move three
    times (two encodes, one decode), in this order."""
    ns = new_stats if new_stats is not None else {}
    z_c, _ = disentangled_encode(sd, video_content, training, new_stats=ns, **kw)
    _, z_m = disentangled_encode(_running(sd, ns), video_motion, training, new_stats=ns, **kw)
    return disentangled_decode(_running(sd, ns), z_c, z_m, video_content.shape[1], training,
                               kw.get('lstm_layers', 1), ns)


def _alpha_mix(a: Tensor, b: Tensor, num_steps: int) -> Tensor:
    """(1 - alpha) a + alpha b for alpha in linspace(0, 1, num_steps), rows concatenated
    (models/disentangled_vae.py:170-175)."""
    alphas = torch.linspace(0, 1, num_steps, dtype=a.dtype)
    return torch.cat([(1 - al) * a + al * b for al in alphas], dim=0)


def interpolate_content(sd: SD, video1: Tensor, video2: Tensor, num_steps: int = 10,
                        use_motion_from: str = 'first', **kw) -> Tensor:
    """models/disentangled_vae.py:158-181 (eval mode; single-clip inputs)."""
    c1, m1 = disentangled_encode(sd, video1, False, **kw)
    c2, m2 = disentangled_encode(sd, video2, False, **kw)
    z_m = (m1 if use_motion_from == 'first' else m2).expand(num_steps, -1)
    return disentangled_decode(sd, _alpha_mix(c1, c2, num_steps), z_m, video1.shape[1], False,
                               kw.get('lstm_layers', 1))


def interpolate_motion(sd: SD, video1: Tensor, video2: Tensor, num_steps: int = 10,
                       use_content_from: str = 'first', **kw) -> Tensor:
    """models/disentangled_vae.py:183-206."""
    c1, m1 = disentangled_encode(sd, video1, False, **kw)
    c2, m2 = disentangled_encode(sd, video2, False, **kw)
    z_c = (c1 if use_content_from == 'first' else c2).expand(num_steps, -1)
    return disentangled_decode(sd, z_c, _alpha_mix(m1, m2, num_steps), video1.shape[1], False,
                               kw.get('lstm_layers', 1))


def tuple_interpolate(kind: str, sd: SD, video1: Tensor, video2: Tensor, num_steps: int = 10,
                      lstm_layers: int = 1) -> Tensor:
    """BaselineVAE.interpolate / LSTMVAE.interpolate; models/baseline_vae.py:166-191, models/lstm_vae.py:130-149."""
    fwd = (lambda v: baseline_forward(sd, v, None, False)) if kind == 'baseline' else \
        (lambda v: lstm_vae_forward(sd, v, None, False, lstm_layers))
    z1, z2 = fwd(video1)[1], fwd(video2)[1]
    z = _alpha_mix(z1, z2, num_steps)
    T = video1.shape[1]
    return baseline_decode(sd, z, T, False) if kind == 'baseline' else lstm_vae_decode(sd, z, T, False, lstm_layers)


# ---------------------------------------------------------------------------
# losses
# ---------------------------------------------------------------------------
def recon_term(recon: Tensor, target: Tensor, kind: str) -> Tensor:
    """mean-reduced MSE or BCE (log terms clamped at -100 like F.binary_cross_entropy)."""
    if kind == 'mse':
        return ((recon - target) ** 2).mean()
    if kind == 'bce':
        lp = torch.clamp(torch.log(recon), min=-100.0)
        lq = torch.clamp(torch.log(1 - recon), min=-100.0)
        return (-(target * lp + (1 - target) * lq)).mean()
    raise ValueError(f"Unknown recon_loss_type: {kind}")


def temporal_term(recon: Tensor, target: Tensor) -> Tensor:
    """models/disentangled_vae.py:264-268: MSE of frame differences times (T-1)*C*H*W."""
    rd = recon[:, 1:] - recon[:, :-1]
    td = target[:, 1:] - target[:, :-1]
    return ((rd - td) ** 2).mean() * rd[0].numel()


def recon_metrics(recon: Tensor, target: Tensor, max_val: float = 1.0) -> Tuple[float, float, float]:
    """(mse, psnr, temporal consistency of recon) as the reference's evaluation loop computes them per
    batch (evaluation/metrics.py:14-15 mse, :18-22 psnr, :48-50 temporal_consistency)."""
    m = F.mse_loss(recon, target, reduction='mean').item()
    p = float('inf') if m == 0 else 10 * np.log10(max_val ** 2 / m)
    tc = (recon[:, 1:] - recon[:, :-1]).abs().mean().item()
    return m, float(p), tc


def kl_per_dim(mu: Tensor, logvar: Tensor) -> Tensor:
    return -0.5 * (1 + logvar - mu ** 2 - logvar.exp())


def disentangled_loss(out: Dict[str, Tensor], target: Tensor, beta: float = 1.0,
                      beta_content: Optional[float] = None, beta_motion: Optional[float] = None,
                      recon_loss_type: str = 'mse', temporal_weight: float = 0.0,
                      independence_weight: float = 0.0) -> Dict[str, Tensor]:
    """compute_loss of models/disentangled_vae.py:222-292 (no free bits on this path)."""
    recon = out['recon']
    num_pixels = recon[0].numel()
    bc = beta_content if beta_content is not None else beta
    bm = beta_motion if beta_motion is not None else beta
    rl = recon_term(recon, target, recon_loss_type)
    rl_scaled = rl * num_pixels
    kl_c = kl_per_dim(out['mu_content'], out['logvar_content']).sum(dim=1).mean()
    kl_m = kl_per_dim(out['mu_motion'], out['logvar_motion']).sum(dim=1).mean()
    kl = bc * kl_c + bm * kl_m
    temporal = recon.new_zeros(())
    if temporal_weight > 0:
        temporal = temporal_term(recon, target)
    indep = recon.new_zeros(())
    if independence_weight > 0:
        zc, zm = out['z_content'], out['z_motion']
        zc_n = (zc - zc.mean(dim=0)) / (zc.std(dim=0) + 1e-8)      # unbiased std, :272-273
        zm_n = (zm - zm.mean(dim=0)) / (zm.std(dim=0) + 1e-8)
        corr = zc_n.t() @ zm_n / zc.shape[0]
        indep = (corr ** 2).mean()
    total = rl_scaled + kl + temporal_weight * temporal + independence_weight * indep
    return {'total_loss': total, 'recon_loss': rl_scaled, 'recon_loss_per_pixel': rl, 'kl_loss': kl,
            'kl_content': kl_c, 'kl_motion': kl_m, 'temporal_loss': temporal,
            'independence_loss': indep, 'beta': torch.tensor(beta)}


def tuple_loss(recon: Tensor, target: Tensor, mu: Tensor, logvar: Tensor, beta: float = 1.0,
               recon_loss_type: str = 'mse', temporal_weight: float = 0.0, free_bits: float = 0.0,
               use_free_bits: bool = False, with_temporal_key: bool = True) -> Dict[str, Tensor]:
    """compute_loss of models/lstm_vae.py:167-220 (with_temporal_key) and models/baseline_vae.py:208-252."""
    num_pixels = recon[0].numel()
    rl = recon_term(recon, target, recon_loss_type)
    rl_scaled = rl * num_pixels
    kpd_raw = kl_per_dim(mu, logvar)
    kpd = kpd_raw
    if use_free_bits and free_bits > 0:
        # the floor is a float32 scalar tensor in the reference (models/lstm_vae.py:194), also under fp64
        kpd = torch.maximum(kpd_raw, torch.tensor(free_bits))
    kl = kpd.sum(dim=1).mean()
    kl_raw = kpd_raw.sum(dim=1).mean()
    res = {'recon_loss': rl_scaled, 'recon_loss_per_pixel': rl, 'kl_loss': kl, 'kl_loss_raw': kl_raw,
           'kl_per_dim': kpd.mean(), 'beta': torch.tensor(beta)}
    total = rl_scaled + beta * kl
    if with_temporal_key:
        temporal = recon.new_zeros(())
        if temporal_weight > 0:
            temporal = temporal_term(recon, target)
        total = total + temporal_weight * temporal
        res['temporal_loss'] = temporal
    res['total_loss'] = total
    return res


# ---------------------------------------------------------------------------
# optimiser step and schedules
# ---------------------------------------------------------------------------
def clip_grad_norm(grads: Sequence[Tensor], max_norm: float) -> Tuple[Tensor, List[Tensor]]:
    """nn.utils.clip_grad_norm_ as called at training/trainer.py:198-202 (2-norm of per-tensor 2-norms)."""
    total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g) for g in grads]))
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    return total, [g * coef for g in grads]


def adam_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float, beta1: float,
              beta2: float, eps: float, weight_decay: float) -> Tuple[Tensor, Tensor, Tensor]:
    """torch.optim.Adam (L2 weight decay, not AdamW) as configured at training/trainer.py:56-61."""
    if weight_decay != 0:
        g = g + weight_decay * p
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    p = p - (lr / bc1) * (m / denom)
    return p, m, v


def beta_schedule(epoch: int, beta_vae: float, kl_start_epoch: int, use_kl_annealing: bool,
                  warmup_epochs: int, annealing_type: str) -> float:
    """Trainer._get_beta; training/trainer.py:115-149."""
    if epoch < kl_start_epoch:
        return 0.0
    if not use_kl_annealing or warmup_epochs <= 0:
        return beta_vae
    e = epoch - kl_start_epoch
    if annealing_type == 'linear':
        return beta_vae * (e / warmup_epochs) if e < warmup_epochs else beta_vae
    if annealing_type == 'cosine':
        if e >= warmup_epochs:
            return beta_vae
        return beta_vae * (1 - math.cos(e / warmup_epochs * math.pi)) / 2
    if annealing_type == 'cyclical':
        pos = e % warmup_epochs
        return beta_vae * 0.5 * (1 - math.cos(pos / warmup_epochs * math.pi))
    return beta_vae


def cosine_lr(epoch_steps: int, base_lr: float, t_max: int, eta_min: float) -> float:
    """Closed form of CosineAnnealingLR(T_max, eta_min) after ``epoch_steps`` scheduler steps."""
    return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * epoch_steps / t_max)) / 2


# ---------------------------------------------------------------------------
# Moving-MNIST generator (host, numpy, float64 trajectory state)
# ---------------------------------------------------------------------------
def moving_mnist_video(rng: np.random.RandomState, bank: np.ndarray, sequence_length: int = 20,
                       frame_size: int = 64, num_digits: int = 2, digit_size: int = 28,
                       velocity_range: Tuple[float, float] = (2.0, 5.0)):
    """MovingMNIST.generate_video; data/moving_mnist.py:84-206.

    Draw order: per digit randint(len(bank)); per digit uniform x then y; per digit uniform speed
    then angle.  Per frame: render at int(pos) with max-compositing, then bounce-advance.
    Returns (video (T,1,H,W) float32, digit indices, position history (T,D,2), velocity history).
    """
    max_pos = frame_size - digit_size
    idx = [int(rng.randint(len(bank))) for _ in range(num_digits)]
    pos = [np.array([rng.uniform(0, max_pos), rng.uniform(0, max_pos)]) for _ in range(num_digits)]
    vel = []
    for _ in range(num_digits):
        mag = rng.uniform(velocity_range[0], velocity_range[1])
        ang = rng.uniform(0, 2 * np.pi)
        vel.append(np.array([mag * np.cos(ang), mag * np.sin(ang)]))
    video = np.zeros((sequence_length, 1, frame_size, frame_size), dtype=np.float32)
    pos_hist = np.zeros((sequence_length, num_digits, 2))
    vel_hist = np.zeros((sequence_length, num_digits, 2))
    for t in range(sequence_length):
        frame = video[t, 0]
        for d in range(num_digits):
            x, y = int(pos[d][0]), int(pos[d][1])
            xs, ys = max(0, x), max(0, y)
            xe, ye = min(frame_size, x + digit_size), min(frame_size, y + digit_size)
            sub = bank[idx[d]][ys - y:ys - y + (ye - ys), xs - x:xs - x + (xe - xs)]
            frame[ys:ye, xs:xe] = np.maximum(frame[ys:ye, xs:xe], sub)
        pos_hist[t] = np.stack(pos)
        vel_hist[t] = np.stack(vel)
        for d in range(num_digits):
            p = pos[d] + vel[d]
            v = vel[d].copy()
            for a in range(2):
                if p[a] < 0:
                    p[a] = -p[a]
                    v[a] = -v[a]
                elif p[a] > max_pos:
                    p[a] = 2 * max_pos - p[a]
                    v[a] = -v[a]
            pos[d] = np.clip(p, 0, max_pos)
            vel[d] = v
    return video, idx, pos_hist, vel_hist


def moving_mnist_item(seed: Optional[int], index: int, bank: np.ndarray, **kw):
    """MovingMNIST.__getitem__ with deterministic=True; data/moving_mnist.py:208-221 (seed 0 == no seed)."""
    rng = np.random.RandomState(seed + index if seed else index)
    return moving_mnist_video(rng, bank, **kw)


# ---------------------------------------------------------------------------
# whole training step (the CPU baseline bench.py times, and the N-rank parity oracle)
# ---------------------------------------------------------------------------
def model_forward(kind: str, sd: SD, video: Tensor, eps: Sequence[Optional[Tensor]], training: bool,
                  lstm_layers: int = 1, bidirectional: bool = True, new_stats: Optional[SD] = None):
    if kind == 'disentangled':
        return disentangled_forward(sd, video, eps[0], eps[1], training, lstm_layers, bidirectional,
                                    new_stats=new_stats)
    if kind == 'lstm':
        return lstm_vae_forward(sd, video, eps[0], training, lstm_layers, bidirectional, new_stats)
    if kind == 'baseline':
        return baseline_forward(sd, video, eps[0], training, new_stats)
    raise ValueError(f"Unknown model type: {kind}")


def model_loss(kind: str, out, video: Tensor, beta: float, tcfg: Dict) -> Dict[str, Tensor]:
    """Trainer._compute_loss dispatch; training/trainer.py:151-179."""
    if kind == 'disentangled':
        return disentangled_loss(out, video, beta=beta, beta_content=tcfg.get('beta_content'),
                                 beta_motion=tcfg.get('beta_motion'),
                                 recon_loss_type=tcfg.get('recon_loss_type', 'mse'),
                                 temporal_weight=tcfg.get('temporal_weight', 0.0),
                                 independence_weight=tcfg.get('independence_weight', 0.0))
    recon, _, mu, logvar = out
    return tuple_loss(recon, video, mu, logvar, beta=beta,
                      recon_loss_type=tcfg.get('recon_loss_type', 'mse'),
                      free_bits=tcfg.get('free_bits', 0.0), use_free_bits=tcfg.get('use_free_bits', False),
                      with_temporal_key=(kind == 'lstm'))


def trainable_keys(sd: SD) -> List[str]:
    return [k for k in sd if not (k.endswith('running_mean') or k.endswith('running_var')
                                  or k.endswith('num_batches_tracked'))]


def train_step(kind: str, sd: SD, opt_state: Dict[str, Dict], video: Tensor,
               eps: Sequence[Optional[Tensor]], beta: float, tcfg: Dict, lr: float,
               lstm_layers: int = 1, bidirectional: bool = True,
               micro_batches: int = 1) -> Tuple[SD, Dict[str, Dict], Dict[str, Tensor], Dict[str, Tensor]]:
    """One Trainer.train_epoch iteration; training/trainer.py:192-204.

    With micro_batches=N the batch is split into N equal shards, each run independently with its
    own BatchNorm statistics, and the gradients are averaged before clip + Adam: the semantics of
    N data-parallel ranks.  Returns (new state dict, new optimiser state, losses of shard 0, grads).
    """
    keys = trainable_keys(sd)
    leaves = {k: sd[k].detach().clone().requires_grad_(True) for k in keys}
    work = dict(sd)
    work.update(leaves)
    grads_sum = None
    losses0 = None
    new_stats: SD = {}
    shards = video.chunk(micro_batches, dim=0)
    for r, v in enumerate(shards):
        sl = slice(r * v.shape[0], (r + 1) * v.shape[0])
        e = [None if x is None else x[sl] for x in eps]
        ns: SD = {}
        out = model_forward(kind, work, v, e, True, lstm_layers, bidirectional, ns)
        losses = model_loss(kind, out, v, beta, tcfg)
        g = torch.autograd.grad(losses['total_loss'], [leaves[k] for k in keys], allow_unused=True)
        g = [torch.zeros_like(leaves[k]) if gi is None else gi for k, gi in zip(keys, g)]
        grads_sum = g if grads_sum is None else [a + b for a, b in zip(grads_sum, g)]
        if r == 0:
            losses0 = {k: v2.detach() for k, v2 in losses.items()}
            new_stats = ns
    grads = [g / micro_batches for g in grads_sum]
    clip = tcfg.get('gradient_clip', 0.0)
    if clip > 0:
        _, grads = clip_grad_norm(grads, clip)
    new_sd = dict(sd)
    new_sd.update(new_stats)
    new_opt: Dict[str, Dict] = {}
    for k, g in zip(keys, grads):
        st = opt_state.get(k) or {'step': 0, 'exp_avg': torch.zeros_like(sd[k]),
                                  'exp_avg_sq': torch.zeros_like(sd[k])}
        step = st['step'] + 1
        p, m, v = adam_step(sd[k].detach(), g, st['exp_avg'], st['exp_avg_sq'], step, lr,
                            tcfg.get('beta1', 0.9), tcfg.get('beta2', 0.999), 1e-8,
                            tcfg.get('weight_decay', 0.0))
        new_sd[k] = p
        new_opt[k] = {'step': step, 'exp_avg': m, 'exp_avg_sq': v}
    return new_sd, new_opt, losses0, dict(zip(keys, grads))
