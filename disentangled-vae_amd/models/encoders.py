"""Encoder blocks with the reference's constructors and state_dict layout, computed by HIP kernels.

The torch.nn modules created here are parameter/buffer containers only (same names, shapes, init and
RNG draw order as the reference, models/encoders.py); their own forward() is never called.  All
arithmetic goes through hipvae.functional, which has no CPU path.
"""
from typing import List, Optional, Tuple

import torch
import torch.nn as nn

from hipvae import functional as HF
from hipvae import ops, rng


def _conv_bn_stack(make_conv, channels_in: int, channels: List[int], use_batch_norm: bool) -> nn.Sequential:
    if not use_batch_norm:
        raise NotImplementedError("the MI355X path fuses BatchNorm into the conv kernels; use_batch_norm=False "
                                  "is not supported")
    mods = []
    c_in = channels_in
    for c_out in channels:
        mods += [make_conv(c_in, c_out), nn.BatchNorm2d(c_out), nn.LeakyReLU(0.2, inplace=True)]
        c_in = c_out
    return nn.Sequential(*mods)


def _stack_tensors(seq: nn.Sequential):
    """conv weights, BN gammas, BN betas and BN buffer handles of a (conv, bn, act)* Sequential."""
    convs = [seq[i] for i in range(0, len(seq), 3)]
    bns = [seq[i] for i in range(1, len(seq), 3)]
    handles = [HF.BNState(b.running_mean, b.running_var, b.num_batches_tracked) for b in bns]
    return [c.weight for c in convs], [b.weight for b in bns], [b.bias for b in bns], handles


def _check_k4s2p1(kernel_size, stride, padding):
    if (kernel_size, stride, padding) != (4, 2, 1):
        raise NotImplementedError("the MI355X conv kernels implement kernel 4, stride 2, padding 1 only")


def _frames_nhwc(x: torch.Tensor) -> torch.Tensor:
    """[N,C,H,W] -> [N,H,W,C] (a free view for single-channel frames)."""
    if x.shape[1] == 1:
        return x.reshape(x.shape[0], x.shape[2], x.shape[3], 1)
    return x.permute(0, 2, 3, 1).contiguous()


class _ConvEncoderBase(nn.Module):
    final_hw = 4
    # whether a staged data-parallel step may cut this encoder's backward between its Linear layer and its
    # convolution stack (hipvae.functional.encoder_cnn).  An encoder that sees one frame per clip switches it off:
    # its whole backward is a chain of small kernels that runs best beside another branch's recurrence.
    stage_cut = True

    def __init__(self, in_channels, channels, kernel_size, stride, padding, output_dim, use_batch_norm):
        super().__init__()
        _check_k4s2p1(kernel_size, stride, padding)
        self.output_dim = output_dim
        self.use_batch_norm = use_batch_norm
        self.conv = _conv_bn_stack(lambda a, b: nn.Conv2d(a, b, kernel_size, stride, padding, bias=False),
                                   in_channels, list(channels), use_batch_norm)
        self.flat_size = channels[-1] * self.final_hw * self.final_hw
        self.fc = nn.Sequential(nn.Linear(self.flat_size, output_dim), nn.LeakyReLU(0.2, inplace=True))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        ws, gs, bs, handles = _stack_tensors(self.conv)
        return HF.encoder_cnn(_frames_nhwc(x), self.training, handles, self.fc[0].weight, self.fc[0].bias,
                              *ws, *gs, *bs, stage_cut=self.stage_cut)


class CNNEncoder(_ConvEncoderBase):
    """Four stride-2 convs to a 4x4 map, then Linear (reference: models/encoders.py:11-53)."""

    def __init__(self, in_channels: int = 1, channels: List[int] = [32, 64, 128, 256], kernel_size: int = 4,
                 stride: int = 2, padding: int = 1, output_dim: int = 512, use_batch_norm: bool = True):
        super().__init__(in_channels, channels, kernel_size, stride, padding, output_dim, use_batch_norm)


class LightCNNEncoder(_ConvEncoderBase):
    """Two stride-2 convs to a 16x16 map, then Linear (reference: models/encoders.py:56-97)."""
    final_hw = 16

    def __init__(self, in_channels: int = 1, channels: List[int] = [16, 32], kernel_size: int = 4,
                 stride: int = 2, padding: int = 1, output_dim: int = 512, use_batch_norm: bool = True):
        super().__init__(in_channels, channels, kernel_size, stride, padding, output_dim, use_batch_norm)


class VideoEncoder(nn.Module):
    """Per-frame CNN over [B,T,...] with optional temporal pooling (reference: models/encoders.py:100-125)."""

    def __init__(self, cnn_encoder: nn.Module, pool_type: str = 'none'):
        super().__init__()
        self.cnn_encoder = cnn_encoder
        self.pool_type = pool_type

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        B, T = x.shape[:2]
        feats = self.cnn_encoder(x.reshape(B * T, *x.shape[2:])).view(B, T, -1)
        if self.pool_type == 'mean':
            feats = feats.mean(dim=1)
        elif self.pool_type == 'max':
            feats = feats.max(dim=1)[0]
        return feats


class LSTMEncoder(nn.Module):
    """(Bi)directional LSTM whose result is the last layer's final hidden state(s), or with return_sequence the
    output sequence, and nn.LSTM's (h_n, c_n) beside either (reference: models/encoders.py:128-173).  A caller
    that only uses the final state sets `return_states = False` and gets (None, None) for the states, which saves
    the launches that gather them (the models of this package do)."""
    return_states = True

    def __init__(self, input_dim: int = 512, hidden_dim: int = 256, num_layers: int = 1,
                 bidirectional: bool = True, dropout: float = 0.0):
        super().__init__()
        self.hidden_dim = hidden_dim
        self.num_layers = num_layers
        self.bidirectional = bidirectional
        self.num_directions = 2 if bidirectional else 1
        if dropout and num_layers > 1:
            raise NotImplementedError("inter-layer dropout is not implemented in the HIP LSTM")
        self.lstm = nn.LSTM(input_size=input_dim, hidden_size=hidden_dim, num_layers=num_layers, batch_first=True,
                            bidirectional=bidirectional, dropout=0.0)
        self.output_dim = hidden_dim * self.num_directions

    def _flat_weights(self):
        ws = []
        for l in range(self.num_layers):
            for sfx in (['', '_reverse'] if self.bidirectional else ['']):
                ws += [getattr(self.lstm, f'{n}_l{l}{sfx}') for n in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh')]
        return ws

    def forward(self, x: torch.Tensor, return_sequence: bool = False):
        if not (return_sequence or self.return_states):
            final = HF.LSTMEncoder.apply(x, self.num_layers, self.bidirectional, False, *self._flat_weights())
            return final, (None, None)
        final, output, h_n, c_n = HF.LSTMEncoder.apply(x, self.num_layers, self.bidirectional, True,
                                                       *self._flat_weights())
        return (output if return_sequence else final), (h_n, c_n)


class VAEEncoder(nn.Module):
    """mu / logvar heads (reference: models/encoders.py:176-204)."""

    def __init__(self, input_dim: int = 512, latent_dim: int = 128, hidden_dim: Optional[int] = None):
        super().__init__()
        self.latent_dim = latent_dim
        if hidden_dim is not None:
            self.fc = nn.Sequential(nn.Linear(input_dim, hidden_dim), nn.LeakyReLU(0.2, inplace=True))
            head_in = hidden_dim
        else:
            self.fc = nn.Identity()
            head_in = input_dim
        self.fc_mu = nn.Linear(head_in, latent_dim)
        self.fc_logvar = nn.Linear(head_in, latent_dim)

    def forward(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        h = x
        if not isinstance(self.fc, nn.Identity):
            h = HF.linear(h, self.fc[0].weight, self.fc[0].bias, ops.ACT_LRELU)
        mu = HF.linear(h, self.fc_mu.weight, self.fc_mu.bias)
        logvar = HF.linear(h, self.fc_logvar.weight, self.fc_logvar.bias)
        return mu, logvar


def reparameterize(mu: torch.Tensor, logvar: torch.Tensor) -> torch.Tensor:
    """mu under no_grad, else mu + eps*exp(logvar/2) with eps ~ N(0,1) (reference: models/encoders.py:207-214)."""
    if not torch.is_grad_enabled():
        return mu
    return HF.Reparam.apply(mu, logvar, rng.draw_like(mu))


class ContentEncoder(nn.Module):
    """Temporal mean (or middle frame) -> CNN -> heads (reference: models/encoders.py:217-265)."""

    def __init__(self, in_channels: int = 1, channels: List[int] = [32, 64, 128, 256], feature_dim: int = 512,
                 latent_dim: int = 128, aggregation: str = 'mean'):
        super().__init__()
        self.aggregation = aggregation
        self.latent_dim = latent_dim
        self.cnn_encoder = CNNEncoder(in_channels=in_channels, channels=channels, output_dim=feature_dim)
        self.cnn_encoder.stage_cut = False         # one frame per clip: see _ConvEncoderBase.stage_cut
        self.vae_head = VAEEncoder(input_dim=feature_dim, latent_dim=latent_dim)

    def forward(self, video: torch.Tensor, return_distribution: bool = True, _mean: Optional[torch.Tensor] = None):
        if self.aggregation == 'mean':
            frame = _mean if _mean is not None else ops.split_mean_diff(video.contiguous())[0]
        elif self.aggregation == 'middle':
            frame = video[:, video.shape[1] // 2].contiguous()
        else:
            raise ValueError(f"Unknown aggregation: {self.aggregation}")
        mu, logvar = self.vae_head(self.cnn_encoder(frame))
        z = reparameterize(mu, logvar)
        return (z, (mu, logvar)) if return_distribution else (z, None)


class MotionEncoder(nn.Module):
    """Frame differences -> light CNN -> LSTM -> heads (reference: models/encoders.py:268-327)."""

    def __init__(self, in_channels: int = 1, channels: List[int] = [16, 32], feature_dim: int = 512,
                 lstm_hidden: int = 256, lstm_layers: int = 1, bidirectional: bool = True, latent_dim: int = 128):
        super().__init__()
        self.latent_dim = latent_dim
        self.cnn_encoder = LightCNNEncoder(in_channels=in_channels, channels=channels, output_dim=feature_dim)
        self.lstm = LSTMEncoder(input_dim=feature_dim, hidden_dim=lstm_hidden, num_layers=lstm_layers,
                                bidirectional=bidirectional)
        self.lstm.return_states = False            # only the final state is used here
        self.vae_head = VAEEncoder(input_dim=self.lstm.output_dim, latent_dim=latent_dim)

    def compute_frame_differences(self, video: torch.Tensor) -> torch.Tensor:
        return ops.split_mean_diff(video.contiguous())[1]

    def forward(self, video: torch.Tensor, return_distribution: bool = True, _diff: Optional[torch.Tensor] = None):
        B = video.shape[0]
        diff = _diff if _diff is not None else self.compute_frame_differences(video)
        n = diff.shape[1]
        feats = self.cnn_encoder(diff.view(B * n, *diff.shape[2:])).view(B, n, -1)
        h, _ = self.lstm(feats)
        mu, logvar = self.vae_head(h)
        z = reparameterize(mu, logvar)
        return (z, (mu, logvar)) if return_distribution else (z, None)
