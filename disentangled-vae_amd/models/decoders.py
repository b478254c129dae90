"""Decoder blocks with the reference's constructors and state_dict layout (models/decoders.py),
computed by HIP kernels.  The torch.nn modules are parameter containers only."""
from typing import List, Optional

import torch
import torch.nn as nn

from hipvae import functional as HF
from hipvae import ops
from .encoders import _check_k4s2p1, _conv_bn_stack, _stack_tensors


class CNNDecoder(nn.Module):
    """Linear -> 4x4 map -> transposed convs (BN, LeakyReLU) -> transposed conv with bias -> sigmoid
    (reference: models/decoders.py:11-70)."""

    def __init__(self, input_dim: int = 512, channels: List[int] = [256, 128, 64, 32], out_channels: int = 1,
                 kernel_size: int = 4, stride: int = 2, padding: int = 1, use_batch_norm: bool = True,
                 output_activation: str = 'sigmoid'):
        super().__init__()
        _check_k4s2p1(kernel_size, stride, padding)
        if output_activation != 'sigmoid':
            raise NotImplementedError("the fused final layer implements the sigmoid output only")
        if out_channels != 1:
            raise NotImplementedError("the fused final layer implements one output channel")
        self.use_batch_norm = use_batch_norm
        self.output_activation = output_activation
        self.initial_size = 4
        self.fc = nn.Sequential(nn.Linear(input_dim, channels[0] * 16), nn.LeakyReLU(0.2, inplace=True))
        if not use_batch_norm:
            raise NotImplementedError("use_batch_norm=False is not supported by the fused kernels")
        mods = []
        for c_in, c_out in zip(channels[:-1], channels[1:]):
            mods += [nn.ConvTranspose2d(c_in, c_out, kernel_size, stride, padding, bias=False),
                     nn.BatchNorm2d(c_out), nn.LeakyReLU(0.2, inplace=True)]
        self.deconv = nn.Sequential(*mods)
        self.final = nn.ConvTranspose2d(channels[-1], out_channels, kernel_size, stride, padding, bias=True)
        self.output_act = nn.Sigmoid()

    def forward(self, x: torch.Tensor, replicas: int = 1) -> torch.Tensor:
        """`replicas` (not in the reference): x stands for that many identical copies of each of its rows, see
        hipvae.functional.DecoderCNN."""
        ws, gs, bs, handles = _stack_tensors(self.deconv)
        return HF.bind_tail(HF.DecoderCNN.apply(x, self.training, handles, int(replicas), self.fc[0].weight,
                                                self.fc[0].bias, self.final.weight, self.final.bias, *ws, *gs, *bs))


def _lstm_layer_weights(lstm: nn.LSTM):
    ws = []
    for l in range(lstm.num_layers):
        ws += [getattr(lstm, f'{n}_l{l}') for n in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh')]
    return ws


def _decode_states(mod, z: torch.Tensor, num_layers: int, seq_len: int) -> torch.Tensor:
    h0 = HF.linear(z, mod.fc_hidden.weight, mod.fc_hidden.bias)
    c0 = HF.linear(z, mod.fc_cell.weight, mod.fc_cell.bias)
    return HF.LSTMDecoder.apply(h0, c0, mod.initial_input, seq_len, num_layers, *_lstm_layer_weights(mod.lstm))


class LSTMDecoder(nn.Module):
    """z -> (h0, c0) -> autoregressive LSTM -> Linear + LeakyReLU per step
    (reference: models/decoders.py:73-137)."""

    def __init__(self, input_dim: int = 256, hidden_dim: int = 256, output_dim: int = 512, num_layers: int = 1,
                 sequence_length: int = 20, dropout: float = 0.0):
        super().__init__()
        if dropout and num_layers > 1:
            raise NotImplementedError("inter-layer dropout is not implemented in the HIP LSTM")
        self.hidden_dim = hidden_dim
        self.num_layers = num_layers
        self.sequence_length = sequence_length
        self.output_dim = output_dim
        self.fc_hidden = nn.Linear(input_dim, hidden_dim * num_layers)
        self.fc_cell = nn.Linear(input_dim, hidden_dim * num_layers)
        self.lstm = nn.LSTM(input_size=hidden_dim, hidden_size=hidden_dim, num_layers=num_layers,
                            batch_first=True, dropout=0.0)
        self.fc_out = nn.Sequential(nn.Linear(hidden_dim, output_dim), nn.LeakyReLU(0.2, inplace=True))
        self.initial_input = nn.Parameter(torch.zeros(1, hidden_dim))
        nn.init.xavier_uniform_(self.initial_input)

    def forward(self, z: torch.Tensor, sequence_length: Optional[int] = None) -> torch.Tensor:
        T = sequence_length or self.sequence_length
        states = _decode_states(self, z, self.num_layers, T)
        return HF.linear(states, self.fc_out[0].weight, self.fc_out[0].bias, ops.ACT_LRELU)


class VideoDecoder(nn.Module):
    """LSTMDecoder + CNNDecoder with optional content concatenation (reference: models/decoders.py:140-193)."""

    def __init__(self, latent_dim: int = 256, lstm_hidden: int = 256, frame_feature_dim: int = 512,
                 cnn_channels: List[int] = [256, 128, 64, 32], sequence_length: int = 20,
                 use_content_modulation: bool = False, content_dim: int = 128):
        super().__init__()
        self.use_content_modulation = use_content_modulation
        self.content_dim = content_dim
        self.lstm_decoder = LSTMDecoder(input_dim=latent_dim, hidden_dim=lstm_hidden, output_dim=frame_feature_dim,
                                        sequence_length=sequence_length)
        cnn_in = frame_feature_dim + content_dim if use_content_modulation else frame_feature_dim
        self.cnn_decoder = CNNDecoder(input_dim=cnn_in, channels=cnn_channels)
        self.sequence_length = sequence_length

    def forward(self, z, z_content=None, sequence_length=None):
        B = z.size(0)
        T = sequence_length or self.sequence_length
        ff = self.lstm_decoder(z, T)
        if self.use_content_modulation and z_content is not None:
            ff = torch.cat([ff, z_content.unsqueeze(1).expand(-1, T, -1)], dim=-1)
        frames = self.cnn_decoder(ff.reshape(B * T, -1))
        return HF.as_clips(frames, B, T)


class DisentangledDecoder(nn.Module):
    """cat(z_c, z_m) seeds the autoregressive LSTM; each step's feature is concatenated with z_c and
    decoded to a frame (reference: models/decoders.py:196-281)."""

    def __init__(self, content_dim: int = 128, motion_dim: int = 128, lstm_hidden: int = 256,
                 frame_feature_dim: int = 512, cnn_channels: List[int] = [256, 128, 64, 32],
                 sequence_length: int = 20, num_lstm_layers: int = 1):
        super().__init__()
        self.content_dim = content_dim
        self.motion_dim = motion_dim
        self.lstm_hidden = lstm_hidden
        self.num_lstm_layers = num_lstm_layers
        self.sequence_length = sequence_length
        combined = content_dim + motion_dim
        self.fc_hidden = nn.Linear(combined, lstm_hidden * num_lstm_layers)
        self.fc_cell = nn.Linear(combined, lstm_hidden * num_lstm_layers)
        self.lstm = nn.LSTM(input_size=lstm_hidden, hidden_size=lstm_hidden, num_layers=num_lstm_layers,
                            batch_first=True)
        self.fc_frame = nn.Sequential(nn.Linear(lstm_hidden, frame_feature_dim), nn.LeakyReLU(0.2, inplace=True))
        self.initial_input = nn.Parameter(torch.zeros(1, lstm_hidden))
        nn.init.xavier_uniform_(self.initial_input)
        self.cnn_decoder = CNNDecoder(input_dim=frame_feature_dim + content_dim, channels=cnn_channels)

    def forward(self, z_content: torch.Tensor, z_motion: torch.Tensor,
                sequence_length: Optional[int] = None) -> torch.Tensor:
        B = z_content.size(0)
        T = sequence_length or self.sequence_length
        states = _decode_states(self, torch.cat([z_content, z_motion], dim=-1), self.num_lstm_layers, T)
        ff = HF.linear(states, self.fc_frame[0].weight, self.fc_frame[0].bias, ops.ACT_LRELU)
        ff = torch.cat([ff, z_content.unsqueeze(1).expand(-1, T, -1)], dim=-1)
        frames = self.cnn_decoder(ff.reshape(B * T, -1))
        return HF.as_clips(frames, B, T)
