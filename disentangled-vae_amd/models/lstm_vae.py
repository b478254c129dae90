"""LSTMVAE: per-frame CNN -> (bi)LSTM -> latent -> autoregressive LSTM -> per-frame CNN decoder,
on the MI355X kernels.  Surface as in the reference (models/lstm_vae.py:14-164)."""
from typing import List, Optional

import torch
import torch.nn as nn

from hipvae.losses import lstm_loss as compute_loss  # noqa: F401  (reference name)
from hipvae import functional as HF
from hipvae.parallel import stage_boundary
from .decoders import CNNDecoder, LSTMDecoder
from .encoders import CNNEncoder, LSTMEncoder, VAEEncoder, reparameterize


class LSTMVAE(nn.Module):

    def __init__(self, in_channels: int = 1, latent_dim: int = 256, encoder_channels: List[int] = [32, 64, 128, 256],
                 decoder_channels: List[int] = [256, 128, 64, 32], feature_dim: int = 512, lstm_hidden: int = 256,
                 lstm_layers: int = 1, bidirectional: bool = True, sequence_length: int = 20, dropout: float = 0.0):
        super().__init__()
        self.latent_dim = latent_dim
        self.sequence_length = sequence_length
        self.feature_dim = feature_dim
        self.model_type = 'lstm_vae'
        self.cnn_encoder = CNNEncoder(in_channels=in_channels, channels=encoder_channels, output_dim=feature_dim)
        self.lstm_encoder = LSTMEncoder(input_dim=feature_dim, hidden_dim=lstm_hidden, num_layers=lstm_layers,
                                        bidirectional=bidirectional, dropout=dropout)
        self.lstm_encoder.return_states = False    # only the final state is used here
        self.vae_head = VAEEncoder(input_dim=self.lstm_encoder.output_dim, latent_dim=latent_dim)
        self.lstm_decoder = LSTMDecoder(input_dim=latent_dim, hidden_dim=lstm_hidden, output_dim=feature_dim,
                                        num_layers=lstm_layers, sequence_length=sequence_length, dropout=dropout)
        self.cnn_decoder = CNNDecoder(input_dim=feature_dim, channels=decoder_channels, out_channels=in_channels)

    def _frame_features(self, video: torch.Tensor) -> torch.Tensor:
        B, T = video.shape[:2]
        return self.cnn_encoder(video.reshape(B * T, *video.shape[2:])).view(B, T, -1)

    def encode(self, video: torch.Tensor):
        h, _ = self.lstm_encoder(self._frame_features(video))
        mu, logvar = self.vae_head(h)
        return reparameterize(mu, logvar), mu, logvar

    def decode(self, z: torch.Tensor, sequence_length: Optional[int] = None) -> torch.Tensor:
        B = z.size(0)
        T = sequence_length or self.sequence_length
        ff = self.lstm_decoder(z, sequence_length=T)
        frames = self.cnn_decoder(ff.reshape(B * T, -1))
        return HF.as_clips(frames, B, T)

    def forward(self, video: torch.Tensor):
        z, mu, logvar = self.encode(video)
        z, mu, logvar = stage_boundary(z, mu, logvar)      # where a data-parallel step cuts backward in two
        return self.decode(z, sequence_length=video.shape[1]), z, mu, logvar

    def sample(self, num_samples: int = 1, device: torch.device = None, sequence_length: Optional[int] = None):
        if device is None:
            device = next(self.parameters()).device
        z = torch.randn(num_samples, self.latent_dim, device=device)
        with torch.no_grad():
            return self.decode(z, sequence_length)

    def interpolate(self, video1, video2, num_steps: int = 10):
        with torch.no_grad():
            z1, z2 = self.encode(video1)[0], self.encode(video2)[0]
            alphas = torch.linspace(0, 1, num_steps, device=z1.device)
            z = torch.cat([(1 - a) * z1 + a * z2 for a in alphas], dim=0)
            return self.decode(z, sequence_length=video1.shape[1])

    def get_latent(self, video):
        with torch.no_grad():
            return self.encode(video)[1]

    def get_frame_features(self, video):
        with torch.no_grad():
            return self._frame_features(video)
