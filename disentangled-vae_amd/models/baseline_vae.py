"""BaselineVAE: per-frame CNN encoder, per-frame heads averaged over time, the same latent decoded
for every frame; on the MI355X kernels.  Surface as in the reference (models/baseline_vae.py:15-205)."""
from typing import List, Optional

import torch
import torch.nn as nn

from hipvae.losses import baseline_loss as compute_loss  # noqa: F401  (reference name)
from hipvae.parallel import stage_boundary
from .decoders import CNNDecoder
from .encoders import CNNEncoder, VAEEncoder, reparameterize


class BaselineVAE(nn.Module):

    def __init__(self, in_channels: int = 1, latent_dim: int = 256, encoder_channels: List[int] = [32, 64, 128, 256],
                 decoder_channels: List[int] = [256, 128, 64, 32], feature_dim: int = 512, sequence_length: int = 20):
        super().__init__()
        self.latent_dim = latent_dim
        self.sequence_length = sequence_length
        self.model_type = 'baseline_vae'
        self.cnn_encoder = CNNEncoder(in_channels=in_channels, channels=encoder_channels, output_dim=feature_dim)
        self.vae_head = VAEEncoder(input_dim=feature_dim, latent_dim=latent_dim)
        self.cnn_decoder = CNNDecoder(input_dim=latent_dim, channels=decoder_channels, out_channels=in_channels)

    def encode(self, video: torch.Tensor, return_all_frames: bool = False):
        B, T = video.shape[:2]
        feats = self.cnn_encoder(video.reshape(B * T, *video.shape[2:]))
        fmu, flv = self.vae_head(feats)
        fmu, flv = fmu.view(B, T, -1), flv.view(B, T, -1)
        mu, logvar = fmu.mean(dim=1), flv.mean(dim=1)
        z = reparameterize(mu, logvar)
        if return_all_frames:
            return z, mu, logvar, fmu, flv
        return z, mu, logvar

    def decode(self, z: torch.Tensor, sequence_length: Optional[int] = None) -> torch.Tensor:
        B = z.size(0)
        T = sequence_length or self.sequence_length
        # the reference decodes z repeated T times (models/baseline_vae.py:114-121): T identical frames per clip.
        # One copy is decoded and repeated; backward sums the copies' gradients through the expand.
        frames = self.cnn_decoder(z, replicas=T)
        return frames.unsqueeze(1).expand(B, T, *frames.shape[1:]).contiguous()

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
