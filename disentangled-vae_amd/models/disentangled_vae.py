"""DisentangledVAE: content and motion encoders + joint decoder, on the MI355X kernels.

Same constructor, attributes, method set and 7-key forward() dict as the reference
(models/disentangled_vae.py:28-219); the loss is hipvae.losses.disentangled_loss.
"""
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from hipvae import functional as HF
from hipvae import ops
from hipvae.losses import disentangled_loss as compute_loss  # noqa: F401  (reference name)
from hipvae.parallel import stage_boundary
from .decoders import DisentangledDecoder
from .encoders import ContentEncoder, MotionEncoder


class DisentangledVAE(nn.Module):

    def __init__(self, in_channels: int = 1, content_dim: int = 128, motion_dim: int = 128,
                 content_channels: List[int] = [32, 64, 128, 256], motion_channels: List[int] = [16, 32],
                 decoder_channels: List[int] = [256, 128, 64, 32], content_feature_dim: int = 512,
                 motion_feature_dim: int = 512, lstm_hidden: int = 256, lstm_layers: int = 1,
                 bidirectional: bool = True, sequence_length: int = 20, content_aggregation: str = 'mean'):
        super().__init__()
        self.content_dim = content_dim
        self.motion_dim = motion_dim
        self.latent_dim = content_dim + motion_dim
        self.sequence_length = sequence_length
        self.model_type = 'disentangled_vae'
        self.content_encoder = ContentEncoder(in_channels=in_channels, channels=content_channels,
                                              feature_dim=content_feature_dim, latent_dim=content_dim,
                                              aggregation=content_aggregation)
        self.motion_encoder = MotionEncoder(in_channels=in_channels, channels=motion_channels,
                                            feature_dim=motion_feature_dim, lstm_hidden=lstm_hidden,
                                            lstm_layers=lstm_layers, bidirectional=bidirectional,
                                            latent_dim=motion_dim)
        self.decoder = DisentangledDecoder(content_dim=content_dim, motion_dim=motion_dim, lstm_hidden=lstm_hidden,
                                           frame_feature_dim=content_feature_dim, cnn_channels=decoder_channels,
                                           sequence_length=sequence_length, num_lstm_layers=lstm_layers)

    def encode(self, video: torch.Tensor):
        # one pass over the clip yields both the temporal mean and the frame differences
        mean, diff = ops.split_mean_diff(video.contiguous())
        if HF.USE_SIDE_ENCODER and video.is_cuda:
            with HF.side_branch(video) as br:
                br.take(mean)
                z_c, (mu_c, lv_c) = self.content_encoder(video, _mean=mean)      # content draw first
                br.hand_over(z_c, mu_c, lv_c)
            z_m, (mu_m, lv_m) = self.motion_encoder(video, _diff=diff)
            br.join()
            return z_c, z_m, (mu_c, lv_c), (mu_m, lv_m)
        z_c, (mu_c, lv_c) = self.content_encoder(video, _mean=mean)      # content draw first
        z_m, (mu_m, lv_m) = self.motion_encoder(video, _diff=diff)
        return z_c, z_m, (mu_c, lv_c), (mu_m, lv_m)

    def decode(self, z_content, z_motion, sequence_length: Optional[int] = None):
        return self.decoder(z_content, z_motion, sequence_length)

    def forward(self, video: torch.Tensor) -> Dict[str, torch.Tensor]:
        z_c, z_m, (mu_c, lv_c), (mu_m, lv_m) = self.encode(video)
        # encoder | decoder + loss: where a data-parallel step cuts backward in two (a no-op otherwise)
        z_c, z_m, mu_c, lv_c, mu_m, lv_m = stage_boundary(z_c, z_m, mu_c, lv_c, mu_m, lv_m)
        recon = self.decode(z_c, z_m, sequence_length=video.shape[1])
        return {'recon': recon, 'z_content': z_c, 'z_motion': z_m, 'mu_content': mu_c, 'logvar_content': lv_c,
                'mu_motion': mu_m, 'logvar_motion': lv_m}

    def sample(self, num_samples: int = 1, device: torch.device = None, sequence_length: Optional[int] = None,
               z_content: Optional[torch.Tensor] = None, z_motion: Optional[torch.Tensor] = None):
        if device is None:
            device = next(self.parameters()).device
        if z_content is None:
            z_content = torch.randn(num_samples, self.content_dim, device=device)
        if z_motion is None:
            z_motion = torch.randn(num_samples, self.motion_dim, device=device)
        with torch.no_grad():
            return self.decode(z_content, z_motion, sequence_length)

    def swap_content(self, video_content: torch.Tensor, video_motion: torch.Tensor) -> torch.Tensor:
        with torch.no_grad():
            z_c = self.encode(video_content)[0]
            z_m = self.encode(video_motion)[1]
            return self.decode(z_c, z_m, video_content.shape[1])

    def swap_motion(self, video_content, video_motion):
        # alias with the same argument order, as in the reference (:151-156)
        return self.swap_content(video_content, video_motion)

    def _interp(self, a: torch.Tensor, b: torch.Tensor, num_steps: int) -> torch.Tensor:
        alphas = torch.linspace(0, 1, num_steps, device=a.device)
        return torch.cat([(1 - al) * a + al * b for al in alphas], dim=0)

    def interpolate_content(self, video1, video2, num_steps: int = 10, use_motion_from: str = 'first'):
        with torch.no_grad():
            c1, m1, _, _ = self.encode(video1)
            c2, m2, _, _ = self.encode(video2)
            z_m = (m1 if use_motion_from == 'first' else m2).expand(num_steps, -1)
            return self.decode(self._interp(c1, c2, num_steps), z_m.contiguous(), video1.shape[1])

    def interpolate_motion(self, video1, video2, num_steps: int = 10, use_content_from: str = 'first'):
        with torch.no_grad():
            c1, m1, _, _ = self.encode(video1)
            c2, m2, _, _ = self.encode(video2)
            z_c = (c1 if use_content_from == 'first' else c2).expand(num_steps, -1)
            return self.decode(z_c.contiguous(), self._interp(m1, m2, num_steps), video1.shape[1])

    def get_content_latent(self, video):
        with torch.no_grad():
            return self.content_encoder(video, return_distribution=True)[1][0]

    def get_motion_latent(self, video):
        with torch.no_grad():
            return self.motion_encoder(video, return_distribution=True)[1][0]

    def get_latents(self, video) -> Tuple[torch.Tensor, torch.Tensor]:
        return self.get_content_latent(video), self.get_motion_latent(video)


class DisentangledVAEWrapper(nn.Module):
    """Tuple-API adapter (reference: models/disentangled_vae.py:295-328)."""

    def __init__(self, model: DisentangledVAE):
        super().__init__()
        self.model = model
        self.latent_dim = model.latent_dim
        self.content_dim = model.content_dim
        self.motion_dim = model.motion_dim
        self.model_type = 'disentangled_vae'

    def forward(self, video):
        o = self.model(video)
        cat = lambda a, b: torch.cat([o[a], o[b]], dim=1)
        return (o['recon'], cat('z_content', 'z_motion'), cat('mu_content', 'mu_motion'),
                cat('logvar_content', 'logvar_motion'))

    def encode(self, video):
        z_c, z_m, (mu_c, lv_c), (mu_m, lv_m) = self.model.encode(video)
        return torch.cat([z_c, z_m], 1), torch.cat([mu_c, mu_m], 1), torch.cat([lv_c, lv_m], 1)

    def decode(self, z, sequence_length: Optional[int] = None):
        return self.model.decode(z[:, :self.content_dim].contiguous(), z[:, self.content_dim:].contiguous(),
                                 sequence_length)

    def sample(self, num_samples: int = 1, device=None, sequence_length: Optional[int] = None):
        return self.model.sample(num_samples, device, sequence_length)
