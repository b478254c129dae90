"""Model factory with the reference's names (models/__init__.py:38-76)."""
from .encoders import (CNNEncoder, LightCNNEncoder, VideoEncoder, LSTMEncoder, VAEEncoder, ContentEncoder,
                       MotionEncoder, reparameterize)
from .decoders import CNNDecoder, LSTMDecoder, VideoDecoder, DisentangledDecoder
from .baseline_vae import BaselineVAE
from .lstm_vae import LSTMVAE
from .disentangled_vae import DisentangledVAE, DisentangledVAEWrapper
from hipvae.losses import baseline_loss, lstm_loss, disentangled_loss

_MODELS = {'baseline': BaselineVAE, 'lstm': LSTMVAE, 'disentangled': DisentangledVAE}
_LOSSES = {'baseline': baseline_loss, 'lstm': lstm_loss, 'disentangled': disentangled_loss}


def get_model(model_type: str, **kwargs):
    if model_type not in _MODELS:
        raise ValueError(f"Unknown model type: {model_type}")
    return _MODELS[model_type](**kwargs)


def get_loss_fn(model_type: str):
    if model_type not in _LOSSES:
        raise ValueError(f"Unknown model type: {model_type}")
    return _LOSSES[model_type]


__all__ = ['CNNEncoder', 'LightCNNEncoder', 'VideoEncoder', 'LSTMEncoder', 'VAEEncoder', 'ContentEncoder',
           'MotionEncoder', 'reparameterize', 'CNNDecoder', 'LSTMDecoder', 'VideoDecoder', 'DisentangledDecoder',
           'BaselineVAE', 'LSTMVAE', 'DisentangledVAE', 'DisentangledVAEWrapper', 'baseline_loss', 'lstm_loss',
           'disentangled_loss', 'get_model', 'get_loss_fn']
