"""Experiment configuration: three flat sections (data, model, training) that round-trip through
JSON.  Field names, defaults and the JSON shape are those of the reference (training/config.py:7-127),
so its config files load unchanged; unknown keys are rejected, missing keys take the defaults."""
import json
import os
from dataclasses import asdict, dataclass, field, fields
from typing import Any, Dict, List, Optional


def _list(*v):
    return field(default_factory=lambda: list(v))


@dataclass
class DataConfig:
    root: str = './data'
    train_size: int = 10000
    val_size: int = 1000
    test_size: int = 1000
    sequence_length: int = 20
    frame_size: int = 64
    num_digits: int = 2
    batch_size: int = 32
    num_workers: int = 4
    seed: int = 42


@dataclass
class ModelConfig:
    model_type: str = 'disentangled'          # 'baseline' | 'lstm' | 'disentangled'
    latent_dim: int = 256                     # baseline / lstm
    content_dim: int = 128                    # disentangled
    motion_dim: int = 128
    encoder_channels: List[int] = _list(32, 64, 128, 256)
    decoder_channels: List[int] = _list(256, 128, 64, 32)
    motion_channels: List[int] = _list(16, 32)
    feature_dim: int = 512
    lstm_hidden: int = 256
    lstm_layers: int = 1
    bidirectional: bool = True
    in_channels: int = 1
    sequence_length: int = 20
    content_aggregation: str = 'mean'         # 'mean' | 'middle'


@dataclass
class TrainingConfig:
    learning_rate: float = 1e-4
    beta1: float = 0.9
    beta2: float = 0.999
    weight_decay: float = 1e-5
    beta_vae: float = 4.0
    beta_content: Optional[float] = None
    beta_motion: Optional[float] = None
    temporal_weight: float = 0.0
    independence_weight: float = 0.0
    recon_loss_type: str = 'mse'              # 'mse' | 'bce'
    use_kl_annealing: bool = True
    warmup_epochs: int = 10
    annealing_type: str = 'linear'            # 'linear' | 'cosine' | 'cyclical'
    kl_start_epoch: int = 0
    use_free_bits: bool = True
    free_bits: float = 0.5
    num_epochs: int = 100
    gradient_clip: float = 1.0
    lr_scheduler: str = 'exponential'         # 'exponential' | 'cosine' | 'none'
    lr_decay: float = 0.98
    lr_min: float = 1e-6
    device: str = 'mps'                       # kept for file compatibility; this build trains on 'cuda'
    save_every: int = 10
    log_every: int = 100
    visualize_every: int = 5
    checkpoint_dir: str = './checkpoints'
    results_dir: str = './results'


_SECTIONS = (('data', DataConfig), ('model', ModelConfig), ('training', TrainingConfig))


@dataclass
class ExperimentConfig:
    name: str = 'experiment'
    data: DataConfig = field(default_factory=DataConfig)
    model: ModelConfig = field(default_factory=ModelConfig)
    training: TrainingConfig = field(default_factory=TrainingConfig)

    def to_dict(self) -> Dict[str, Any]:
        out: Dict[str, Any] = {'name': self.name}
        for key, _ in _SECTIONS:
            out[key] = dict(vars(getattr(self, key)))
        return out

    def save(self, path: str):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, 'w') as f:
            json.dump(self.to_dict(), f, indent=2)

    @classmethod
    def load(cls, path: str) -> 'ExperimentConfig':
        with open(path) as f:
            raw = json.load(f)
        cfg = cls(name=raw['name'])
        for key, klass in _SECTIONS:
            setattr(cfg, key, klass(**raw[key]))      # unknown keys raise TypeError, as in the reference
        return cfg


def _preset(name: str, model_type: str, **model_over) -> ExperimentConfig:
    cfg = ExperimentConfig(name=name)
    cfg.model.model_type = model_type
    for k, v in model_over.items():
        setattr(cfg.model, k, v)
    return cfg


def get_baseline_config(name: str = 'baseline_vae') -> ExperimentConfig:
    return _preset(name, 'baseline', latent_dim=256)


def get_lstm_config(name: str = 'lstm_vae') -> ExperimentConfig:
    return _preset(name, 'lstm', latent_dim=256, lstm_hidden=256, bidirectional=True)


def get_disentangled_config(name: str = 'disentangled_vae') -> ExperimentConfig:
    cfg = _preset(name, 'disentangled', content_dim=128, motion_dim=128)
    cfg.training.independence_weight = 0.1
    return cfg


def get_ablation_configs() -> Dict[str, ExperimentConfig]:
    """The reference's ablation grid (training/config.py:158-187): beta, latent size, aggregation, direction."""
    grid: Dict[str, ExperimentConfig] = {}
    for beta in (1.0, 2.0, 4.0, 8.0, 16.0):
        c = get_disentangled_config(f'ablation_beta_{beta}')
        c.training.beta_vae = beta
        grid[f'beta_{beta}'] = c
    for dim in (64, 128, 256):
        c = get_disentangled_config(f'ablation_dim_{dim}')
        c.model.content_dim = c.model.motion_dim = dim
        grid[f'dim_{dim}'] = c
    for agg in ('mean', 'middle'):
        c = get_disentangled_config(f'ablation_agg_{agg}')
        c.model.content_aggregation = agg
        grid[f'agg_{agg}'] = c
    for bidir in (True, False):
        c = get_disentangled_config(f'ablation_bidir_{bidir}')
        c.model.bidirectional = bidir
        grid[f'bidir_{bidir}'] = c
    return grid


BASELINE_CONFIG = get_baseline_config()
LSTM_CONFIG = get_lstm_config()
DISENTANGLED_CONFIG = get_disentangled_config()
