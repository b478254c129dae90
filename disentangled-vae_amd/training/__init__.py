"""Training package: configuration dataclasses, the Trainer and the command-line entry point
(`python -m training.train --config <json> --device cuda`, run from this directory)."""
from .config import (DataConfig, ModelConfig, TrainingConfig, ExperimentConfig, get_baseline_config,
                     get_lstm_config, get_disentangled_config, get_ablation_configs)

__all__ = ['DataConfig', 'ModelConfig', 'TrainingConfig', 'ExperimentConfig', 'get_baseline_config',
           'get_lstm_config', 'get_disentangled_config', 'get_ablation_configs', 'Trainer', 'create_trainer']


def __getattr__(name):
    # the Trainer pulls in the HIP library; keep `import training.config` usable without it
    if name in ('Trainer', 'create_trainer'):
        from . import trainer
        return getattr(trainer, name)
    raise AttributeError(name)
