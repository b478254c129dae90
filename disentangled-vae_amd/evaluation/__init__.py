"""Evaluation helpers that stay on the device (reference package: evaluation/)."""
from .metrics import (compare_models, computational_efficiency, generation_diversity, mse, psnr,  # noqa: F401
                      reconstruction_metrics, temporal_consistency)
