"""Evaluation helpers that stay on the device (reference package: evaluation/)."""
from .metrics import mse, psnr, reconstruction_metrics, temporal_consistency  # noqa: F401
