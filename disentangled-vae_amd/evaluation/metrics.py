"""Reconstruction metrics of the evaluation loop, computed on the GPU.

Same names, arguments and returned keys as the reference (evaluation/metrics.py:14-93).  The reference
makes three passes per batch (two F.mse_loss calls and one difference tensor), reads each result back
with .item() and computes SSIM per frame on the host with scikit-image.  Here one kernel pass per batch
(hipvae.ops.recon_metrics) yields mean squared error, PSNR and temporal consistency together; the
evaluation loop keeps the per-batch values on the device and reads them back once at the end.
SSIM is a host-side scikit-image computation in the reference and is not part of this path: the 'ssim'
keys are returned as NaN.
"""
from typing import Dict, Optional

import numpy as np
import torch

from hipvae import ops


def _pair(pred: torch.Tensor, target: torch.Tensor):
    """both clips as contiguous fp32 [B, T, ...] on the GPU (a lower-rank tensor counts as one clip)"""
    if pred.shape != target.shape:
        raise ValueError(f"pred and target differ in shape: {tuple(pred.shape)} vs {tuple(target.shape)}")
    if pred.dim() < 3:
        pred, target = pred.reshape(1, 1, -1), target.reshape(1, 1, -1)
    return pred.contiguous().float(), target.contiguous().float()


def mse(pred: torch.Tensor, target: torch.Tensor) -> float:
    return float(ops.recon_metrics(*_pair(pred, target))[0].item())


def psnr(pred: torch.Tensor, target: torch.Tensor, max_val: float = 1.0) -> float:
    return float(ops.recon_metrics(*_pair(pred, target), max_val=max_val)[1].item())


def temporal_consistency(video: torch.Tensor) -> float:
    v = video.contiguous().float()
    return float(ops.recon_metrics(v, v)[2].item())


def reconstruction_metrics(model: torch.nn.Module, dataloader, device: torch.device,
                           max_batches: Optional[int] = None) -> Dict[str, float]:
    """Per-batch mse / psnr / temporal consistency of the model's reconstructions, averaged over
    batches (reference: evaluation/metrics.py:53-93)."""
    model.eval()
    rows = []
    with torch.no_grad():
        for i, batch in enumerate(dataloader):
            if max_batches and i >= max_batches:
                break
            video = (batch[0] if isinstance(batch, (tuple, list)) else batch).to(device)
            out = model(video)
            recon = out['recon'] if getattr(model, 'model_type', 'unknown') == 'disentangled_vae' else out[0]
            rows.append(ops.recon_metrics(recon.contiguous(), video.contiguous()))
    if not rows:
        raise ValueError("reconstruction_metrics: the dataloader produced no batches")
    vals = torch.stack(rows).double().cpu().numpy()              # the only read-back
    nan = float('nan')
    return {'mse': float(np.mean(vals[:, 0])), 'psnr': float(np.mean(vals[:, 1])), 'ssim': nan,
            'temporal_consistency': float(np.mean(vals[:, 2])), 'mse_std': float(np.std(vals[:, 0])),
            'psnr_std': float(np.std(vals[:, 1])), 'ssim_std': nan}
