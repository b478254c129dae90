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
    ops.check_lstm_seq_timeouts()        # a recurrence launch that was not fully resident voids the numbers
    nan = float('nan')
    return {'mse': float(np.mean(vals[:, 0])), 'psnr': float(np.mean(vals[:, 1])), 'ssim': nan,
            'temporal_consistency': float(np.mean(vals[:, 2])), 'mse_std': float(np.std(vals[:, 0])),
            'psnr_std': float(np.std(vals[:, 1])), 'ssim_std': nan}


def generation_diversity(model: torch.nn.Module, num_samples: int = 100,
                         device: Optional[torch.device] = None) -> Dict[str, float]:
    """Variance across prior samples, mean pairwise distance and temporal variance of decoded prior draws
    (reference: evaluation/metrics.py:95-121).  The statistics are a few reductions over the decoded clips
    and stay on the device until one read-back."""
    if device is None:
        device = next(model.parameters()).device
    model.eval()
    with torch.no_grad():
        samples = model.sample(num_samples=num_samples, device=device)
        flat = samples.reshape(num_samples, -1)
        dist = torch.cdist(flat, flat)
        upper = torch.triu(torch.ones_like(dist), diagonal=1).bool()
        vals = torch.stack([samples.var(dim=0).mean(), dist[upper].mean(), samples.var(dim=1).mean()]).cpu()
    return {'sample_variance': float(vals[0]), 'mean_pairwise_distance': float(vals[1]),
            'temporal_variance': float(vals[2])}


def computational_efficiency(model: torch.nn.Module, video: torch.Tensor, num_runs: int = 100,
                             device: Optional[torch.device] = None) -> Dict[str, float]:
    """Parameter counts and the mean / standard deviation of the time of encode(), forward() and sample()
    on `video`'s batch (reference: evaluation/metrics.py:124-188, same keys, milliseconds).

    The reference brackets every call with a host clock and a device synchronisation, so what it reports
    for small batches is mostly launch and synchronisation latency.  Here every run is bracketed by two
    events on the stream and the host synchronises once per phase: the numbers are the GPU time of each
    call as a queued serving loop sees it.  Ten untimed warm-up calls first, as in the reference."""
    if device is None:
        device = next(model.parameters()).device
    device = torch.device(device)
    if device.type != 'cuda':
        raise RuntimeError("computational_efficiency: this build runs on the GPU only")
    model.eval()
    video = video.to(device)
    out = {'total_parameters': sum(p.numel() for p in model.parameters()),
           'trainable_parameters': sum(p.numel() for p in model.parameters() if p.requires_grad)}

    def timed(fn):
        pairs = []
        with torch.no_grad():
            for _ in range(num_runs):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                fn()
                e.record()
                pairs.append((s, e))
        torch.cuda.synchronize(device)
        ms = np.array([s.elapsed_time(e) for s, e in pairs], dtype=np.float64)
        return float(ms.mean()), float(ms.std())

    with torch.no_grad():
        for _ in range(10):
            model(video)
    phases = (('encode', (lambda: model.encode(video)) if hasattr(model, 'encode') else (lambda: model(video))),
              ('forward', lambda: model(video)),
              ('generation', lambda: model.sample(num_samples=video.size(0), device=device)))
    for name, fn in phases:
        out[f'{name}_time_ms'], out[f'{name}_time_std_ms'] = timed(fn)
    return out


def compare_models(models: Dict[str, torch.nn.Module], dataloader, device: torch.device,
                   max_batches: int = 50) -> Dict[str, Dict[str, float]]:
    """reconstruction_metrics + generation_diversity + computational_efficiency per model
    (reference: evaluation/metrics.py:191-212; the timing sample is the loader's first four clips)."""
    results = {}
    sample_video = next(iter(dataloader))[0][:4].to(device)
    for name, model in models.items():
        print(f"\nEvaluating {name}")
        model = model.to(device)
        results[name] = {**reconstruction_metrics(model, dataloader, device, max_batches),
                         **generation_diversity(model, num_samples=50, device=device),
                         **computational_efficiency(model, sample_video, num_runs=50, device=device)}
    return results
