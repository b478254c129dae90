"""Tensor-level wrappers over the C ABI (include/dvae_hip.h).

Every function takes CUDA(=HIP) fp32 tensors, checks what the kernels assume, and enqueues on
torch's current stream.  PyTorch is used for device memory and streams only; there is no
alternative code path: a CPU tensor is an error.
"""
import ctypes
import os
from typing import List, Optional, Sequence, Tuple

import torch

from ._lib import check, lib

ACT_NONE, ACT_LRELU, ACT_SIGMOID = 0, 1, 2
Tensor = torch.Tensor


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _chk(*ts: Optional[Tensor]):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("hipvae ops need tensors on the GPU (no CPU path exists)")
        if t.dtype != torch.float32:
            raise RuntimeError(f"hipvae ops are fp32; got {t.dtype}")
        if not t.is_contiguous():
            raise RuntimeError("hipvae ops need contiguous tensors")


def _p(t: Optional[Tensor]):
    return None if t is None else t.data_ptr()


# bf16 storage of the large few-channel activations / gradients of the decoder (bf16-MFMA mode only; the flags of
# include/dvae_hip.h).  DVAE_BF16_STORAGE=0 keeps every tensor fp32.
ST_X, ST_Y, ST_BNY, ST_INY, ST_DX = 1, 2, 4, 8, 16
USE_BF16_STORAGE = os.environ.get('DVAE_BF16_STORAGE', '1') != '0'


def bf16_storage() -> bool:
    return USE_BF16_STORAGE and lib.dvae_get_mfma_precision() == 1


def _chk_st(*ts: Optional[Tensor]):
    """like _chk, for tensors that may be stored as bfloat16"""
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda or t.dtype not in (torch.float32, torch.bfloat16) or not t.is_contiguous():
            raise RuntimeError("hipvae ops need contiguous fp32 (or, where the kernel takes it, bf16) tensors on the GPU")


def _st(flag: int, t: Optional[Tensor]) -> int:
    return flag if t is not None and t.dtype == torch.bfloat16 else 0


_WS = {}
_WS_RETIRED = []
_WS_SLOT = ['main']


class workspace_slot:
    """Ops launched inside the block take their scratch from the named slot instead of 'main': work that runs
    on a second stream beside the main chain must not share the main chain's scratch."""

    def __init__(self, slot: str):
        self.slot = slot

    def __enter__(self):
        _WS_SLOT.append(self.slot)
        return self

    def __exit__(self, *exc):
        _WS_SLOT.pop()
        return False


# streams whose ops take their scratch from a slot of their own whenever they run there (forward under a stream
# context, backward wherever autograd replays the forward's stream): raw stream handle -> slot name
_STREAM_SLOT = {}


def register_stream_slot(stream: 'torch.cuda.Stream', slot: str) -> None:
    _STREAM_SLOT[stream.cuda_stream] = slot


def workspace(nfloats: int, device, slot: Optional[str] = None) -> Tensor:
    """Grow-only scratch buffer per (device, slot); contents are only valid within one op."""
    if slot is None:
        slot = _WS_SLOT[-1]
        if _STREAM_SLOT and slot == 'main':
            slot = _STREAM_SLOT.get(torch.cuda.current_stream(device).cuda_stream, 'main')
    key = (torch.device(device).index, slot)
    buf = _WS.get(key)
    if buf is None or buf.numel() < nfloats:
        if buf is not None:
            _WS_RETIRED.append(buf)      # a captured graph may still point at it: never hand it back
        buf = torch.empty(max(int(nfloats), 1024), dtype=torch.float32, device=device)
        _WS[key] = buf
    return buf


_TICKETS = {}
# DVAE_ONE_LAUNCH_SUMS=1: bias-gradient and BatchNorm-backward column sums in one launch each (the workgroup that
# finishes last adds the partial rows inside the launch) instead of two.  Off by default: on MI355X the replayed
# step was 0.3 % (batch 256) to 0.8 % (batch 32) SLOWER with it in a same-box A/B -- a dependent tiny kernel costs
# about 1.6 us inside a replayed graph, less than the in-launch hand-off and the serial tail it buys.
USE_TICKETS = os.environ.get('DVAE_ONE_LAUNCH_SUMS', '0') == '1'


def tickets(device) -> Optional[Tensor]:
    """The arrival counters of the one-launch reductions: zeroed 32-bit words per (device, workspace slot) that every
    kernel using them leaves at zero again.  Launches that share them are ordered on one stream (a slot is what a
    stream's work takes its scratch from), so they never meet."""
    if not USE_TICKETS:
        return None
    slot = _WS_SLOT[-1]
    if _STREAM_SLOT and slot == 'main':
        slot = _STREAM_SLOT.get(torch.cuda.current_stream(device).cuda_stream, 'main')
    key = (torch.device(device).index, slot)
    t = _TICKETS.get(key)
    if t is None:
        t = _TICKETS[key] = torch.zeros(int(lib.dvae_colsum_tickets()), dtype=torch.int32, device=device)
    return t


# ---------------------------------------------------------------------------- split
def split_mean_diff(video: Tensor) -> Tuple[Tensor, Tensor]:
    """video [B,T,C,H,W] -> (mean over T [B,C,H,W], frame differences [B,T-1,C,H,W])."""
    _chk(video)
    B, T = video.shape[:2]
    P = video[0, 0].numel()
    mean = torch.empty((B,) + tuple(video.shape[2:]), dtype=torch.float32, device=video.device)
    diff = torch.empty((B, max(T - 1, 0)) + tuple(video.shape[2:]), dtype=torch.float32, device=video.device)
    check(lib.dvae_split_mean_diff_f32(_p(video), _p(mean), _p(diff), B, T, P, _stream()), 'split_mean_diff')
    return mean, diff


def render_clips(bank: Tensor, digit_idx: Tensor, positions: Tensor, frame_size: int) -> Tensor:
    """bank [N,D,D] f32, digit_idx [B,nd] int32, positions [B,T,nd,2] int32 (x, y) -> clips [B,T,1,S,S]."""
    _chk(bank)
    for t in (digit_idx, positions):
        if not t.is_cuda or t.dtype != torch.int32 or not t.is_contiguous():
            raise RuntimeError("render_clips: digit_idx and positions must be contiguous int32 tensors on the GPU")
    B, T, nd, _ = positions.shape
    video = torch.empty(B, T, 1, frame_size, frame_size, dtype=torch.float32, device=bank.device)
    check(lib.dvae_render_clips_f32(_p(bank), bank.shape[0], digit_idx.data_ptr(), positions.data_ptr(), _p(video),
                                    B, T, nd, frame_size, bank.shape[1], _stream()), 'render_clips')
    return video


# ---------------------------------------------------------------------------- losses
def recon_loss_fwd(recon: Tensor, target: Tensor, kind: int, with_temporal: bool, out: Tensor):
    _chk(recon, target, out)
    B, T = recon.shape[:2]
    P = recon[0, 0].numel()
    ws = workspace(lib.dvae_recon_loss_ws_floats(B, T, P), recon.device)
    check(lib.dvae_recon_loss_fwd_f32(_p(recon), _p(target), _p(out), _p(ws), B, T, P, kind,
                                      int(with_temporal), _stream()), 'recon_loss_fwd')


def recon_metrics(recon: Tensor, target: Tensor, max_val: float = 1.0, out: Optional[Tensor] = None) -> Tensor:
    """recon, target [B,T,...] -> device tensor [mse, psnr, temporal consistency of recon] in one pass
    (evaluation/metrics.py:14-22, :48-50 of the reference); nothing is read back."""
    _chk(recon, target)
    if recon.shape != target.shape:
        raise RuntimeError(f"recon_metrics: shapes differ: {tuple(recon.shape)} vs {tuple(target.shape)}")
    B, T = recon.shape[:2]
    P = recon[0, 0].numel()
    if out is None:
        out = torch.empty(3, dtype=torch.float32, device=recon.device)
    ws = workspace(lib.dvae_recon_loss_ws_floats(B, T, P), recon.device)
    check(lib.dvae_recon_metrics_f32(_p(recon), _p(target), _p(out), _p(ws), B, T, P, float(max_val), _stream()),
          'recon_metrics')
    return out


def recon_metrics_noalloc(recon: Tensor, target: Tensor, out: Tensor, max_val: float = 1.0) -> Tensor:
    """recon_metrics through the entry point that allocates nothing: the caller brings `out` (3 floats)
    and the partial sums go to the library's own per-GPU array instead of a workspace tensor.  For callers
    that cannot allocate; calls that overlap in time on one GPU race on that array."""
    _chk(recon, target, out)
    if recon.shape != target.shape or out.numel() < 3:
        raise RuntimeError("recon_metrics_noalloc: recon/target shapes differ or out has fewer than 3 elements")
    B, T = recon.shape[:2]
    P = recon[0, 0].numel()
    with torch.cuda.device(recon.device):
        check(lib.dvae_recon_metrics_noalloc_f32(_p(recon), _p(target), _p(out), B, T, P, float(max_val),
                                                 torch.cuda.current_stream().cuda_stream), 'recon_metrics_noalloc')
    return out


def recon_loss_bwd(recon: Tensor, target: Tensor, gscale: Optional[Tensor], kind: int,
                   temporal_weight: float) -> Tensor:
    _chk(recon, target, gscale)
    B, T = recon.shape[:2]
    P = recon[0, 0].numel()
    d = torch.empty_like(recon)
    check(lib.dvae_recon_loss_bwd_f32(_p(recon), _p(target), _p(gscale), _p(d), B, T, P, kind,
                                      float(temporal_weight), _stream()), 'recon_loss_bwd')
    return d


def recon_loss_bwd_logit(recon: Tensor, target: Tensor, gscale: Optional[Tensor], kind: int,
                         temporal_weight: float):
    """recon_loss_bwd taken through the sigmoid that produced `recon`, in one pass: (dlogit, dbias)."""
    _chk(recon, target, gscale)
    B, T = recon.shape[:2]
    P = recon[0, 0].numel()
    d = torch.empty_like(recon)
    db = torch.empty(1, dtype=torch.float32, device=recon.device)
    ws = workspace(lib.dvae_recon_loss_bwd_logit_ws_floats(B, P), recon.device)
    check(lib.dvae_recon_loss_bwd_logit_f32(_p(recon), _p(target), _p(gscale), _p(d), _p(db), _p(ws), B, T, P, kind,
                                            float(temporal_weight), _stream()), 'recon_loss_bwd_logit')
    return d, db


def reparam_fwd(mu, logvar, eps):
    _chk(mu, logvar, eps)
    z = torch.empty_like(mu)
    check(lib.dvae_reparam_fwd_f32(_p(mu), _p(logvar), _p(eps), _p(z), mu.numel(), _stream()), 'reparam_fwd')
    return z


def reparam_bwd(dz, logvar, eps):
    _chk(dz, logvar, eps)
    dmu = torch.empty_like(dz)
    dlv = torch.empty_like(dz)
    check(lib.dvae_reparam_bwd_f32(_p(dz), _p(logvar), _p(eps), _p(dmu), _p(dlv), dz.numel(), _stream()),
          'reparam_bwd')
    return dmu, dlv


def kl_fwd(mu, logvar, free_bits: float, out: Tensor):
    _chk(mu, logvar, out)
    B, D = mu.shape
    check(lib.dvae_kl_fwd_f32(_p(mu), _p(logvar), _p(out), B, D, float(free_bits), _stream()), 'kl_fwd')


def _dev_coef(latents, beta_dev):
    """(plain coefficients, mask of the latents whose coefficient is the device scalar beta_dev): a latent whose
    coefficient is given as None takes it from beta_dev."""
    mask = 0
    for p, l in enumerate(latents):
        if l[2] is None:
            if beta_dev is None:
                raise RuntimeError("a latent without a coefficient needs beta_dev")
            mask |= 1 << p
    if mask:
        _chk(beta_dev)
        if beta_dev.numel() != 1:
            raise RuntimeError("beta_dev must hold one float")
    return [0.0 if l[2] is None else float(l[2]) for l in latents], mask


def kl_bwd(mu, logvar, gscale, coef: Optional[float], free_bits: float, beta_dev: Optional[Tensor] = None):
    """coef None: the coefficient is the device scalar beta_dev (read by the kernel at run time)"""
    _chk(mu, logvar, gscale)
    B, D = mu.shape
    dmu = torch.empty_like(mu)
    dlv = torch.empty_like(mu)
    (c,), mask = _dev_coef([(mu, logvar, coef)], beta_dev)
    check(lib.dvae_kl_bwd_f32(_p(mu), _p(logvar), _p(gscale), c, _p(beta_dev) if mask else None, _p(dmu), _p(dlv),
                              B, D, float(free_bits), _stream()), 'kl_bwd')
    return dmu, dlv


def vae_loss_fwd(recon, target, kind: int, with_temporal: bool, latents, free_bits: float, temporal_weight: float,
                 indep: Optional[Tensor], indep_weight: float, out: Tensor, total_out: Optional[Tensor],
                 beta_dev: Optional[Tensor] = None):
    """The objective of one compute_loss call in two launches.  latents: [(mu, logvar, beta)] (one entry: 7 values
    in `out`, two: 12; layouts in include/dvae_hip.h); beta None: that latent's weight is the device scalar
    beta_dev."""
    _chk(recon, target, indep, out, total_out, *[t for l in latents for t in l[:2]])
    B, T = recon.shape[:2]
    P = recon[0, 0].numel()
    assert len(latents) in (1, 2) and all(mu.shape[0] == B and mu.shape == lv.shape for mu, lv, _ in latents)
    assert out.numel() >= (7 if len(latents) == 1 else 12)
    ws = workspace(lib.dvae_recon_loss_ws_floats(B, T, P), recon.device)
    coefs, mask = _dev_coef(latents, beta_dev)
    mu0, lv0, _ = latents[0]
    mu1, lv1, _ = latents[1] if len(latents) == 2 else (None, None, 0.0)
    check(lib.dvae_vae_loss_fwd_f32(_p(recon), _p(target), B, T, P, kind, int(with_temporal), _p(mu0), _p(lv0),
                                    mu0.shape[1], coefs[0], _p(mu1), _p(lv1), 0 if mu1 is None else mu1.shape[1],
                                    coefs[1] if len(coefs) == 2 else 0.0, _p(beta_dev) if mask else None, mask,
                                    float(free_bits), float(temporal_weight), _p(indep),
                                    float(indep_weight), _p(out), _p(total_out), _p(ws), _stream()), 'vae_loss_fwd')


def kl_bwd2(latents, gscale, free_bits: float, beta_dev: Optional[Tensor] = None):
    """[(mu, logvar, coef)] (one or two) -> [(dmu, dlogvar)] of coef * KL (batch mean) * gscale[0], one launch;
    coef None: the device scalar beta_dev"""
    _chk(gscale, *[t for l in latents for t in l[:2]])
    B = latents[0][0].shape[0]
    outs = [(torch.empty_like(mu), torch.empty_like(mu)) for mu, _, _ in latents]
    coefs, mask = _dev_coef(latents, beta_dev)
    mu0, lv0, _ = latents[0]
    mu1, lv1, _ = latents[1] if len(latents) == 2 else (None, None, 0.0)
    o1 = outs[1] if len(latents) == 2 else (None, None)
    check(lib.dvae_kl_bwd2_f32(_p(mu0), _p(lv0), coefs[0], _p(outs[0][0]), _p(outs[0][1]), mu0.shape[1], _p(mu1),
                               _p(lv1), coefs[1] if len(coefs) == 2 else 0.0, _p(o1[0]), _p(o1[1]),
                               0 if mu1 is None else mu1.shape[1], _p(beta_dev) if mask else None, mask,
                               _p(gscale), B, float(free_bits), _stream()), 'kl_bwd2')
    return outs


def indep_fwd(zc, zm, out: Tensor) -> Tensor:
    """Returns the saved workspace (correlation + statistics) needed by indep_bwd."""
    _chk(zc, zm, out)
    B, Dc = zc.shape
    Dm = zm.shape[1]
    ws = torch.empty(lib.dvae_indep_ws_floats(Dc, Dm), dtype=torch.float32, device=zc.device)
    check(lib.dvae_indep_fwd_f32(_p(zc), _p(zm), _p(out), _p(ws), B, Dc, Dm, _stream()), 'indep_fwd')
    return ws


def indep_bwd(zc, zm, ws, gscale, coef: float):
    _chk(zc, zm, ws, gscale)
    B, Dc = zc.shape
    Dm = zm.shape[1]
    dzc = torch.empty_like(zc)
    dzm = torch.empty_like(zm)
    check(lib.dvae_indep_bwd_f32(_p(zc), _p(zm), _p(ws), _p(gscale), float(coef), _p(dzc), _p(dzm), B, Dc, Dm,
                                 _stream()), 'indep_bwd')
    return dzc, dzm


# ---------------------------------------------------------------------------- BatchNorm glue
PRECISIONS = {'fp32': 0, 'f32': 0, 'float32': 0, 'bf16': 1, 'bfloat16': 1, 'fp32_split': 2}


def set_precision(name: str) -> None:
    """Operand precision of the MFMA GEMM / convolution kernels: 'fp32' (default, exact products on the
    fp32 matrix instruction), 'bf16' (operands rounded to bf16 when staged into LDS, fp32 accumulation,
    fp32 tensors) or 'fp32_split' (Linear-layer GEMMs form each fp32 product from six bf16 x bf16
    products of a three-term split, fp32 accumulation; everything else as in 'fp32')."""
    if name not in PRECISIONS:
        raise ValueError(f"Unknown precision: {name}. Choose from {sorted(PRECISIONS)}")
    check(lib.dvae_set_mfma_precision(PRECISIONS[name]), 'set_mfma_precision')


def get_precision() -> str:
    code = lib.dvae_get_mfma_precision()
    return {0: 'fp32', 1: 'bf16', 2: 'fp32_split'}[code]


def bn_finalize(partials: Tensor, nparts: int, M: int, gamma, beta, running_mean, running_var, nbt,
                momentum: float, eps: float, replicas: int = 1):
    """-> (scale, shift, mean, invstd); updates the running buffers in place when given.  `replicas`: the layer's
    batch is that many identical copies of the M rows behind the partial sums (only the unbiased correction of the
    running variance sees it)."""
    _chk(partials, gamma, beta, running_mean, running_var)
    C = gamma.numel()
    out = torch.empty(4, C, dtype=torch.float32, device=gamma.device)
    check(lib.dvae_bn_finalize_rep_f32(_p(partials), nparts, M, int(replicas), C, _p(gamma), _p(beta),
                                       _p(running_mean), _p(running_var), None if nbt is None else nbt.data_ptr(),
                                       float(momentum), float(eps), _p(out[0]), _p(out[1]), _p(out[2]), _p(out[3]),
                                       _stream()), 'bn_finalize')
    return out[0], out[1], out[2], out[3]


def bn_eval_affine(gamma, beta, running_mean, running_var, eps: float):
    _chk(gamma, beta, running_mean, running_var)
    C = gamma.numel()
    out = torch.empty(2, C, dtype=torch.float32, device=gamma.device)
    check(lib.dvae_bn_eval_affine_f32(_p(gamma), _p(beta), _p(running_mean), _p(running_var), float(eps),
                                      _p(out[0]), _p(out[1]), C, _stream()), 'bn_eval_affine')
    return out[0], out[1]


def affine_act(x: Tensor, scale, shift, act: int) -> Tensor:
    _chk(x, scale, shift)
    C = x.shape[-1]
    out = torch.empty_like(x)
    check(lib.dvae_affine_act_f32(_p(x), _p(scale), _p(shift), act, _p(out), x.numel() // C, C, _stream()),
          'affine_act')
    return out


def bn_bwd(du: Tensor, y: Tensor, scale, shift, mean, invstd, gamma, partials: Optional[Tensor] = None):
    """Overwrites du with dL/dy; returns (dgamma, dbeta).  partials: the [rows][2][C] statistics a
    *_dgrad_bnstat call reduced while it produced du (skips the reduction pass)."""
    _chk(du, y, scale, shift, mean, invstd, gamma, partials)
    C = y.shape[-1]
    M = y.numel() // C
    dgb = torch.empty(2, C, dtype=torch.float32, device=y.device)
    if partials is not None:
        check(lib.dvae_bn_bwd_from_partials_f32(_p(du), _p(y), _p(scale), _p(shift), _p(mean), _p(invstd),
                                                _p(partials), partials.shape[0], _p(dgb[0]), _p(dgb[1]), M, C,
                                                _stream()), 'bn_bwd_from_partials')
        return dgb[0], dgb[1]
    ws = workspace(lib.dvae_colsum_ws_floats(M, C), y.device)
    check(lib.dvae_bn_bwd_f32(_p(du), _p(y), _p(scale), _p(shift), _p(mean), _p(invstd), _p(gamma), _p(dgb[0]),
                              _p(dgb[1]), _p(ws), _p(tickets(y.device)), M, C, _stream()), 'bn_bwd')
    return dgb[0], dgb[1]


def bias_act_bwd(dout: Tensor, out: Optional[Tensor], act: int, want_bias: bool,
                 dst: Optional[Tensor] = None) -> Optional[Tensor]:
    """dst <- dout * act'(out) (in place over dout when dst is None); returns column sums (bias gradient) if wanted."""
    _chk(dout, out, dst)
    N = dout.shape[-1]
    M = dout.numel() // N
    if dst is not None and dst.numel() != dout.numel():
        raise RuntimeError("bias_act_bwd: dst must have dout's size")
    db = torch.empty(N, dtype=torch.float32, device=dout.device) if want_bias else None
    ws = workspace(lib.dvae_colsum_ws_floats(M, N), dout.device)
    check(lib.dvae_bias_act_bwd_to_f32(_p(dout), _p(dout if dst is None else dst), _p(out), act, _p(db), _p(ws),
                                       _p(tickets(dout.device)), M, N, _stream()), 'bias_act_bwd')
    return db


def sigmoid_bwd(drecon: Tensor, recon: Tensor, want_bias: bool = True):
    _chk(drecon, recon)
    dl = torch.empty_like(drecon)
    db = torch.empty(1, dtype=torch.float32, device=recon.device) if want_bias else None
    ws = workspace(4096, recon.device)
    check(lib.dvae_sigmoid_bwd_f32(_p(drecon), _p(recon), _p(dl), _p(db), _p(ws), recon.numel(), _stream()),
          'sigmoid_bwd')
    return dl, db


# ---------------------------------------------------------------------------- layout helpers
def permute3(x: Tensor, dims: Tuple[int, int, int], perm: Tuple[int, int, int]) -> Tensor:
    """x viewed as contiguous [D0,D1,D2] -> contiguous x.permute(perm)."""
    _chk(x)
    D0, D1, D2 = dims
    assert x.numel() == D0 * D1 * D2
    out = torch.empty(dims[perm[0]], dims[perm[1]], dims[perm[2]], dtype=torch.float32, device=x.device)
    check(lib.dvae_permute3_f32(_p(x), _p(out), D0, D1, D2, perm[0], perm[1], perm[2], _stream()), 'permute3')
    return out


def weighted_sum(terms: Sequence[Tensor], coefs: Sequence[float], out: Tensor):
    """out[0] = sum_i coefs[i] * terms[i][0] for up to 8 device scalars."""
    n = len(terms)
    ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in terms])
    cf = (ctypes.c_float * n)(*[float(c) for c in coefs])
    check(lib.dvae_weighted_sum_f32(n, ptrs, cf, _p(out), _stream()), 'weighted_sum')


# ---------------------------------------------------------------------------- GEMMs
def _rows(t: Tensor) -> Tuple[int, int]:
    """(rows, row stride) of a 2-D tensor whose last dim is contiguous."""
    if t.dim() != 2 or t.stride(1) != 1:
        raise RuntimeError("expected a 2-D tensor with a contiguous last dimension")
    return t.shape[0], t.stride(0)


def _chk2(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda or t.dtype != torch.float32:
            raise RuntimeError("hipvae GEMMs need fp32 GPU tensors (no CPU path exists)")


def gemm_nt(a: Tensor, b: Tensor, bias: Optional[Tensor] = None, act: int = ACT_NONE,
            out: Optional[Tensor] = None) -> Tensor:
    """act(a[M,K] @ b[N,K]^T + bias)."""
    _chk2(a, b, bias, out)
    M, lda = _rows(a)
    N, ldb = _rows(b)
    K = a.shape[1]
    assert b.shape[1] == K
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    ldc = out.stride(0)
    ws = workspace(lib.dvae_gemm_ws_floats(M, N, K), a.device)
    check(lib.dvae_gemm_nt_f32(_p(a), lda, _p(b), ldb, _p(bias), act, _p(out), ldc, M, N, K, _p(ws), _stream()),
          'gemm_nt')
    return out


def gemm_nn(a: Tensor, b: Tensor, out: Optional[Tensor] = None, accumulate: bool = False) -> Tensor:
    """a[M,K] @ b[K,N]."""
    _chk2(a, b, out)
    M, lda = _rows(a)
    K, ldb = _rows(b)
    N = b.shape[1]
    assert a.shape[1] == K
    if out is None:
        assert not accumulate
        out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    ws = workspace(lib.dvae_gemm_ws_floats(M, N, K), a.device)
    check(lib.dvae_gemm_nn_f32(_p(a), lda, _p(b), ldb, _p(out), out.stride(0), M, N, K, int(accumulate), _p(ws),
                               _stream()), 'gemm_nn')
    return out


def gemm_tn(a: Tensor, b: Tensor, out: Optional[Tensor] = None, accumulate: bool = False) -> Tensor:
    """a[K,M]^T @ b[K,N]."""
    _chk2(a, b, out)
    K, lda = _rows(a)
    K2, ldb = _rows(b)
    assert K == K2
    M, N = a.shape[1], b.shape[1]
    if out is None:
        assert not accumulate
        out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    ws = workspace(lib.dvae_gemm_ws_floats(M, N, K), a.device)
    check(lib.dvae_gemm_tn_f32(_p(a), lda, _p(b), ldb, _p(out), out.stride(0), M, N, K, int(accumulate), _p(ws),
                               _stream()), 'gemm_tn')
    return out


# ---------------------------------------------------------------------------- conv family
def pack_conv_w(w: Tensor) -> Tensor:
    """[A][Bc][4][4] -> [A][4][4][Bc]"""
    _chk(w)
    A, Bc = w.shape[:2]
    wp = torch.empty(A, 4, 4, Bc, dtype=torch.float32, device=w.device)
    check(lib.dvae_pack_conv_w_f32(_p(w), _p(wp), A, Bc, _stream()), 'pack_conv_w')
    return wp


def pack_deconv_w(w: Tensor) -> Tensor:
    """[Kc][Nc][4][4] -> [4 phases][Nc][2][2][Kc]"""
    _chk(w)
    Kc, Nc = w.shape[:2]
    wp = torch.empty(4, Nc, 2, 2, Kc, dtype=torch.float32, device=w.device)
    check(lib.dvae_pack_deconv_w_f32(_p(w), _p(wp), Kc, Nc, _stream()), 'pack_deconv_w')
    return wp


# DVAE_PACK_MANY=0: one launch per weight (for A/B measurements)
PACK_IN_ONE_LAUNCH = os.environ.get('DVAE_PACK_MANY', '1') != '0'


def pack_many(items) -> list:
    """items: [(weight, transposed)]; -> the packed weights, pack_conv_w(weight) where transposed is false and
    pack_deconv_w(weight) where it is true, all written by one launch."""
    if not items:
        return []
    if not PACK_IN_ONE_LAUNCH:
        return [pack_deconv_w(w) if tr else pack_conv_w(w) for w, tr in items]
    ws = [w for w, _ in items]
    _chk(*ws)
    n = len(items)
    outs = []
    for w, tr in items:
        d0, d1 = w.shape[:2]
        shape = (4, d1, 2, 2, d0) if tr else (d0, 4, 4, d1)
        outs.append(torch.empty(*shape, dtype=torch.float32, device=w.device))
    PA, LA, IA = ctypes.c_void_p * n, ctypes.c_int64 * n, ctypes.c_int * n
    check(lib.dvae_pack_w_many_f32(n, PA(*[w.data_ptr() for w in ws]), PA(*[o.data_ptr() for o in outs]),
                                   LA(*[w.shape[0] for w in ws]), LA(*[w.shape[1] for w in ws]),
                                   IA(*[1 if tr else 0 for _, tr in items]), _stream()), 'pack_w_many')
    return outs


def conv4s2_fwd(x: Tensor, scale, shift, act: int, wp: Tensor, want_stats: bool, out_bf16: bool = False):
    """x [N,H,W,Cin] -> (y [N,H/2,W/2,Cout], stat partials or None, number of partial rows).  out_bf16: y is stored
    as bfloat16 (single-channel-input kernel in bf16 mode; statistics from the fp32 accumulators)."""
    _chk(scale, shift, wp)
    _chk_st(x)
    N, H, W, Cin = x.shape
    Cout = wp.shape[0]
    y = torch.empty(N, H // 2, W // 2, Cout, dtype=torch.bfloat16 if out_bf16 else torch.float32, device=x.device)
    nparts = lib.dvae_conv_stat_parts_written(N * (H // 2) * (W // 2), 1)
    stats = torch.empty(nparts, 2, Cout, dtype=torch.float32, device=x.device) if want_stats else None
    need = lib.dvae_conv_fwd_ws_floats(0, N, H, W, Cin, Cout)
    ws = workspace(need, x.device) if need else None
    check(lib.dvae_conv4s2_fwd_ws_f32(_p(x), _p(scale), _p(shift), act, _p(wp), _p(y), _p(stats), N, H, W, Cin, Cout,
                                      _p(ws), need, _st(ST_X, x) | _st(ST_Y, y), _stream()), 'conv4s2_fwd')
    return y, stats, nparts


def deconv4s2_fwd(x: Tensor, scale, shift, act: int, wp: Tensor, want_stats: bool, out_bf16: bool = False):
    """x [N,H,W,Cin] -> (y [N,2H,2W,Cout], stat partials or None, number of partial rows).  out_bf16: y is stored
    as bfloat16 (few-channel row kernels in bf16 mode; the statistics still come from the fp32 accumulators)."""
    _chk(scale, shift, wp)
    _chk_st(x)
    N, H, W, Cin = x.shape
    Cout = wp.shape[1]
    y = torch.empty(N, 2 * H, 2 * W, Cout, dtype=torch.bfloat16 if out_bf16 else torch.float32, device=x.device)
    nparts = lib.dvae_conv_stat_parts_written(N * H * W, 4)
    stats = torch.empty(nparts, 2, Cout, dtype=torch.float32, device=x.device) if want_stats else None
    need = lib.dvae_conv_fwd_ws_floats(1, N, H, W, Cin, Cout)
    ws = workspace(need, x.device) if need else None
    check(lib.dvae_deconv4s2_fwd_ws_f32(_p(x), _p(scale), _p(shift), act, _p(wp), _p(y), _p(stats), N, H, W, Cin,
                                        Cout, _p(ws), need, _st(ST_X, x) | _st(ST_Y, y), _stream()), 'deconv4s2_fwd')
    return y, stats, nparts


def conv4s2_dgrad_bnstat(x: Tensor, wp: Tensor, bn_y: Tensor, bn_aff, out_bf16: bool = False):
    """conv4s2_fwd used as a data gradient that also reduces the BatchNorm-backward statistics of the
    layer it back-propagates into: returns (du, partials [rows,2,Cout]).  bn_aff = (scale, shift, mean,
    invstd) of that layer's forward pass, bn_y its raw output (shaped like du)."""
    _chk(x, wp, *bn_aff)
    _chk_st(bn_y)
    N, H, W, Cin = x.shape
    Cout = wp.shape[0]
    y = torch.empty(N, H // 2, W // 2, Cout, dtype=torch.bfloat16 if out_bf16 else torch.float32, device=x.device)
    assert bn_y.shape == y.shape
    nparts = lib.dvae_conv_stat_parts_written(N * (H // 2) * (W // 2), 1)
    stats = torch.empty(nparts, 2, Cout, dtype=torch.float32, device=x.device)
    sc, sh, mean, invstd = bn_aff
    check(lib.dvae_conv4s2_dgrad_bnstat_f32(_p(x), _p(wp), _p(y), _p(stats), N, H, W, Cin, Cout, _p(bn_y), _p(sc),
                                            _p(sh), _p(mean), _p(invstd), _st(ST_Y, y) | _st(ST_BNY, bn_y), _stream()),
          'conv4s2_dgrad_bnstat')
    return y, stats


def bn_bwd_coeffs(partials: Tensor, aff, M: int):
    """From the partial rows of a *_dgrad_bnstat call and the layer's (scale, shift, mean, invstd):
    (dgamma, dbeta, A, B) with dL/dy = scale*dz + A*y + B."""
    sc, sh, mean, invstd = aff
    _chk(partials, sc, mean, invstd)
    C = sc.shape[0]
    out = torch.empty(4, C, dtype=torch.float32, device=sc.device)
    check(lib.dvae_bn_bwd_coeffs_f32(_p(partials), partials.shape[0], _p(sc), _p(mean), _p(invstd), _p(out[0]),
                                     _p(out[1]), _p(out[2]), _p(out[3]), M, C, _stream()), 'bn_bwd_coeffs')
    return out[0], out[1], out[2], out[3]


def bn_bwd_reduce_coeffs(du: Tensor, y: Tensor, aff):
    """bn_bwd_reduce + bn_bwd_coeffs in one launch: (dgamma, dbeta, A, B)."""
    sc, sh, mean, invstd = aff
    _chk(du, y, sc, sh, mean, invstd)
    C = y.shape[-1]
    M = y.numel() // C
    if not USE_TICKETS:
        return bn_bwd_coeffs(bn_bwd_reduce(du, y, aff), aff, M)
    out = torch.empty(4, C, dtype=torch.float32, device=sc.device)
    ws = workspace(lib.dvae_colsum_ws_floats(M, C), y.device)
    check(lib.dvae_bn_bwd_reduce_coeffs_f32(_p(du), _p(y), _p(sc), _p(sh), _p(mean), _p(invstd), _p(out[0]), _p(out[1]),
                                            _p(out[2]), _p(out[3]), _p(ws), _p(tickets(y.device)), M, C, _stream()),
          'bn_bwd_reduce_coeffs')
    return out[0], out[1], out[2], out[3]


def bn_bwd_reduce(du: Tensor, y: Tensor, aff) -> Tensor:
    """The BatchNorm-backward sums alone: partial rows [rows][2][C] for bn_bwd_coeffs."""
    sc, sh, mean, invstd = aff
    _chk(du, y, sc, sh, mean, invstd)
    C = y.shape[-1]
    M = y.numel() // C
    parts = torch.empty(lib.dvae_bn_bwd_reduce_rows(M, C), 2, C, dtype=torch.float32, device=y.device)
    check(lib.dvae_bn_bwd_reduce_f32(_p(du), _p(y), _p(sc), _p(sh), _p(mean), _p(invstd), _p(parts), M, C, _stream()),
          'bn_bwd_reduce')
    return parts


def deconv4s2_bnin_supported(x: Tensor, Cout: int) -> bool:
    N, H, W, Cin = x.shape
    return bool(lib.dvae_deconv4s2_bnin_supported(H, W, Cin, Cout))


def deconv4s2_dgrad_bnstat_bnin(du_in: Tensor, in_y: Tensor, in_aff, coef_a: Tensor, coef_b: Tensor, wp: Tensor,
                                bn_y: Tensor, bn_aff):
    """deconv4s2_dgrad_bnstat on an input that still needs its own BatchNorm backward (see
    conv4s2_dgrad_bnstat_bnin); returns (du, partials, dx).  A bfloat16 bn_y makes du bfloat16 too."""
    _chk(du_in, in_y, wp, coef_a, coef_b, *in_aff, *bn_aff)
    _chk_st(bn_y)
    assert du_in.shape == in_y.shape
    N, H, W, Cin = du_in.shape
    Cout = wp.shape[1]
    y = torch.empty(N, 2 * H, 2 * W, Cout, dtype=bn_y.dtype, device=du_in.device)
    assert bn_y.shape == y.shape
    dx = torch.empty_like(du_in)
    nparts = lib.dvae_conv_stat_parts_written(N * H * W, 4)
    stats = torch.empty(nparts, 2, Cout, dtype=torch.float32, device=du_in.device)
    sc, sh, mean, invstd = bn_aff
    check(lib.dvae_deconv4s2_dgrad_bnstat_bnin_f32(_p(du_in), _p(in_y), _p(in_aff[0]), _p(in_aff[1]), _p(coef_a),
                                                   _p(coef_b), _p(dx), _p(wp), _p(y), _p(stats), N, H, W, Cin, Cout,
                                                   _p(bn_y), _p(sc), _p(sh), _p(mean), _p(invstd),
                                                   _st(ST_Y, y) | _st(ST_BNY, bn_y), _stream()),
          'deconv4s2_dgrad_bnstat_bnin')
    return y, stats, dx


def conv4s2_wgrad_bnp(du: Tensor, p_y: Tensor, p_aff, coef_a: Tensor, coef_b: Tensor, Q: Tensor) -> Tensor:
    """conv4s2_wgrad with a single-channel Q whose P = du still needs its layer's BatchNorm backward."""
    _chk(Q, coef_a, coef_b, *p_aff)
    _chk_st(du, p_y)
    N, Hs, Ws, C1 = du.shape
    assert du.shape == p_y.shape and Q.shape == (N, 2 * Hs, 2 * Ws, 1)
    dw = torch.empty(C1, 1, 4, 4, dtype=torch.float32, device=du.device)
    ws = workspace(lib.dvae_conv_wgrad_ws_floats(N, Hs, Ws, C1, 1), du.device)
    check(lib.dvae_conv4s2_wgrad_bnp_f32(_p(du), _p(p_y), _p(p_aff[0]), _p(p_aff[1]), _p(coef_a), _p(coef_b), _p(Q),
                                         _p(dw), _p(ws), N, Hs, Ws, C1, _st(ST_X, du) | _st(ST_BNY, p_y), _stream()),
          'conv4s2_wgrad_bnp')
    return dw


def conv4s2_wgrad_bnq_supported(P: Tensor, C2: int) -> bool:
    N, Hs, Ws, C1 = P.shape
    return bool(lib.dvae_conv4s2_wgrad_bnq_supported(N, Hs, Ws, C1, C2))


def conv4s2_wgrad_bnq(P: Tensor, p_scale, p_shift, p_act: int, du: Tensor, q_y: Tensor, q_aff, coef_a: Tensor,
                      coef_b: Tensor):
    """conv4s2_wgrad whose Q = du still needs its layer's BatchNorm backward; returns (dw, dx) with dx the
    finished dL/d(q_y)."""
    _chk(P, p_scale, p_shift, coef_a, coef_b, *q_aff)
    _chk_st(du, q_y)
    N, Hs, Ws, C1 = P.shape
    C2 = du.shape[3]
    assert du.shape == q_y.shape == (N, 2 * Hs, 2 * Ws, C2)
    dw = torch.empty(C1, C2, 4, 4, dtype=torch.float32, device=P.device)
    dx = torch.empty(du.shape, dtype=torch.float32, device=du.device)        # (always floats: a tiled kernel reads it)
    ws = workspace(lib.dvae_conv_wgrad_ws_floats(N, Hs, Ws, C1, C2), P.device)
    check(lib.dvae_conv4s2_wgrad_bnq_f32(_p(P), _p(p_scale), _p(p_shift), p_act, _p(du), _p(q_y), _p(q_aff[0]),
                                         _p(q_aff[1]), _p(coef_a), _p(coef_b), _p(dx), _p(dw), _p(ws), N, Hs, Ws, C1,
                                         C2, _st(ST_Y, du) | _st(ST_INY, q_y), _stream()), 'conv4s2_wgrad_bnq')
    return dw, dx


def conv4s2_bnin_supported(x: Tensor, Cout: int) -> bool:
    N, H, W, Cin = x.shape
    return bool(lib.dvae_conv4s2_bnin_supported(H, W, Cin, Cout))


def conv4s2_dgrad_bnstat_bnin(du_in: Tensor, in_y: Tensor, in_aff, coef_a: Tensor, coef_b: Tensor, wp: Tensor,
                              bn_y: Tensor, bn_aff):
    """conv4s2_dgrad_bnstat on an input that still needs its own BatchNorm backward: du_in is the gradient
    of the activation of the layer whose raw output is in_y; returns (du, partials, dx) where dx is the
    finished dL/d(in_y) (for the weight-gradient call)."""
    _chk(wp, coef_a, coef_b, *in_aff, *bn_aff)
    _chk_st(du_in, in_y, bn_y)
    assert du_in.shape == in_y.shape
    N, H, W, Cin = du_in.shape
    Cout = wp.shape[0]
    y = torch.empty(N, H // 2, W // 2, Cout, dtype=bn_y.dtype, device=du_in.device)      # (bfloat16 when bn_y is)
    assert bn_y.shape == y.shape
    dx = torch.empty_like(du_in)                # (bfloat16 when du_in is)
    nparts = lib.dvae_conv_stat_parts_written(N * (H // 2) * (W // 2), 1)
    stats = torch.empty(nparts, 2, Cout, dtype=torch.float32, device=du_in.device)
    sc, sh, mean, invstd = bn_aff
    check(lib.dvae_conv4s2_dgrad_bnstat_bnin_f32(_p(du_in), _p(in_y), _p(in_aff[0]), _p(in_aff[1]), _p(coef_a),
                                                 _p(coef_b), _p(dx), _p(wp), _p(y), _p(stats), N, H, W, Cin, Cout,
                                                 _p(bn_y), _p(sc), _p(sh), _p(mean), _p(invstd),
                                                 _st(ST_X, du_in) | _st(ST_INY, in_y) | _st(ST_DX, dx)
                                                 | _st(ST_BNY, bn_y) | _st(ST_Y, y), _stream()),
          'conv4s2_dgrad_bnstat_bnin')
    return y, stats, dx


def deconv4s2_dgrad_bnstat(x: Tensor, wp: Tensor, bn_y: Tensor, bn_aff):
    """deconv4s2_fwd used as a data gradient, with the same fused statistics (see conv4s2_dgrad_bnstat)."""
    _chk(x, wp, bn_y, *bn_aff)
    N, H, W, Cin = x.shape
    Cout = wp.shape[1]
    y = torch.empty(N, 2 * H, 2 * W, Cout, dtype=torch.float32, device=x.device)
    assert bn_y.shape == y.shape
    nparts = lib.dvae_conv_stat_parts_written(N * H * W, 4)
    stats = torch.empty(nparts, 2, Cout, dtype=torch.float32, device=x.device)
    sc, sh, mean, invstd = bn_aff
    check(lib.dvae_deconv4s2_dgrad_bnstat_f32(_p(x), _p(wp), _p(y), _p(stats), N, H, W, Cin, Cout, _p(bn_y), _p(sc),
                                              _p(sh), _p(mean), _p(invstd), _stream()), 'deconv4s2_dgrad_bnstat')
    return y, stats


def conv4s2_wgrad(P: Tensor, p_scale, p_shift, p_act: int, Q: Tensor, q_scale, q_shift, q_act: int) -> Tensor:
    """P [N,Hs,Ws,C1], Q [N,2Hs,2Ws,C2] -> dw [C1][C2][4][4]."""
    _chk(p_scale, p_shift, q_scale, q_shift)
    _chk_st(P, Q)
    N, Hs, Ws, C1 = P.shape
    C2 = Q.shape[3]
    assert Q.shape[1] == 2 * Hs and Q.shape[2] == 2 * Ws and Q.shape[0] == N
    dw = torch.empty(C1, C2, 4, 4, dtype=torch.float32, device=P.device)
    ws = workspace(lib.dvae_conv_wgrad_ws_floats(N, Hs, Ws, C1, C2), P.device)
    check(lib.dvae_conv4s2_wgrad_f32(_p(P), _p(p_scale), _p(p_shift), p_act, _p(Q), _p(q_scale), _p(q_shift),
                                     q_act, _p(dw), _p(ws), N, Hs, Ws, C1, C2, _st(ST_X, P) | _st(ST_Y, Q), _stream()),
          'conv4s2_wgrad')
    return dw


def deconv4s2_final_fwd(x: Tensor, scale, shift, act: int, w: Tensor, bias: Optional[Tensor],
                        out_act: int) -> Tensor:
    """x [N,H,W,Cin], w [Cin][1][4][4] -> y [N,2H,2W]."""
    _chk(scale, shift, w, bias)
    _chk_st(x)
    N, H, W, Cin = x.shape
    y = torch.empty(N, 2 * H, 2 * W, dtype=torch.float32, device=x.device)
    check(lib.dvae_deconv4s2_final_fwd_f32(_p(x), _p(scale), _p(shift), act, _p(w), _p(bias), out_act, _p(y), N, H,
                                           W, Cin, _st(ST_X, x), _stream()), 'deconv4s2_final_fwd')
    return y


# ---------------------------------------------------------------------------- LSTM steps
def lstm_step_fwd(xproj, x, w_ih, h_prev, c_prev, w_hh, b_ih, b_hh, gates_out, h_out, c_out):
    """All row tensors are 2-D views with contiguous last dim (row strides are passed through)."""
    _chk2(xproj, x, w_ih, h_prev, c_prev, w_hh, b_ih, b_hh, gates_out, h_out, c_out)
    B, H = h_out.shape
    I = 0 if x is None else x.shape[1]
    ld = lambda t: 0 if t is None else t.stride(0)
    check(lib.dvae_lstm_step_fwd_f32(_p(xproj), ld(xproj), _p(x), ld(x), _p(w_ih), I, _p(h_prev), ld(h_prev),
                                     _p(c_prev), ld(c_prev), _p(w_hh), _p(b_ih), _p(b_hh), _p(gates_out),
                                     _p(h_out), ld(h_out), _p(c_out), ld(c_out), B, H, _stream()),
          'lstm_step_fwd')


def lstm_step_bwd_pointwise(dh, dc, gates, c_prev, c, dgates, dc_prev):
    _chk2(dh, dc, gates, c_prev, c, dgates, dc_prev)
    B, H = c.shape
    ld = lambda t: 0 if t is None else t.stride(0)
    check(lib.dvae_lstm_step_bwd_pointwise_f32(_p(dh), ld(dh), _p(dc), ld(dc), _p(gates), _p(c_prev), ld(c_prev),
                                               _p(c), ld(c), _p(dgates), _p(dc_prev), ld(dc_prev), B, H, _stream()),
          'lstm_step_bwd_pointwise')


def lstm_bwd_step(dg_t, wt, dh_prev, dc, gates, c_prev, c, dgates, dc_prev):
    """dh_prev += dg_t @ wt^T, then the pointwise backward of the step dh_prev belongs to (arguments as in
    lstm_step_bwd_pointwise, for that step)."""
    _chk2(dg_t, wt, dh_prev, dc, gates, c_prev, c, dgates, dc_prev)
    B, K = dg_t.shape
    H = wt.shape[0]
    assert wt.shape[1] == K and dh_prev.shape == (B, H) and gates.is_contiguous() and dgates.is_contiguous()
    assert dg_t.stride(1) == 1 and wt.stride(1) == 1 and dh_prev.stride(1) == 1
    ld = lambda t: 0 if t is None else t.stride(0)
    check(lib.dvae_lstm_bwd_step_f32(_p(dg_t), dg_t.stride(0), _p(wt), wt.stride(0), _p(dh_prev), dh_prev.stride(0),
                                     _p(dc), ld(dc), _p(gates), _p(c_prev), ld(c_prev), _p(c), ld(c), _p(dgates),
                                     _p(dc_prev), ld(dc_prev), B, H, K, _stream()), 'lstm_bwd_step')


def lstm_bwd_step2(dg_a, wt_a, dg_b, wt_b, dh_prev, dc, gates, c_prev, c, dgates, dc_prev):
    """dh_prev += dg_a @ wt_a^T + dg_b @ wt_b^T (the second pair may be None), then the pointwise backward of the
    step dh_prev belongs to (arguments as in lstm_step_bwd_pointwise, for that step)."""
    _chk2(dg_a, wt_a, dg_b, wt_b, dh_prev, dc, gates, c_prev, c, dgates, dc_prev)
    B, Ka = dg_a.shape
    H = wt_a.shape[0]
    assert wt_a.shape[1] == Ka and dh_prev.shape == (B, H) and gates.is_contiguous() and dgates.is_contiguous()
    assert dg_a.stride(1) == 1 and wt_a.stride(1) == 1 and dh_prev.stride(1) == 1
    Kb = 0
    if dg_b is not None:
        Kb = dg_b.shape[1]
        assert dg_b.shape[0] == B and wt_b.shape == (H, Kb) and dg_b.stride(1) == 1 and wt_b.stride(1) == 1
    ld = lambda t: 0 if t is None else t.stride(0)
    check(lib.dvae_lstm_bwd_step2_f32(_p(dg_a), dg_a.stride(0), _p(wt_a), wt_a.stride(0), Ka, _p(dg_b), ld(dg_b),
                                      _p(wt_b), ld(wt_b), Kb, _p(dh_prev), dh_prev.stride(0), _p(dc), ld(dc),
                                      _p(gates), _p(c_prev), ld(c_prev), _p(c), ld(c), _p(dgates), _p(dc_prev),
                                      ld(dc_prev), B, H, _stream()), 'lstm_bwd_step2')


def skinny_gemm_nt(a: Tensor, wt: Tensor, out: Tensor, accumulate: bool):
    """out[B,N] (+)= a[B,K] @ wt[N,K]^T; rows of a, wt and out may be strided (unit stride inside a row)."""
    _chk2(a, wt, out)
    B, K = a.shape
    N = wt.shape[0]
    assert wt.shape[1] == K and out.shape == (B, N)
    assert a.stride(1) == 1 and wt.stride(1) == 1 and out.stride(1) == 1
    check(lib.dvae_skinny_gemm_nt_f32(_p(a), a.stride(0), _p(wt), wt.stride(0), _p(out), out.stride(0), B, N, K,
                                      int(accumulate), _stream()), 'skinny_gemm_nt')


def lstm_seq_supported(nchains: int, T: int, B: int, H: int) -> bool:
    return bool(lib.dvae_lstm_seq_supported(nchains, T, B, H))


def _ptr_array(ts):
    return (ctypes.c_void_p * len(ts))(*[None if t is None else t.data_ptr() for t in ts])


def lstm_seq_fwd(chains, T: int, B: int, H: int):
    """One launch for the whole recurrence of up to two independent chains.  Each chain is a dict with
    w_rec [4H,H], gates [T,B,4H], hseq (a [T,B,>=H] view: row and step strides are taken from it), cseq
    [T,B,H], and optionally xproj [T,B,4H], b_a / b_b [4H], h_init / c_init [B,H] views, reverse."""
    n = len(chains)
    get = lambda k: [c.get(k) for c in chains]
    for c in chains:
        _chk2(c.get('xproj'), c['w_rec'], c.get('b_a'), c.get('b_b'), c.get('h_init'), c.get('c_init'), c['gates'],
              c['hseq'], c['cseq'])
        assert c['w_rec'].is_contiguous() and c['gates'].is_contiguous() and c['cseq'].is_contiguous()
        assert c['hseq'].shape[:2] == (T, B) and c['hseq'].stride(2) == 1
        assert c.get('xproj') is None or c['xproj'].is_contiguous()
    hs0 = chains[0]['hseq']
    ld_h, step_h = hs0.stride(1), hs0.stride(0)
    assert all(c['hseq'].stride(1) == ld_h and c['hseq'].stride(0) == step_h for c in chains)
    hi = [c.get('h_init') for c in chains]
    ci = [c.get('c_init') for c in chains]
    assert all((h is None) == (hi[0] is None) for h in hi) and all((x is None) == (ci[0] is None) for x in ci)
    ld_hi = 0 if hi[0] is None else hi[0].stride(0)
    ld_ci = 0 if ci[0] is None else ci[0].stride(0)
    assert all(h is None or (h.stride(0) == ld_hi and h.stride(1) == 1) for h in hi)
    assert all(x is None or (x.stride(0) == ld_ci and x.stride(1) == 1) for x in ci)
    ws = workspace(lib.dvae_lstm_seq_ws_floats(n, B, H), chains[0]['w_rec'].device, 'lstm_seq')
    rev = (ctypes.c_int * n)(*[int(bool(c.get('reverse', False))) for c in chains])
    opt = lambda k: None if all(v is None for v in get(k)) else _ptr_array(get(k))
    check(lib.dvae_lstm_seq_fwd_f32(n, opt('xproj'), _ptr_array(get('w_rec')), opt('b_a'), opt('b_b'), opt('h_init'),
                                    ld_hi, opt('c_init'), ld_ci, _ptr_array(get('gates')), _ptr_array(get('hseq')),
                                    ld_h, step_h, _ptr_array(get('cseq')), rev, T, B, H, _p(ws), _stream()),
          'lstm_seq_fwd')
    return ws


def lstm_seq_bwd(chains, T: int, B: int, H: int, accumulate_dh_init: bool = False):
    """Backpropagation through time of up to two independent chains in one launch.  Each chain is a dict with
    w_rec [4H,H], gates [T,B,4H], cseq [T,B,H], dh_ext (a [T,B,>=H] view), dgates [T,B,4H] (out), and optionally
    c_init [B,H] view, dh_init / dc_init [B,H] views (out), reverse."""
    n = len(chains)
    get = lambda k: [c.get(k) for c in chains]
    for c in chains:
        _chk2(c['w_rec'], c['gates'], c['cseq'], c.get('c_init'), c['dh_ext'], c['dgates'], c.get('dh_init'),
              c.get('dc_init'))
        assert c['w_rec'].is_contiguous() and c['gates'].is_contiguous() and c['cseq'].is_contiguous()
        assert c['dgates'].is_contiguous() and c['dgates'].shape == (T, B, 4 * H)
        assert c['dh_ext'].shape[:2] == (T, B) and c['dh_ext'].stride(2) == 1
    d0 = chains[0]['dh_ext']
    ld_dh, step_dh = d0.stride(1), d0.stride(0)
    assert all(c['dh_ext'].stride(1) == ld_dh and c['dh_ext'].stride(0) == step_dh for c in chains)

    def common_ld(k):
        ts = get(k)
        assert all((t is None) == (ts[0] is None) for t in ts)
        if ts[0] is None:
            return 0
        assert all(t.stride(1) == 1 and t.stride(0) == ts[0].stride(0) for t in ts)
        return ts[0].stride(0)

    ld_ci, ld_dhi, ld_dci = common_ld('c_init'), common_ld('dh_init'), common_ld('dc_init')
    ws = workspace(lib.dvae_lstm_seq_bwd_ws_floats(n, B, H), chains[0]['w_rec'].device, 'lstm_seq')
    rev = (ctypes.c_int * n)(*[int(bool(c.get('reverse', False))) for c in chains])
    opt = lambda k: None if all(v is None for v in get(k)) else _ptr_array(get(k))
    check(lib.dvae_lstm_seq_bwd_f32(n, _ptr_array(get('w_rec')), _ptr_array(get('gates')), _ptr_array(get('cseq')),
                                    opt('c_init'), ld_ci, _ptr_array(get('dh_ext')), ld_dh, step_dh,
                                    _ptr_array(get('dgates')), opt('dh_init'), ld_dhi, opt('dc_init'), ld_dci,
                                    int(accumulate_dh_init), rev, T, B, H, _p(ws), _stream()), 'lstm_seq_bwd')
    return ws


def lstm_seq_timeouts() -> int:
    """Workgroups of the one-launch recurrence kernels that gave up a bounded wait on the current GPU since the
    library was loaded (0 in a healthy run; the affected launches' outputs are invalid).  Synchronises."""
    n = lib.dvae_lstm_seq_timeouts()
    if n < 0:
        check(int(n), 'lstm_seq_timeouts')
    return int(n)


def lstm_resident_workgroups(kind: int, H: int, max_per_cu: int = 0) -> int:
    """Workgroups of the one-launch recurrence kernels (kind 0: one layer; 1: the stacked decoder) that fit the
    current device together, at most max_per_cu per compute unit when that is > 0."""
    return int(lib.dvae_lstm_resident_workgroups(kind, H, max_per_cu))


def check_lstm_seq_timeouts():
    """Raises if a workgroup of a one-launch recurrence kernel gave up a wait since the last check (the counter
    is cleared, so the next check reports new failures only).  Synchronises."""
    n = lstm_seq_timeouts()
    if n:
        check(lib.dvae_lstm_seq_timeouts_clear(), 'lstm_seq_timeouts_clear')
        raise RuntimeError(f"{n} workgroup(s) of a one-launch LSTM kernel ran into the bound of a wait: the launch was "
                           "not fully resident (another kernel held compute units for more than 200 ms); results of "
                           "that step are invalid (the kernels filled them with NaN).  DVAE_SEQ_LSTM=0 selects one "
                           "launch per step.")


def lstm_dec2_supported(T: int, B: int, H: int) -> bool:
    return bool(lib.dvae_lstm_dec2_supported(T, B, H))


def lstm_dec2_fwd(weights, h0: Tensor, c0: Tensor, init: Tensor, gates: Tensor, hs: Tensor, cs: Tensor):
    """The two-layer autoregressive decoder loop in one launch.  weights: w_ih, w_hh, b_ih, b_hh of layer 0, then of
    layer 1; h0 / c0 [B, 2H]; init [H]; outputs gates [2,T,B,4H], hs / cs [2,T,B,H]."""
    _chk(*weights, h0, c0, init, gates, hs, cs)
    L, T, B, H = hs.shape
    assert L == 2 and len(weights) == 8 and h0.shape == (B, 2 * H) and c0.shape == (B, 2 * H) and init.numel() == H
    assert gates.shape == (2, T, B, 4 * H) and cs.shape == hs.shape
    assert all(weights[4 * l + k].shape == (4 * H, H) for l in range(2) for k in range(2))
    ws = workspace(lib.dvae_lstm_dec2_ws_floats(B, H), h0.device, 'lstm_seq')
    arr = lambda k: _ptr_array([weights[k], weights[4 + k]])
    check(lib.dvae_lstm_dec2_fwd_f32(arr(0), arr(1), arr(2), arr(3), _p(h0), h0.stride(0), _p(c0), c0.stride(0),
                                     _p(init), _p(gates), _p(hs), _p(cs), T, B, H, _p(ws), _stream()), 'lstm_dec2_fwd')
    return ws


def lstm_dec2_bwd(weights, gates: Tensor, cs: Tensor, c0: Tensor, dstates: Tensor, dgates: Tensor, dh0: Tensor,
                  dc0: Tensor, dx0: Tensor):
    """Backward of lstm_dec2_fwd in one launch.  dstates [B,T,H] (the gradient of the top layer's outputs, batch-major
    as the decoder returns them); outputs dgates [2,T,B,4H], dh0 / dc0 [B,2H], dx0 [B,H]."""
    _chk(*[w for w in weights if w is not None], gates, cs, c0, dstates, dgates, dh0, dc0, dx0)
    L, T, B, H = cs.shape
    assert L == 2 and dstates.shape == (B, T, H) and dgates.shape == gates.shape == (2, T, B, 4 * H)
    assert c0.shape == dh0.shape == dc0.shape == (B, 2 * H) and dx0.shape == (B, H)
    ws = workspace(lib.dvae_lstm_dec2_ws_floats(B, H), c0.device, 'lstm_seq')
    arr = lambda k: _ptr_array([weights[k], weights[4 + k]])
    check(lib.dvae_lstm_dec2_bwd_f32(arr(0), arr(1), _p(gates), _p(cs), _p(c0), c0.stride(0), _p(dstates), T * H, H,
                                     _p(dgates), _p(dh0), dh0.stride(0), _p(dc0), dc0.stride(0), _p(dx0), T, B, H,
                                     _p(ws), _stream()), 'lstm_dec2_bwd')
    return ws


# ---------------------------------------------------------------------------- optimiser
class TensorTable:
    """Host-side pointer/size arrays for the multi-tensor kernels, rebuilt only when pointers move."""

    def __init__(self):
        self.key = None
        self.arrays = None

    def get(self, lists: Sequence[Sequence[Tensor]]):
        key = tuple(t.data_ptr() for ts in lists for t in ts)
        if key != self.key:
            n = len(lists[0])
            arrs = []
            for ts in lists:
                for t in ts:
                    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
                        raise RuntimeError("multi-tensor ops need contiguous fp32 GPU tensors")
                arrs.append((ctypes.c_void_p * n)(*[t.data_ptr() for t in ts]))
            numel = (ctypes.c_int64 * n)(*[t.numel() for t in lists[0]])
            self.key = key
            self.arrays = (n, arrs, numel)
        return self.arrays


def grad_norm(grads: List[Tensor], max_norm: float, out: Tensor, table: Optional[TensorTable] = None):
    """out[0] = total 2-norm, out[1] = clip coefficient."""
    table = table or TensorTable()
    n, (g,), numel = table.get([grads])
    ws = workspace(lib.dvae_grad_norm_ws_floats(n, numel), grads[0].device, 'norm')
    check(lib.dvae_grad_norm_f32(n, g, numel, float(max_norm), _p(out), _p(ws), _stream()), 'grad_norm')


def scale_tensors(tensors: List[Tensor], coef: Tensor, table: Optional[TensorTable] = None):
    table = table or TensorTable()
    n, (g,), numel = table.get([tensors])
    check(lib.dvae_scale_tensors_f32(n, g, numel, _p(coef), _stream()), 'scale_tensors')


def copy_tensors(dsts: List[Tensor], srcs: List[Tensor], table: Optional[TensorTable] = None):
    """dsts[i] <- srcs[i] for a list of equally sized fp32 tensors, one launch per 48 of them."""
    table = table or TensorTable()
    for d, s_ in zip(dsts, srcs):
        if d.numel() != s_.numel():
            raise RuntimeError("copy_tensors: sizes differ")
    n, (d, s_), numel = table.get([dsts, srcs])
    check(lib.dvae_copy_tensors_f32(n, d, s_, numel, _stream()), 'copy_tensors')


def adam_step(params, grads, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, bc1, bc2,
              grad_coef: Optional[Tensor], table: Optional[TensorTable] = None, dyn: Optional[Tensor] = None):
    """dyn: device tensor {step count, learning rate, 1-beta1^t, sqrt(1-beta2^t)} maintained by
    adam_advance; when given it overrides lr/bc1/bc2."""
    table = table or TensorTable()
    n, (p, g, m, v), numel = table.get([params, grads, exp_avg, exp_avg_sq])
    check(lib.dvae_adam_step_f32(n, p, g, m, v, numel, float(lr), float(beta1), float(beta2), float(eps),
                                 float(weight_decay), float(bc1), float(bc2), _p(grad_coef), _p(dyn), _stream()),
          'adam_step')


def adam_advance(dyn: Tensor, beta1: float, beta2: float):
    """dyn[0] += 1 and the two bias corrections for the new step count, on the stream."""
    if not (dyn.is_cuda and dyn.dtype == torch.float32 and dyn.numel() >= 4 and dyn.is_contiguous()):
        raise RuntimeError("adam_advance: dyn must be four contiguous fp32 values on the GPU")
    check(lib.dvae_adam_advance_f32(_p(dyn), float(beta1), float(beta2), _stream()), 'adam_advance')


# ---------------------------------------------------------------------------- launch trace (benchmarks)
def trace_begin(max_launches: int = 1 << 16):
    """From here to trace_collect() every kernel launch of the library is timed with two events on its
    stream and recorded with its symbol and declared algorithmic work.  Not under graph capture."""
    check(lib.dvae_trace_begin(int(max_launches)), 'trace_begin')


def trace_collect():
    """Ends the trace; list of (kernel symbol, milliseconds, work, unit) with unit 1 = FLOP, 2 = byte,
    0 = none declared.  Waits for the traced launches to finish."""
    n = lib.dvae_trace_end()
    name = ctypes.create_string_buffer(512)
    ms, work, unit = ctypes.c_float(), ctypes.c_double(), ctypes.c_int()
    out = []
    for i in range(n):
        check(lib.dvae_trace_read(i, name, 512, ctypes.byref(ms), ctypes.byref(work), ctypes.byref(unit)), 'trace_read')
        out.append((name.value.decode(), float(ms.value), float(work.value), int(unit.value)))
    return out
