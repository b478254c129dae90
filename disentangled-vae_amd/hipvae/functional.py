"""Autograd building blocks of the MI355X hot path.

Each Function covers one fusion region of the network and runs only HIP kernels from
libdvae_hip.so in both directions:

  encoder_cnn  [Conv(k4,s2,p1) -> BN -> LeakyReLU]* (EncoderConvs) -> flatten -> Linear -> LeakyReLU (EncoderFC)
  DecoderCNN   Linear -> LeakyReLU -> [ConvT -> BN -> LeakyReLU]* -> ConvT(bias) -> sigmoid
  Linear       x W^T + b (+ LeakyReLU)
  LSTMEncoder  (bi)directional multi-layer LSTM over a sequence, returns the final hidden states
  LSTMDecoder  autoregressive multi-layer LSTM (x_t = top-layer h_{t-1})
  Reparam      z = mu + eps * exp(logvar / 2)

Activations are NHWC with frames folded into the leading dimension; a layer's BatchNorm +
LeakyReLU is never materialised between conv layers: the conv kernel writes the raw output plus
per-channel partial sums, a finalize kernel folds the statistics into (scale, shift), and the next
kernel applies them while loading.
"""
import os
from typing import List, Optional, Tuple

import torch

from . import ops

Tensor = torch.Tensor
BN_EPS = 1e-5
BN_MOMENTUM = 0.1
# one launch per LSTM recurrence (weights resident, in-launch hand-off of the hidden state) where the
# shape admits it; False selects one launch per time step everywhere
USE_SEQ_KERNELS = os.environ.get('DVAE_SEQ_LSTM', '1') != '0'


def _need_gpu(t: Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(f"{what}: input is on {t.device}; this package runs on the GPU only "
                           "(there is no CPU fallback)")


class BNState:
    """Non-tensor handle for a BatchNorm layer's buffers (updated in place by the kernels)."""

    def __init__(self, running_mean: Tensor, running_var: Tensor, num_batches_tracked: Tensor):
        self.rm, self.rv, self.nbt = running_mean, running_var, num_batches_tracked


def _bn_affine(stats, nparts, M, gamma, beta, bn: BNState, training: bool, replicas: int = 1):
    if training:
        return ops.bn_finalize(stats, nparts, M, gamma, beta, bn.rm, bn.rv, bn.nbt, BN_MOMENTUM, BN_EPS, replicas)
    sc, sh = ops.bn_eval_affine(gamma, beta, bn.rm, bn.rv, BN_EPS)
    return sc, sh, None, None


# DVAE_OFF_PATH=1: weight gradients on a second stream beside the chain of data gradients.  Off by default: as
# parallel branches of the captured step they made it slower on MI355X (batch 32: 3.95 against 3.55 ms, batch
# 256: 13.8 against 13.4 ms -- every fork/join is a cross-queue dependency that costs more than the overlap buys).
USE_OFF_PATH = os.environ.get('DVAE_OFF_PATH', '0') == '1'
_OFF_STREAMS = {}


def _seq_ok(chains: int, T: int, B: int, H: int) -> bool:
    """Whether a recurrence runs as one launch.  Every workgroup of such a launch must become resident: with
    several ranks the collectives' kernels share the compute units with the backward pass, so only grids of at
    most one workgroup per compute unit are used there (single process: up to two)."""
    if not (USE_SEQ_KERNELS and ops.lstm_seq_supported(chains, T, B, H)):
        return False
    if _several_ranks():
        return chains * ((B + 31) // 32) * (H // 8) <= ops.lstm_resident_workgroups(0, H, 1)
    return True


def _seq_groups(D: int, T: int, B: int, H: int):
    """How the D directions of a recurrence share out over one-launch kernels: [[0, 1]] (one launch), [[0], [1]]
    (one launch per direction, when both together would exceed what may be resident), or None (per-step kernels)."""
    if _seq_ok(D, T, B, H):
        return [list(range(D))]
    if D == 2 and _seq_ok(1, T, B, H):
        return [[0], [1]]
    return None


def _several_ranks() -> bool:
    """(DVAE_ASSUME_SEVERAL_RANKS=1: the kernel choices of a multi-rank job in a single process, for measuring them
    on one GPU)"""
    if os.environ.get('DVAE_ASSUME_SEVERAL_RANKS') == '1':
        return True
    return (torch.distributed.is_available() and torch.distributed.is_initialized()
            and torch.distributed.get_world_size() > 1)


def _dec2_ok(T: int, B: int, H: int) -> bool:
    """The same question for the stacked decoder's one-launch kernels.  They hold a whole compute unit each, so with
    several ranks (collectives' kernels on the same chip) only grids of at most one workgroup per two compute
    units are used."""
    if not (USE_SEQ_KERNELS and ops.lstm_dec2_supported(T, B, H)):
        return False
    if _several_ranks():
        return ((B + 31) // 32) * (H // 8) <= ops.lstm_resident_workgroups(1, H, 1) // 2
    return True


class _OffPath:
    """Inside one backward node: work whose result only the optimiser needs (weight gradients, their slab
    reductions and re-orderings) is issued on a second stream, so that it runs beside the chain of data
    gradients instead of in it (under graph capture: a parallel branch).  The node joins before it returns, so
    callers see finished gradients on their own stream.  Such work takes its scratch from its own slot, and
    every tensor it reads is kept referenced until the join (the allocator knows nothing of the second stream).
    """

    def __init__(self, like: Tensor, enabled: bool = True):
        self.main = torch.cuda.current_stream(like.device)
        self.side = None
        if enabled and USE_OFF_PATH:
            key = like.device.index if like.device.index is not None else torch.cuda.current_device()
            if key not in _OFF_STREAMS:
                _OFF_STREAMS[key] = torch.cuda.Stream(device=like.device)
            self.side = _OFF_STREAMS[key]
        self.keep = []
        self.forked = False

    def run(self, fn, *reads):
        """fn() on the second stream, ordered behind everything issued on the main stream so far"""
        if self.side is None:
            return fn()
        self.side.wait_stream(self.main)
        self.forked = True
        self.keep.extend(reads)
        with torch.cuda.stream(self.side), ops.workspace_slot('off_path'):
            out = fn()
        self.keep.append(out)
        return out

    def join(self):
        if self.forked:
            self.main.wait_stream(self.side)
        self.keep = []
        self.forked = False


# The content encoder of the disentangled model (one frame per clip: a chain of small,
# latency-bound kernels) runs on a second stream beside the motion encoder, forward and -- because autograd
# replays a node's backward on the stream its forward ran on -- backward.  One fork and one join each way
# (batch 32: 3.62 -> 3.39 ms per step, batch 256: 13.36 -> 13.11 ms).  DVAE_SIDE_ENCODER=0 keeps one stream.
USE_SIDE_ENCODER = os.environ.get('DVAE_SIDE_ENCODER', '1') != '0'
_SIDE_STREAMS = {}


class side_branch:
    """with side_branch(t) as br: ...ops on the second stream...; br.hand_over(outputs); later br.join()."""

    def __init__(self, like: Tensor):
        self.main = torch.cuda.current_stream(like.device)
        key = like.device.index if like.device.index is not None else torch.cuda.current_device()
        if key not in _SIDE_STREAMS:
            _SIDE_STREAMS[key] = torch.cuda.Stream(device=like.device)
            ops.register_stream_slot(_SIDE_STREAMS[key], 'side_branch')
        self.side = _SIDE_STREAMS[key]
        self.ctx = torch.cuda.stream(self.side)

    def __enter__(self):
        self.side.wait_stream(self.main)
        self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        return self.ctx.__exit__(*exc)

    def take(self, *tensors: Tensor):
        """inputs from the main stream that the second stream (and its backward) will read"""
        for t in tensors:
            t.record_stream(self.side)

    def hand_over(self, *tensors: Tensor):
        """results that the main stream will read: the allocator must not recycle them on the second stream's
        schedule alone"""
        for t in tensors:
            t.record_stream(self.main)

    def join(self):
        self.main.wait_stream(self.side)


# =================================================================================================
def _fuse_bn_stats(channels: int) -> bool:
    """Whether the data-gradient kernel that produces a gradient with `channels` channels should also
    reduce the BatchNorm-backward statistics of the layer below.  It pays for the wide-and-shallow
    tensors (the reduction pass it replaces reads two of them from HBM); on the 128-wide tiles the
    longer epilogue costs more than that pass."""
    return channels <= 64


class EncoderConvs(torch.autograd.Function):
    """The convolution stack of CNNEncoder / LightCNNEncoder of the reference (models/encoders.py:28-39, 72-83):
    frames [N, H, W, Cin] -> the last layer's BatchNorm + LeakyReLU output [N, h, w, C] (NHWC)."""

    @staticmethod
    def forward(ctx, x: Tensor, training: bool, bns: List[BNState], *wgb: Tensor):
        # wgb = conv weights (L), then BN gammas (L), then BN betas (L)
        _need_gpu(x, 'EncoderConvs')
        L = len(bns)
        conv_w, gammas, betas = wgb[:L], wgb[L:2 * L], wgb[2 * L:3 * L]
        ys, affs = [], []
        cur, sc, sh, act = x.contiguous(), None, None, ops.ACT_NONE
        # every layer's weight in the layout its kernel reads, and (when a backward follows) in the transposed
        # layout its data gradient reads, all packed by one launch
        need_bwd = training and any(ctx.needs_input_grad)
        packs = ops.pack_many([(w.detach(), False) for w in conv_w]
                              + ([(w.detach(), True) for w in conv_w[1:]] if need_bwd else []))
        ctx.wpd = [None] + packs[L:]
        # bf16 mode, the few-channel encoder on full frames (1 -> 16 -> 32 channels from 64 x 64: the motion encoder):
        # the first block's raw output, the largest tensor of this stack, and the gradient of its activation are
        # kept as bfloat16; every kernel that touches them takes that
        st16 = (need_bwd and ops.bf16_storage() and L == 2 and x.shape[1:] == (64, 64, 1)
                and tuple(conv_w[0].shape[:2]) == (16, 1) and tuple(conv_w[1].shape[:2]) == (32, 16)
                and bool(ops.lib.dvae_deconv4s2_bnin_supported(16, 16, 32, 16)))      # (backward's route)
        for l in range(L):
            y, stats, nparts = ops.conv4s2_fwd(cur, sc, sh, act, packs[l], training, out_bf16=st16 and l == 0)
            M = y.numel() // y.shape[-1]
            aff = _bn_affine(stats, nparts, M, gammas[l].detach(), betas[l].detach(), bns[l], training)
            ys.append(y)
            affs.append(aff)
            cur, sc, sh, act = y, aff[0], aff[1], ops.ACT_LRELU
        u = ops.affine_act(cur, sc, sh, ops.ACT_LRELU)
        if training:
            ctx.save_for_backward(x, *ys, *conv_w)
            ctx.affs = affs
            ctx.gammas = [g.detach() for g in gammas]
            ctx.L = L
        else:
            ctx.L = -1
        return u

    @staticmethod
    def backward(ctx, du: Tensor):
        if ctx.L < 0:
            raise RuntimeError("EncoderConvs: backward is only implemented for training mode (batch statistics)")
        L = ctx.L
        saved = ctx.saved_tensors
        x = saved[0]
        ys, conv_w = saved[1:1 + L], saved[1 + L:1 + 2 * L]
        affs = ctx.affs
        du = du.contiguous()                                     # grad wrt post-activation of last conv
        off = _OffPath(du)
        d_conv, d_gamma, d_beta = [None] * L, [None] * L, [None] * L
        parts = None                      # BatchNorm-backward statistics reduced by the producer of du
        for l in range(L - 1, -1, -1):
            sc, sh, mean, invstd = affs[l]
            M = du.numel() // du.shape[-1]
            # Where the consumer of dL/dy can form it from (du, y) as it loads them, the elementwise
            # BatchNorm-backward pass is skipped: only the batch sums and two coefficient vectors are made.
            if l == 0 and x.shape[-1] == 1 and du.shape[-1] <= 128:
                if parts is None:
                    d_gamma[0], d_beta[0], ca, cb = ops.bn_bwd_reduce_coeffs(du, ys[0], affs[0])
                else:
                    d_gamma[0], d_beta[0], ca, cb = ops.bn_bwd_coeffs(parts, affs[0], M)
                d_conv[0] = off.run(lambda du=du, ca=ca, cb=cb: ops.conv4s2_wgrad_bnp(du, ys[0], affs[0], ca, cb, x),
                                    du, ca, cb)
                continue
            if (l > 0 and _fuse_bn_stats(ys[l - 1].shape[-1])
                    and ops.deconv4s2_bnin_supported(du, conv_w[l].shape[1])):
                if parts is None:
                    d_gamma[l], d_beta[l], ca, cb = ops.bn_bwd_reduce_coeffs(du, ys[l], affs[l])
                else:
                    d_gamma[l], d_beta[l], ca, cb = ops.bn_bwd_coeffs(parts, affs[l], M)
                wpd = ctx.wpd[l]
                du, parts, dy = ops.deconv4s2_dgrad_bnstat_bnin(du, ys[l], affs[l], ca, cb, wpd, ys[l - 1], affs[l - 1])
                psc, psh = affs[l - 1][0], affs[l - 1][1]
                d_conv[l] = off.run(lambda dy=dy, l=l, psc=psc, psh=psh: ops.conv4s2_wgrad(
                    dy, None, None, ops.ACT_NONE, ys[l - 1], psc, psh, ops.ACT_LRELU), dy)
                continue
            if l == L - 1:
                du = du.clone()                                  # the incoming gradient is not ours to overwrite
            d_gamma[l], d_beta[l] = ops.bn_bwd(du, ys[l], sc, sh, mean, invstd, ctx.gammas[l], parts)
            dy = du                                              # now dL/d(raw conv output)
            if l > 0:
                psc, psh = affs[l - 1][0], affs[l - 1][1]
                d_conv[l] = off.run(lambda dy=dy, l=l, psc=psc, psh=psh: ops.conv4s2_wgrad(
                    dy, None, None, ops.ACT_NONE, ys[l - 1], psc, psh, ops.ACT_LRELU), dy)
                wpd = ctx.wpd[l]
                if _fuse_bn_stats(ys[l - 1].shape[-1]):
                    du, parts = ops.deconv4s2_dgrad_bnstat(dy, wpd, ys[l - 1], affs[l - 1])
                else:
                    (du, _, _), parts = ops.deconv4s2_fwd(dy, None, None, ops.ACT_NONE, wpd, False), None
            else:
                d_conv[0] = ops.conv4s2_wgrad(dy, None, None, ops.ACT_NONE, x, None, None, ops.ACT_NONE)
        off.join()
        return (None, None, None, *d_conv, *d_gamma, *d_beta)


class EncoderFC(torch.autograd.Function):
    """The Linear + LeakyReLU that follows the convolution stack (models/encoders.py:43-46, 87-90) on its NHWC
    output: the reference flattens NCHW (c,h,w) columns, so the weight's columns are permuted to (h,w,c) on the
    way in and its gradient back on the way out."""

    @staticmethod
    def forward(ctx, u: Tensor, fc_w: Tensor, fc_b: Tensor):
        _need_gpu(u, 'EncoderFC')
        N, h, w, C = u.shape
        S = h * w
        F_out = fc_w.shape[0]
        if fc_w.shape[1] != C * S:
            raise RuntimeError(f"encoder fc expects {fc_w.shape[1]} inputs but the conv stack produced {C * S}")
        w_perm = ops.permute3(fc_w.detach(), (F_out, C, S), (0, 2, 1)).view(F_out, S * C)
        u_flat = u.contiguous().view(N, S * C)
        feat = ops.gemm_nt(u_flat, w_perm, fc_b.detach(), ops.ACT_LRELU)
        ctx.save_for_backward(u_flat, w_perm, feat)
        ctx.fc_shape = (F_out, C, S)
        ctx.u_shape = tuple(u.shape)
        return feat

    @staticmethod
    def backward(ctx, dfeat: Tensor):
        u_flat, w_perm, feat = ctx.saved_tensors
        F_out, C, S = ctx.fc_shape
        off = _OffPath(dfeat)
        dfeat = dfeat.contiguous()
        dpre = torch.empty_like(dfeat)                           # the incoming gradient is not ours to overwrite
        d_fc_b = ops.bias_act_bwd(dfeat, feat, ops.ACT_LRELU, True, dst=dpre)
        d_fc_w = off.run(lambda: ops.permute3(ops.gemm_tn(dpre, u_flat), (F_out, S, C), (0, 2, 1)).view(F_out, C * S),
                         dpre)                                   # via [F, S*C]
        du = ops.gemm_nn(dpre, w_perm).view(ctx.u_shape) if ctx.needs_input_grad[0] else None
        off.join()
        return du, d_fc_w, d_fc_b


def encoder_cnn(x: Tensor, training: bool, bns: List[BNState], fc_w: Tensor, fc_b: Tensor, *wgb: Tensor,
                stage_cut: bool = True) -> Tensor:
    """CNNEncoder / LightCNNEncoder of the reference (models/encoders.py:11-97) as two autograd nodes, with (stage_cut)
    the second cut of a staged data-parallel step between them: the Linear layer's weight gradient (the largest
    gradient tensor of the encoders) is then complete, and can travel, before the convolution stack's backward
    runs."""
    from .parallel import stage_boundary
    u = EncoderConvs.apply(x, training, bns, *wgb)
    if stage_cut:
        (u,) = stage_boundary(u, level=2)
    return EncoderFC.apply(u, fc_w, fc_b)


# =================================================================================================
class SigmoidTail:
    """Hand-off between the decoder's last layer and a loss node that consumes its output directly.

    The reconstruction leaves DecoderCNN through a sigmoid, and the loss's gradient pass reads the reconstruction
    anyway, so it can write d/d(logit) = d/d(recon) * recon * (1 - recon) at once and save the separate pass over
    three clip-sized tensors.  The decoder attaches one of these records to the tensor it returns (`as_clips`
    carries it onto the [B, T, ...] view); a loss node that finds it writes the logit-form gradient, notes the
    tensor here and returns it as its gradient; DecoderCNN.backward recognises its own record's tensor by address.
    Any other gradient that autograd adds on the way (the reconstruction used twice) arrives as a different
    tensor and is corrected for, so the result is the same whichever path runs.

    What travels through autograd as "the gradient of recon" on this path is the logit form, so the path is only
    taken when nobody can see or change that tensor (`tail_usable`): the reconstruction has no tensor hooks and does
    not retain its gradient, and the decoder's node is part of the backward pass that is running (it is not when
    the caller asks for `torch.autograd.grad(loss, recon)`).  A record is stamped with the backward pass that wrote
    it; DecoderCNN.backward drops one left by a pass that never reached it (an exception on the way).  Not covered:
    `torch.autograd.grad(loss, [recon, *parameters])`, which runs the decoder's node AND captures recon's gradient
    -- DVAE_FUSE_TAIL=0 for such callers."""
    __slots__ = ('dlogit', 'dbias', 'node', 'task')

    def __init__(self):
        self.dlogit: Optional[Tensor] = None
        self.dbias: Optional[Tensor] = None
        self.node = None          # DecoderCNN's autograd node (bind_tail)
        self.task = -1            # the backward pass (autograd graph task) that wrote dlogit


FUSE_TAIL = os.environ.get('DVAE_FUSE_TAIL', '1') != '0'


def _backward_pass_id() -> int:
    f = getattr(torch._C, '_current_graph_task_id', None)
    return int(f()) if f is not None else -1


def _observed(t: Tensor) -> bool:
    """whether somebody will see (retain_grad) or may rewrite (a tensor hook) the gradient that reaches t"""
    return bool(t.retains_grad) or bool(getattr(t, '_backward_hooks', None))


def bind_tail(frames: Tensor) -> Tensor:
    """notes DecoderCNN's autograd node in the hand-off record of the tensor it returned"""
    tail = getattr(frames, '_dvae_tail', None)
    if tail is not None:
        tail.node = frames.grad_fn
    return frames


def sigmoid_tail(recon: Tensor) -> Optional[SigmoidTail]:
    """The hand-off record of a reconstruction that came straight from DecoderCNN (None otherwise, or when its
    gradient is observed).  Called by the loss functions' forward: whatever an earlier backward pass left in the
    record is dropped here."""
    tail = getattr(recon, '_dvae_tail', None) if FUSE_TAIL else None
    if tail is None:
        return None
    tail.dlogit = tail.dbias = None
    return None if _observed(recon) else tail


def tail_usable(tail: Optional[SigmoidTail], recon: Tensor) -> bool:
    """At the loss node's backward: may the logit-form gradient be handed over in this backward pass?"""
    if tail is None or tail.dlogit is not None or _observed(recon):
        return False
    will_run = getattr(torch._C, '_will_engine_execute_node', None)
    if tail.node is not None and will_run is not None:
        try:
            if not will_run(tail.node):
                return False
        except RuntimeError:          # not inside a backward pass that knows its nodes
            return False
    return True


def as_clips(frames: Tensor, B: int, T: int) -> Tensor:
    """[B*T, C, H, W] frames of DecoderCNN as [B, T, C, H, W] clips, keeping the hand-off record."""
    clips = frames.view(B, T, *frames.shape[1:])
    tail = getattr(frames, '_dvae_tail', None)
    if tail is not None:
        clips._dvae_tail = tail
    return clips


class DecoderCNN(torch.autograd.Function):
    """CNNDecoder of the reference (models/decoders.py:11-70); returns frames [N, 1, 2^k*4, ...].

    `replicas` > 1: the caller's batch is that many identical copies of each row of `feat` (BaselineVAE repeats one
    latent per clip T times, models/baseline_vae.py:114-115) and hands over one copy.  The frames of the copies
    are equal, the batch statistics of the repeated batch are those of one copy, and what backward needs of the
    copies' gradients is their sum (which is what the caller's expand hands back), so the whole decoder runs on
    N rows instead of N*replicas; only the running variance's unbiased correction uses the full count."""

    @staticmethod
    def forward(ctx, feat: Tensor, training: bool, bns: List[BNState], replicas: int, fc_w: Tensor, fc_b: Tensor,
                final_w: Tensor, final_b: Tensor, *wgb: Tensor):
        _need_gpu(feat, 'DecoderCNN')
        L = len(bns)
        dec_w, gammas, betas = wgb[:L], wgb[L:2 * L], wgb[2 * L:3 * L]
        N, Kin = feat.shape
        C0 = dec_w[0].shape[0] if L > 0 else final_w.shape[0]
        S = 16
        if fc_w.shape[0] != C0 * S:
            raise RuntimeError("decoder fc output does not match channels[0]*4*4")
        if final_w.shape[1] != 1:
            raise RuntimeError("the final transposed convolution kernel supports one output channel")
        # rows of the fc weight are (c,h,w) in the reference; ours are (h,w,c)
        w_perm = ops.permute3(fc_w.detach(), (C0, S, Kin), (1, 0, 2)).view(S * C0, Kin)
        b_perm = ops.permute3(fc_b.detach(), (C0, S, 1), (1, 0, 2)).view(S * C0)
        feat_c = feat.contiguous()
        h0 = ops.gemm_nt(feat_c, w_perm, b_perm, ops.ACT_LRELU)   # [N, 4*4*C0] == NHWC [N,4,4,C0]
        cur = h0.view(N, 4, 4, C0)
        sc, sh, act = None, None, ops.ACT_NONE
        ys, affs = [], []
        need_bwd = training and any(ctx.needs_input_grad)
        packs = ops.pack_many([(w.detach(), True) for w in dec_w]
                              + ([(w.detach(), False) for w in dec_w] if need_bwd else []))   # as in EncoderCNN
        ctx.wp = packs[L:]
        # bf16 mode: the last block's raw output (64 -> 32 channels, the largest tensor of the step, read four times
        # more) and the two gradients shaped like it are kept in HBM as bfloat16 where every kernel around them
        # takes that: the 64 -> 32 row kernel with 16-wide input, then the fixed 32-channel final layer
        st16 = (need_bwd and ops.bf16_storage() and L >= 2 and tuple(dec_w[-1].shape[:2]) == (64, 32)
                and 4 * 2 ** (L - 1) == 16 and replicas == 1
                and bool(ops.lib.dvae_conv4s2_bnin_supported(32, 32, 32, 64)))   # (backward's route for this block)
        ctx.st16 = st16
        # ... and, where the block before it is the 128 -> 64 one on 8 x 8 frames (whose backward takes the route of
        # the weight-gradient kernel that folds the BatchNorm backward in), that block's raw output and du as well
        st16b = (st16 and L >= 3 and tuple(dec_w[-2].shape[:2]) == (128, 64)
                 and bool(ops.lib.dvae_conv4s2_wgrad_bnq_supported(N, 8, 8, 128, 64)))
        for l in range(L):
            y, stats, nparts = ops.deconv4s2_fwd(cur, sc, sh, act, packs[l], training,
                                                 out_bf16=(st16 and l == L - 1) or (st16b and l == L - 2))
            M = y.numel() // y.shape[-1]
            aff = _bn_affine(stats, nparts, M, gammas[l].detach(), betas[l].detach(), bns[l], training, replicas)
            ys.append(y)
            affs.append(aff)
            cur, sc, sh, act = y, aff[0], aff[1], ops.ACT_LRELU
        recon = ops.deconv4s2_final_fwd(cur, sc, sh, act, final_w.detach(), final_b.detach(), ops.ACT_SIGMOID)
        recon = recon.view(N, 1, recon.shape[1], recon.shape[2])
        if training:
            ctx.save_for_backward(feat_c, w_perm, h0, recon, final_w, *ys, *dec_w)
            ctx.affs = affs
            ctx.gammas = [g.detach() for g in gammas]
            ctx.L = L
            ctx.dims = (C0, S, Kin)
            ctx.tail = recon._dvae_tail = SigmoidTail()
        else:
            ctx.L = -1
        return recon

    @staticmethod
    def backward(ctx, drecon: Tensor):
        if ctx.L < 0:
            raise RuntimeError("DecoderCNN: backward is only implemented for training mode (batch statistics)")
        L = ctx.L
        saved = ctx.saved_tensors
        feat, w_perm, h0, recon, final_w = saved[:5]
        ys, dec_w = saved[5:5 + L], saved[5 + L:5 + 2 * L]
        affs = ctx.affs
        C0, S, Kin = ctx.dims
        N = feat.shape[0]
        off = _OffPath(drecon)
        tail, fused = ctx.tail, None
        if tail.dlogit is not None:
            fused, fused_b = tail.dlogit, tail.dbias
            tail.dlogit = tail.dbias = None
            if tail.task != _backward_pass_id():
                fused = None                       # left by a backward pass that never got here
        if fused is not None and drecon.data_ptr() == fused.data_ptr() and drecon.numel() == fused.numel():
            dlogit, d_final_b = fused.view_as(recon), fused_b          # the loss node wrote d/d(logit) already
        elif fused is not None:
            # the loss node's logit-form gradient plus gradients of other consumers of the reconstruction
            rest, rest_b = ops.sigmoid_bwd((drecon - fused.view_as(drecon)).contiguous().view_as(recon), recon)
            dlogit, d_final_b = rest + fused.view_as(recon), rest_b + fused_b
        else:
            dlogit, d_final_b = ops.sigmoid_bwd(drecon.contiguous(), recon)
        Ho, Wo = recon.shape[2], recon.shape[3]
        dlogit4 = dlogit.view(N, Ho, Wo, 1)
        # input of the final layer: last raw deconv output with its BN+LReLU, or h0 itself when L == 0
        if L > 0:
            xin, xsc, xsh, xact = ys[-1], affs[-1][0], affs[-1][1], ops.ACT_LRELU
        else:
            xin, xsc, xsh, xact = h0.view(N, 4, 4, C0), None, None, ops.ACT_NONE
        d_final_w = off.run(lambda: ops.conv4s2_wgrad(xin, xsc, xsh, xact, dlogit4, None, None, ops.ACT_NONE), dlogit)
        # the [Cin][1][4][4] weight is already the packed [Cout_c=Cin][4][4][Cin_c=1] conv layout; every
        # data-gradient kernel also reduces the BatchNorm-backward statistics of the layer below it
        parts = None
        if L > 0 and _fuse_bn_stats(ys[-1].shape[-1]):
            du, parts = ops.conv4s2_dgrad_bnstat(dlogit4, final_w.contiguous(), ys[-1], affs[-1], out_bf16=ctx.st16)
        else:
            du, _, _ = ops.conv4s2_fwd(dlogit4, None, None, ops.ACT_NONE, final_w.contiguous(), False)
        d_dec, d_gamma, d_beta = [None] * L, [None] * L, [None] * L
        for l in range(L - 1, -1, -1):
            sc, sh, mean, invstd = affs[l]
            wp = ctx.wp[l]                        # [Cin_d][4][4][Cout_d]: the adjoint is a plain conv
            if l > 0:
                pin, psc, psh, pact = ys[l - 1], affs[l - 1][0], affs[l - 1][1], ops.ACT_LRELU
            else:
                pin, psc, psh, pact = h0.view(N, 4, 4, C0), None, None, ops.ACT_NONE
            if (parts is not None and l > 0 and _fuse_bn_stats(ys[l - 1].shape[-1])
                    and ops.conv4s2_bnin_supported(du, wp.shape[0])):
                # the data-gradient kernel takes du through this layer's BatchNorm backward as it loads
                # it and hands the finished dL/dy to the weight gradient: no elementwise pass
                d_gamma[l], d_beta[l], ca, cb = ops.bn_bwd_coeffs(parts, affs[l], du.numel() // du.shape[-1])
                du, parts, dy = ops.conv4s2_dgrad_bnstat_bnin(du, ys[l], affs[l], ca, cb, wp, ys[l - 1], affs[l - 1])
                d_dec[l] = off.run(lambda dy=dy, pin=pin, psc=psc, psh=psh, pact=pact: ops.conv4s2_wgrad(
                    pin, psc, psh, pact, dy, None, None, ops.ACT_NONE), dy)
                continue
            if parts is not None and ops.conv4s2_wgrad_bnq_supported(pin, du.shape[-1]):
                # here it is the weight-gradient kernel that takes du through the BatchNorm backward as it
                # loads it, and hands the finished dL/dy to the data gradient
                d_gamma[l], d_beta[l], ca, cb = ops.bn_bwd_coeffs(parts, affs[l], du.numel() // du.shape[-1])
                d_dec[l], dy = ops.conv4s2_wgrad_bnq(pin, psc, psh, pact, du, ys[l], affs[l], ca, cb)
            else:
                d_gamma[l], d_beta[l] = ops.bn_bwd(du, ys[l], sc, sh, mean, invstd, ctx.gammas[l], parts)
                dy = du
                d_dec[l] = off.run(lambda dy=dy, pin=pin, psc=psc, psh=psh, pact=pact: ops.conv4s2_wgrad(
                    pin, psc, psh, pact, dy, None, None, ops.ACT_NONE), dy)
            if l > 0 and _fuse_bn_stats(ys[l - 1].shape[-1]):
                du, parts = ops.conv4s2_dgrad_bnstat(dy, wp, ys[l - 1], affs[l - 1])
            else:
                (du, _, _), parts = ops.conv4s2_fwd(dy, None, None, ops.ACT_NONE, wp, False), None
        dpre = du.view(N, S * C0)                  # grad wrt h0 (post-LeakyReLU), freshly allocated
        d_bperm = ops.bias_act_bwd(dpre, h0, ops.ACT_LRELU, True)
        d_fc_w = off.run(lambda: ops.permute3(ops.gemm_tn(dpre, feat), (S, C0, Kin), (1, 0, 2)).view(C0 * S, Kin),
                         dpre, du)                 # via [S*C0, Kin]
        d_fc_b = ops.permute3(d_bperm, (S, C0, 1), (1, 0, 2)).view(C0 * S)
        dfeat = ops.gemm_nn(dpre, w_perm)
        off.join()
        return (dfeat, None, None, None, d_fc_w, d_fc_b, d_final_w, d_final_b, *d_dec, *d_gamma, *d_beta)


# =================================================================================================
class Linear(torch.autograd.Function):
    """nn.Linear (+ optional LeakyReLU(0.2)) on the f32-MFMA GEMM."""

    @staticmethod
    def forward(ctx, x: Tensor, w: Tensor, b: Optional[Tensor], act: int):
        _need_gpu(x, 'Linear')
        shape = x.shape
        x2 = x.reshape(-1, shape[-1]).contiguous()
        out = ops.gemm_nt(x2, w.detach().contiguous(), None if b is None else b.detach(), act)
        ctx.save_for_backward(x2, w, out if act != ops.ACT_NONE else None)
        ctx.act = act
        ctx.has_bias = b is not None
        ctx.in_shape = shape
        return out.view(*shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dout: Tensor):
        x2, w, out = ctx.saved_tensors
        d = dout.reshape(-1, w.shape[0]).contiguous()
        if ctx.act != ops.ACT_NONE:
            d, d_in = torch.empty_like(d), d                     # the incoming gradient is not ours to overwrite
            db = ops.bias_act_bwd(d_in, out, ctx.act, ctx.has_bias, dst=d)
        else:
            db = ops.bias_act_bwd(d, out, ctx.act, ctx.has_bias)
        off = _OffPath(d, ctx.needs_input_grad[0] and ctx.needs_input_grad[1])
        dw = off.run(lambda: ops.gemm_tn(d, x2), d) if ctx.needs_input_grad[1] else None
        dx = ops.gemm_nn(d, w.contiguous()).view(ctx.in_shape) if ctx.needs_input_grad[0] else None
        off.join()
        return dx, dw, db, None


def linear(x, w, b, act=ops.ACT_NONE):
    return Linear.apply(x, w, b, act)


class Reparam(torch.autograd.Function):
    """z = mu + eps * exp(logvar / 2)  (models/encoders.py:211-213 with the draw made explicit)."""

    @staticmethod
    def forward(ctx, mu, logvar, eps):
        _need_gpu(mu, 'Reparam')
        mu, logvar, eps = mu.contiguous(), logvar.contiguous(), eps.contiguous()
        ctx.save_for_backward(logvar, eps)
        return ops.reparam_fwd(mu, logvar, eps)

    @staticmethod
    def backward(ctx, dz):
        logvar, eps = ctx.saved_tensors
        dmu, dlv = ops.reparam_bwd(dz.contiguous(), logvar, eps)
        return dmu, dlv, None


# =================================================================================================
def _zeros(*shape, like: Tensor):
    return torch.zeros(*shape, dtype=torch.float32, device=like.device)


def _empty(*shape, like: Tensor):
    return torch.empty(*shape, dtype=torch.float32, device=like.device)


_SIDE_STREAMS = {}


def _side_stream(dev: torch.device) -> 'torch.cuda.Stream':
    """One extra stream per device for work that is independent of what the current stream runs
    (the second direction of a bidirectional LSTM layer)."""
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=dev)
    return _SIDE_STREAMS[key]


class _TwoLanes:
    """Fork/join around two independent launch sequences: lane 0 stays on the current stream, lane 1
    runs on the side stream.  Buffers touched inside must have been allocated before entering, and
    nothing launched inside may use the shared workspace."""

    def __init__(self, like: Tensor, enabled: bool):
        self.main = torch.cuda.current_stream(like.device)
        self.side = _side_stream(like.device) if enabled else None

    def __enter__(self):
        if self.side is not None:
            self.side.wait_stream(self.main)
        return self

    def lane(self, d: int):
        return torch.cuda.stream(self.side if (d == 1 and self.side is not None) else self.main)

    def __exit__(self, *exc):
        if self.side is not None:
            self.main.wait_stream(self.side)
        return False


class LSTMEncoder(torch.autograd.Function):
    """nn.LSTM(batch_first, bidirectional, num_layers) from zero state; returns cat of the last
    layer's final hidden state per direction (models/encoders.py:156-173) and, with want_all, also what nn.LSTM
    itself returns: the last layer's output sequence [B, T, D*H] and (h_n, c_n) [L*D, B, H].  Gradients flow back
    from the final state, the sequence and h_n; c_n is returned for inspection only (not differentiable).

    Weights come flat, per layer and direction: w_ih, w_hh, b_ih, b_hh.  Internally time-major.
    """

    @staticmethod
    def forward(ctx, x: Tensor, num_layers: int, bidirectional: bool, want_all: bool, *weights: Tensor):
        _need_gpu(x, 'LSTMEncoder')
        B, T, I0 = x.shape
        D = 2 if bidirectional else 1
        H = weights[1].shape[1]
        inp = ops.permute3(x.contiguous(), (B, T, I0), (1, 0, 2))       # [T, B, I]
        layers = []
        for l in range(num_layers):
            I = inp.shape[2]
            hseq = _empty(T, B, D * H, like=x)
            cseq = _empty(D, T, B, H, like=x)
            gates = _empty(D, T, B, 4 * H, like=x)
            # input projections of both directions first (they may use the shared split-K workspace),
            # then the two recurrences side by side: they share nothing but the read-only input
            ws = [[w.detach() for w in weights[(l * D + d) * 4:(l * D + d) * 4 + 4]] for d in range(D)]
            xproj = [ops.gemm_nt(inp.view(T * B, I), ws[d][0].contiguous()).view(T, B, 4 * H) for d in range(D)]
            w_hh = [ws[d][1].contiguous() for d in range(D)]
            # the whole recurrence of both directions in one launch, recurrent weights in registers; where the two
            # chains together exceed what may be resident (several ranks: room is left for the collectives' kernels),
            # one launch per direction, one after the other
            groups = _seq_groups(D, T, B, H)
            if groups is not None:
                for ds in groups:
                    ops.lstm_seq_fwd([dict(xproj=xproj[d], w_rec=w_hh[d], b_a=ws[d][2].contiguous(),
                                           b_b=ws[d][3].contiguous(), gates=gates[d],
                                           hseq=hseq[:, :, d * H:(d + 1) * H], cseq=cseq[d], reverse=(d == 1))
                                      for d in ds], T, B, H)
                layers.append((inp, hseq, cseq, gates))
                inp = hseq
                continue
            with _TwoLanes(x, D == 2) as lanes:
                for i in range(T):
                    for d in range(D):
                        t = T - 1 - i if d == 1 else i
                        prev = None if i == 0 else (t + 1 if d == 1 else t - 1)
                        h_prev = None if prev is None else hseq[prev][:, d * H:(d + 1) * H]
                        c_prev = None if prev is None else cseq[d, prev]
                        with lanes.lane(d):
                            ops.lstm_step_fwd(xproj[d][t], None, None, h_prev, c_prev, w_hh[d], ws[d][2], ws[d][3],
                                              gates[d, t], hseq[t][:, d * H:(d + 1) * H], cseq[d, t])
            layers.append((inp, hseq, cseq, gates))
            inp = hseq
        last = layers[-1][1]
        if bidirectional:
            final = torch.cat([last[T - 1][:, :H], last[0][:, H:]], dim=1)
        else:
            final = last[T - 1].clone()
        ctx.layers = layers
        ctx.cfg = (B, T, I0, D, H, num_layers)
        ctx.want_all = want_all
        ctx.save_for_backward(*weights)
        if not want_all:
            return final
        out_seq = ops.permute3(last, (T, B, D * H), (1, 0, 2))                      # [B, T, D*H]
        # the chronologically last step of a chain: T-1 forward, 0 for the reverse direction
        h_n = torch.stack([layers[l][1][0 if d == 1 else T - 1][:, d * H:(d + 1) * H]
                           for l in range(num_layers) for d in range(D)])
        c_n = torch.stack([layers[l][2][d, 0 if d == 1 else T - 1] for l in range(num_layers) for d in range(D)])
        ctx.mark_non_differentiable(c_n)
        return final, out_seq, h_n, c_n

    @staticmethod
    def backward(ctx, dfinal: Optional[Tensor], dseq: Optional[Tensor] = None, dh_n: Optional[Tensor] = None,
                 dc_n: Optional[Tensor] = None):
        B, T, I0, D, H, L = ctx.cfg
        weights = ctx.saved_tensors
        grads = [None] * len(weights)
        like = next(t for t in (dfinal, dseq, dh_n) if t is not None)
        dfinal = dfinal.contiguous() if dfinal is not None else _zeros(B, D * H, like=like)
        dout = None                                    # gradient wrt this layer's output sequence
        for l in range(L - 1, -1, -1):
            inp, hseq, cseq, gates = ctx.layers[l]
            I = inp.shape[2]
            dh = dout if dout is not None else _zeros(T, B, D * H, like=dfinal)
            if l == L - 1:
                if dseq is not None:
                    dh += ops.permute3(dseq.contiguous(), (B, T, D * H), (1, 0, 2))
                dh[T - 1][:, :H] += dfinal[:, :H]
                if D == 2:
                    dh[0][:, H:] += dfinal[:, H:]
            if dh_n is not None:
                for d in range(D):
                    dh[0 if d == 1 else T - 1][:, d * H:(d + 1) * H] += dh_n[l * D + d]
            dinp = None
            # the two directions' recurrences side by side, then the weight-gradient GEMMs in turn
            dgs = [_empty(T, B, 4 * H, like=dfinal) for _ in range(D)]
            dcs = [[_empty(B, H, like=dfinal), _empty(B, H, like=dfinal)] for _ in range(D)]
            w_hhs = [weights[(l * D + d) * 4 + 1].contiguous() for d in range(D)]
            groups = _seq_groups(D, T, B, H)
            seq = groups is not None
            if seq:
                # both directions' chains in one launch (or one launch each, see forward), recurrent weights in registers
                for ds in groups:
                    ops.lstm_seq_bwd([dict(w_rec=w_hhs[d], gates=gates[d], cseq=cseq[d],
                                           dh_ext=dh[:, :, d * H:(d + 1) * H], dgates=dgs[d], reverse=(d == 1))
                                      for d in ds], T, B, H)
            w_hh_t = [] if seq else [w.t().contiguous() for w in w_hhs]   # [H, 4H]: k-contiguous rows for dh = dG * W_hh
            with _TwoLanes(dfinal, D == 2 and not seq) as lanes:
                for i in (() if seq else range(T - 1, -1, -1)):   # walk the recurrences backwards
                    for d in range(D):
                        t = T - 1 - i if d == 1 else i
                        tp = None if i == 0 else (t + 1 if d == 1 else t - 1)
                        sl = slice(d * H, (d + 1) * H)
                        dc = dcs[d]
                        with lanes.lane(d):
                            if i == T - 1:                        # the last step starts the chain
                                ops.lstm_step_bwd_pointwise(dh[t][:, sl], None, gates[d, t],
                                                            None if tp is None else cseq[d, tp], cseq[d, t],
                                                            dgs[d][t], dc[i & 1])
                            if tp is not None:
                                # dh[tp] += dG_t W_hh, then straight through the cell of step tp
                                tpp = None if i == 1 else (tp + 1 if d == 1 else tp - 1)
                                ops.lstm_bwd_step(dgs[d][t], w_hh_t[d], dh[tp][:, sl], dc[i & 1], gates[d, tp],
                                                  None if tpp is None else cseq[d, tpp], cseq[d, tp],
                                                  dgs[d][tp], dc[(i - 1) & 1])
            for d in range(D):
                k = (l * D + d) * 4
                w_ih = weights[k].contiguous()
                dg = dgs[d]
                sl = slice(d * H, (d + 1) * H)
                dg2 = dg.view(T * B, 4 * H)
                grads[k] = ops.gemm_tn(dg2, inp.view(T * B, I))                      # dW_ih
                # dW_hh = sum_t dG_t^T h_prev(t): forward uses h[t-1], reverse uses h[t+1]
                hs2 = hseq.view(T * B, D * H)[:, sl]
                if T > 1:
                    if d == 0:
                        grads[k + 1] = ops.gemm_tn(dg2[B:], hs2[:(T - 1) * B])
                    else:
                        grads[k + 1] = ops.gemm_tn(dg2[:(T - 1) * B], hs2[B:])
                else:
                    grads[k + 1] = torch.zeros_like(w_hhs[d])
                db = ops.bias_act_bwd(dg2, None, ops.ACT_NONE, True)
                grads[k + 2] = db
                grads[k + 3] = db.clone()
                if l > 0 or ctx.needs_input_grad[0]:
                    if dinp is None:
                        dinp = ops.gemm_nn(dg2, w_ih)
                    else:
                        ops.gemm_nn(dg2, w_ih, out=dinp, accumulate=True)
            dout = None if dinp is None else dinp.view(T, B, I)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops.permute3(dout, (T, B, I0), (1, 0, 2))
        return (dx, None, None, None, *grads)


class LSTMDecoder(torch.autograd.Function):
    """Autoregressive decoder recurrence (models/decoders.py:118-134, :253-269).

    h0c0_h, h0c0_c: [B, L*H] (fc_hidden / fc_cell outputs; layer l seeds from columns l*H:(l+1)*H);
    initial_input [1, H]; weights per layer: w_ih, w_hh, b_ih, b_hh.  Returns states [B, T, H].
    """

    @staticmethod
    def forward(ctx, h0: Tensor, c0: Tensor, init: Tensor, T: int, num_layers: int, *weights: Tensor):
        _need_gpu(h0, 'LSTMDecoder')
        B = h0.shape[0]
        L = num_layers
        H = h0.shape[1] // L
        h0, c0 = h0.contiguous(), c0.contiguous()
        init_c = init.detach().contiguous()
        hs = _empty(L, T, B, H, like=h0)
        cs = _empty(L, T, B, H, like=h0)
        gates = _empty(L, T, B, 4 * H, like=h0)
        w = [t.detach().contiguous() for t in weights]
        x0 = init_c.expand(B, H)                      # row stride 0: every row reads the same vector
        # a single layer is fed its own previous output: x_t == h_{t-1} for t > 0, so
        # x W_ih^T + h W_hh^T = h (W_ih + W_hh)^T and the step is H deep instead of 2H
        w_comb = (w[0] + w[1]) if (L == 1 and T > 1) else None
        seq = w_comb is not None and _seq_ok(1, T - 1, B, H)
        dec2 = L == 2 and _dec2_ok(T, B, H)
        if dec2:
            # both layers' cells of all T steps as one chain in one launch, the four weight slices in registers
            ops.lstm_dec2_fwd(w, h0, c0, init_c.view(-1), gates, hs, cs)
        for t in range(0 if dec2 else 1 if seq else T):
            for l in range(L):
                h_prev = h0[:, l * H:(l + 1) * H] if t == 0 else hs[l, t - 1]
                c_prev = c0[:, l * H:(l + 1) * H] if t == 0 else cs[l, t - 1]
                if w_comb is not None and t > 0:
                    ops.lstm_step_fwd(None, None, None, h_prev, c_prev, w_comb, w[2], w[3],
                                      gates[l, t], hs[l, t], cs[l, t])
                    continue
                if l == 0:
                    x = x0 if t == 0 else hs[L - 1, t - 1]
                else:
                    x = hs[l - 1, t]
                ops.lstm_step_fwd(None, x, w[4 * l], h_prev, c_prev, w[4 * l + 1], w[4 * l + 2], w[4 * l + 3],
                                  gates[l, t], hs[l, t], cs[l, t])
        if seq:
            # steps 1 .. T-1 feed on their own previous output: one launch, W_ih + W_hh in registers
            ops.lstm_seq_fwd([dict(w_rec=w_comb, b_a=w[2], b_b=w[3], h_init=hs[0, 0], c_init=cs[0, 0],
                                   gates=gates[0, 1:], hseq=hs[0, 1:], cseq=cs[0, 1:])], T - 1, B, H)
        states = ops.permute3(hs[L - 1], (T, B, H), (1, 0, 2))         # [B, T, H]
        ctx.save_for_backward(h0, c0, init_c, hs, cs, gates, *weights)
        ctx.cfg = (B, T, L, H)
        ctx.w_comb = w_comb if seq else None      # W_ih + W_hh once per step, for the backward launch too
        return states

    @staticmethod
    def backward(ctx, dstates: Tensor):
        B, T, L, H = ctx.cfg
        saved = ctx.saved_tensors
        h0, c0, init_c, hs, cs, gates = saved[:6]
        weights = [t.contiguous() for t in saved[6:]]
        dg = _empty(L, T, B, 4 * H, like=h0)
        dh0 = _empty(B, L * H, like=h0)
        dc0 = _empty(B, L * H, like=h0)
        dx0 = _empty(B, H, like=h0)
        dec2 = L == 2 and _dec2_ok(T, B, H)
        if dec2:
            # the whole chain of cells backwards in one launch
            ops.lstm_dec2_bwd(weights, gates, cs, c0, dstates.contiguous(), dg, dh0, dc0, dx0)
            return LSTMDecoder._weight_grads(B, T, L, H, h0, init_c, hs, dg, dh0, dc0, dx0)
        if L == 1:
            dhs = ops.permute3(dstates.contiguous(), (B, T, H), (1, 0, 2)).view(1, T, B, H)
        else:
            dhs = _zeros(L, T, B, H, like=h0)
            dhs[L - 1] = ops.permute3(dstates.contiguous(), (B, T, H), (1, 0, 2))
        dcbuf = [[_empty(B, H, like=h0), _empty(B, H, like=h0)] for _ in range(L)]
        # dh = dG * W needs W's columns as k-contiguous rows: transposed copies, once per backward;
        # for a single layer both paths of a step t > 0 end in h_{t-1}, so they share W_ih + W_hh
        w_ih_t = [weights[4 * l].t().contiguous() for l in range(L)]
        w_hh_t = [weights[4 * l + 1].t().contiguous() for l in range(L)]
        fused = L == 1 and T > 1          # single layer: one launch per step (product + next pointwise) ...
        seq = fused and _seq_ok(1, T - 1, B, H)   # ... or one for all of them
        w_comb_t = (w_ih_t[0] + w_hh_t[0]) if (fused and not seq) else None
        if seq:
            # steps T-1 .. 1 in one launch (W_ih + W_hh in registers); it leaves dL/dh_0 and dL/dc_0 for step 0
            w_comb = ctx.w_comb if ctx.w_comb is not None else weights[0] + weights[1]
            ops.lstm_seq_bwd([dict(w_rec=w_comb, gates=gates[0, 1:], cseq=cs[0, 1:], c_init=cs[0, 0],
                                   dh_ext=dhs[0, 1:], dgates=dg[0, 1:], dh_init=dhs[0, 0], dc_init=dcbuf[0][1])],
                             T - 1, B, H, accumulate_dh_init=True)
            ops.lstm_step_bwd_pointwise(dhs[0, 0], dcbuf[0][1], gates[0, 0], c0[:, :H], cs[0, 0], dg[0, 0], dc0[:, :H])
        stacked = L >= 2
        if stacked:
            # Stacked decoder: walk the cells (t, l) from the last one back.  A hidden state h_{l,t} receives at
            # most two products -- the input path of the layer above at the same step (for the top layer: of
            # layer 0 at the next step, which it feeds) and its own layer's recurrent path from the next step --
            # so one launch per cell adds both and goes straight through the cell's pointwise backward.
            ops.lstm_step_bwd_pointwise(dhs[L - 1, T - 1], None, gates[L - 1, T - 1],
                                        c0[:, (L - 1) * H:] if T == 1 else cs[L - 1, T - 2], cs[L - 1, T - 1],
                                        dg[L - 1, T - 1], dc0[:, (L - 1) * H:] if T == 1 else dcbuf[L - 1][(T - 1) & 1])
            for t in range(T - 1, -1, -1):
                for l in range(L - 1, -1, -1):
                    if t == T - 1 and l == L - 1:
                        continue
                    first = (dg[l + 1, t], w_ih_t[l + 1]) if l < L - 1 else (dg[0, t + 1], w_ih_t[0])
                    second = (dg[l, t + 1], w_hh_t[l]) if t < T - 1 else (None, None)
                    ops.lstm_bwd_step2(first[0], first[1], second[0], second[1], dhs[l, t],
                                       None if t == T - 1 else dcbuf[l][(t + 1) & 1], gates[l, t],
                                       c0[:, l * H:(l + 1) * H] if t == 0 else cs[l, t - 1], cs[l, t], dg[l, t],
                                       dc0[:, l * H:(l + 1) * H] if t == 0 else dcbuf[l][t & 1])
            # what reaches the initial states and the learned first input
            for l in range(L):
                ops.skinny_gemm_nt(dg[l, 0], w_hh_t[l], dh0[:, l * H:(l + 1) * H], False)
            ops.skinny_gemm_nt(dg[0, 0], w_ih_t[0], dx0, False)
        for t in (() if stacked else (0,) if seq else range(T - 1, -1, -1)):
            for l in range(L - 1, -1, -1):
                w_ih, w_hh = w_ih_t[l], w_hh_t[l]
                c_prev = c0[:, l * H:(l + 1) * H] if t == 0 else cs[l, t - 1]
                dc_in = None if t == T - 1 else dcbuf[l][(t + 1) & 1]
                dc_out = dc0[:, l * H:(l + 1) * H] if t == 0 else dcbuf[l][t & 1]
                if (not fused or t == T - 1) and not seq:
                    ops.lstm_step_bwd_pointwise(dhs[l, t], dc_in, gates[l, t], c_prev, cs[l, t], dg[l, t], dc_out)
                if fused and t > 0:
                    # dh_{t-1} += dG_t (W_ih + W_hh), then straight through the cell of step t-1
                    ops.lstm_bwd_step(dg[0, t], w_comb_t, dhs[0, t - 1], dc_out, gates[0, t - 1],
                                      c0[:, :H] if t == 1 else cs[0, t - 2], cs[0, t - 1], dg[0, t - 1],
                                      dc0[:, :H] if t == 1 else dcbuf[0][(t - 1) & 1])
                    continue
                # recurrent path -> h_{t-1} of this layer
                rec_target = dh0[:, l * H:(l + 1) * H] if t == 0 else dhs[l, t - 1]
                rec_acc = t > 0
                # input path -> layer below at the same step, or the top layer's previous output
                if l > 0:
                    ops.skinny_gemm_nt(dg[l, t], w_hh, rec_target, rec_acc)
                    ops.skinny_gemm_nt(dg[l, t], w_ih, dhs[l - 1, t], True)
                elif t > 0:
                    if L == 1:
                        ops.skinny_gemm_nt(dg[0, t], w_comb_t, rec_target, True)
                    else:
                        ops.skinny_gemm_nt(dg[0, t], w_hh, rec_target, True)
                        ops.skinny_gemm_nt(dg[0, t], w_ih, dhs[L - 1, t - 1], True)
                else:
                    ops.skinny_gemm_nt(dg[0, 0], w_hh, rec_target, False)
                    ops.skinny_gemm_nt(dg[0, 0], w_ih, dx0, False)
        return LSTMDecoder._weight_grads(B, T, L, H, h0, init_c, hs, dg, dh0, dc0, dx0)

    @staticmethod
    def _weight_grads(B, T, L, H, h0, init_c, hs, dg, dh0, dc0, dx0):
        """the hoisted part of the backward: every weight gradient is one GEMM over all steps' gate gradients"""
        dinit = ops.bias_act_bwd(dx0, None, ops.ACT_NONE, True).view(1, H)
        grads = [None] * (4 * L)
        for l in range(L):
            dg2 = dg[l].view(T * B, 4 * H)
            hs2 = hs[l].view(T * B, H)
            # dW_hh: h_prev(t) = hs[l, t-1] for t >= 1, h0 slice for t = 0
            dw_hh = ops.gemm_tn(dg2[:B], h0[:, l * H:(l + 1) * H])
            if T > 1:
                ops.gemm_tn(dg2[B:], hs2[:(T - 1) * B], out=dw_hh, accumulate=True)
            grads[4 * l + 1] = dw_hh
            if l > 0:
                grads[4 * l] = ops.gemm_tn(dg2, hs[l - 1].view(T * B, H))
            else:
                # x_0 = initial_input for every row: outer(colsum(dG_0), init)
                colsum0 = ops.bias_act_bwd(dg2[:B], None, ops.ACT_NONE, True).view(1, 4 * H)
                dw_ih = ops.gemm_tn(colsum0, init_c.view(1, H))
                if T > 1:
                    ops.gemm_tn(dg2[B:], hs[L - 1].view(T * B, H)[:(T - 1) * B], out=dw_ih, accumulate=True)
                grads[0] = dw_ih
            db = ops.bias_act_bwd(dg2, None, ops.ACT_NONE, True)
            grads[4 * l + 2] = db
            grads[4 * l + 3] = db.clone()
        return (dh0, dc0, dinit, None, None, *grads)
