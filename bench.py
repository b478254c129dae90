"""Training-throughput benchmark of the MI355X hot path.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step = zero_grad + forward + loss + backward + (gradient all-reduce) + clip + Adam on one batch of
synthetic 20x64x64 clips that is already resident in HBM.  Rank 0 prints ONE JSON line.  Weak
scaling: the per-GPU batch is fixed, the global batch grows with N.

Besides the headline clips/s the line carries
  roofline      the kernel symbol that takes the most time in a step, timed live with events around
                every launch of it on the stream it runs on: algorithmic FLOPs / time vs the fp32 MFMA peak;
  kernels       the same for the other GEMM-shaped and HBM-bound kernel groups (short table);
  cpu_baseline  the CPU oracle (oracle/vae_oracle.py, a restatement of the reference pinned to the
                reference's own vectors) timed on this box's host cores on a few B=32 steps.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, 'disentangled-vae_amd'))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3     # v_mfma_f32_32x32x2_f32, dense
PEAK_HBM_GBS = 8000.0


def load_config(name):
    with open(os.path.join(ROOT, 'disentangled-vae_amd', 'configs', name)) as f:
        return json.load(f)


def model_kwargs(cfg):
    m = cfg['model']
    kw = dict(in_channels=m['in_channels'], sequence_length=m['sequence_length'])
    if m['model_type'] == 'baseline':
        kw.update(latent_dim=m['latent_dim'], encoder_channels=m['encoder_channels'],
                  decoder_channels=m['decoder_channels'], feature_dim=m['feature_dim'])
    elif m['model_type'] == 'lstm':
        kw.update(latent_dim=m['latent_dim'], encoder_channels=m['encoder_channels'],
                  decoder_channels=m['decoder_channels'], feature_dim=m['feature_dim'],
                  lstm_hidden=m['lstm_hidden'], lstm_layers=m['lstm_layers'], bidirectional=m['bidirectional'])
    else:
        kw.update(content_dim=m['content_dim'], motion_dim=m['motion_dim'], content_channels=m['encoder_channels'],
                  motion_channels=m['motion_channels'], decoder_channels=m['decoder_channels'],
                  content_feature_dim=m['feature_dim'], motion_feature_dim=m['feature_dim'],
                  lstm_hidden=m['lstm_hidden'], lstm_layers=m['lstm_layers'], bidirectional=m['bidirectional'],
                  content_aggregation=m['content_aggregation'])
    return kw


def synthetic_clips(B, T, size, seed):
    g = torch.Generator().manual_seed(seed)
    u1 = torch.rand(B, T, 1, size, size, generator=g)
    u2 = torch.rand(B, T, 1, size, size, generator=g)
    return (u1 > 0.9).float() * u2


def loss_call(kind, losses_mod, out, video, t, beta):
    if kind == 'disentangled':
        return losses_mod.disentangled_loss(out, video, beta=beta, beta_content=beta, beta_motion=beta,
                                            recon_loss_type=t['recon_loss_type'],
                                            temporal_weight=t['temporal_weight'],
                                            independence_weight=t['independence_weight'])
    fn = losses_mod.lstm_loss if kind == 'lstm' else losses_mod.baseline_loss
    recon, z, mu, logvar = out
    return fn(recon, video, mu, logvar, beta=beta, recon_loss_type=t['recon_loss_type'],
              free_bits=t['free_bits'], use_free_bits=t['use_free_bits'])


# --------------------------------------------------------------------------------------------
# per-kernel timing: events around every launch of the GEMM-shaped and streaming entry points
# --------------------------------------------------------------------------------------------
class KernelTimer:
    def __init__(self, ops):
        self.ops = ops
        self.records = []          # (key, start_event, end_event, work, unit)
        self.saved = {}
        self.shapes = []           # DVAE_BENCH_DEBUG: tensor-argument shapes of every timed launch

    def _wrap(self, name, keyfn):
        orig = getattr(self.ops, name)
        self.saved[name] = orig

        def wrapped(*a, **k):
            key, work, unit = keyfn(*a, **k)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = orig(*a, **k)
            e.record()
            self.records.append((key, s, e, work, unit))
            if os.environ.get('DVAE_BENCH_DEBUG'):
                self.shapes.append([tuple(t.shape) for t in a if torch.is_tensor(t)])
            return r
        setattr(self.ops, name, wrapped)

    def install(self):
        def tile(c, rows=None, phases=1, stats=True):
            """tile the launcher picks: by output channels, narrowed while fewer than 256 workgroups; a
            layer without statistics whose 128-wide tiles end in a partly filled round of the 512
            workgroup slots finishes with a second launch of 64-wide tiles"""
            bn = 128 if c > 64 else (64 if c > 32 else 32)
            if rows is not None:
                row_tiles = -(-rows // 128) * phases
                while bn > 32 and row_tiles * -(-c // bn) < 256:
                    bn //= 2
                tiles = row_tiles * -(-c // 128)
                if bn == 128 and not stats and tiles >= 512 and 512 <= 5 * (tiles % 512) and 10 * (tiles % 512) <= 3584:
                    return 'T128+T64'
            return f'T{bn}'

        def conv_key(x, sc, sh, act, wp, stats):
            N, H, W, Cin = x.shape
            Cout = wp.shape[0]
            if Cin == 1:        # sixteen taps deep: bound by the write of the output
                return "conv_c1_kernel", float(x.numel() * 4 + N * (H // 2) * (W // 2) * Cout * 4), 'byte'
            fl = 2.0 * N * (H // 2) * (W // 2) * Cout * 16 * Cin
            if (Cin, Cout) in ((32, 64), (16, 32)) and 32 <= W <= 64:     # few channels: rows through an LDS ring
                return f"conv_rows_kernel<{Cin},{Cout}>", fl, 'flop'
            return f"conv_fwd_kernel<{tile(Cout, N * (H // 2) * (W // 2), 1, stats)},conv>", fl, 'flop'

        def conv_bn_key(x, wp, bn_y, aff):
            key, work, unit = conv_key(x, None, None, 0, wp, True)
            if unit == 'byte':
                return key + ',bnstat', work + float(bn_y.numel() * 4), unit
            return key[:-1] + ',bnstat>', work, unit

        def conv_bnin_key(du_in, in_y, in_aff, ca, cb, wp, bn_y, bn_aff):
            key, work, unit = conv_key(du_in, None, None, 0, wp, True)
            return key[:-1] + ',bnstat,bnin>', work, unit

        def deconv_bn_key(x, wp, bn_y, aff):
            key, work, unit = deconv_key(x, None, None, 0, wp, True)
            return key[:-1] + ',bnstat>', work, unit

        def deconv_key(x, sc, sh, act, wp, stats):
            N, H, W, Cin = x.shape
            Cout = wp.shape[1]
            fl = 2.0 * N * 4 * H * W * Cout * 4 * Cin
            if (Cin, Cout, W) == (128, 64, 8):
                return "deconv_rows8_kernel<128>", fl, 'flop'
            if (Cin, Cout) in ((32, 16), (64, 32)) and 16 <= W <= 32:   # few channels: rows through an LDS ring
                return f"deconv_rows_kernel<{Cin},{Cout}>", fl, 'flop'
            return f"conv_fwd_kernel<{tile(Cout, N * H * W, 4, stats)},deconv>", fl, 'flop'

        def wgrad_key(P, ps, ph, pa, Q, qs, qh, qa):
            N, Hs, Ws, C1 = P.shape
            C2 = Q.shape[3]
            fl = 2.0 * N * Hs * Ws * C1 * 16 * C2
            if (C1, C2, Ws, Hs) == (256, 128, 4, 4) and N >= 32:
                return "wgrad_img4_kernel", fl, 'flop'
            if (C1, C2, Ws) == (128, 64, 8):
                return "wgrad_rows8_kernel", fl, 'flop'
            if (C1, C2) in ((64, 32), (32, 16)) and Ws in (16, 32):       # few channels: rows through LDS rings
                return f"wgrad_rows_kernel<{C1},{C2}>", fl, 'flop'
            if C2 == 1:
                var = 'T128x32,c1'
            else:
                var = 'T32x128' if C1 <= 32 else ('T64x128' if C1 <= 64 else 'T128x128')
            return f"conv_wgrad_kernel<{var}>", fl, 'flop'

        def gemm_key(kind):
            def f(a, b, *rest, **kw):
                if kind == 'nt':
                    M, K = a.shape
                    N = b.shape[0]
                elif kind == 'nn':
                    M, K = a.shape
                    N = b.shape[1]
                else:
                    K, M = a.shape
                    N = b.shape[1]
                tiles16 = -(-M // 16) * -(-N // 16)      # the launcher's rule for its small-product kernel
                name = 'small_gemm_kernel' if tiles16 <= 1024 and tiles16 * K <= (1 << 20) else 'gemm_kernel'
                return f"{name}<{kind}>", 2.0 * M * N * K, 'flop'
            return f

        def final_key(x, sc, sh, act, w, b, oa):
            return "deconv_final_kernel", float(x.numel() * 4 + x.shape[0] * 4 * x.shape[1] * x.shape[2] * 4), 'byte'

        def split_key(v):
            return "split_mean_diff_kernel", float(v.numel() * 4 * 2), 'byte'

        def bn_bwd_key(du, y, sc, sh, mean, invstd, gamma, partials=None):
            if partials is not None:       # statistics came from the producing kernel: 2 reads + 1 write
                return "bn_bwd (apply)", float(y.numel() * 4 * 3), 'byte'
            return "bn_bwd (reduce+apply)", float(y.numel() * 4 * 5), 'byte'    # 2 reads + 2 reads + 1 write

        def loss_fwd_key(recon, target, *r):
            return "recon_loss_fwd_kernel", float(recon.numel() * 4 * 2), 'byte'

        def loss_bwd_key(recon, target, *r):
            return "recon_loss_bwd_kernel", float(recon.numel() * 4 * 3), 'byte'

        def lstm_key(*a):
            h_out = a[9]
            B, H = h_out.shape
            x = a[1]
            K = H + (0 if x is None else x.shape[1])
            return "lstm_step_fwd_kernel", 2.0 * B * 4 * H * K, 'flop'

        def skinny_key(a, wt, out, acc):
            B, K = a.shape
            return "skinny_gemm_nt_kernel", 2.0 * B * wt.shape[0] * K, 'flop'

        self._wrap('conv4s2_fwd', conv_key)
        self._wrap('deconv4s2_fwd', deconv_key)
        self._wrap('conv4s2_dgrad_bnstat', conv_bn_key)
        self._wrap('deconv4s2_dgrad_bnstat', deconv_bn_key)
        self._wrap('conv4s2_dgrad_bnstat_bnin', conv_bnin_key)

        def deconv_bnin_key(du_in, in_y, in_aff, ca, cb, wp, bn_y, bn_aff):
            key, work, unit = deconv_key(du_in, None, None, 0, wp, True)
            return key[:-1] + ',bnstat,bnin>', work, unit

        def wgrad_bnq_key(P, ps, ph, pa, du, q_y, q_aff, ca, cb):
            key, work, unit = wgrad_key(P, ps, ph, pa, du, None, None, 0)
            return key + ',bnq', work, unit

        def wgrad_bnp_key(du, p_y, p_aff, ca, cb, Q):
            key, work, unit = wgrad_key(du, None, None, 0, Q, None, None, 0)
            return key[:-1] + ',bnp>', work, unit

        self._wrap('deconv4s2_dgrad_bnstat_bnin', deconv_bnin_key)
        self._wrap('conv4s2_wgrad_bnq', wgrad_bnq_key)
        self._wrap('conv4s2_wgrad_bnp', wgrad_bnp_key)
        self._wrap('conv4s2_wgrad', wgrad_key)
        self._wrap('gemm_nt', gemm_key('nt'))
        self._wrap('gemm_nn', gemm_key('nn'))
        self._wrap('gemm_tn', gemm_key('tn'))
        self._wrap('deconv4s2_final_fwd', final_key)
        self._wrap('split_mean_diff', split_key)
        self._wrap('bn_bwd', bn_bwd_key)
        self._wrap('recon_loss_fwd', loss_fwd_key)
        self._wrap('recon_loss_bwd', loss_bwd_key)
        self._wrap('lstm_step_fwd', lstm_key)
        self._wrap('skinny_gemm_nt', skinny_key)
        self._wrap('lstm_bwd_step', lambda dg_t, wt, *r: ('lstm_bwd_step_kernel', 2.0 * dg_t.shape[0] * wt.shape[0] * dg_t.shape[1], 'flop'))

    def remove(self):
        for n, f in self.saved.items():
            setattr(self.ops, n, f)

    def summary(self, steps):
        """One row per kernel group.  Every step issues the same launches in the same order, so the
        records split into `steps` equal runs; a group's time per step is the MEDIAN over the steps of
        its per-step sum (one stalled launch, e.g. behind an allocation, does not move it)."""
        torch.cuda.synchronize()
        per = len(self.records) // steps
        if per * steps != len(self.records):
            raise RuntimeError(f"{len(self.records)} timed launches do not split into {steps} equal steps")
        agg = {}
        for i, (key, s, e, work, unit) in enumerate(self.records):
            a = agg.setdefault(key, {'ms': [0.0] * steps, 'launches': 0, 'work': 0.0, 'unit': unit})
            ms = s.elapsed_time(e)
            a['ms'][i // per] += ms
            a['launches'] += 1
            a['work'] += work
            if os.environ.get('DVAE_BENCH_DEBUG'):
                rate = work / ms / (1e9 if unit == 'flop' else 1e6)
                print('launch', i // per, key, round(ms * 1e3, 1), 'us', round(rate, 1),
                      'TFLOP/s' if unit == 'flop' else 'GB/s', self.shapes[i], file=sys.stderr)
        rows = []
        for key, a in agg.items():
            ms = sorted(a['ms'])[steps // 2]
            sec, work, launches = ms / 1e3, a['work'] / steps, a['launches'] // steps
            if a['unit'] == 'flop':
                ach, peak, u, bound = work / sec / 1e12, PEAK_FP32_MFMA_TFLOPS, 'TFLOP/s', 'mfma'
            else:
                ach, peak, u, bound = work / sec / 1e9, PEAK_HBM_GBS, 'GB/s', 'hbm'
            rows.append({'kernel': key, 'bound': bound, 'achieved': round(ach, 2), 'peak': peak, 'unit': u,
                         'frac': round(ach / peak, 4), 'ms_per_step': round(ms, 4),
                         'launches_per_step': launches,
                         'avg_launch_us': round(ms * 1e3 / launches, 2)})
        rows.sort(key=lambda r: -r['ms_per_step'])
        return rows


def pmc_traffic(label):
    """HBM bytes per launch of a kernel group from the newest committed counter summary
    (profiles/*pmc_traffic.json: separate rocprofv3 --pmc passes of this same command; counters cannot
    be collected inside a timed run).  (None, None) when there is no entry."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*pmc_traffic.json')))
    if not files:
        return None, None
    try:
        entry = json.load(open(files[-1]))['kernels'].get(label)
    except (OSError, ValueError, KeyError):
        return None, None
    if not entry:
        return None, None
    return entry['hbm_bytes_per_launch'], os.path.relpath(files[-1], ROOT)


# --------------------------------------------------------------------------------------------
def cpu_baseline(kind, cfg, steps=3, batch=32):
    """The CPU oracle's training step timed on the host cores (bounded sample)."""
    from oracle import vae_oracle as O
    import models
    torch.manual_seed(0)
    kw = model_kwargs(cfg)
    # weights with the reference's default initialisation; parameter containers only, no HIP compute
    m = models.get_model(kind, **kw)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    T, size = cfg['data']['sequence_length'], cfg['data']['frame_size']
    video = synthetic_clips(batch, T, size, 0)
    dim = kw.get('content_dim', kw.get('latent_dim'))
    eps = [torch.randn(batch, dim, generator=torch.Generator().manual_seed(100 + k)) for k in range(2)]
    t = dict(cfg['training'])
    t['beta_content'] = t['beta_motion'] = 4.0
    layers = kw.get('lstm_layers', 1)
    opt = {}
    times = []
    for i in range(steps + 1):
        t0 = time.perf_counter()
        sd, opt, _, _ = O.train_step(kind, sd, opt, video, eps, 4.0, t, t['learning_rate'], lstm_layers=layers,
                                     bidirectional=kw.get('bidirectional', True))
        times.append(time.perf_counter() - t0)
    med = sorted(times[1:])[len(times[1:]) // 2]
    return {'value': round(batch / med, 2), 'unit': 'clips/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': f'{steps} training steps (after 1 warm-up) of the CPU oracle, batch {batch}, fp32, '
                      f'median {med * 1e3:.0f} ms/step'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--config', default='disentangled_simple.json')
    ap.add_argument('--batch', type=int, default=256, help='clips per GPU')
    ap.add_argument('--mode', default='train', choices=['train', 'infer'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-table', action='store_true')
    ap.add_argument('--bucket-mb', type=float, default=16.0)
    ap.add_argument('--precision', default='fp32', choices=['fp32', 'bf16'],
                    help='operand precision of the MFMA kernels for the headline number')
    ap.add_argument('--no-extras', action='store_true', help='skip the inference and bf16-variant measurements')
    ap.add_argument('--host-batches', default='', choices=['', 'serial', 'prefetch'],
                    help='feed every step from pinned host memory over PCIe: copied on the step stream (serial) or one '
                         'batch ahead on a side stream (prefetch); the headline value keeps batches resident in HBM')
    ap.add_argument('--force-dist', action='store_true',
                    help='initialise the RCCL process group and use the data-parallel step even with one rank '
                         '(exercises the multi-GPU code path on a single GPU)')
    ap.add_argument('--graph', type=int, default=-1, help='0: eager launches (several ranks: bucketed all-reduces '
                    'overlapped with backward); otherwise replay the captured step (several ranks: two hipGraphs '
                    'around one gradient all-reduce)')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run "
                         f"--nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    use_dp = world > 1 or args.force_dist
    if use_dp:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    import models
    from hipvae import losses as L, ops
    from hipvae.optim import FusedAdam
    from hipvae.parallel import GradSync

    ops.set_precision(args.precision)
    cfg = load_config(args.config)
    kind = cfg['model']['model_type']
    t = cfg['training']
    T, size = cfg['data']['sequence_length'], cfg['data']['frame_size']
    B = args.batch
    torch.manual_seed(0)
    model = models.get_model(kind, **model_kwargs(cfg)).to(dev)
    n_params = sum(p.numel() for p in model.parameters())
    opt = FusedAdam(model.parameters(), lr=t['learning_rate'], betas=(t['beta1'], t['beta2']),
                    weight_decay=t['weight_decay'])
    use_graph_early = args.mode == 'train' and args.graph != 0
    sync = GradSync(model, bucket_mb=args.bucket_mb) if (use_dp and not use_graph_early) else None
    ring = [synthetic_clips(B, T, size, 1000 * rank + i).to(dev) for i in range(4)]
    beta = 4.0

    def train_step(i):
        video = ring[i % len(ring)]
        if sync is not None:
            sync.zero_grad()
        else:
            opt.zero_grad(set_to_none=True)
        out = model(video)
        losses = loss_call(kind, L, out, video, t, beta)
        losses['total_loss'].backward()
        if sync is not None:
            sync.finish()
        if t['gradient_clip'] > 0:
            opt.clip_grad_norm_(t['gradient_clip'], fold_into_step=True)
        opt.step()
        return losses

    def infer_step(i):
        with torch.no_grad():
            return model(ring[i % len(ring)])

    # default: the captured step everywhere (with several ranks: two graphs around one all-reduce);
    # --graph 0 is the eager step (with several ranks: bucketed all-reduces overlapped with backward)
    use_graph = args.mode == 'train' and args.graph != 0

    def make_train_step():
        model.train()
        if not use_graph:
            return train_step
        from hipvae.graph import GraphedDataParallelStep, GraphedTrainStep
        cls = GraphedDataParallelStep if use_dp else GraphedTrainStep
        gstep = cls(model, opt, lambda m, v: loss_call(kind, L, m(v), v, t, beta), ring[0],
                    gradient_clip=t['gradient_clip'])
        return lambda i: gstep(ring[i % len(ring)])

    def barrier():
        if use_dp:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(step_fn, steps, warmup):
        """warm-up, then `steps` steps between barrier + synchronize; max over ranks; seconds"""
        out = None
        for i in range(warmup):
            step_fn(i)
        barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            out = step_fn(i)
        barrier()
        el = time.perf_counter() - t0
        if use_dp:
            tt = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = tt.item()
        return el, out

    if args.mode == 'train':
        step = make_train_step()
    else:
        model.eval()
        step = infer_step
    if args.host_batches:
        # PCIe-inclusive variant: the same step fed from pinned host memory
        host_ring = [v.cpu().pin_memory() for v in ring]
        inner = step
        if args.host_batches == 'serial':
            dev_buf = torch.empty_like(ring[0])

            def step(i):                                    # noqa: F811
                dev_buf.copy_(host_ring[i % len(host_ring)], non_blocking=True)
                ring[i % len(ring)] = dev_buf
                return inner(i)
        else:
            copy_stream = torch.cuda.Stream()
            bufs = [torch.empty_like(ring[0]) for _ in range(2)]
            events = [None, None]

            def upload(i):
                copy_stream.wait_stream(torch.cuda.current_stream())      # the buffer's last reader is queued
                with torch.cuda.stream(copy_stream):
                    bufs[i % 2].copy_(host_ring[i % len(host_ring)], non_blocking=True)
                    events[i % 2] = torch.cuda.Event()
                    events[i % 2].record(copy_stream)
            upload(0)

            def step(i):                                    # noqa: F811
                torch.cuda.current_stream().wait_event(events[i % 2])
                ring[i % len(ring)] = bufs[i % 2]
                upload(i + 1)       # waits for step i-1 (the other buffer's last reader), then runs beside step i
                return inner(i)
    elapsed, last = measure(step, args.steps, args.warmup)
    value = world * B * args.steps / elapsed
    prec_label = 'fp32 MFMA' if args.precision == 'fp32' else 'bf16 MFMA operands, fp32 accumulate and storage'

    result = {
        'metric': ('training' if args.mode == 'train' else 'inference') +
                  ' clips/sec (20x64x64 Moving MNIST) at 1/2/4/8 MI355X vs host CPU',
        'value': round(value, 2), 'unit': 'clips/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f32' if args.precision == 'fp32' else 'bf16', 'data': 'synthetic',
        'config': {'workload': f"{cfg['name']} ({args.config}) {args.mode} step, beta=4, {T}x{size}x{size} clips, "
                               f"batch {B}/GPU, {prec_label}, {('batches from pinned host memory (' + args.host_batches + '), ') if args.host_batches else ''}{('hipGraph replay' + (' (backward graph, one all-reduce, update graph)' if use_dp else '')) if use_graph else 'eager launches'}",
                   'global_batch': B * world, 'params': n_params,
                   'parallelism': f'dp{world}'},
    }
    if args.mode == 'train':
        result['loss'] = round(float(last['total_loss'].item()), 3)

    if rank == 0 and world == 1 and not args.no_kernel_table and args.precision == 'fp32':
        step = train_step if args.mode == 'train' else infer_step      # kernels are timed on eager launches
        step(0)                      # untimed: the eager path's buffers come out of the allocator's cache afterwards
        kt = KernelTimer(ops)
        kt.install()
        nprof = 5
        for i in range(nprof):
            step(i)
        rows = kt.summary(nprof)
        kt.remove()
        top = rows[0]
        traffic, traffic_src = pmc_traffic(top['kernel'])
        result['roofline'] = {'kernel': top['kernel'], 'bound': top['bound'], 'achieved': top['achieved'],
                              'peak': top['peak'], 'unit': top['unit'], 'frac': top['frac'], 'traffic': traffic,
                              'traffic_source': traffic_src,
                              'avg_launch_us': top['avg_launch_us'], 'launches_per_step': top['launches_per_step'],
                              'ms_per_step': top['ms_per_step']}
        result['kernels'] = [{k: r[k] for k in ('kernel', 'bound', 'achieved', 'unit', 'frac', 'ms_per_step')}
                             for r in rows[:12]]
    if args.mode == 'train' and not args.no_extras and world == 1:
        # the same job's inference rate (eval-mode forward, no_grad) and its bf16-MFMA variant
        model.eval()
        el, _ = measure(infer_step, args.steps, min(args.warmup, 3))
        result['inference'] = {'value': round(world * B * args.steps / el, 2), 'unit': 'clips/s',
                               'ms_per_step': round(el / args.steps * 1e3, 3),
                               'workload': f'eval-mode forward (no_grad), batch {B}/GPU, {prec_label}'}
        model.train()
        if args.precision == 'fp32':
            ops.set_precision('bf16')
            el, out = measure(make_train_step(), args.steps, args.warmup)
            result['bf16_variant'] = {'value': round(world * B * args.steps / el, 2), 'unit': 'clips/s',
                                      'ms_per_step': round(el / args.steps * 1e3, 3),
                                      'loss': round(float(out['total_loss'].item()), 3),
                                      'workload': 'same training step with bf16 MFMA operands (fp32 accumulate, '
                                                  'fp32 tensors); not the headline value'}
            ops.set_precision('fp32_split')
            el, out = measure(make_train_step(), args.steps, args.warmup)
            result['fp32_split_variant'] = {
                'value': round(world * B * args.steps / el, 2), 'unit': 'clips/s',
                'ms_per_step': round(el / args.steps * 1e3, 3), 'loss': round(float(out['total_loss'].item()), 3),
                'workload': 'same training step with the Linear-layer GEMMs forming each fp32 product from six '
                            'bf16 MFMA products of a three-term operand split (fp32-level error); not the headline '
                            'value'}
            ops.set_precision('fp32')
    if use_dp:
        dist.barrier()
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.mode == 'train':
        result['cpu_baseline'] = cpu_baseline(kind, cfg)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if use_dp:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
