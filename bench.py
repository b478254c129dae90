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
PEAK_BF16_MFMA_TFLOPS = 2500.0    # v_mfma_f32_32x32x16_bf16, dense
PEAK_HBM_GBS = 8000.0
# algorithmic GFLOP per clip (conv, transposed conv, Linear and LSTM products only; forward x 3 for a
# training step) of the three JSON configurations at 20 frames of 64 x 64.  The baseline model decodes one latent
# per clip into twenty identical frames: its decoder counts once per clip (0.286 encoder + 0.272 / 20 decoder),
# as it is executed, not twenty times as the reference runs it (0.558).
# (gflop_per_clip() derives them from a configuration's layer list: 1.524 / 4.571 for disentangled_simple.json, 2.397 /
# 7.191 for lstm_simple.json, 0.299 / 0.898 for small_baseline.json, 44.21 / 132.64 for the 128 x 128 x 40 one.)
PRECISION_LABEL = {'fp32': ('f32', 'fp32 MFMA'),
                   'bf16': ('bf16', 'bf16 MFMA operands, fp32 accumulate and storage'),
                   'fp32_split': ('f32', 'fp32 products from a three-term bf16 split of each operand '
                                         '(Linear layers, tiled convolutions), fp32 MFMA elsewhere')}


def gflop_per_clip(cfg):
    """(forward, training) algorithmic GFLOP per clip of a configuration: 2 FLOP per multiply-accumulate of every
    convolution (Ho*Wo*Cout*16*Cin), transposed convolution (4 sub-pixel phases of Hi*Wi*Cout*4*Cin), Linear layer and
    LSTM gate product, nothing else; training = 3 x forward (data and weight gradient of every product)."""
    m, d = cfg['model'], cfg['data']
    T, S = d['sequence_length'], d['frame_size']
    F, H, L = m['feature_dim'], m['lstm_hidden'], m['lstm_layers']
    D = 2 if m['bidirectional'] else 1

    def conv_stack(channels, cin=1):
        macs, s = 0, S
        for c in channels:
            s //= 2
            macs += s * s * c * 16 * cin
            cin = c
        return macs, cin * s * s

    def decoder_frame(width_in):
        ch = m['decoder_channels']
        macs, s = width_in * ch[0] * 16, 4
        for a, b in zip(ch[:-1], ch[1:]):
            s *= 2
            macs += s * s * b * 4 * a
        s *= 2
        return macs + s * s * 4 * ch[-1]

    enc_lstm = sum(D * 4 * H * ((F if l == 0 else D * H) + H) for l in range(L))      # per step
    dec_lstm = L * 4 * H * 2 * H                                                       # per step
    kind = m['model_type']
    if kind == 'disentangled':
        cd, md = m['content_dim'], m['motion_dim']
        content, flat = conv_stack(m['encoder_channels'])
        content += flat * F + 2 * F * cd
        motion, flat = conv_stack(m['motion_channels'])
        motion += flat * F
        macs = content + (T - 1) * (motion + enc_lstm) + 2 * D * H * md
        macs += 2 * (cd + md) * H * L + T * (dec_lstm + H * F + decoder_frame(F + cd))
    else:
        lat = m['latent_dim']
        frame, flat = conv_stack(m['encoder_channels'])
        frame += flat * F
        if kind == 'baseline':
            macs = T * (frame + 2 * F * lat) + decoder_frame(lat)        # one decoded frame per clip, as executed
        else:
            macs = T * (frame + enc_lstm) + 2 * D * H * lat
            macs += 2 * lat * H * L + T * (dec_lstm + H * F + decoder_frame(F))
    return round(2 * macs / 1e9, 3), round(6 * macs / 1e9, 3)


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
# per-kernel timing: the library's own launch trace (two events around every kernel launch, on the
# stream it is launched on), grouped by kernel symbol as a profiler prints it
# --------------------------------------------------------------------------------------------
def short_symbol(name):
    """a demangled kernel name without its parameter list and return type (the key of the tables here
    and of profiles/*pmc_traffic.json)"""
    import re
    return re.sub(r'\(.*\)$', '', name).replace('void ', '').strip()


def trace_steps(step_fn, nsteps, peak_tflops):
    """Runs nsteps eager steps under the library's launch trace.  One row per kernel symbol: time per step
    = MEDIAN over the steps of the symbol's per-step sum (every step issues the same launches in the same
    order, so the records split into nsteps equal runs); achieved = declared algorithmic work / time."""
    from hipvae import ops
    ops.trace_begin(1 << 16)
    for i in range(nsteps):
        step_fn(i)
    torch.cuda.synchronize()
    recs = ops.trace_collect()
    per = len(recs) // nsteps
    if per * nsteps != len(recs):
        raise RuntimeError(f"{len(recs)} traced launches do not split into {nsteps} equal steps")
    agg = {}
    for i, (name, ms, work, unit) in enumerate(recs):
        key = short_symbol(name)
        a = agg.setdefault(key, {'ms': [0.0] * nsteps, 'launches': 0, 'flop': 0.0, 'byte': 0.0})
        a['ms'][i // per] += ms
        a['launches'] += 1
        if unit == 1:
            a['flop'] += work
        elif unit == 2:
            a['byte'] += work
    rows = []
    for key, a in agg.items():
        ms = sorted(a['ms'])[nsteps // 2]
        launches = a['launches'] // nsteps
        row = {'kernel': key, 'ms_per_step': round(ms, 4), 'launches_per_step': launches,
               'avg_launch_us': round(ms * 1e3 / max(launches, 1), 2)}
        if a['flop'] > 0:
            ach = a['flop'] / nsteps / (ms / 1e3) / 1e12
            row.update(bound='mfma', achieved=round(ach, 2), peak=peak_tflops, unit='TFLOP/s',
                       frac=round(ach / peak_tflops, 4), work_per_launch=a['flop'] / a['launches'])
        elif a['byte'] > 0:
            ach = a['byte'] / nsteps / (ms / 1e3) / 1e9
            row.update(bound='hbm', achieved=round(ach, 2), peak=PEAK_HBM_GBS, unit='GB/s',
                       frac=round(ach / PEAK_HBM_GBS, 4), work_per_launch=a['byte'] / a['launches'])
        else:
            row.update(bound=None, achieved=None, peak=None, unit=None, frac=None, work_per_launch=0.0)
        rows.append(row)
    rows.sort(key=lambda r: -r['ms_per_step'])
    return rows


def pmc_traffic(label):
    """HBM bytes per launch of a kernel group from the newest committed counter summary
    (profiles/*pmc_traffic.json: separate rocprofv3 --pmc passes of this same command; counters cannot
    be collected inside a timed run).  (None, None) when there is no entry."""
    import glob
    files = sorted(f for f in glob.glob(os.path.join(ROOT, 'profiles', '*pmc_traffic.json')) if 'bf16' not in f)
    if not files:
        return None, None
    try:
        entry = json.load(open(files[-1]))['kernels'].get(label)
    except (OSError, ValueError, KeyError):
        return None, None
    if not entry:
        return None, None
    return entry['hbm_bytes_per_launch'], os.path.relpath(files[-1], ROOT)


def hbm_floor(ms_per_step, bf16):
    """The step against the HBM roofline: bytes one step moved through HBM (every kernel's FETCH_SIZE/WRITE_SIZE
    counters from the newest committed counter summary of this command, fp32 or bf16 variant) / 8 TB/s, as a fraction
    of the measured step.  None without such a summary."""
    import glob
    files = sorted(f for f in glob.glob(os.path.join(ROOT, 'profiles', '*pmc_traffic.json')) if ('bf16' in f) == bf16)
    for f in reversed(files):
        try:
            nbytes = json.load(open(f)).get('hbm_bytes_per_step')
        except (OSError, ValueError):
            continue
        if nbytes:
            floor_ms = nbytes / (PEAK_HBM_GBS * 1e9) * 1e3
            return {'bound': 'hbm', 'traffic': nbytes, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                    'achieved': round(nbytes / (ms_per_step * 1e-3) / 1e9, 1), 'floor_ms': round(floor_ms, 3),
                    'frac': round(floor_ms / ms_per_step, 4), 'traffic_source': os.path.relpath(f, ROOT)}
    return None


# --------------------------------------------------------------------------------------------
def synthetic_digit_bank(n=512, seed=1234):
    """[n,28,28] float32 in [0,1]: blurred random strokes with MNIST's share of lit pixels (there is no MNIST on
    the box and no network; the generator's cost does not depend on what the digits show)."""
    import numpy as np
    rs = np.random.RandomState(seed)
    bank = np.zeros((n, 28, 28), dtype=np.float32)
    for i in range(n):
        img = np.zeros((28, 28), dtype=np.float32)
        p = rs.uniform(6, 22, size=2)
        for _ in range(40):
            p = np.clip(p + rs.uniform(-2.5, 2.5, size=2), 4, 24)
            y, x = int(p[0]), int(p[1])
            img[y - 1:y + 2, x - 1:x + 2] = np.maximum(img[y - 1:y + 2, x - 1:x + 2], rs.uniform(0.5, 1.0))
        bank[i] = img
    return (np.round(bank * 255) / 255).astype(np.float32)


def trainer_epoch(config_name, batch, precision='fp32', train_size=10000, val_size=1000, epochs=2, use_graph=True,
                  host_render=False):
    """What `python -m training.train --config <json> --device cuda` delivers: Trainer.train_epoch() and
    Trainer.validate() with the Trainer's defaults (steps and validation batches replayed from hipGraphs, clips
    rendered on the GPU from host-simulated trajectories, losses accumulated on the device) over train_size /
    val_size generated clips, beta = 4.  Epoch 0 pays for the captures; the last epoch is the rate reported.
    Wall-clock around each call, device synchronised on both sides."""
    import tempfile
    from data import get_moving_mnist_dataloaders
    from hipvae import ops
    from training.config import ExperimentConfig
    from training.trainer import create_trainer
    ops.set_precision(precision)
    cfg = ExperimentConfig.load(os.path.join(ROOT, 'disentangled-vae_amd', 'configs', config_name))
    tmp = tempfile.mkdtemp(prefix='dvae_bench_')
    cfg.data.train_size, cfg.data.val_size, cfg.data.test_size, cfg.data.batch_size = train_size, val_size, 8, batch
    t = cfg.training
    t.device, t.checkpoint_dir, t.results_dir = 'cuda', os.path.join(tmp, 'ck'), os.path.join(tmp, 'res')
    t.beta_vae, t.use_kl_annealing, t.kl_start_epoch = 4.0, False, 0
    if cfg.model.model_type == 'disentangled':
        t.beta_content = t.beta_motion = 4.0
    dev = torch.device('cuda', torch.cuda.current_device())
    torch.manual_seed(0)
    train_loader, val_loader, _ = get_moving_mnist_dataloaders(
        digit_bank=synthetic_digit_bank(), batch_size=batch, train_size=train_size, val_size=val_size, test_size=8,
        sequence_length=cfg.data.sequence_length, frame_size=cfg.data.frame_size, num_digits=cfg.data.num_digits,
        num_workers=cfg.data.num_workers, seed=cfg.data.seed, pin_memory=host_render,
        render_device=None if host_render else dev)
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):                # the Trainer's own prints
        tr = create_trainer(cfg, train_loader, val_loader)
    tr.use_graph = use_graph
    rows = []
    for epoch in range(epochs):
        tr.current_epoch = epoch
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = tr.train_epoch()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        val = tr.validate()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        rows.append((t1 - t0, t2 - t1, out['total_loss'], val['total_loss']))
    tr.wait_for_checkpoint()
    if getattr(tr, 'writer', None) is not None:
        tr.writer.close()
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    ops.set_precision('fp32')
    tt, tv, loss, vloss = rows[-1]
    return {'batch': batch, 'train_clips': train_size, 'val_clips': val_size,
            'train_epoch_s': round(tt, 4), 'train_clips_per_s': round(train_size / tt, 1),
            'validate_s': round(tv, 4), 'validate_clips_per_s': round(val_size / tv, 1),
            'first_epoch_train_s': round(rows[0][0], 4), 'first_epoch_validate_s': round(rows[0][1], 4),
            'steps_per_epoch': len(train_loader), 'train_loss': round(loss, 3), 'val_loss': round(vloss, 3),
            'graph_fallback': tr.graph_fallback_reason,
            'path': ('hipGraph replay' if use_graph else 'eager launches') + ', ' +
                    ('clips rendered by DataLoader workers, uploaded one batch ahead' if host_render
                     else 'clips rendered on the GPU')}


def cpu_baseline(kind, cfg, steps=10, warmup=3, batch=32):
    """The CPU oracle (oracle/vae_oracle.py, a restatement of the reference pinned to the reference's own
    vectors) timed on the host cores: training step = forward + loss + backward + clip + Adam; inference =
    eval-mode forward.  3 warm-up + 10 timed steps each, median."""
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
    for i in range(warmup + steps):
        t0 = time.perf_counter()
        sd, opt, _, _ = O.train_step(kind, sd, opt, video, eps, 4.0, t, t['learning_rate'], lstm_layers=layers,
                                     bidirectional=kw.get('bidirectional', True))
        times.append(time.perf_counter() - t0)
    med = sorted(times[warmup:])[steps // 2]
    itimes = []
    with torch.no_grad():
        for i in range(warmup + steps):
            t0 = time.perf_counter()
            O.model_forward(kind, sd, video, [None, None], False, lstm_layers=layers)
            itimes.append(time.perf_counter() - t0)
    imed = sorted(itimes[warmup:])[steps // 2]
    return {'value': round(batch / med, 2), 'unit': 'clips/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': f'{steps} training steps (after {warmup} warm-up) of the CPU oracle, batch {batch}, fp32, '
                      f'median {med * 1e3:.0f} ms/step; inference leg: {steps} eval-mode forwards, median '
                      f'{imed * 1e3:.0f} ms',
            'inference': {'value': round(batch / imed, 2), 'unit': 'clips/s'}}


def other_sections(steps, warmup, batch, config_name='disentangled_simple.json'):
    """The measurements beside the headline that need models of their own (`bench.py --section others`): the three
    batch-32 configurations, LSTMVAE at `batch`, the 128 x 128 x 40 configuration at batch 32 (hipGraph replay on a ring
    of four synthetic batches, like the headline) and one epoch of the Trainer at batch 32 and `batch`."""
    import gc
    import torch
    dev = torch.device('cuda', torch.cuda.current_device())
    import models
    from hipvae import losses as L
    from hipvae.graph import GraphedTrainStep
    from hipvae.optim import FusedAdam
    beta = 4.0

    def release(what):
        # graphs, models and rings of a finished section go back to the device before the next one captures
        gc.collect()
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        print(f'[bench] after {what}: {torch.cuda.memory_reserved() / 2 ** 30:.1f} GiB reserved', file=sys.stderr, flush=True)

    def timed(step_fn):
        for i in range(warmup):
            step_fn(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step_fn(i)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    def graphed_rate(cfg_name, b):
        c2 = load_config(cfg_name)
        k2, t2 = c2['model']['model_type'], c2['training']
        T2, S2 = c2['data']['sequence_length'], c2['data']['frame_size']
        torch.manual_seed(0)
        m2 = models.get_model(k2, **model_kwargs(c2)).to(dev).train()
        o2 = FusedAdam(m2.parameters(), lr=t2['learning_rate'], betas=(t2['beta1'], t2['beta2']),
                       weight_decay=t2['weight_decay'])
        r2 = [synthetic_clips(b, T2, S2, 7000 + i).to(dev) for i in range(4)]
        g2 = GraphedTrainStep(m2, o2, lambda mm, v: loss_call(k2, L, mm(v), v, t2, beta), r2[0],
                              gradient_clip=t2['gradient_clip'])
        ms2 = timed(lambda i: g2(r2[i % 4])) / steps * 1e3
        gf2 = gflop_per_clip(c2)[1]
        del g2, m2, o2, r2
        release(f'{cfg_name} at batch {b}')
        return {'batch': b, 'clip': f'{T2}x{S2}x{S2}', 'ms_per_step': round(ms2, 3),
                'value': round(b / ms2 * 1e3, 1), 'unit': 'clips/s', 'gflop_per_clip': gf2,
                'whole_step_frac': round(gf2 * b / ms2 / PEAK_FP32_MFMA_TFLOPS, 4)}

    out = {'batch32_configs': {n: graphed_rate(n, 32) for n in ('disentangled_simple.json', 'lstm_simple.json',
                                                                'small_baseline.json')}}
    out['other_configs'] = {'lstm_simple.json': graphed_rate('lstm_simple.json', batch),
                            'disentangled_128x128x40.json': graphed_rate('disentangled_128x128x40.json', 32)}
    try:
        out['trainer_epoch'] = {f'batch{b}': trainer_epoch(config_name, b) for b in sorted({32, batch})}
    except Exception as e:                                          # noqa: BLE001
        out['trainer_epoch'] = {'error': f'{type(e).__name__}: {e}'[:300]}
    return out


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
    ap.add_argument('--kernel-rows', type=int, default=16, help='rows of the per-kernel table in the result line')
    ap.add_argument('--bucket-mb', type=float, default=16.0)
    ap.add_argument('--precision', default='fp32', choices=['fp32', 'bf16', 'fp32_split'],
                    help='operand precision of the MFMA kernels for the headline number')
    ap.add_argument('--no-extras', action='store_true', help='skip the inference and bf16-variant measurements')
    ap.add_argument('--host-batches', default='', choices=['', 'serial', 'prefetch'],
                    help='feed every step from pinned host memory over PCIe: copied on the step stream (serial) or one '
                         'batch ahead on a side stream (prefetch); the headline value keeps batches resident in HBM')
    ap.add_argument('--force-dist', action='store_true',
                    help='initialise the RCCL process group and use the data-parallel step even with one rank '
                         '(exercises the multi-GPU code path on a single GPU)')
    ap.add_argument('--section', default='', choices=['', 'others'],
                    help="'others': only the measurements beside the headline")
    ap.add_argument('--trainer', action='store_true',
                    help='measure Trainer.train_epoch() + validate() (what the CLI runs: 10,000 / 1,000 generated clips) '
                         'at batch 32 and at --batch instead of the device-resident ring, and print that as the line')
    ap.add_argument('--graph', type=int, default=-1, help='0: eager launches (several ranks: bucketed all-reduces '
                    'overlapped with backward); otherwise replay the captured step (several ranks: two hipGraphs '
                    'around one gradient all-reduce)')
    args = ap.parse_args()
    import faulthandler
    faulthandler.enable()            # a crash inside native code leaves the Python stacks of all threads on stderr
    # this process captures a dozen steps one after the other: none of their graphs is destroyed before it ends
    # (see hipvae.graph.keep_graphs_alive; DVAE_BENCH_DROP_GRAPHS=1 lets them go as their sections end)
    if os.environ.get('DVAE_BENCH_DROP_GRAPHS') != '1':
        from hipvae.graph import keep_graphs_alive
        keep_graphs_alive()

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
    saved_stdout = None
    if use_dp:
        # RCCL prints a version banner on stdout when it creates its communicator: everything the libraries write
        # goes to stderr, stdout carries the one result line
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    import models
    from hipvae import losses as L, ops
    from hipvae.optim import FusedAdam
    from hipvae.parallel import GradSync

    ops.set_precision(args.precision)
    cfg = load_config(args.config)
    if args.section == 'others':
        if world != 1:
            raise SystemExit("--section others is a single-GPU measurement")
        print(json.dumps(other_sections(args.steps, args.warmup, args.batch, args.config)), flush=True)
        return
    if args.trainer:
        if world != 1:
            raise SystemExit("--trainer measures the single-GPU command line")
        rows = {f'batch{b}': trainer_epoch(args.config, b, args.precision, use_graph=args.graph != 0)
                for b in sorted({32, args.batch})}
        head = rows[f'batch{args.batch}']
        print(json.dumps({
            'metric': 'training clips/sec (20x64x64 Moving MNIST) through Trainer.train_epoch()', 'unit': 'clips/s',
            'value': head['train_clips_per_s'], 'n_gpus': 1, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': PRECISION_LABEL[args.precision][0], 'data': 'synthetic',
            'steps': head['steps_per_epoch'], 'warmup': head['steps_per_epoch'],
            'ms_per_step': round(head['train_epoch_s'] / head['steps_per_epoch'] * 1e3, 3),
            'config': {'workload': f"{cfg['name']} ({args.config}): one epoch of the Trainer, beta=4, {head['path']}"},
            'trainer_epoch': rows}), flush=True)
        return
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
    graph_fallback = []          # why the captured data-parallel step was given up, if it was
    dp_runs_mb = []              # sizes of the gradient runs of the captured data-parallel step, one per backward stage

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
        nonlocal sync, use_graph
        model.train()
        if not use_graph:
            return train_step
        from hipvae.graph import GraphedDataParallelStep, GraphedTrainStep
        cls = GraphedDataParallelStep if use_dp else GraphedTrainStep
        loss_of = lambda m, v: loss_call(kind, L, m(v), v, t, beta)
        if not use_dp:
            gstep = cls(model, opt, loss_of, ring[0], gradient_clip=t['gradient_clip'])
            return lambda i: gstep(ring[i % len(ring)])
        # Several ranks.  Ranks may only change path together, and only at a point where all of them stand between
        # the same two collectives:
        #  1. pre-flight: one eager forward + backward per rank, no collective anywhere in it.  A rank that cannot
        #     run the model at all (out of memory, a kernel refusing a shape) shows here; the flag is reduced with
        #     MIN and every rank stops with the same message.
        #  2. the constructor's warm-up steps issue collectives; a rank that fails BETWEEN two of them leaves its
        #     peers inside one, and no flag can be exchanged any more: that rank re-raises, the job ends (the
        #     launcher takes the peers down), nothing falls back.
        #  3. capture issues no collective.  Should it fail on any rank (the collectives' own threads share the device
        #     with the capturing thread), every rank has finished the same warm-up collectives, the flag exchange
        #     below is the next collective on all of them, and all take the eager data-parallel step instead - the
        #     same kernels launched one by one, bucketed all-reduces overlapped with backward; the result line says so.
        # DVAE_BENCH_FAIL_CAPTURE rehearses these: '1' (before the constructor), 'capture[:rank]', 'warmup[:rank]'.
        knob, _, knob_rank = os.environ.get('DVAE_BENCH_FAIL_CAPTURE', '').partition(':')
        knob_here = knob != '' and (knob_rank == '' or int(knob_rank) == rank)

        def agreed(flag_ok):
            ok = torch.tensor([1 if flag_ok else 0], device=dev, dtype=torch.int32)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            return int(ok.item()) == 1

        why = ''
        try:
            opt.zero_grad(set_to_none=True)
            loss_of(model, ring[0])['total_loss'].backward()
            opt.zero_grad(set_to_none=True)
            torch.cuda.synchronize()
        except Exception as e:                                          # noqa: BLE001
            why = f'{type(e).__name__}: {e}'
        if not agreed(why == ''):
            raise RuntimeError(f'rank {rank}: the training step does not run on every rank ({why or "failed on another rank"})')

        class Rehearsed(cls):
            def _capture(self):
                if knob == 'capture' and knob_here:
                    raise RuntimeError('DVAE_BENCH_FAIL_CAPTURE=capture')
                super()._capture()

            def _step(self):
                losses = super()._step()
                if knob == 'warmup' and knob_here:
                    raise RuntimeError('DVAE_BENCH_FAIL_CAPTURE=warmup')
                return losses

        gstep = None
        try:
            if knob == '1' and knob_here:
                raise RuntimeError('DVAE_BENCH_FAIL_CAPTURE=1')
            gstep = (Rehearsed if knob in ('capture', 'warmup') else cls)(model, opt, loss_of, ring[0],
                                                                         gradient_clip=t['gradient_clip'])
        except Exception as e:                                          # noqa: BLE001
            why = f'{type(e).__name__}: {e}'
            if not (knob == '1' and knob_here) and getattr(e, 'dvae_phase', 'warmup') != 'capture':
                print(f'[bench] rank {rank}: the data-parallel step failed between its warm-up collectives ({why}); '
                      f'the other ranks cannot be told: stopping', file=sys.stderr, flush=True)
                raise
            print(f'[bench] rank {rank}: captured data-parallel step failed ({why}); eager step instead',
                  file=sys.stderr, flush=True)
        if agreed(gstep is not None):
            dp_runs_mb[:] = [round((hi - lo) * 4 / 2 ** 20, 2) for lo, hi in gstep.staged.runs]
            return lambda i: gstep(ring[i % len(ring)])
        del gstep
        torch.cuda.synchronize()
        use_graph = False
        graph_fallback.append(why or 'capture failed on another rank')
        sync = GradSync(model, bucket_mb=args.bucket_mb)
        return train_step

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

    def make_infer_step():
        model.eval()
        if args.graph == 0:
            return infer_step
        from hipvae.graph import GraphedForward
        gfwd = GraphedForward(model, ring[0])
        return lambda i: gfwd(ring[i % len(ring)])

    if args.mode == 'train':
        step = make_train_step()
    else:
        step = make_infer_step()
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
    ops.check_lstm_seq_timeouts()          # a one-launch LSTM kernel that was not fully resident would void the run
    value = world * B * args.steps / elapsed
    dtype_label, prec_label = PRECISION_LABEL[args.precision]
    peak_tflops = PEAK_BF16_MFMA_TFLOPS if args.precision == 'bf16' else PEAK_FP32_MFMA_TFLOPS

    result = {
        'metric': ('training' if args.mode == 'train' else 'inference') +
                  ' clips/sec (20x64x64 Moving MNIST) at 1/2/4/8 MI355X vs host CPU',
        'value': round(value, 2), 'unit': 'clips/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': dtype_label, 'data': 'synthetic',
        'config': {'workload': f"{cfg['name']} ({args.config}) {args.mode} step, beta=4, {T}x{size}x{size} clips, "
                               f"batch {B}/GPU, {prec_label}, {('batches from pinned host memory (' + args.host_batches + '), ') if args.host_batches else ''}{('hipGraph replay' + (' (one graph per backward stage -- decoder side, encoders down to the convolution stack of the per-frame encoder, that stack -- the gradients of each stage all-reduced beside the next stage, update graph)' if use_dp else '')) if use_graph else 'eager launches'}",
                   'global_batch': B * world, 'params': n_params,
                   'parallelism': f'dp{world}'},
    }
    if args.mode == 'train':
        result['loss'] = round(float(last['total_loss'].item()), 3)

    if use_dp:
        result['rccl_world_size'] = dist.get_world_size()
        if graph_fallback:
            result['graph_fallback'] = graph_fallback[0][:300]
        result['config']['collective'] = (
            f'{len(dp_runs_mb)} all-reduces (AVG) per step over RCCL on a side stream, one per backward stage '
            f'({" / ".join(str(x) for x in dp_runs_mb)} MiB): each travels beside the stage that follows it, only the '
            f'last one is reduced after backward has ended'
            if use_graph else 'bucketed all-reduces (AVG) over RCCL, overlapped with backward')
    gf = gflop_per_clip(cfg)
    whole = None
    if gf is not None:
        tfl = gf[1 if args.mode == 'train' else 0] * B * world / (elapsed / args.steps) / 1e3
        whole = {'achieved': round(tfl, 2), 'peak': peak_tflops * world, 'unit': 'TFLOP/s',
                 'frac': round(tfl / (peak_tflops * world), 4),
                 'gflop_per_clip': gf[1 if args.mode == 'train' else 0]}
        if args.mode == 'train' and kind == 'disentangled' and B == 256 and args.precision in ('fp32', 'bf16'):
            # the same step against the other roof (counter bytes of the committed profile of this configuration)
            whole['hbm'] = hbm_floor(elapsed / args.steps * 1e3, bf16=args.precision == 'bf16')
    if rank == 0 and world == 1 and not args.no_kernel_table:
        step = train_step if args.mode == 'train' else infer_step      # kernels are timed on eager launches
        # ... of one stream: with the weight gradients on their second stream a launch's events would also
        # time the kernels it shares the chip with
        from hipvae import functional as _fn
        off_path, _fn.USE_OFF_PATH = _fn.USE_OFF_PATH, False
        # ... and on the stream the captured step was warmed up on: autograd keeps each parameter's accumulation node
        # with the stream it first ran on, and warns when a gradient arrives from another one
        from hipvae.graph import capture_stream
        side = capture_stream(dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step(0)                  # untimed: the eager path's buffers come out of the allocator's cache afterwards
            rows = trace_steps(step, 5, peak_tflops)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        _fn.USE_OFF_PATH = off_path
        top = next((r for r in rows if r['bound'] is not None), rows[0])
        traffic, traffic_src = pmc_traffic(top['kernel'])
        result['roofline'] = {'kernel': top['kernel'], 'bound': top['bound'], 'achieved': top['achieved'],
                              'peak': top['peak'], 'unit': top['unit'], 'frac': top['frac'], 'traffic': traffic,
                              'traffic_source': traffic_src, 'algorithmic_work_per_launch': top['work_per_launch'],
                              'avg_launch_us': top['avg_launch_us'], 'launches_per_step': top['launches_per_step'],
                              'ms_per_step': top['ms_per_step'], 'whole_step': whole}
        traced_ms = sum(r['ms_per_step'] for r in rows)
        result['kernels'] = [{k: r[k] for k in ('kernel', 'bound', 'achieved', 'unit', 'frac', 'ms_per_step',
                                                'launches_per_step', 'avg_launch_us')} for r in rows[:args.kernel_rows]]
        result['kernels_note'] = (f'{len(rows)} kernel symbols of this library, {round(traced_ms, 3)} ms per eager step '
                                  f'in all; the first {min(args.kernel_rows, len(rows))} listed')
    elif whole is not None:
        result['roofline'] = {'whole_step': whole}
    if args.mode == 'train' and not args.no_extras and world == 1:
        # the same job's inference rate (eval-mode forward, no_grad) and its bf16-MFMA variant
        el, _ = measure(make_infer_step(), args.steps, min(args.warmup, 3))
        itfl = gf[0] * B * world / (el / args.steps) / 1e3
        result['inference'] = {'value': round(world * B * args.steps / el, 2), 'unit': 'clips/s',
                               'ms_per_step': round(el / args.steps * 1e3, 3),
                               'roofline': {'bound': 'mfma', 'achieved': round(itfl, 2), 'peak': peak_tflops * world,
                                            'unit': 'TFLOP/s', 'frac': round(itfl / (peak_tflops * world), 4),
                                            'gflop_per_clip': gf[0]},
                               'workload': f'eval-mode forward (no_grad), batch {B}/GPU, {prec_label}'
                                           + (', hipGraph replay' if args.graph != 0 else '')}
        model.train()
        if args.precision == 'fp32':
            ops.set_precision('bf16')
            el, out = measure(make_train_step(), args.steps, args.warmup)
            result['bf16_variant'] = {'value': round(world * B * args.steps / el, 2), 'unit': 'clips/s',
                                      'ms_per_step': round(el / args.steps * 1e3, 3),
                                      'loss': round(float(out['total_loss'].item()), 3),
                                      'workload': 'same training step with bf16 MFMA operands (fp32 accumulate, '
                                                  'fp32 tensors); not the headline value'}
            if kind == 'disentangled' and B == 256 and size == 64 and T == 20:
                # the bf16 step is bound by its bytes, not by the matrix pipe: priced against HBM
                result['bf16_variant']['roofline'] = hbm_floor(result['bf16_variant']['ms_per_step'], bf16=True)
            ops.set_precision('fp32_split')
            el, out = measure(make_train_step(), args.steps, args.warmup)
            result['fp32_split_variant'] = {
                'value': round(world * B * args.steps / el, 2), 'unit': 'clips/s',
                'ms_per_step': round(el / args.steps * 1e3, 3), 'loss': round(float(out['total_loss'].item()), 3),
                'workload': 'same training step with the Linear-layer GEMMs and the tiled convolutions forming each fp32 '
                            'product from six bf16 MFMA products of a three-term operand split (fp32-level error); '
                            'not the headline value'}
            ops.set_precision('fp32')
        if args.config == 'disentangled_simple.json' and args.precision == 'fp32':
            # the reference's own batch-32 configurations (DisentangledVAE, LSTMVAE, fp32) and the small baseline beside
            # them, LSTMVAE at the headline's batch, the 128 x 128 x 40 configuration at batch 32, and the
            # Trainer's own epoch loop.  In this process: starting a child process from one that has initialised the GPU
            # was seen to end in a segmentation fault inside the fork now and then (`--section others` runs these
            # measurements alone).
            try:
                extra = other_sections(args.steps, args.warmup, B, args.config)
            except Exception as e:                                          # noqa: BLE001
                extra = {'other_sections_error': f'{type(e).__name__}: {e}'[:300]}
            result.update(extra)
            if 'trainer_epoch' in extra and 'error' not in extra['trainer_epoch'] and 'batch32_configs' in extra:
                ring_rate = {B: result['value'], 32: extra['batch32_configs']['disentangled_simple.json']['value']}
                for b in sorted({32, B}):
                    row = extra['trainer_epoch'].get(f'batch{b}')
                    if row:
                        row['vs_ring'] = round(row['train_clips_per_s'] / ring_rate[b], 4)
    if use_dp:
        dist.barrier()
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.mode == 'train':
        result['cpu_baseline'] = cpu_baseline(kind, cfg)
    if saved_stdout is not None:
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if use_dp:
        os.dup2(2, 1)                         # (the teardown may print again)
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
