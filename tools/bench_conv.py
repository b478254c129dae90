"""Micro-benchmark of the convolution-family kernels on the decoder / encoder layer shapes of the
20x64x64 DisentangledVAE at batch 256 (5120 decoder frames), one line per (kernel, layer, variant).

usage: python tools/bench_conv.py [--frames 5120] [--reps 10] [--only deconv|conv|wgrad]
"""
import argparse
import time
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'disentangled-vae_amd'))
from hipvae import ops  # noqa: E402


def timeit(fn, reps, warm_s=0.2):
    # launches for a fifth of a second first: right after a pause the chip's clock is still low and the first
    # timings read up to 15 % slow
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < warm_s:
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True)
    e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=5120)
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--only', default='')
    ap.add_argument('--precision', default='fp32')
    a = ap.parse_args()
    N = a.frames
    ops.set_precision(a.precision)
    dev = 'cuda'
    # (Cin, Cout, H_in of the transposed conv)
    layers = [(256, 128, 4), (128, 64, 8), (64, 32, 16), (32, 16, 16)]     # last: the motion encoder's second layer
    rows = []
    for Cin, Cout, H in layers:
        x = torch.randn(N, H, H, Cin, device=dev)
        sc = torch.rand(Cin, device=dev) + 0.5
        sh = torch.randn(Cin, device=dev) * 0.1
        w = torch.randn(Cin, Cout, 4, 4, device=dev) * 0.05
        wpd = ops.pack_deconv_w(w)
        flops = 2.0 * N * H * H * 4 * 4 * Cin * Cout
        dy = torch.randn(N, 2 * H, 2 * H, Cout, device=dev)
        osc = torch.rand(Cout, device=dev) + 0.5
        osh = torch.randn(Cout, device=dev) * 0.1
        wpc = ops.pack_conv_w(w)          # dgrad of the transposed conv: conv with [Cin][Cout] roles
        if a.only in ('', 'deconv'):
            for name, s1, s2, act, st in (('plain', None, None, 0, False), ('bn+lrelu', sc, sh, 1, False),
                                          ('bn+lrelu+stats', sc, sh, 1, True)):
                t = timeit(lambda: ops.deconv4s2_fwd(x, s1, s2, act, wpd, st), a.reps)
                rows.append((f'deconv fwd {Cin}->{Cout} H{H} {name}', t, flops / t / 1e6))
        if a.only in ('', 'conv'):
            for name, st in (('plain', False), ('stats', True)):
                t = timeit(lambda: ops.conv4s2_fwd(dy, None, None, 0, wpc, st), a.reps)
                rows.append((f'conv (dgrad) {Cout}->{Cin} H{2 * H} {name}', t, flops / t / 1e6))
        if a.only in ('', 'conv', 'bnstat'):
            ybn = torch.randn(N, H, H, Cin, device=dev)
            aff = tuple(torch.rand(Cin, device=dev) + 0.5 for _ in range(4))
            t = timeit(lambda: ops.conv4s2_dgrad_bnstat(dy, wpc, ybn, aff), a.reps)
            rows.append((f'conv (dgrad) {Cout}->{Cin} H{2 * H} +bn-bwd stats', t, flops / t / 1e6))
        if a.only in ('', 'wgrad'):
            for name, s1, s2, act in (('plain', None, None, 0), ('bn+lrelu on P', sc, sh, 1)):
                t = timeit(lambda: ops.conv4s2_wgrad(x, s1, s2, act, dy, None, None, 0), a.reps)
                rows.append((f'wgrad P{Cin} Q{Cout} Hs{H} {name}', t, flops / t / 1e6))
    if a.only in ('', 'bnstat', 'c1'):
        # data gradient of the final layer: single-channel input, Cout = 32, write bound
        dl = torch.randn(N, 64, 64, 1, device=dev)
        wf = torch.randn(32, 4, 4, 1, device=dev) * 0.1
        ybn = torch.randn(N, 32, 32, 32, device=dev)
        aff = tuple(torch.rand(32, device=dev) + 0.5 for _ in range(4))
        fl = 2.0 * N * 32 * 32 * 16 * 32
        t = timeit(lambda: ops.conv4s2_fwd(dl, None, None, 0, wf, False), a.reps)
        rows.append(('conv c1 (final dgrad) 1->32 H64 plain', t, fl / t / 1e6))
        t = timeit(lambda: ops.conv4s2_dgrad_bnstat(dl, wf, ybn, aff), a.reps)
        rows.append(('conv c1 (final dgrad) 1->32 H64 +bn-bwd stats', t, fl / t / 1e6))
    for name, t, tf in rows:
        print(f'{name:48s} {t:9.1f} us {tf:7.1f} TFLOP/s  ({tf / 157.3:.3f} of peak)', flush=True)


if __name__ == '__main__':
    main()
