"""Micro-benchmark of the Linear-layer GEMMs on the shapes of the 20x64x64 DisentangledVAE at batch 256,
one line per (kind, shape).

usage: python tools/bench_gemm.py [--reps 20] [--precision fp32|bf16|fp32_split] [--error]
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'disentangled-vae_amd'))
from hipvae import ops  # noqa: E402


def timeit(fn, reps, warm_s=0.3):
    # the chip needs a few hundred milliseconds of load to settle its clock: timings taken right after a pause read
    # up to 15 % slow (4864 x 8192 x 512 NT: 418 us after one warm-up launch, 361 us after two seconds of them)
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < warm_s:
        for _ in range(10):
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
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--precision', default='fp32')
    ap.add_argument('--layers', type=int, default=0, help='only the first N layer shapes (0: all)')
    ap.add_argument('--pad', type=int, default=0, help='operands are views of buffers whose rows are this many floats '
                    'longer (row strides that are not powers of two)')
    ap.add_argument('--error', action='store_true', help='also print the relative L2 error against an fp64 product')
    a = ap.parse_args()
    ops.set_precision(a.precision)
    dev = 'cuda'
    # (rows of the activation, in features, out features) of the Linear layers that matter
    layers = [(4864, 8192, 512), (5120, 640, 4096), (4864, 512, 1024), (4864, 256, 1024), (5120, 256, 512),
              (256, 4096, 512)]
    for M, K, N in (layers[:a.layers] if a.layers else layers):
        def rnd(r, c, scale=1.0):
            return (torch.randn(r, c + a.pad, device=dev) * scale)[:, :c]
        x = rnd(M, K)
        w = rnd(N, K, 0.05)
        b = torch.randn(N, device=dev)
        dy = rnd(M, N)
        fl = 2.0 * M * N * K
        for name, fn in (('nt  y = x w^T + b ', lambda: ops.gemm_nt(x, w, b, ops.ACT_LRELU)),
                         ('nn  dx = dy w     ', lambda: ops.gemm_nn(dy, w)),
                         ('tn  dw = dy^T x   ', lambda: ops.gemm_tn(dy, x))):
            t = timeit(fn, a.reps)
            err = ''
            if a.error:
                x64, w64, dy64 = x.double(), w.double(), dy.double()
                want = {'nt': lambda: torch.nn.functional.leaky_relu(x64 @ w64.t() + b.double(), 0.2),
                        'nn': lambda: dy64 @ w64, 'tn': lambda: dy64.t() @ x64}[name[:2]]()
                got = fn().double()
                err = f'  rel L2 err vs fp64 {float((got - want).norm() / want.norm()):.2e}'
            print(f'{name} M={M:5d} K={K:5d} N={N:5d}  {t:8.1f} us  {fl / t / 1e6:7.1f} TFLOP/s{err}', flush=True)


if __name__ == '__main__':
    main()
