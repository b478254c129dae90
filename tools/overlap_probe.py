"""Does a replayed hipGraph overlap with a pinned host-to-device copy on another stream?
Times a graph of matrix products alone, the copies alone, and both queued together.
usage: python tools/overlap_probe.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'disentangled-vae_amd'))
from hipvae import ops  # noqa: E402

x = torch.randn(4864, 8192, device='cuda')
w = torch.randn(512, 8192, device='cuda')
host = torch.empty(256, 20, 1, 64, 64).pin_memory()
dst = torch.empty_like(host, device='cuda')
copy_stream = torch.cuda.Stream()


def work():
    for _ in range(20):
        ops.gemm_nt(x, w, None, ops.ACT_NONE)


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    work()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    work()


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    copy_stream.synchronize()
    return a.elapsed_time(b) / reps


def copies(n=4):
    with torch.cuda.stream(copy_stream):
        for _ in range(n):
            dst.copy_(host, non_blocking=True)


def copies_timed():
    copy_stream.wait_stream(torch.cuda.current_stream())
    copies()
    torch.cuda.current_stream().wait_stream(copy_stream)


def both(launch):
    copy_stream.wait_stream(torch.cuda.current_stream())
    copies()
    launch()
    torch.cuda.current_stream().wait_stream(copy_stream)


print(f'graph alone        : {timed(g.replay):7.2f} ms')
print(f'eager alone        : {timed(work):7.2f} ms')
print(f'4 copies alone     : {timed(copies_timed):7.2f} ms')
print(f'graph + 4 copies   : {timed(lambda: both(g.replay)):7.2f} ms')
print(f'eager + 4 copies   : {timed(lambda: both(work)):7.2f} ms')
