"""Cost per step of a chain of dependent LSTM cell launches replayed from a hipGraph (the way the
training step runs them), forward and backward, at the DisentangledVAE's sizes (B = 256, H = 256).
usage: python tools/bench_lstm.py [--steps 100]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'disentangled-vae_amd'))
from hipvae import ops  # noqa: E402


def replay_time(fn, reps=20):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=100)
    a = ap.parse_args()
    B, H, T = 256, 256, a.steps
    dev = 'cuda'
    xproj = torch.randn(T, B, 4 * H, device=dev) * 0.1
    w_hh = torch.randn(4 * H, H, device=dev) * 0.05
    bi, bh = torch.zeros(4 * H, device=dev), torch.zeros(4 * H, device=dev)
    hseq, cseq = torch.zeros(T, B, H, device=dev), torch.zeros(T, B, H, device=dev)
    gates = torch.zeros(T, B, 4 * H, device=dev)

    def fwd():
        for t in range(T):
            ops.lstm_step_fwd(xproj[t], None, None, None if t == 0 else hseq[t - 1], None if t == 0 else cseq[t - 1],
                              w_hh, bi, bh, gates[t], hseq[t], cseq[t])
    us = replay_time(fwd)
    print(f'forward : {us / T:6.2f} us per step ({T} dependent steps in one graph)', flush=True)
    for Bs in (32, 64, 128, 256):
        if not ops.lstm_seq_supported(1, T, Bs, H):
            continue
        spec = dict(xproj=xproj[:, :Bs].contiguous(), w_rec=w_hh, b_a=bi, b_b=bh, gates=gates[:, :Bs].contiguous(),
                    hseq=hseq[:, :Bs].contiguous(), cseq=cseq[:, :Bs].contiguous())
        us = replay_time(lambda: ops.lstm_seq_fwd([spec], T, Bs, H))
        print(f'forward, whole recurrence in one launch, B={Bs:3d}: {us / T:6.2f} us per step', flush=True)
        if ops.lstm_seq_supported(2, T, Bs, H):
            two = [spec, dict(spec, gates=torch.empty_like(spec['gates']), hseq=torch.empty_like(spec['hseq']),
                              cseq=torch.empty_like(spec['cseq']), reverse=True)]
            us = replay_time(lambda: ops.lstm_seq_fwd(two, T, Bs, H))
            print(f'   two chains (both directions) in one launch, B={Bs:3d}: {us / T:6.2f} us per step', flush=True)

    w_t = w_hh.t().contiguous()
    dh = torch.randn(T, B, H, device=dev) * 0.01
    dgs = torch.zeros(T, B, 4 * H, device=dev)
    dc = [torch.zeros(B, H, device=dev), torch.zeros(B, H, device=dev)]

    def bwd():
        ops.lstm_step_bwd_pointwise(dh[T - 1], None, gates[T - 1], cseq[T - 2], cseq[T - 1], dgs[T - 1], dc[(T - 1) & 1])
        for i in range(T - 1, 0, -1):
            ops.lstm_bwd_step(dgs[i], w_t, dh[i - 1], dc[i & 1], gates[i - 1], None if i == 1 else cseq[i - 2],
                              cseq[i - 1], dgs[i - 1], dc[(i - 1) & 1])
    us = replay_time(bwd)
    print(f'backward: {us / T:6.2f} us per step', flush=True)

    for Bs in (32, 64, 128, 256):
        if not ops.lstm_seq_supported(1, T, Bs, H):
            continue
        spec = dict(w_rec=w_hh, gates=gates[:, :Bs].contiguous(), cseq=cseq[:, :Bs].contiguous(),
                    dh_ext=dh[:, :Bs].contiguous(), dgates=dgs[:, :Bs].contiguous())
        us = replay_time(lambda: ops.lstm_seq_bwd([spec], T, Bs, H))
        print(f'backward, whole recurrence in one launch, B={Bs:3d}: {us / T:6.2f} us per step', flush=True)
        if ops.lstm_seq_supported(2, T, Bs, H):
            two = [spec, dict(spec, dgates=torch.empty_like(spec['dgates']), reverse=True)]
            us = replay_time(lambda: ops.lstm_seq_bwd(two, T, Bs, H))
            print(f'   two chains (both directions) in one launch, B={Bs:3d}: {us / T:6.2f} us per step', flush=True)

    # the stacked (two-layer) autoregressive decoder: one launch each way, 2 T cells in a chain
    Td = 20
    for Bs in (32, 256):
        if not ops.lstm_dec2_supported(Td, Bs, H):
            continue
        wts = [torch.randn(4 * H, H, device=dev) * 0.05, torch.randn(4 * H, H, device=dev) * 0.05,
               torch.zeros(4 * H, device=dev), torch.zeros(4 * H, device=dev)] * 2
        h0, c0 = torch.randn(Bs, 2 * H, device=dev) * 0.1, torch.randn(Bs, 2 * H, device=dev) * 0.1
        init = torch.randn(H, device=dev) * 0.1
        gates2 = torch.empty(2, Td, Bs, 4 * H, device=dev)
        hs2, cs2 = torch.empty(2, Td, Bs, H, device=dev), torch.empty(2, Td, Bs, H, device=dev)
        us = replay_time(lambda: ops.lstm_dec2_fwd(wts, h0, c0, init, gates2, hs2, cs2))
        print(f'stacked decoder forward,  one launch, B={Bs:3d}: {us / (2 * Td):6.2f} us per cell', flush=True)
        ext = torch.randn(Bs, Td, H, device=dev) * 0.1
        dg2 = torch.empty(2, Td, Bs, 4 * H, device=dev)
        dh0, dc0 = torch.empty(Bs, 2 * H, device=dev), torch.empty(Bs, 2 * H, device=dev)
        dx0 = torch.empty(Bs, H, device=dev)
        us = replay_time(lambda: ops.lstm_dec2_bwd(wts, gates2, cs2, c0, ext, dg2, dh0, dc0, dx0))
        print(f'stacked decoder backward, one launch, B={Bs:3d}: {us / (2 * Td):6.2f} us per cell', flush=True)

    # for scale: the same number of empty-ish dependent launches (a 1-block fill kernel)
    z = torch.zeros(64, device=dev)

    def empty():
        for _ in range(T):
            z.add_(1.0)
    us = replay_time(empty)
    print(f'dependent trivial launches: {us / T:6.2f} us per launch', flush=True)


if __name__ == '__main__':
    main()
