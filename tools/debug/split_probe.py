"""Tiled convolution with few row tiles: the plain entry point against the one that shares K out (batch-32 shapes)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'disentangled-vae_amd'))
from hipvae import ops  # noqa: E402
from hipvae._lib import lib  # noqa: E402


def t(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for (N, H, Cin, Cout, tr) in [(640, 8, 128, 256, 0), (640, 16, 64, 128, 0), (640, 4, 256, 128, 1), (256, 16, 64, 128, 0), (256, 8, 128, 256, 0), (32, 8, 128, 256, 0), (32, 16, 64, 128, 0)]:
    x = torch.randn(N, H, H, Cin, device='cuda')
    w = torch.randn(Cout, Cin, 4, 4, device='cuda') * 0.05 if not tr else torch.randn(Cin, Cout, 4, 4, device='cuda') * 0.05
    wp = ops.pack_deconv_w(w) if tr else ops.pack_conv_w(w)
    Ho = 2 * H if tr else H // 2
    y = torch.empty(N, Ho, Ho, Cout, device='cuda')
    s = torch.cuda.current_stream().cuda_stream
    plain = lib.dvae_deconv4s2_fwd_f32 if tr else lib.dvae_conv4s2_fwd_f32
    withws = lib.dvae_deconv4s2_fwd_ws_f32 if tr else lib.dvae_conv4s2_fwd_ws_f32
    need = lib.dvae_conv_fwd_ws_floats(tr, N, H, H, Cin, Cout)
    ws = torch.empty(max(need, 4), device='cuda')
    a = t(lambda: plain(x.data_ptr(), None, None, 0, wp.data_ptr(), y.data_ptr(), None, N, H, H, Cin, Cout, s))
    b = t(lambda: withws(x.data_ptr(), None, None, 0, wp.data_ptr(), y.data_ptr(), None, N, H, H, Cin, Cout,
                         ws.data_ptr(), need, s))
    print(f'N={N} H={H} {Cin}->{Cout} transposed={tr}: plain {a:7.1f} us, with scratch ({need * 4 / 1e6:.0f} MB) {b:7.1f} us')
