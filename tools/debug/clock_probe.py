"""The shader clock the chip holds inside the Linear-layer GEMMs of the headline step, on random operands.

Needs a diagnostic build of the library (gemm.hip with -DDVAE_GEMM_STAMP=1) named by DVAE_HIP_LIB: every workgroup
stamps the shader-clock counter and the 100 MHz real-time counter around its main loop; after about two seconds of
back-to-back launches the median ratio is the clock.
"""
import ctypes
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'disentangled-vae_amd'))
from hipvae import ops  # noqa: E402
from hipvae._lib import lib  # noqa: E402


def main():
    dev = 'cuda'
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
    lib.dvae_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.dvae_debug_read_stamps.restype = ctypes.c_int
    for M, K, N in ((4864, 8192, 512), (5120, 640, 4096)):
        x = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev) * 0.05
        dy = torch.randn(M, N, device=dev)
        fl = 2.0 * M * N * K
        for name, fn, wgs in (('nt', lambda: ops.gemm_nt(x, w), None), ('nn', lambda: ops.gemm_nn(dy, w), None),
                              ('tn', lambda: ops.gemm_tn(dy, x), None)):
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 0
            while time.perf_counter() - t0 < seconds:
                for _ in range(50):
                    fn()
                torch.cuda.synchronize()
                n += 50
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(50):
                fn()
            e.record()
            torch.cuda.synchronize()
            us = s.elapsed_time(e) / 50 * 1e3
            buf = (ctypes.c_ulonglong * 8192)()
            rc = lib.dvae_debug_read_stamps(ctypes.cast(buf, ctypes.c_void_p), 8192)
            assert rc == 0, rc
            ghz = [buf[2 * i] / buf[2 * i + 1] * 0.1 for i in range(256) if buf[2 * i + 1] > 0]
            cyc = statistics.median(buf[2 * i] for i in range(256))
            print(f'{name} M={M} K={K} N={N}: {us:7.1f} us {fl / us / 1e6:6.1f} TFLOP/s; in-kernel clock median '
                  f'{statistics.median(ghz):.3f} GHz (min {min(ghz):.3f}, max {max(ghz):.3f}); main loop '
                  f'{cyc / 1e3:.0f} k shader cycles per workgroup', flush=True)


if __name__ == '__main__':
    main()
