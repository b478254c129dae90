"""What the long rows of the motion encoder's Linear layer (4864 x 8192, rows 32 KB apart) cost its GEMMs.

The same launches (same grid, same loop, same flops) with the row stride of an operand changed, results ignored:
  real      A rows 32 KB apart (the layer as it is)
  stride 0  every row of the operand is the same 32 KB (one page, resident in L2)
  packed    rows 128 B apart (overlapping reads of a 0.7 MB buffer: resident in L2, a tile's rows share pages)
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'disentangled-vae_amd'))
from hipvae import ops  # noqa: E402
from hipvae._lib import lib  # noqa: E402


def t(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    dev = 'cuda'
    M, K, N = 4864, 8192, 512
    s = torch.cuda.current_stream().cuda_stream
    x = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) * 0.05
    dy = torch.randn(M, N, device=dev)
    fl = 2.0 * M * N * K

    # forward NT: y[M,N] = x[M,K] w[N,K]^T
    y = torch.empty(M, N, device=dev)
    ws = torch.empty(max(lib.dvae_gemm_ws_floats(M, N, K), 4), device=dev)
    for name, lda, ldb in (('real', K, K), ('A stride 0', 0, K), ('B stride 0', K, 0), ('both stride 0', 0, 0),
                           ('A packed (128 B)', 32, K), ('both packed', 32, 32)):
        us = t(lambda: lib.dvae_gemm_nt_f32(x.data_ptr(), lda, w.data_ptr(), ldb, None, 0, y.data_ptr(), N, M, N, K,
                                            ws.data_ptr(), s))
        print(f'nt {name:18s} {us:7.1f} us  {fl / us / 1e6:6.1f} TFLOP/s', flush=True)

    # weight gradient TN: dw[N,K] = dy[M,N]^T x[M,K]   (k index = M, both operands row-contiguous)
    dw = torch.empty(N, K, device=dev)
    ws2 = torch.empty(max(lib.dvae_gemm_ws_floats(N, K, M), 4), device=dev)
    for name, ldb in (('real', K), ('x stride 0', 0)):
        us = t(lambda: lib.dvae_gemm_tn_f32(dy.data_ptr(), N, x.data_ptr(), ldb, dw.data_ptr(), K, N, K, M, 0,
                                            ws2.data_ptr(), s))
        print(f'tn {name:18s} {us:7.1f} us  {fl / us / 1e6:6.1f} TFLOP/s', flush=True)

    # data gradient NN: dx[M,K] = dy[M,N] w[N,K]
    dx = torch.empty(M, K, device=dev)
    ws3 = torch.empty(max(lib.dvae_gemm_ws_floats(M, K, N), 4), device=dev)
    for name, ldb, ldc in (('real', K, K), ('w stride 0', 0, K), ('dx rows 128 B apart', K, 32)):
        big = torch.empty(M * 32 + K + 64, device=dev) if ldc != K else dx
        us = t(lambda: lib.dvae_gemm_nn_f32(dy.data_ptr(), N, w.data_ptr(), ldb, big.data_ptr(), ldc, M, K, N, 0,
                                            ws3.data_ptr(), s))
        print(f'nn {name:18s} {us:7.1f} us  {fl / us / 1e6:6.1f} TFLOP/s', flush=True)


if __name__ == '__main__':
    main()
