"""Time of the un-folded BatchNorm backward (reduction pass + elementwise pass) and of the bias/activation
backward at the shapes the B=256 step uses, with the bytes each pass moves."""
import sys, os, importlib
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
pkg = importlib.import_module('disentangled-vae_amd')
ops = importlib.import_module('disentangled-vae_amd.hipvae.ops')


def t(fn, n=20):
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


for M, C in [(5120 * 64, 128), (5120 * 256, 64), (5120 * 1024, 32), (5120 * 16, 256), (5120, 512)]:
    du = torch.randn(M, C, device='cuda')
    y = torch.randn(M, C, device='cuda')
    aff = tuple(torch.rand(C, device='cuda') + 0.5 for _ in range(4))
    gamma = torch.ones(C, device='cuda')
    mb = M * C * 4 / 1e6
    us = t(lambda: ops.bn_bwd_reduce(du, y, aff))
    print(f'M={M:8d} C={C:4d} reduce {us:7.1f} us {2 * mb / us:.2f} TB/s', end='  ')
    us2 = t(lambda: ops.bn_bwd(du, y, *aff, gamma))
    print(f'reduce+apply {us2:7.1f} us  apply ~{us2 - us:7.1f} us {3 * mb / max(us2 - us, 1e-3):.2f} TB/s', end='  ')
    us3 = t(lambda: ops.bias_act_bwd(du, y, 1, True))
    print(f'bias_act_bwd {us3:7.1f} us {3 * mb / us3:.2f} TB/s')
