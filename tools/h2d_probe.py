"""Rate of a pinned-host to device copy of one batch of 256 clips (84 MB) on this box.
usage: python tools/h2d_probe.py"""
import torch, time
x = torch.empty(256, 20, 1, 64, 64).pin_memory()
d = torch.empty_like(x, device='cuda')
torch.cuda.synchronize()
for n in (1, 3):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        d.copy_(x, non_blocking=True)
    e.record(); torch.cuda.synchronize()
    print('H2D 84 MB pinned:', s.elapsed_time(e) / 10, 'ms', 84e-3 * 1.0 / (s.elapsed_time(e) / 10 / 1e3) / 1e0, 'GB/s' )
