"""Write / read / copy bandwidth of plain torch kernels on a tensor the size of the largest activation (671 MB)."""
import torch

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

N = 5120 * 32 * 32 * 32
x = torch.empty(N, device='cuda')
y = torch.empty(N, device='cuda')
mb = N * 4 / 1e6
us = t(lambda: x.fill_(1.0)); print(f'fill   {us:7.1f} us  write {mb / us:.2f} TB/s')
us = t(lambda: x.sum()); print(f'sum    {us:7.1f} us  read  {mb / us:.2f} TB/s')
us = t(lambda: y.copy_(x)); print(f'copy   {us:7.1f} us  total {2 * mb / us:.2f} TB/s')
us = t(lambda: torch.add(x, 1.0, out=y)); print(f'add    {us:7.1f} us  total {2 * mb / us:.2f} TB/s')
