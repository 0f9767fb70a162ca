"""Streaming write / copy / read bandwidth of the box with plain torch kernels (sizes far beyond the 256 MB Infinity Cache):
the ceilings the row traffic of the edge kernels is measured against."""
import torch, time
dev = torch.device('cuda')
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for gb in (0.6, 2.4):
    n = int(gb * 1e9 / 4)
    x = torch.empty(n, device=dev); y = torch.empty(n, device=dev)
    x.fill_(1.0)
    t = timeit(lambda: x.fill_(2.0)); print(f'fill   {gb:.1f} GB: {t:.3f} ms  {gb / t:.2f} TB/s written')
    t = timeit(lambda: y.copy_(x)); print(f'copy   {gb:.1f} GB: {t:.3f} ms  {2 * gb / t:.2f} TB/s read+written')
    t = timeit(lambda: x.sum()); print(f'sum    {gb:.1f} GB: {t:.3f} ms  {gb / t:.2f} TB/s read')
    # 1 read stream, 2 write streams (the forward's ratio is 1.15 : 2.0)
    z = torch.empty(n, device=dev)
    t = timeit(lambda: (torch.add(x, 1.0, out=y), torch.mul(x, 2.0, out=z))); print(f'r+2w   {gb:.1f} GB: {t:.3f} ms  {4 * gb / t:.2f} TB/s (2 reads + 2 writes of the size)')
    del x, y, z
