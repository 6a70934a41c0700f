"""HBM write bandwidth of a plain fill (torch) on the box: the practical ceiling for a write-only kernel"""
import torch, time
for mb in (268, 537, 2148, 4296):
    n = mb * 1000 * 1000 // 8
    x = torch.empty(n, dtype=torch.float64, device="cuda")
    for _ in range(3): x.fill_(1.5)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): x.fill_(2.5)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("fill %5d MB: %.1f us  %.2f TB/s" % (mb, ms * 1e3, n * 8 / ms / 1e9))
    y = torch.empty_like(x)
    e0.record()
    for _ in range(10): y.copy_(x)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("copy %5d MB: %.1f us  %.2f TB/s read+write" % (mb, ms * 1e3, 2 * n * 8 / ms / 1e9))
