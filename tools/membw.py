#!/usr/bin/env python3
"""Calibrate achievable HBM rates on this chip with plain torch kernels: pure write (fill), pure read (sum), copy."""
import torch
n = 1 << 30     # 8 GiB of float64
x = torch.empty(n, dtype=torch.float64, device="cuda")
y = torch.empty(n, dtype=torch.float64, device="cuda")
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(reps):
        a.record(); fn(); b.record(); b.synchronize()
        best = min(best, a.elapsed_time(b))
    return best
gb = n * 8 / 1e9
w = t(lambda: x.fill_(1.5)); print("fill  (write %.1f GB): %.3f ms -> %.0f GB/s" % (gb, w, gb / w * 1e3))
r = t(lambda: x.sum()); print("sum   (read  %.1f GB): %.3f ms -> %.0f GB/s" % (gb, r, gb / r * 1e3))
c = t(lambda: y.copy_(x)); print("copy  (r+w  %.1f GB): %.3f ms -> %.0f GB/s" % (2 * gb, c, 2 * gb / c * 1e3))
