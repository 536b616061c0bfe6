#!/usr/bin/env python3
"""What this box's memory system gives plain streaming kernels (torch's own): read-only, write-only, copy (1 : 1) --
the context for the channelizer's 40 % reads / 60 % writes.  Run on the GPU box."""
import torch
n = 1 << 28                      # 1 GiB of float32
x = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
y = torch.empty_like(x)
def timed(f, reps=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
gb = n * 4 / 1e9
t = timed(lambda: y.copy_(x));   print(f"copy  (read 1 : write 1): {t:.3f} ms  {2 * gb / t:.2f} TB/s of traffic")
t = timed(lambda: y.fill_(1.0)); print(f"fill  (write only)      : {t:.3f} ms  {gb / t:.2f} TB/s")
t = timed(lambda: x.sum());      print(f"sum   (read only)       : {t:.3f} ms  {gb / t:.2f} TB/s")
t = timed(lambda: torch.add(x, 1.0, out=y)); print(f"add   (read 1 : write 1): {t:.3f} ms  {2 * gb / t:.2f} TB/s of traffic")
