"""Development aid: what a plain device-to-device copy reaches on this GPU at the tensor sizes of the backbone
(calibration for the streaming kernels' bytes/s)."""
import torch
for mb in (25, 50, 100, 200, 400, 1000):
    n = mb * 2 ** 20 // 2
    x = torch.empty(n, dtype=torch.bfloat16, device="cuda").normal_()
    y = torch.empty_like(x)
    for _ in range(3):
        y.copy_(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        y.copy_(x)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"copy {mb:5d} MB in + {mb} MB out: {ms * 1e3:8.1f} us  {2 * mb * 2 ** 20 / ms / 1e9:7.2f} TB/s (read+write)")
