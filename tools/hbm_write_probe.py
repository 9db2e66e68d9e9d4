"""Development aid: what a plain write-only / copy kernel reaches on this box (yardstick for the HBM-write-bound operators)."""
import torch
x = torch.empty(32 * 146 * 146 * 512, dtype=torch.bfloat16, device="cuda")
y = torch.empty_like(x)
def t(f, n=20):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
gb = x.numel() * 2 / 1e9
ms = t(lambda: x.fill_(1.0)); print(f"fill_ {gb:.3f} GB: {ms:.4f} ms = {gb / ms * 1e3:.0f} GB/s written")
ms = t(lambda: x.zero_()); print(f"zero_ {gb:.3f} GB: {ms:.4f} ms = {gb / ms * 1e3:.0f} GB/s written")
ms = t(lambda: y.copy_(x)); print(f"copy_ {gb:.3f} GB: {ms:.4f} ms = {2 * gb / ms * 1e3:.0f} GB/s read+written")
ms = t(lambda: torch.add(x, 1.0, out=y)); print(f"add   {gb:.3f} GB: {ms:.4f} ms = {2 * gb / ms * 1e3:.0f} GB/s read+written")
ms = t(lambda: x.sum()); print(f"sum   {gb:.3f} GB: {ms:.4f} ms = {gb / ms * 1e3:.0f} GB/s read")
