"""Development aid: run the fused MBConv at one EfficientViT-B1 shape (B = 32) a few times (for rocprofv3 counter passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import util as U
B, H, W, Cin, Cmid, Cout, stride, res = [int(v) for v in (sys.argv[1:9] if len(sys.argv) > 8 else "32 252 252 32 128 32 1 1".split())]
g = torch.Generator().manual_seed(0)
x = torch.randn(B, H, W, Cin, generator=g).to("cuda", torch.bfloat16)
w1 = (torch.randn(Cmid, Cin, generator=g) * (2 / Cin) ** 0.5).numpy(); b1 = (torch.randn(Cmid, generator=g) * 0.1).numpy()
wd = (torch.randn(Cmid, 9, generator=g) * 0.4).numpy(); bd = (torch.randn(Cmid, generator=g) * 0.1).numpy()
w2 = (torch.randn(Cout, Cmid, generator=g) / Cmid ** 0.5).numpy(); b2 = (torch.randn(Cout, generator=g) * 0.1).numpy()
OH, OW = (H + stride - 1) // stride, (W + stride - 1) // stride
out = torch.empty((B, OH, OW, Cout), dtype=torch.bfloat16, device="cuda")
for _ in range(3):
    U.check(U.lib().esam3_op_mbconv_fused(1, U.P(x), U.H(w1), U.H(b1), U.H(wd), U.H(bd), U.H(w2), U.H(b2), U.P(out), B, H, W, Cin, Cmid, Cout, stride, res, None), "mb")
torch.cuda.synchronize()
print("ok", float(out.float().abs().mean()))
