"""Development aid: per-kernel register / spill / LDS summary of a hipcc -S output (the .amdhsa metadata block)."""
import re
import sys

t = open(sys.argv[1]).read()
md = t[t.index("amdhsa.kernels:"):]
for blk in md.split("  - .agpr_count:")[1:]:
    g = lambda k: (re.search(k + r":\s+(\S+)", blk) or [None, "?"])[1]
    name = g(r"\.name")
    if len(sys.argv) > 2 and sys.argv[2] not in name:
        continue
    print(f"{name[14:70]:58s} vgpr {g(r'.vgpr_count'):>4s} vspill {g(r'.vgpr_spill_count'):>4s} sgpr {g(r'.sgpr_count'):>4s} sspill {g(r'.sgpr_spill_count'):>3s} lds {g(r'.group_segment_fixed_size'):>6s} scratch {g(r'.private_segment_fixed_size')}")
