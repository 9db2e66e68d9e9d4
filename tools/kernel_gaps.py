"""Idle time BETWEEN the kernels of a headline step, from a rocprofv3 --kernel-trace of bench.py.

    cd /tmp && rocprofv3 --kernel-trace -d <dir> -o t --output-format csv -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline
    python tools/kernel_gaps.py <dir> [--out profiles/r06/kernel_gaps.txt]

A step is found by its one dominant dispatch (the longest gemm256p launch); the report covers the dispatches between consecutive
dominant launches of the timed region: span, summed kernel time, summed gaps, and the gaps grouped by (previous kernel -> next kernel)."""
import argparse
import csv
import glob
import os
import re
from collections import defaultdict


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0][:48]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--out")
    args = ap.parse_args()
    f = sorted(glob.glob(os.path.join(args.dir, "**", "*_kernel_trace.csv"), recursive=True))[0]
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in csv.DictReader(open(f))]
    rows.sort()
    durs = sorted((e - s for s, e, n in rows if n.startswith("gemm256p_kernel<0, false")), reverse=True)
    if not durs:
        raise SystemExit("no gemm256p dispatch in the trace")
    thr = 0.8 * durs[0]
    dom = [i for i, (s, e, n) in enumerate(rows) if n.startswith("gemm256p_kernel<0, false") and e - s >= thr]
    lines = [f"{len(rows)} dispatches, {len(dom)} dominant launches (one per encode)"]
    # consecutive dominant launches that are one step apart (same number of dispatches between them = the steady timed region)
    seg = [(dom[k], dom[k + 1]) for k in range(len(dom) - 1)]
    counts = defaultdict(int)
    for a, b in seg:
        counts[b - a] += 1
    per_step = max(counts, key=lambda c: (counts[c], -c))
    seg = [(a, b) for a, b in seg if b - a == per_step][-6:]
    lines.append(f"dispatches per step: {per_step}; {len(seg)} steady steps analysed")
    gaps = defaultdict(lambda: [0, 0.0])
    for a, b in seg:
        span = rows[b][0] - rows[a][0]
        busy = sum(rows[i][1] - rows[i][0] for i in range(a, b))
        idle = 0
        for i in range(a, b):
            g = rows[i + 1][0] - rows[i][1]
            if g > 0:
                idle += g
                k = (rows[i][2], rows[i + 1][2])
                gaps[k][0] += 1
                gaps[k][1] += g
        lines.append(f"step: span {span / 1e6:.3f} ms, kernels {busy / 1e6:.3f} ms, gaps {idle / 1e6:.3f} ms ({100.0 * idle / span:.1f} %), "
                     f"mean gap {idle / max(per_step, 1) / 1e3:.2f} us")
    n = max(len(seg), 1)
    lines.append("largest gap totals per step (previous kernel -> next kernel): count per step, us per step")
    for k, (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:40]:
        lines.append(f"  {t / n / 1e3:8.2f} us  x{c / n:4.1f}  {k[0]} -> {k[1]}")
    text = "\n".join(lines) + "\n"
    print(text)
    if args.out:
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        open(args.out, "w").write(text)


if __name__ == "__main__":
    main()
