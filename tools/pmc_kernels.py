#!/usr/bin/env python
"""Per-kernel counter table from several rocprofv3 --pmc passes of the same command (one pass per counter group, as
MI355X_MICROARCH.md prescribes: SQ 8 slots, TCC FETCH_SIZE 3 + WRITE_SIZE 2 in separate passes).

    python tools/pmc_kernels.py --pass <dir> [--pass <dir> ...] --match mbconv3s,mla1,mla2d,kvprep,stem_dsconv [--out profiles/r06/x.txt]

For every kernel symbol whose name contains one of the --match substrings: dispatches, average duration (from the pass's own
timestamps) and, summed over the dispatches of ONE run and divided by the dispatch count, each counter; plus the ratios the
backbone work is steered by: LDS bank-conflict cycles / LDS active cycles, the wave-time split (issuing / issue-stalled / parked),
the matrix-pipe share, bytes fetched (x2: the gfx950 FETCH_SIZE correction) and written per launch.
"""
import argparse
import csv
import glob
import os
import re
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pass", dest="passes", action="append", required=True)
    ap.add_argument("--match", required=True)
    ap.add_argument("--out")
    args = ap.parse_args()
    keys = [k for k in args.match.split(",") if k]
    agg = defaultdict(lambda: defaultdict(float))   # kernel -> counter -> sum over dispatches
    cnt = defaultdict(lambda: defaultdict(set))     # kernel -> counter -> dispatch ids
    dur = defaultdict(dict)                         # kernel -> dispatch id -> ns (any pass)
    for d in args.passes:
        files = sorted(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True))
        if not files:
            print(f"# no counter_collection.csv under {d}")
            continue
        with open(files[0]) as f:
            for r in csv.DictReader(f):
                k = short(r["Kernel_Name"])
                if not any(s in k for s in keys):
                    continue
                c = r["Counter_Name"]
                agg[k][c] += float(r["Counter_Value"])
                cnt[k][c].add((d, r["Dispatch_Id"]))
                dur[k][(d, r["Dispatch_Id"])] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    lines = []
    for k in sorted(agg, key=lambda k: -sum(dur[k].values())):
        per = {c: agg[k][c] / max(len(cnt[k][c]), 1) for c in agg[k]}
        nd = max(len(s) for s in cnt[k].values())
        avg_us = sum(dur[k].values()) / max(len(dur[k]), 1) / 1e3
        lines.append(f"{k}: {nd} dispatches per pass, {avg_us:.1f} us average under the counters")
        lines.append("    " + "  ".join(f"{c}={per[c]:.0f}" for c in sorted(per)))
        dv = []
        if per.get("SQ_LDS_IDX_ACTIVE"):
            dv.append(f"lds_bank_conflict/lds_active={per.get('SQ_LDS_BANK_CONFLICT', 0) / per['SQ_LDS_IDX_ACTIVE']:.3f}")
        if per.get("SQ_WAVE_CYCLES"):
            wc = per["SQ_WAVE_CYCLES"]
            dv.append("wave time issuing/issue-stalled/parked=" + "/".join(
                f"{per.get(c, 0) / wc:.2f}" for c in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY")))
            if "SQ_WAIT_INST_LDS" in per:
                dv.append(f"lds-issue-stall={per['SQ_WAIT_INST_LDS'] / wc:.3f}")
        if per.get("GRBM_GUI_ACTIVE") and "SQ_VALU_MFMA_BUSY_CYCLES" in per:
            dv.append(f"mfma_pipe_busy={per['SQ_VALU_MFMA_BUSY_CYCLES'] / (per['GRBM_GUI_ACTIVE'] / 8.0 * 1024.0):.3f}")
            dv.append(f"clock_ghz={per['GRBM_GUI_ACTIVE'] / 8.0 / (avg_us * 1e3):.2f}")
        if "FETCH_SIZE" in per:
            dv.append(f"read_MB(2xFETCH)={2 * per['FETCH_SIZE'] * 1024 / 1e6:.1f}")
        if "WRITE_SIZE" in per:
            dv.append(f"written_MB={per['WRITE_SIZE'] * 1024 / 1e6:.1f}")
        if dv:
            lines.append("    -> " + "  ".join(dv))
    text = "\n".join(lines) + "\n"
    print(text)
    if args.out:
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        with open(args.out, "w") as f:
            f.write("# " + " ".join(os.sys.argv) + "\n" + text)


if __name__ == "__main__":
    main()
