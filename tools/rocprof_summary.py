#!/usr/bin/env python
"""Condense rocprofv3 output of `bench.py` into the summaries committed under profiles/.

    python tools/rocprof_summary.py --stats <dir with *_kernel_stats.csv / *_kernel_trace.csv> \
        --fetch <dir of the --pmc FETCH_SIZE pass> --write <dir of the --pmc WRITE_SIZE pass> \
        --steps K --round r01

Writes profiles/<round>_kernel_stats.csv (verbatim rocprofv3 --stats table), and
profiles/pmc_dominant_kernel.json: HBM traffic of the dominant launch (the level-0 3x3 conv of
the sam3 neck, one gemm256_kernel<bf16, ACT_NONE> dispatch per step -- the longest one) with the
gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md (HBM section: wide coalesced reads are
tallied at half their bytes -> doubled; WRITE_SIZE taken as is), plus that dispatch's average
duration from the kernel trace for comparison with bench.py's HIP-event figure.
"""
import argparse
import csv
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOM = "gemm256_kernel<unsigned short, 0>"


def one(pattern):
    files = sorted(glob.glob(pattern, recursive=True))
    if not files:
        raise SystemExit(f"no file matches {pattern}")
    return files[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stats", required=True)
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--steps", type=int, required=True, help="timed + warm-up steps of the --stats run")
    ap.add_argument("--round", default="r01")
    args = ap.parse_args()
    prof = os.path.join(ROOT, "profiles")
    os.makedirs(prof, exist_ok=True)
    shutil.copy(one(os.path.join(args.stats, "**", "*_kernel_stats.csv")), os.path.join(prof, f"{args.round}_kernel_stats.csv"))

    # per-dispatch durations of the dominant kernel symbol; the level-0 3x3 conv is its longest launch
    durs = []
    with open(one(os.path.join(args.stats, "**", "*_kernel_trace.csv"))) as f:
        for r in csv.DictReader(f):
            if DOM in r["Kernel_Name"]:
                durs.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    durs.sort(reverse=True)
    top = durs[: args.steps]
    out = {
        "kernel": "gemm256_kernel<bf16, ACT_NONE> (256x256x64 implicit GEMM), launch = neck level-0 3x3 conv 256->256 "
                  "@288^2, B=32 (M = 2,654,208 rows, K = 2304): the longest dispatch of this symbol in every step",
        "kernel_trace": {"dispatches_of_symbol": len(durs), "steps_in_run": args.steps,
                         "dominant_launch_avg_ns": sum(top) / max(len(top), 1),
                         "dominant_launch_min_ns": min(top) if top else None,
                         "dominant_launch_max_ns": max(top) if top else None,
                         "note": "rocprofv3 --stats averages all shapes launched through this symbol "
                                 "(see <round>_kernel_stats.csv); this is the per-shape figure from the trace"},
    }
    if args.fetch and args.write:
        def rows(d):
            with open(one(os.path.join(d, "**", "*_counter_collection.csv"))) as f:
                return [r for r in csv.DictReader(f)]
        fr, wr = rows(args.fetch), rows(args.write)
        cand = [r for r in fr if DOM in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
        # one warm-up + one timed step: the dominant launch appears twice; take the last (timed) one
        best = sorted(cand, key=lambda r: float(r["Counter_Value"]))[-2:]
        best = max(best, key=lambda r: int(r["Dispatch_Id"]))
        wmatch = [r for r in wr if r["Dispatch_Id"] == best["Dispatch_Id"] and r["Counter_Name"] == "WRITE_SIZE"]
        assert wmatch and DOM in wmatch[0]["Kernel_Name"], "dispatch order differs between the PMC passes"
        fetch_kb, write_kb = float(best["Counter_Value"]), float(wmatch[0]["Counter_Value"])
        out.update({
            "command": "rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) --output-format csv -- "
                       "python bench.py --steps 1 --warmup 1 --no-cpu-baseline",
            "dispatch_id": int(best["Dispatch_Id"]),
            "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb,
            "correction": "gfx950: FETCH_SIZE tallies 128-B requests of wide coalesced reads (global_load 16 B/lane and "
                          "buffer_load...lds alike) at 64 B -> doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE as is",
            "hbm_bytes_per_launch": 2.0 * fetch_kb * 1024.0 + write_kb * 1024.0,
            "duration_ns_under_pmc": int(best["End_Timestamp"]) - int(best["Start_Timestamp"]),
        })
    with open(os.path.join(prof, "pmc_dominant_kernel.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
