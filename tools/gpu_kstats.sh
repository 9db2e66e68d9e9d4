#!/bin/bash
# true kernel durations of the headline step (graph replay) from rocprofv3 --kernel-trace --stats: per-step totals by kernel
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02/kstats
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O -o s --output-format csv -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" > $O/stats.log 2>&1
python - <<P
import csv, glob, re
f = glob.glob("$O/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernels total per step (ms, /7 passes incl. warmup+profile pass):", round(tot / 7e6, 3))
for r in rows[:45]:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Name"]); n = re.sub(r"^void ", "", n); n = re.sub(r"\(.*", "", n)
    print(f'{float(r["TotalDurationNs"]) / 7e6:8.4f} ms/step  calls {int(r["Calls"]):5d}  avg {float(r["AverageNs"]) / 1e3:8.1f} us  {n[:90]}')
P
find $O -name "*_kernel_trace.csv" -delete
