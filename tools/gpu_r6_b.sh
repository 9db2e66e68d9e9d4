#!/bin/bash
# round 6, GPU call B: GPU timeline of API-level steps (where the device idles) + the double-buffered PCIe-inclusive leg
O=gpurun_out/r06
mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/tl -o t --output-format csv -- python $R/tools/api_timeline.py run > $R/$O/b_api_timeline_host.txt 2>&1
cd $R
grep "host step" $O/b_api_timeline_host.txt
python tools/api_timeline.py report /tmp/tl > $O/b_api_timeline.txt 2>&1
cat $O/b_api_timeline.txt | cut -c1-200
timeout 400 python bench.py --no-cpu-baseline > $O/b_bench.json 2> $O/b_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06/b_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], {k: v for k, v in d["config"].items() if k.endswith("images_per_s")})
PY
