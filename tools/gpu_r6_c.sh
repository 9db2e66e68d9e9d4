#!/bin/bash
# round 6, GPU call C: the API path after (1) the helper-thread staging of chunk 2 and (2) hand-backs started before any wait
O=gpurun_out/r06
mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_e2e_gpu.py -q -m gpu -k "batch or processor or api or host or alias or preprocess" --timeout 500 > $O/c_tests.txt 2>&1
tail -4 $O/c_tests.txt | cut -c1-300
timeout 300 python tools/api_level_probe.py 2>&1 | grep -v amdgpu > $O/c_api_probe.txt
head -12 $O/c_api_probe.txt | cut -c1-200
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/tl -o t --output-format csv -- python $R/tools/api_timeline.py run > $R/$O/c_api_timeline_host.txt 2>&1
cd $R
grep "host step" $O/c_api_timeline_host.txt | tail -3
python tools/api_timeline.py report /tmp/tl > $O/c_api_timeline.txt 2>&1
tail -30 $O/c_api_timeline.txt | cut -c1-160
timeout 500 python bench.py --no-cpu-baseline > $O/c_bench.json 2> $O/c_bench.err
tail -3 $O/c_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06/c_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], {k: v for k, v in d["config"].items() if k.endswith("images_per_s") or "FAILED" in str(v)})
PY
