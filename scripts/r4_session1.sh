#!/bin/bash
# Round 4, GPU session 1: copy-rate probe for the pinned staging ring, first-call breakdown, baseline bench line.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4_s1; mkdir -p $O
cd $R
timeout -k 10 200 ./build_ab/h2d_probe 300 > $O/h2d_probe.log 2>&1 || echo "probe rc=$?"
tail -40 $O/h2d_probe.log
timeout -k 10 200 python scripts/first_call_probe.py > $O/first_call.json 2> $O/first_call.err || echo "first_call rc=$?"
CELLMAP_NO_HINT=1 timeout -k 10 200 python scripts/first_call_probe.py > $O/first_call_nohint.json 2>> $O/first_call.err || echo "first_call nohint rc=$?"
cat $O/first_call.json $O/first_call_nohint.json
timeout -k 10 400 python bench.py --steps 10 --warmup 3 > $O/bench_base.json 2> $O/bench_base.err || echo "bench rc=$?"
python - <<'PY'
import json, os
o = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r4_s1/bench_base.json"
d = json.loads(open(o).read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "first_call_ms", "host_inclusive", "stage_ms_mean")}, d["roofline"]["frac"], d["roofline"]["kernel_ms"])
PY
