#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_final; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1 || echo "tests rc=$?"
tail -3 $O/gpu_tests.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err || echo "bench rc=$?"
python - <<'PY'
import json, os
d = json.loads(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r4_final/bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "first_call_ms", "value_host_inclusive")}, "frac", round(d["roofline"]["frac"], 4), "kernel_ms", round(d["roofline"]["kernel_ms"], 1), "traffic", d["roofline"]["traffic"], "stale", d["roofline"]["traffic_source"]["stale"], "parity", d["parity_check"]["ok"], "cpu", round(d["cpu_baseline"]["value"]))
PY
