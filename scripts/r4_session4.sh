#!/bin/bash
# Round 4, GPU session 4: first call vs steady state after the structure pre-test + code preload; clustered data still takes
# the structured route; then the whole GPU suite.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4_s4; mkdir -p $O
cd $R
timeout -k 10 200 python scripts/first_call_probe.py > $O/first_call.json 2> $O/first_call.err || echo "first_call rc=$?"
CELLMAP_NO_HINT=1 timeout -k 10 200 python scripts/first_call_probe.py > $O/first_call_nohint.json 2>> $O/first_call.err || echo "nohint rc=$?"
CELLMAP_NO_PRETEST=1 CELLMAP_NO_PRELOAD=1 timeout -k 10 200 python scripts/first_call_probe.py > $O/first_call_old.json 2>> $O/first_call.err || echo "old rc=$?"
python - <<'PY'
import json, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r4_s4/"
for f in ("first_call.json", "first_call_nohint.json", "first_call_old.json"):
    d = json.loads(open(O + f).read().strip().splitlines()[-1])
    print(f, "create", round(d["handle_create_ms"], 1), "call1", round(d["call1"]["wall_ms"], 2), {k: round(v, 2) for k, v in d["call1"]["stages"].items()})
    print("    call2", round(d["call2"]["wall_ms"], 2), "mean 4-13", round(d["mean_calls_4_13_wall_ms"], 2), {k: round(v, 2) for k, v in d["mean_calls_4_13_stages"].items()})
PY
for sg in 2.0 1.0 0.1; do
CELLMAP_DBG=16 timeout -k 10 200 python scripts/bench_knn.py --nq 750000 --nref 1500000 --d 50 --k 30 --uniform 0 --clusters 40 --spread 5 --sigma $sg --reps 2 > $O/clustered_$sg.log 2>&1 || echo "clustered $sg rc=$?"
grep -E "pre-test" $O/clustered_$sg.log | tail -1
tail -1 $O/clustered_$sg.log | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): r=json.loads(l); print('sigma $sg knn_ms', round(r['knn_ms'],1), {k: round(v,1) for k,v in r['stage_ms'].items() if v > 0.05}, r['stats'])
    else: print(l[:300])"
done
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1 || echo "tests rc=$?"
tail -8 $O/gpu_tests.log
