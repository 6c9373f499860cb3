#!/bin/bash
# Round 4, GPU session 11: multi-device handle (sharded copies, staged route, self-check), sharded bench record with per-rank clocks.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4_s11; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_multi.py tests/test_gpu_boundary.py -x -q -m gpu > $O/tests_multi.log 2>&1 || echo "tests rc=$?"
tail -5 $O/tests_multi.log
timeout -k 10 400 python bench.py --force-dist --steps 3 --warmup 1 --nq 400000 --nref 400000 --no-host-leg > $O/bench_forcedist.json 2> $O/bench_forcedist.err || echo "force-dist rc=$?"
timeout -k 10 400 python bench.py --gpus 2 --share-gpu --backend gloo --steps 2 --warmup 1 --nq 200000 --nref 300000 --no-host-leg > $O/bench_gloo2.json 2> $O/bench_gloo2.err || echo "gloo2 rc=$?"
python - <<'PY'
import json, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r4_s11/"
for f in ("bench_forcedist.json", "bench_gloo2.json"):
    try:
        d = json.loads(open(O + f).read().strip().splitlines()[-1])
        print(f, d["value"], d["config"]["parallelism"], d["config"]["exchange"], d.get("parity_check", {}).get("ok"))
        for r in d.get("per_rank", []): print("   ", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items()})
        print("   traffic_source", d["roofline"].get("traffic_source"))
    except Exception as e:
        print(f, "FAILED", e); print(open(O + f.replace(".json", ".err")).read()[-1500:])
PY
