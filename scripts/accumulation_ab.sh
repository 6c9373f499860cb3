#!/bin/bash
# Round 5, item 4: the analytic accumulation bound at levels 1-2 against the device-validated model (CELLMAP_DBG=64 switches the
# model back on, host side only): k-NN time, filter level, fallback queries -- config 2's shape and the clustered table's cases
# (d = 50 and d = 128), alternating on one box.
cd $GRAFT_REPO_ROOT; O=gpurun_out/acc_ab; mkdir -p $O; : > $O/lines.jsonl
run() {  # tag, env, args...
  tag=$1; envv=$2; shift 2
  env $envv timeout -k 10 300 python scripts/bench_knn.py "$@" --reps 3 2> $O/err.log | tail -1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); r['tag'] = '$tag'; r['bound'] = 'model' if '$envv' != 'X=1' else 'analytic'
        print(json.dumps(r))" >> $O/lines.jsonl || exit 1
}
for rep in 1 2; do
  for envv in X=1 CELLMAP_DBG=64; do
    run c2 $envv --nq 1500000 --nref 1500000 --d 50 --k 30 --uniform 0
    for sg in 2.0 1.0 0.3 0.1; do run cl_d50_s$sg $envv --nq 750000 --nref 1500000 --d 50 --k 30 --uniform 0 --clusters 40 --spread 5 --sigma $sg; done
    for sg in 1.0 0.3; do run cl_d128_s$sg $envv --nq 500000 --nref 1500000 --d 128 --k 30 --uniform 0 --clusters 40 --spread 5 --sigma $sg; done
    run c4share $envv --nq 750000 --nref 1500000 --d 128 --k 30 --uniform 0
  done
done
python - <<'PY'
import json
rows = [json.loads(l) for l in open("gpurun_out/acc_ab/lines.jsonl")]
print("| case | bound | k-NN ms | filter ms | level | fallback queries |")
print("|---|---|---|---|---|---|")
for r in rows:
    print(f"| {r['tag']} | {r['bound']} | {r['knn_ms']:.1f} | {r['stage_ms']['phase1_mfma']:.1f} | {r['stats']['filter_products']} | {r['stats']['fallback_queries']} |")
PY
