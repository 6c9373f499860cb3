#!/bin/bash
# Round 5, item 5: the K = 128 family on clustered data (40 clusters, centres N(0, 5^2), d = 128, 500 k x 1.5 M, k = 30) with and
# without bucket pruning, and the structure-free share of config 4 (1.5 M x 1.5 M), alternating on one box.
cd $GRAFT_REPO_ROOT; O=gpurun_out/d128; mkdir -p $O; : > $O/lines.jsonl
run() {  # tag, env, args...
  tag=$1; envv=$2; shift 2
  env $envv timeout -k 10 400 python scripts/bench_knn.py "$@" --reps 3 2> $O/err_$tag.log | tail -1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); r['tag'] = '$tag'; r['env'] = '$envv'
        print(json.dumps(r))" >> $O/lines.jsonl || exit 1
}
for rep in 1 2; do
  for envv in X=1 CELLMAP_NO_IVF=1; do
    run iso_share $envv --nq 1500000 --nref 1500000 --d 128 --k 30 --uniform 0
    for sg in 2.0 1.0 0.3; do run cl_s$sg $envv --nq 500000 --nref 1500000 --d 128 --k 30 --uniform 0 --clusters 40 --spread 5 --sigma $sg; done
  done
done
CELLMAP_DBG=16 timeout -k 10 300 python scripts/bench_knn.py --nq 500000 --nref 1500000 --d 128 --k 30 --uniform 0 --clusters 40 --spread 5 --sigma 1.0 --reps 1 > $O/dbg_s1.0.log 2>&1
CELLMAP_DBG=16 timeout -k 10 300 python scripts/bench_knn.py --nq 500000 --nref 1500000 --d 128 --k 30 --uniform 0 --clusters 40 --spread 5 --sigma 0.3 --reps 1 > $O/dbg_s0.3.log 2>&1
grep "cellmap dbg" $O/dbg_s1.0.log | cut -c1-220; echo ---; grep "cellmap dbg" $O/dbg_s0.3.log | cut -c1-220
python - <<'PY'
import json
rows = [json.loads(l) for l in open("gpurun_out/d128/lines.jsonl")]
print("| case | pruning | k-NN ms | stages | level | tiles visited % | fallback queries |")
print("|---|---|---|---|---|---|---|")
for r in rows:
    print(f"| {r['tag']} | {'off' if 'NO_IVF' in r['env'] else 'on'} | {r['knn_ms']:.1f} | { {k: round(v,1) for k,v in r['stage_ms'].items() if v > 0.05} } | {r['stats']['filter_products']} | {r['stats']['tiles_visited_pct']} | {r['stats']['fallback_queries']} |")
PY
