#!/bin/bash
# K = 128 hi*hi level: the rounds x item-cost split choice (CELLMAP_Q4=0 switches it off) on query sets of 1.2 .. 3 rounds
cd $GRAFT_REPO_ROOT; O=gpurun_out/wsplit; mkdir -p $O; : > $O/lines.txt
for nq in 150000 200000 300000 400000 750000 1500000; do
  for envv in X=1 CELLMAP_Q4=0; do
    env $envv timeout -k 10 300 python scripts/bench_knn.py --nq $nq --nref 1500000 --d 128 --k 30 --uniform 0 --reps 3 2>/dev/null | tail -1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('nq $nq', '$envv', 'filter %.1f ms' % r['stage_ms']['phase1_mfma'], 'knn %.1f' % r['knn_ms'], 'S', r['stats']['ref_splits'], 'fallback', r['stats']['fallback_queries'])" | tee -a $O/lines.txt
  done
done
