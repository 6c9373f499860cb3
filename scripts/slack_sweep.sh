#!/bin/bash
# slack_sweep.sh: config-2 k-NN time against the candidate slack of the level-1 filter (CELLMAP_SLACK; default k/2, >= 16).
cd $GRAFT_REPO_ROOT
for round in 1 2; do for s in 16 12 8 6; do
CELLMAP_SLACK=$s timeout -k 10 120 python scripts/bench_knn.py --nq 1500000 --nref 1500000 --d 50 --k 30 --uniform 0 --reps 3 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('slack $s round $round: phase1 %.1f ms  knn %.1f ms  stats %s' % (r['stage_ms']['phase1_mfma'], r['knn_ms'], r['stats']))"
done; done
