#!/bin/bash
for rep in 1 2; do for dbg in 0 528 1040 2064 4112; do
  CELLMAP_DBG=$dbg timeout -k 10 100 python scripts/bench_knn.py --nq 750000 --nref 1500000 --d 50 --k 30 --uniform 0 --reps 2 2>&1 | tail -1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): r=json.loads(l); print('dbg', $dbg, 'p1', round(r['stage_ms']['phase1_mfma'],1))"
done; done
