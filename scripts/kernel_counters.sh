#!/bin/bash
for dbg in ${DBGS:-4 1 2}; do
CELLMAP_FILTER_PRODUCTS=${LV:-1} CELLMAP_DBG=$dbg timeout -k 10 100 python scripts/bench_knn.py --nq 750000 --nref 1500000 --d 50 --k 30 --uniform 0 --reps 2 2>&1 | grep -E "cellmap dbg|stage_ms" | tail -2 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): r=json.loads(l); print('dbg $dbg p1', round(r['stage_ms']['phase1_mfma'],1))
    else: print('   ', l.strip()[14:400])"
done
