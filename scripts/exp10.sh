#!/bin/bash
# level-1 candidate slack sweep: phase-1 time, rerank time, queries sent to the exact fallback
for sl in 16 12 8 4; do
CELLMAP_SLACK=$sl CELLMAP_FILTER_PRODUCTS=1 timeout -k 10 100 python scripts/bench_knn.py --nq 750000 --nref 1500000 --d 50 --k 30 --uniform 0 --reps 2 2>&1 | tail -1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): r=json.loads(l); print('slack $sl', {k: round(v,1) for k,v in r['stage_ms'].items()}, r['stats'])"
done
