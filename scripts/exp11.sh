#!/bin/bash
set -o pipefail
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3 || exit 1
for sl in 12 8; do
CELLMAP_SLACK=$sl CELLMAP_FILTER_PRODUCTS=1 timeout -k 10 100 python scripts/bench_knn.py --nq 750000 --nref 1500000 --d 50 --k 30 --uniform 0 --reps 2 2>&1 | tail -1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): r=json.loads(l); print('slack $sl', {k: round(v,1) for k,v in r['stage_ms'].items()}, r['stats'])"
done
