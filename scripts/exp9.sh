#!/bin/bash
# parity first (kNN-focused GPU tests), then level-1 / level-3 timings of phase 1 at 750k x 1.5M, then default bench
set -o pipefail
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3 || exit 1
for lv in 1 3; do
CELLMAP_FILTER_PRODUCTS=$lv timeout -k 10 100 python scripts/bench_knn.py --nq 750000 --nref 1500000 --d 50 --k 30 --uniform 0 --reps 2 2>&1 | grep -E "stage_ms" | tail -1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): r=json.loads(l); print('level $lv p1', round(r['stage_ms']['phase1_mfma'],1), 'total', r.get('ms'))"
done
timeout -k 10 300 python bench.py 2>&1 | tail -1
