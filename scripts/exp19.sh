#!/bin/bash
for shape in "375000 750000" "1500000 750000"; do set -- $shape
  timeout -k 10 100 python scripts/bench_knn.py --nq $1 --nref $2 --d 50 --k 30 --uniform 0 --reps 3 2>&1 | tail -1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): r=json.loads(l); print('rank-share nq', r['nq'], 'x n_ref', r['n_ref'], 'knn_ms', round(r['knn_ms'],1), {k: round(v,1) for k,v in r['stage_ms'].items() if v > 0.05}, r['stats'])"
done
timeout -k 10 300 python bench.py 2>&1 | tail -1 | cut -c1-2000
