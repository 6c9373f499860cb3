#!/bin/bash
timeout -k 10 200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "slabs" 2>&1 | tail -2
for shape in "1500000 750000" "750000 750000" "375000 750000" "187500 1500000"; do set -- $shape
  timeout -k 10 100 python scripts/bench_knn.py --nq $1 --nref $2 --d 50 --k 30 --uniform 0 --reps 2 2>&1 | tail -1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): r=json.loads(l); print('rank-share', r['nq'], 'x', r['n_ref'], 'knn_ms', round(r['knn_ms'],1), r['stage_ms'])"
done
LOGLEVEL=WARNING timeout -k 10 300 python scripts/bench_host_api.py 2>&1 | tail -1
