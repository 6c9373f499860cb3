#!/bin/bash
# final round-1 measurements quoted in DESIGN.md section 6: per-rank shares, d=128, low-d, host API
for shape in "1500000 750000" "750000 750000" "375000 750000" "187500 1500000"; do set -- $shape
  timeout -k 10 100 python scripts/bench_knn.py --nq $1 --nref $2 --d 50 --k 30 --uniform 0 --reps 2 2>&1 | tail -1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): r=json.loads(l); print('rank-share nq', r['nq'], 'x n_ref', r['n_ref'], 'knn_ms', round(r['knn_ms'],1), {k: round(v,1) for k,v in r['stage_ms'].items() if v > 0.05}, r['stats'])"
done
timeout -k 10 200 python scripts/bench_knn.py --nq 1500000 --nref 1500000 --d 128 --k 30 --uniform 0 --reps 2 2>&1 | tail -1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): r=json.loads(l); print('d128', 'knn_ms', round(r['knn_ms'],1), {k: round(v,1) for k,v in r['stage_ms'].items() if v > 0.05}, r['stats'])"
timeout -k 10 200 python scripts/bench_knn.py --nq 4000000 --nref 4000000 --d 2 --k 90 --self --reps 2 2>&1 | tail -1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): r=json.loads(l); print('lowd 4M d2 k90', 'knn_ms', round(r['knn_ms'],1), {k: round(v,1) for k,v in r['stage_ms'].items() if v > 0.05}, r['stats'])"
LOGLEVEL=WARNING timeout -k 10 300 python scripts/bench_host_api.py 2>&1 | tail -1
