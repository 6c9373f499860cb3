#!/bin/bash
timeout -k 10 200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
for shape in "1500000 187500" "750000 1500000"; do set -- $shape
  CELLMAP_DBG=4 timeout -k 10 100 python scripts/bench_knn.py --nq $1 --nref $2 --d 50 --k 30 --uniform 0 --reps 2 2>&1 | grep -E "cellmap dbg|stage_ms" | tail -2 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): r=json.loads(l); print(r['nq'], r['n_ref'], 'p1', round(r['stage_ms']['phase1_mfma'],1))
    else: print('   ', l.strip()[14:260])"
done
