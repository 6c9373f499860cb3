#!/bin/bash
timeout -k 10 200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
for nref in 375000 750000 1500000; do
timeout -k 10 200 python scripts/bench_knn.py --nq 1500000 --nref $nref --d 50 --k 30 --uniform 0 --reps 2 2>&1 | tail -1 | python -c "
import sys,json; r=json.loads(sys.stdin.read()); print('nref',r['n_ref'],'p1_ms',round(r['stage_ms']['phase1_mfma'],1),'Tpairs/s',round(r['pairs_per_s']/1e12,3))"
done
