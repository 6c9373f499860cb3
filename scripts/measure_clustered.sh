#!/bin/bash
# clustered embeddings table for DESIGN.md section 6 (40 Gaussian clusters, centres N(0, 5^2), d=50, 750k x 1.5M)
for sg in 2.0 1.0 0.3 0.1; do
timeout -k 10 200 python scripts/bench_knn.py --nq 750000 --nref 1500000 --d 50 --k 30 --uniform 0 --clusters 40 --spread 5 --sigma $sg --reps 2 2>&1 | tail -1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): r=json.loads(l); print('sigma $sg knn_ms', round(r['knn_ms'],1), {k: round(v,1) for k,v in r['stage_ms'].items() if v > 0.05}, r['stats'])
    else: print(l[:300])"
done
