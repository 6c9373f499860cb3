#!/bin/bash
# bucket pruning: new tests, clustered table with stage prints, and the unclustered default case (must not slow down)
set -o pipefail
CELLMAP_DBG=16 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -s -k "bucket_pruning or pilot_selects or clustered" 2>&1 | tail -15 || exit 1
for sg in 2.0 1.0 0.3 0.1; do
CELLMAP_DBG=16 timeout -k 10 100 python scripts/bench_knn.py --nq 750000 --nref 1500000 --d 50 --k 30 --uniform 0 --clusters 40 --spread 5 --sigma $sg --reps 2 2>&1 | grep -E "pruning|^\{|stage" | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): r=json.loads(l); print('sigma $sg knn_ms', round(r['knn_ms'],1), {k: round(v,1) for k,v in r['stage_ms'].items() if v > 0.05}, r['stats'])
    else: print(l[:200].rstrip())" || exit 1
done
timeout -k 10 100 python scripts/bench_knn.py --nq 750000 --nref 1500000 --d 50 --k 30 --uniform 0 --reps 3 2>&1 | tail -1 | cut -c1-400
