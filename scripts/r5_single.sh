#!/bin/bash
# single-pass pruned stage (round 5): the tests that cross it, then the clustered shapes with it (default) and without (CELLMAP_DBG=512)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_single; rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 560 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "chained or pruning or centred or clustered or seeded" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for dbg in 0 1024 512; do
for sg in 2.0 1.0 0.3; do
CELLMAP_DBG=$dbg timeout -k 10 200 python scripts/bench_knn.py --nq 1500000 --nref 1500000 --d 50 --k 30 --uniform 0 --clusters 40 --spread 5 --sigma $sg --reps 4 2>&1 | tail -1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): r=json.loads(l); print('dbg $dbg sigma $sg knn_ms', round(r['knn_ms'],1), {k: round(v,1) for k,v in r['stage_ms'].items() if v > 0.05}, r['stats'])
    else: print(l[:300])" | tee -a $O/clustered.log || exit 1
done
done
