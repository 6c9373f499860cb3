#!/bin/bash
# exact-scan hand-over threshold (CELLMAP_FB_MAX): clustered sigma 0.3 / 1.0 and the structure-free config-2 shape
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_fbmax; rm -rf $O; mkdir -p $O
cd $R
for fb in 64 512 1024; do
for args in "--uniform 0 --clusters 40 --spread 5 --sigma 0.3" "--uniform 0 --clusters 40 --spread 5 --sigma 1.0" "--uniform 2000"; do
CELLMAP_FB_MAX=$fb timeout -k 10 200 python scripts/bench_knn.py --nq 1500000 --nref 1500000 --d 50 --k 30 $args --reps 4 2>&1 | tail -1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): r=json.loads(l); print('fb_max $fb $args knn_ms', round(r['knn_ms'],1), {k: round(v,1) for k,v in r['stage_ms'].items() if v > 0.05}, r['stats']['fallback_queries'], r['stats']['pruning_passes'])
    else: print(l[:300])" | tee -a $O/fbmax.log || exit 1
done
done
