#!/bin/bash
# widening of the sample-scan seeds at the bucket-centred level (CELLMAP_SEED_WIDEN x the plain hi*hi bound): time and failures
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_widen; rm -rf $O; mkdir -p $O
cd $R
for w in 0 0.25 0.5 1 2; do
for sg in 1.0 0.5 0.3 0.2; do
CELLMAP_DBG=16 CELLMAP_SEED_WIDEN=$w timeout -k 10 200 python scripts/bench_knn.py --nq 1500000 --nref 1500000 --d 50 --k 30 --uniform 0 --clusters 40 --spread 5 --sigma $sg --reps 3 > $O/out.txt 2> $O/err.txt || exit 1
echo "widen $w sigma $sg: $(grep 'stage 0' $O/err.txt | tail -1 | cut -c15-110)  $(tail -1 $O/out.txt | python -c "
import sys,json
r=json.loads(sys.stdin.read()); print('knn_ms', round(r['knn_ms'],1), {k: round(v,1) for k,v in r['stage_ms'].items() if v > 0.05}, 'passes', r['stats']['pruning_passes'], 'level', r['stats']['filter_products'])")" | tee -a $O/widen.log
done
done
