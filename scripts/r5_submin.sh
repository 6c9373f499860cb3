#!/bin/bash
# smallest flagged subset that is regrouped by bucket and pruned (CELLMAP_IVF_SUBMIN)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_submin; rm -rf $O; mkdir -p $O
cd $R
for sm in 65536 16384 4096; do
for sg in 0.1; do
CELLMAP_IVF_SUBMIN=$sm CELLMAP_DBG=16 timeout -k 10 200 python scripts/bench_knn.py --nq 1500000 --nref 1500000 --d 50 --k 30 --uniform 0 --clusters 40 --spread 5 --sigma $sg --reps 4 > $O/out.txt 2> $O/err.txt || exit 1
echo "== submin $sm sigma $sg" | tee -a $O/submin.log
grep 'stage [0-9]\|pruning' $O/err.txt | tail -4 | cut -c15-110 | tee -a $O/submin.log
tail -1 $O/out.txt | python -c "
import sys,json
r=json.loads(sys.stdin.read()); print('knn_ms', round(r['knn_ms'],1), {k: round(v,1) for k,v in r['stage_ms'].items() if v > 0.05})" | tee -a $O/submin.log
done
done
