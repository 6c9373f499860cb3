#!/bin/bash
# tight clusters (sigma 0.2 / 0.1): single-pass (default) against pass A + B (CELLMAP_DBG=512); every call's stage lines
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_tight; rm -rf $O; mkdir -p $O
cd $R
for dbg in 0 512; do
for sg in 0.2 0.1; do
CELLMAP_DBG=$((dbg+16)) timeout -k 10 200 python scripts/bench_knn.py --nq 1500000 --nref 1500000 --d 50 --k 30 --uniform 0 --clusters 40 --spread 5 --sigma $sg --reps 3 > $O/out.txt 2> $O/err.txt || exit 1
echo "== dbg $dbg sigma $sg" | tee -a $O/tight.log
grep 'stage [0-9]' $O/err.txt | cut -c15-90 | tee -a $O/tight.log
tail -1 $O/out.txt | python -c "
import sys,json
r=json.loads(sys.stdin.read()); print('knn_ms', round(r['knn_ms'],1), {k: round(v,1) for k,v in r['stage_ms'].items() if v > 0.05})" | tee -a $O/tight.log
done
done
