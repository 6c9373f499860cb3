#!/bin/bash
# K' of the single-pass centred stage (CELLMAP_SLACK): certification against list cost
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_slack; rm -rf $O; mkdir -p $O
cd $R
for sl in 0; do
for sg in 1.0 0.3 0.2; do
CELLMAP_KC5=$sl CELLMAP_DBG=16 timeout -k 10 200 python scripts/bench_knn.py --nq 1500000 --nref 1500000 --d 50 --k 30 --uniform 0 --clusters 40 --spread 5 --sigma $sg --reps 3 > $O/out.txt 2> $O/err.txt || exit 1
echo "kc5 +$sl sigma $sg: $(grep 'stage [0-9]' $O/err.txt | tail -2 | cut -c15-62 | tr '\n' '|')  $(tail -1 $O/out.txt | python -c "
import sys,json
r=json.loads(sys.stdin.read()); print('knn_ms', round(r['knn_ms'],1), {k: round(v,1) for k,v in r['stage_ms'].items() if v > 0.05})")" | tee -a $O/slack.log
done
done
