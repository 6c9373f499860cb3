#!/bin/bash
# K = 128 family, clustered, 1.5 M x 1.5 M: single-pass pruned main stage (default) against pass A + B (CELLMAP_DBG=512)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_wide_single; rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wide or direct or ladder or tight or subset or fill_the_chip or long_lists" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for dbg in 0; do
for sg in 1.0 0.3; do
CELLMAP_DBG=$((dbg+16)) timeout -k 10 300 python scripts/bench_knn.py --nq 1500000 --nref 1500000 --d 128 --k 30 --uniform 0 --clusters 40 --spread 5 --sigma $sg --reps 3 > $O/out.txt 2> $O/err.txt || exit 1
echo "dbg $dbg sigma $sg: $(grep 'stage [0-9]\|pruning' $O/err.txt | tail -3 | cut -c15-100 | tr '\n' '|')  $(tail -1 $O/out.txt | python -c "
import sys,json
r=json.loads(sys.stdin.read()); print('knn_ms', round(r['knn_ms'],1), {k: round(v,1) for k,v in r['stage_ms'].items() if v > 0.05}, r['stats']['filter_products'], r['stats']['pruning_passes'])")" | tee -a $O/wide.log
done
done
