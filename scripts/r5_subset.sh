#!/bin/bash
# wide-split seeded subset stages (round 5): tests that cross subset stages, then the mixture shapes with it and without (CELLMAP_DBG=2048)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_subset; rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 700 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "staged or subset or clustered or pruning or centred or seeded or chained or tight or ladder" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for dbg in 0; do
for sg in 1.0 0.3 0.2 0.1; do
CELLMAP_DBG=$((dbg+16)) timeout -k 10 200 python scripts/bench_knn.py --nq 1500000 --nref 1500000 --d 50 --k 30 --uniform 0 --clusters 40 --spread 5 --sigma $sg --reps 3 > $O/out.txt 2> $O/err.txt || exit 1
echo "dbg $dbg sigma $sg: $(grep 'stage [0-9]' $O/err.txt | tail -4 | cut -c15-75 | tr '\n' '|')  $(tail -1 $O/out.txt | python -c "
import sys,json
r=json.loads(sys.stdin.read()); print('knn_ms', round(r['knn_ms'],1), {k: round(v,1) for k,v in r['stage_ms'].items() if v > 0.05})")" | tee -a $O/subset.log
done
done
