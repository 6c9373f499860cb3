#!/bin/bash
# Round 4, GPU session 12: three-accumulator K = 128 kernel: exactness, then A/B against the two-accumulator kernel at config 4's per-GPU shard
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4_s12; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "three_accumulator or wide_rows or four_accumulator" > $O/tests_t3.log 2>&1 || echo "tests rc=$?"
tail -6 $O/tests_t3.log
for round in 1 2; do
for t in 1 0; do
CELLMAP_T3=$t timeout -k 10 200 python scripts/bench_knn.py --nq 1500000 --nref 1500000 --d 128 --k 30 --uniform 0 --reps 3 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('T3=$t round $round: phase1 %.1f ms  knn %.1f ms  fallback %d cand %d' % (r['stage_ms']['phase1_mfma'], r['knn_ms'], r['stats']['fallback_queries'], r['stats']['candidates']))"
done; done
