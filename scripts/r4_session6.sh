#!/bin/bash
# Round 4, GPU session 6: the 16x16x32 re-tile (knn_cand_f16r): accumulation-bound model, exactness, A/B against knn_cand_f16q.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4_s6; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "mfma_f16_accumulation or four_accumulator or r16_filter" > $O/tests_r16.log 2>&1 || echo "tests rc=$?"
grep -E "worst observed|passed|failed|Error|error" $O/tests_r16.log | tail -8
for round in 1 2; do
for r in 1 0; do
CELLMAP_R16=$r timeout -k 10 150 python scripts/bench_knn.py --nq 1500000 --nref 1500000 --d 50 --k 30 --uniform 0 --reps 3 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('R16=$r round $round: phase1 %.1f ms  knn %.1f ms  fallback %d cand %d' % (r['stage_ms']['phase1_mfma'], r['knn_ms'], r['stats']['fallback_queries'], r['stats']['candidates']))"
done; done
