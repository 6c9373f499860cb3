#!/bin/bash
# ablate_r16.sh "NAME[:R16]"...: filter-stage time at config 2 with build_ab/lib_NAME.so (CELLMAP_R16 = the value after the
# colon, default 1), two alternating rounds on one box.
set -o pipefail
cd $GRAFT_REPO_ROOT
for round in 1 2; do
for spec in "$@"; do
  v=${spec%%:*}; r=1; [[ "$spec" == *:* ]] && r=${spec#*:}
  CELLMAP_R16=$r CELLMAP_HIP_LIB=$PWD/build_ab/lib_$v.so timeout -k 10 120 python scripts/bench_knn.py --nq ${NQ:-1500000} --nref ${NREF:-1500000} --d ${D:-50} --k ${K:-30} --uniform 0 --reps 3 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v (R16=$r) round $round: phase1 %.1f ms  knn %.1f ms  fallback %d cand %d' % (r['stage_ms']['phase1_mfma'], r['knn_ms'], r['stats']['fallback_queries'], r['stats']['candidates']))"
done; done
