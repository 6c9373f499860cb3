#!/bin/bash
# ab_grid.sh NAME...: the low-d search (grid_query + re-rank) of config 5 at 4 M cells, build_ab/lib_NAME.so ("main" = in-tree).
set -o pipefail
cd $GRAFT_REPO_ROOT
for round in 1 2; do
for v in "$@"; do
  lib=$PWD/build_ab/lib_$v.so
  if [ $v = main ]; then lib=$PWD/cellmapper_amd/libcellmap_hip.so; fi
  CELLMAP_HIP_LIB=$lib timeout -k 10 200 python scripts/bench_knn.py --nq 4000000 --nref 4000000 --d 2 --k 90 --uniform 1 --reps 3 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v round $round: phase1 %.1f ms  knn %.1f ms  stats %s' % (r['stage_ms']['phase1_mfma'], r['knn_ms'], r['stats']))"
done; done
