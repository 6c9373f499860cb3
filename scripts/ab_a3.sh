#!/bin/bash
# ab_a3.sh NAME...: A3 time at 1 M cells of config 5 with build_ab/lib_NAME.so ("main" = the in-tree library; "old" = the
# in-tree library under CELLMAP_A3_GATHER=1), two alternating rounds on one box.
set -o pipefail
cd $GRAFT_REPO_ROOT
for round in 1 2; do
for v in "$@"; do
  lib=$PWD/build_ab/lib_$v.so; extra=""
  if [ $v = main ]; then lib=$PWD/cellmapper_amd/libcellmap_hip.so; fi
  if [ $v = old ]; then lib=$PWD/cellmapper_amd/libcellmap_hip.so; extra="CELLMAP_A3_GATHER=1"; fi
  env $extra CELLMAP_HIP_LIB=$lib timeout -k 10 200 python bench.py --config 5 --nq ${NQ:-1000000} --steps 2 --warmup 1 --check 0 --cpu-sample 0 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v round $round: a3 %.1f ms  frac %.3f checksum %s' % (r['roofline']['kernel_ms'], r['roofline']['frac'], r['layer_checksum']['value']))"
done; done
