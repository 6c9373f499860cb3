#!/bin/bash
# ab_a3_narrow.sh NAME...: A3 time at 1 M cells of config 5 with 7 500 genes (a gene-column shard of a 4-GPU run: slices of
# ~28 entries), build_ab/lib_NAME.so ("main" = the in-tree library), two alternating rounds.
set -o pipefail
cd $GRAFT_REPO_ROOT
for round in 1 2; do
for v in "$@"; do
  lib=$PWD/build_ab/lib_$v.so
  if [ $v = main ]; then lib=$PWD/cellmapper_amd/libcellmap_hip.so; fi
  CELLMAP_HIP_LIB=$lib timeout -k 10 200 python bench.py --config 5 --nq 1000000 --genes ${GENES:-7500} --density ${DENS:-0.03} --steps 2 --warmup 1 --check 0 --cpu-sample 0 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v round $round: a3 %.1f ms  frac %.3f checksum %s' % (r['roofline']['kernel_ms'], r['roofline']['frac'], r['layer_checksum']['value']))"
done; done
