#!/bin/bash
# Longer randomised cross-checks of round 4 (different seeds from fuzz_r4.sh); logs -> gpurun_out/fuzz_soak_r4/.
set -o pipefail
cd $GRAFT_REPO_ROOT; O=gpurun_out/fuzz_soak_r4; mkdir -p $O
timeout -k 10 500 python tests/fuzz/fuzz_small.py --cases 700 --seed 4277 > $O/small.log 2>&1; echo "small rc=$? bad=$(grep -c '"ok": false' $O/small.log) cases=$(grep -c '"case"' $O/small.log)"
timeout -k 10 500 python tests/fuzz/fuzz_paths.py --cases 60 --seed 4909 > $O/paths.log 2>&1; echo "paths rc=$? bad=$(grep -c '"ok": false' $O/paths.log) cases=$(grep -c '"case"' $O/paths.log)"
timeout -k 10 400 python tests/fuzz/fuzz_metrics.py --cases 300 --seed 4411 > $O/metrics.log 2>&1; echo "metrics rc=$? bad=$(grep -c '"ok": false' $O/metrics.log) cases=$(grep -c '"case"' $O/metrics.log)"
