#!/bin/bash
# Longer randomised cross-checks (different seeds from fuzz_r3.sh); logs -> gpurun_out/fuzz_soak/.
set -o pipefail
cd $GRAFT_REPO_ROOT; O=gpurun_out/fuzz_soak; mkdir -p $O
timeout -k 10 500 python tests/fuzz/fuzz_small.py --cases 700 --seed 2077 > $O/small.log 2>&1; echo "small rc=$? bad=$(grep -c '"ok": false' $O/small.log) cases=$(grep -c '"case"' $O/small.log)"
timeout -k 10 500 python tests/fuzz/fuzz_paths.py --cases 60 --seed 909 > $O/paths.log 2>&1; echo "paths rc=$? bad=$(grep -c '"ok": false' $O/paths.log) cases=$(grep -c '"case"' $O/paths.log)"
