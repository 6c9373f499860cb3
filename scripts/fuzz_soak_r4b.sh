#!/bin/bash
# Soak of the final round-4 library (after the pair-arithmetic refactor, the GQ kernels, the exact scan and the counting metrics):
# new seeds; logs -> gpurun_out/fuzz_soak_r4b/ (copied to profiles/r04_fuzz_final_*.log).
set -o pipefail
cd $GRAFT_REPO_ROOT; O=gpurun_out/fuzz_soak_r4b; mkdir -p $O
timeout -k 10 600 python tests/fuzz/fuzz_small.py --cases 600 --seed 4313 > $O/small.log 2>&1; echo "small rc=$? bad=$(grep -c '"ok": false' $O/small.log) cases=$(grep -c '"case"' $O/small.log)"
timeout -k 10 500 python tests/fuzz/fuzz_paths.py --cases 50 --seed 4919 > $O/paths.log 2>&1; echo "paths rc=$? bad=$(grep -c '"ok": false' $O/paths.log) cases=$(grep -c '"case"' $O/paths.log)"
timeout -k 10 500 python tests/fuzz/fuzz_metrics.py --cases 300 --seed 4421 > $O/metrics.log 2>&1; echo "metrics rc=$? bad=$(grep -c '"ok": false' $O/metrics.log) cases=$(grep -c '"case"' $O/metrics.log)"
timeout -k 10 300 python tests/fuzz/fuzz_graph.py > $O/graph.log 2>&1; echo "graph rc=$? bad=$(grep -c '"ok": false' $O/graph.log) cases=$(grep -c '"case"' $O/graph.log)"
