#!/bin/bash
# Round-5 randomised cross-checks on the GPU box (logs -> gpurun_out/fuzz_r5/, copied to profiles/r05_fuzz_*.log).
set -o pipefail
cd $GRAFT_REPO_ROOT; O=gpurun_out/fuzz_r5; mkdir -p $O
timeout -k 10 400 python tests/fuzz/fuzz_small.py --cases 260 --seed 5031 > $O/small.log 2>&1; echo "small rc=$? bad=$(grep -c '"ok": false' $O/small.log) cases=$(grep -c '"case"' $O/small.log)"
timeout -k 10 700 python tests/fuzz/fuzz_paths.py --cases 60 --seed 5005 > $O/paths.log 2>&1; echo "paths rc=$? bad=$(grep -c '"ok": false' $O/paths.log) cases=$(grep -c '"case"' $O/paths.log)"
timeout -k 10 300 python tests/fuzz/fuzz_metrics.py --cases 120 --seed 5007 > $O/metrics.log 2>&1; echo "metrics rc=$? bad=$(grep -c '"ok": false' $O/metrics.log) cases=$(grep -c '"case"' $O/metrics.log)"
timeout -k 10 300 python tests/fuzz/fuzz_graph.py > $O/graph.log 2>&1; echo "graph rc=$? bad=$(grep -c '"ok": false' $O/graph.log) cases=$(grep -c '"case"' $O/graph.log)"
timeout -k 10 300 python tests/fuzz/fuzz_a3.py > $O/a3.log 2>&1; echo "a3 rc=$? bad=$(grep -c '"ok": false' $O/a3.log) cases=$(grep -c '"case"' $O/a3.log)"
