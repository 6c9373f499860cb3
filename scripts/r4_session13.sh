#!/bin/bash
# Round 4, GPU session 13: whole GPU suite on the final library, then the builder-run bench lines.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4_s13; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1 || echo "tests rc=$?"
tail -4 $O/gpu_tests.log
bash scripts/bench_lines_r4.sh
