#!/bin/bash
# Round-4 (late): the widened shape ranges (no width limit, term-wise metrics beyond 128 columns, large k) through the small-shape and
# metric fuzzers (logs -> gpurun_out/fuzz_r4b/, copied to profiles/r04_fuzz_wide_*.log).
set -o pipefail
cd $GRAFT_REPO_ROOT; O=gpurun_out/fuzz_r4b; mkdir -p $O
timeout -k 10 500 python tests/fuzz/fuzz_small.py --cases 300 --seed 4101 > $O/small.log 2>&1; echo "small rc=$? bad=$(grep -c '"ok": false' $O/small.log) cases=$(grep -c '"case"' $O/small.log)"
timeout -k 10 400 python tests/fuzz/fuzz_metrics.py --cases 160 --seed 4107 > $O/metrics.log 2>&1; echo "metrics rc=$? bad=$(grep -c '"ok": false' $O/metrics.log) cases=$(grep -c '"case"' $O/metrics.log)"
