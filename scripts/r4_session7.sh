#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4_s7; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for r in 1 0; do
CELLMAP_R16=$r timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r$r -- python3 $R/scripts/bench_knn.py --nq 1500000 --nref 1500000 --d 50 --k 30 --uniform 0 --reps 2 > $O/prof_r$r.log 2>&1 || echo "rc=$?"
f=$(find $O/prof_r$r -name "*kernel_stats.csv" | head -1)
echo "R16=$r"; head -6 $f | cut -c1-200
done
find $O -name "*_kernel_trace.csv" -size +10M -delete
