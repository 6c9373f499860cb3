#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_fb; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export CELLMAP_SLACK=8 CELLMAP_FILTER_PRODUCTS=1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/scripts/bench_knn.py --nq 750000 --nref 1500000 --d 50 --k 30 --uniform 0 --reps 1 > $O/out.json 2> $O/err.txt
f=$(find $O -name "*kernel_stats.csv" | head -1)
head -12 "$f" | cut -c1-200
find $O -name "*_kernel_trace.csv" -size +20M -delete
