#!/bin/bash
# kernel-level breakdown of the clustered route (40-cluster mixture, 1.5 M x 1.5 M, d = 50, k = 30): rocprofv3 --kernel-trace --stats
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_clustered; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/scripts/bench_knn.py --nq 1500000 --nref 1500000 --d 50 --k 30 --uniform 0 --clusters 40 --spread 5 --sigma ${SIGMA:-1.0} ${EXTRA:-} --reps 4 > $O/line.json 2> $O/err.log
tail -1 $O/line.json | cut -c1-600
python3 - <<PY
import csv, glob, re
f = sorted(glob.glob("$O/trace/**/*kernel_stats.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms (4 reps + set-up):", tot / 1e6)
for r in rows[:22]:
    name = re.sub(r"\(.*$", "", r["Name"].replace("void ", "").replace("(anonymous namespace)::", ""))[:90]
    print("%-92s calls %5s  total %8.2f ms  avg %8.3f ms  %5.1f %%" % (name, r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6, float(r["Percentage"])))
PY
