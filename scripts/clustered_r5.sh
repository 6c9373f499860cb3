#!/bin/bash
# Round 5 (late): the clustered route on the round's final library -- 40-cluster mixture, 1.5 M x 1.5 M, d = 50, k = 30.
# Single-pass pruned main stage (default) against pass A + pass B (CELLMAP_DBG=512), per sigma; every call's stage lines; then the
# kernel-level breakdown (rocprofv3 --kernel-trace --stats) and the in-kernel counters of the sigma = 1.0 case.
# Output: gpurun_out/clustered_r5/report.md (copied to profiles/r05_clustered_single_pass.md)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/clustered_r5; rm -rf $O; mkdir -p $O
cd $R
{
echo "# Clustered route, round 5 (late): single-pass pruned main stage"
echo
echo "\`scripts/clustered_r5.sh\` on one MI355X box, library of the commit that carries this file.  40 Gaussian clusters, centres N(0, 5^2), within-cluster sigma as given,"
echo "reference and queries share the centres (rN = the ragged variant, \`--ragged\`: Dirichlet(0.7) cluster weights, different on the two sides, per-cluster spread 0.3 .. 3 x sigma, random per-coordinate scale);"
echo "1.5 M x 1.5 M, d = 50, k = 30, float32, \`scripts/bench_knn.py --reps 4\` (mean of the calls after the first;"
echo "stage times of the last call).  \`CELLMAP_DBG=512\` keeps pass A + pass B (the round-4 form, chained pass B).  Stage lines: \`CELLMAP_DBG=16\`."
echo
echo "| sigma | form | k-NN ms | stages (ms) | main stage | uncertified after it | later stages |"
echo "|---|---|---|---|---|---|---|"
} > $O/report.md
for spec in 2.0 1.7 1.5 1.3 1.0 0.5 0.3 0.2 0.17 0.15 0.12 0.1 0.05 r1.0 r0.5 r0.3; do
sg=${spec#r}; extra=""; [ "$spec" != "$sg" ] && extra="--ragged"
for dbg in 0 512; do
CELLMAP_DBG=$((dbg+16)) timeout -k 10 300 python scripts/bench_knn.py --nq 1500000 --nref 1500000 --d 50 --k 30 --uniform 0 --clusters 40 --spread 5 --sigma $sg $extra --reps 4 > $O/out.txt 2> $O/err.txt || exit 1
python3 - "$spec" "$dbg" $O/out.txt $O/err.txt >> $O/report.md <<'PY'
import sys, json, re
sg, dbg, out, err = sys.argv[1:5]
r = json.loads(open(out).read().strip().splitlines()[-1])
lines = [l for l in open(err) if "cellmap dbg" in l and "stage " in l and "queries uncertified" in l]
# the stage lines of the LAST call: from the last "stage 0" on
last0 = max(i for i, l in enumerate(lines) if "stage 0 " in l)
st = lines[last0:]
def short(l):
    m = re.search(r"stage (\d+) level (\d+): (\d+) of (\d+) queries uncertified, (\d+) without", l)
    return m.groups() if m else None
s0 = short(st[0])
later = "; ".join("level %s on %s -> %s left" % (short(l)[1], short(l)[3], short(l)[2]) for l in st[1:]) or "-"
stats = r["stats"]
form = "pass A + B" if dbg != "0" else "single pass"
if not stats["pruning_passes"]: form += " (not pruned: full scan)"
print("| %s | %s | %.1f | %s | level %s, K' = %d, %d %% of the pairs | %s of %s | %s |" % (
    sg, form, r["knn_ms"], {k: round(v, 1) for k, v in r["stage_ms"].items() if v > 0.05}, s0[1], stats["candidates"],
    stats["tiles_visited_pct"], s0[2], s0[3], later))
PY
done
done
{
echo
echo "## Kernel-level breakdown, sigma = 1.0 (rocprofv3 --kernel-trace --stats, 4 calls + set-up)"
echo
echo '```'
} >> $O/report.md
bash scripts/prof_clustered.sh > $O/prof.txt 2>&1 || { tail -20 $O/prof.txt; exit 1; }
grep -v "^\[" $O/prof.txt | tail -n +2 | head -26 >> $O/report.md
{
echo '```'
echo
echo "## In-kernel counters of the pass, sigma = 1.0 (CELLMAP_DBG=4: diagnostics body, one call)"
echo
echo '```'
} >> $O/report.md
for dbg in 20 1044; do
CELLMAP_DBG=$dbg timeout -k 10 300 python scripts/bench_knn.py --nq 1500000 --nref 1500000 --d 50 --k 30 --uniform 0 --clusters 40 --spread 5 --sigma 1.0 --reps 1 2>&1 | grep "cellmap dbg\] filter\|cellmap dbg\] sample" | sed "s/^/[CELLMAP_DBG=$dbg] /" | cut -c1-420 >> $O/report.md
done
echo '```' >> $O/report.md
echo "(CELLMAP_DBG=1044 = 1024 + 20: without the dense seeds.  appended / 1.5 M queries = appends per query; prunes = list merges.)" >> $O/report.md
tail -40 $O/report.md
