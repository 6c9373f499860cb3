#!/bin/bash
# Round-5 soak on the final library: path fuzz (clustered bias, default draw, wide) and the small-shape fuzz with fresh seeds
set -o pipefail
cd $GRAFT_REPO_ROOT; O=gpurun_out/fuzz_soak_r5; mkdir -p $O
timeout -k 10 420 python tests/fuzz/fuzz_paths.py --clustered --cases 150 --seed 7201 > $O/paths_clustered.log 2>&1; rc=$?; echo "clustered rc=$rc bad=$(grep -c '"ok": false' $O/paths_clustered.log) cases=$(grep -c '"case"' $O/paths_clustered.log) single=$(grep -c '"passes": 1' $O/paths_clustered.log) two=$(grep -c '"passes": 2' $O/paths_clustered.log)"; [ $rc -eq 0 ] || [ $rc -eq 124 ] || exit 1
timeout -k 10 300 python tests/fuzz/fuzz_paths.py --cases 120 --seed 7202 > $O/paths.log 2>&1; rc=$?; echo "paths rc=$rc bad=$(grep -c '"ok": false' $O/paths.log) cases=$(grep -c '"case"' $O/paths.log)"; [ $rc -eq 0 ] || [ $rc -eq 124 ] || exit 1
timeout -k 10 200 python tests/fuzz/fuzz_paths.py --wide --cases 40 --seed 7203 > $O/paths_wide.log 2>&1; rc=$?; echo "wide rc=$rc bad=$(grep -c '"ok": false' $O/paths_wide.log) cases=$(grep -c '"case"' $O/paths_wide.log)"; [ $rc -eq 0 ] || [ $rc -eq 124 ] || exit 1
timeout -k 10 200 python tests/fuzz/fuzz_small.py --cases 300 --seed 7204 > $O/small.log 2>&1; rc=$?; echo "small rc=$rc bad=$(grep -c '"ok": false' $O/small.log) cases=$(grep -c '"case"' $O/small.log)"; [ $rc -eq 0 ] || [ $rc -eq 124 ] || exit 1
