#!/bin/bash
# Round 5, after the single-pass pruned stage: path fuzz (default + --wide), forced chains alternating single pass / pass A + B
set -o pipefail
cd $GRAFT_REPO_ROOT; O=gpurun_out/fuzz_r5b; mkdir -p $O
timeout -k 10 600 python tests/fuzz/fuzz_paths.py --cases 75 --seed 5105 > $O/paths.log 2>&1; rc1=$?; echo "paths rc=$rc1 bad=$(grep -c '"ok": false' $O/paths.log) cases=$(grep -c '"case"' $O/paths.log) single=$(grep -c '"passes": 1' $O/paths.log) two=$(grep -c '"passes": 2' $O/paths.log)"
[ $rc1 -eq 0 ] || exit 1
timeout -k 10 400 python tests/fuzz/fuzz_paths.py --wide --cases 30 --seed 5106 > $O/paths_wide.log 2>&1; rc2=$?; echo "wide rc=$rc2 bad=$(grep -c '"ok": false' $O/paths_wide.log) cases=$(grep -c '"case"' $O/paths_wide.log) single=$(grep -c '"passes": 1' $O/paths_wide.log) two=$(grep -c '"passes": 2' $O/paths_wide.log)"
[ $rc2 -eq 0 ] || exit 1
timeout -k 10 500 python tests/fuzz/fuzz_paths.py --clustered --cases 60 --seed 5107 > $O/paths_clustered.log 2>&1; rc3=$?; echo "clustered rc=$rc3 bad=$(grep -c '"ok": false' $O/paths_clustered.log) cases=$(grep -c '"case"' $O/paths_clustered.log) single=$(grep -c '"passes": 1' $O/paths_clustered.log) two=$(grep -c '"passes": 2' $O/paths_clustered.log)"
[ $rc3 -eq 0 ] || exit 1
