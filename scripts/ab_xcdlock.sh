#!/bin/bash
# Same-box A/B of the XCD-phase-locked scan order (CELLMAP_NO_XCDLOCK=1 switches it off), then exactness tests of the
# four-accumulator filter, then the fabric-side traffic of both.
set -o pipefail
cd $GRAFT_REPO_ROOT; O=gpurun_out/xcdlock; mkdir -p $O
run() { timeout -k 10 120 python scripts/bench_knn.py --nq 1500000 --nref 1500000 --d 50 --k 30 --uniform 0 --reps 3 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: phase1 %.1f ms  knn %.1f ms  fallback %d' % (r['stage_ms']['phase1_mfma'], r['knn_ms'], r['stats']['fallback_queries']))"; }
for round in 1 2 3; do
  run lock
  CELLMAP_NO_XCDLOCK=1 run nolock
  CELLMAP_HIP_LIB=$PWD/build_ab/lib_r2.so run r2
done
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "four_accumulator or filter_level or seed" 2>&1 | tail -3
timeout -k 10 600 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "config2 or config_2 or all_rows" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
for v in lock nolock; do
  if [ $v = nolock ]; then export CELLMAP_NO_XCDLOCK=1; else unset CELLMAP_NO_XCDLOCK; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$O/fetch_$v -- python3 $GRAFT_REPO_ROOT/scripts/bench_knn.py --nq 1500000 --nref 1500000 --d 50 --k 30 --uniform 0 --reps 1 > $GRAFT_REPO_ROOT/$O/fetch_$v.log 2>&1 || echo "fetch pass $v rc=$?"
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $GRAFT_REPO_ROOT/$O/clk_$v -- python3 $GRAFT_REPO_ROOT/scripts/bench_knn.py --nq 1500000 --nref 1500000 --d 50 --k 30 --uniform 0 --reps 1 > $GRAFT_REPO_ROOT/$O/clk_$v.log 2>&1 || echo "clk pass $v rc=$?"
done
find $GRAFT_REPO_ROOT/$O -name "*_kernel_trace.csv" -size +30M -delete
python3 - <<'PY'
import csv, glob, os, collections
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/xcdlock"
for name in ("fetch_lock", "fetch_nolock", "clk_lock", "clk_nolock"):
    acc = collections.defaultdict(float); dur = 0.0
    for f in glob.glob(f"{O}/{name}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "knn_cand_f16q" in r.get("Kernel_Name", ""):
                acc[r["Counter_Name"]] += float(r["Counter_Value"])
    for f in glob.glob(f"{O}/{name}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "knn_cand_f16q" in r.get("Kernel_Name", ""):
                dur += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
    print(name, dict(acc), "kernel ms", round(dur, 2))
PY
