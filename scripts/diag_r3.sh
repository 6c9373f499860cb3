#!/bin/bash
# Round-3 diagnostics of the config-2 filter kernel (knn_cand_f16q): in-kernel counters (CELLMAP_DBG=4), the event-free
# skeleton (DBG=1: thresholds at -inf) and the filter-free body (DBG=2), then one PMC pass with the SQ wait / active counters.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/diag_r3; mkdir -p $O
cd $R
for dbg in 4 1 2; do
  CELLMAP_DBG=$dbg timeout -k 10 150 python scripts/bench_knn.py --nq 1500000 --nref 1500000 --d 50 --k 30 --uniform 0 --reps 2 > $O/dbg$dbg.log 2>&1 || echo "dbg $dbg rc=$?"
  grep -E "cellmap dbg" $O/dbg$dbg.log | tail -3; grep stage_ms $O/dbg$dbg.log | tail -1
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmc_wait -- python3 $R/scripts/bench_knn.py --nq 1500000 --nref 1500000 --d 50 --k 30 --uniform 0 --reps 1 > $O/pmc_wait.log 2>&1 || echo "pmc pass rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH --output-format csv -d $O/pmc_inst -- python3 $R/scripts/bench_knn.py --nq 1500000 --nref 1500000 --d 50 --k 30 --uniform 0 --reps 1 > $O/pmc_inst.log 2>&1 || echo "pmc inst rc=$?"
find $O -name "*_kernel_trace.csv" -size +30M -delete
python3 - <<'PY'
import csv, glob, os, collections
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/diag_r3"
for name in ("pmc_wait", "pmc_inst"):
    for f in glob.glob(f"{O}/{name}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            if "knn_cand_f16q" in r.get("Kernel_Name", ""):
                acc[r["Counter_Name"]] += float(r["Counter_Value"])
        print(name, dict(acc))
PY
