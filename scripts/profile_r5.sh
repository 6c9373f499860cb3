#!/bin/bash
# rocprofv3 passes for round 5 (run on the GPU box from the repo root); raw outputs under gpurun_out/prof_r5/, summarised
# into profiles/ by scripts/summarize_r5.py.  Counters are collected in their own passes (with --kernel-trace only);
# FETCH_SIZE and WRITE_SIZE in separate passes (TCC slots, MI355X_MICROARCH.md).
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r5; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
Q="--steps 1 --warmup 1 --cpu-sample 0 --check 0 --no-host-leg"
pass() {  # pass NAME "<rocprof args>" <program args...>
  local name=$1; local rargs=$2; shift 2
  timeout -k 10 420 rocprofv3 --kernel-trace $rargs --output-format csv -d $O/$name -- python3 "$@" > $O/$name.json 2> $O/$name.err || echo "pass $name FAILED rc=$?"
  find $O/$name -name "*_kernel_trace.csv" -size +30M -delete
}
# (a) kernel trace + stats of the DEFAULT bench command (config 2)
pass trace_c2 "--stats" $R/bench.py
# (b) config 2: matrix-pipe / instruction-mix / LDS counters of the filter kernel, then its HBM-side traffic
pass pmc_mfma_c2 "--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" $R/bench.py $Q
pass pmc_inst_c2 "--pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA" $R/bench.py $Q
pass pmc_fetch_c2 "--pmc FETCH_SIZE" $R/bench.py $Q
pass pmc_write_c2 "--pmc WRITE_SIZE" $R/bench.py $Q
# (c) config 4's per-GPU share: 1.5 M x 1.5 M at d = 128 (K = 128 kernel family)
pass trace_d128 "--stats" $R/bench.py --d 128 $Q
pass pmc_mfma_d128 "--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" $R/bench.py --d 128 $Q
pass pmc_inst_d128 "--pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA" $R/bench.py --d 128 $Q
# (d) config 3 (A2, dense layer): stats + HBM traffic at a size that keeps the passes short (same kernel, same row width)
C3="--config 3 --nq 300000 --nref 400000 --steps 1 --warmup 1 --check 0"
pass trace_c3 "--stats" $R/bench.py $C3
pass pmc_fetch_c3 "--pmc FETCH_SIZE" $R/bench.py $C3
pass pmc_write_c3 "--pmc WRITE_SIZE" $R/bench.py $C3
# (e) config 5 (grid search + A3, sparse layer)
C5="--config 5 --nq 400000 --steps 1 --warmup 1 --check 0"
pass trace_c5 "--stats" $R/bench.py $C5
pass pmc_fetch_c5 "--pmc FETCH_SIZE" $R/bench.py $C5
pass pmc_write_c5 "--pmc WRITE_SIZE" $R/bench.py $C5
du -sh $O; ls $O | head -40
