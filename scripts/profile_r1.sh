#!/bin/bash
# rocprofv3 passes for round 1 (run on the GPU box from the repo root); outputs under gpurun_out/prof_r1/
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters.txt 2>&1
grep -i -E "MFMA|FETCH_SIZE|WRITE_SIZE|LDS_BANK|GRBM_GUI|SQ_BUSY_CY|SQ_WAVE_CY|VALU_MFMA" $O/counters.txt | head -40 > $O/counters_short.txt
# (a) kernel trace + stats of the default bench command
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py > $O/bench_under_trace.json 2> $O/trace.err
# (b) MFMA / LDS counters on the same command (own pass, no tracing domains besides kernel-trace)
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --check 0 > $O/pmc_mfma.json 2> $O/pmc_mfma.err
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $O/pmc_lds -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --check 0 > $O/pmc_lds.json 2> $O/pmc_lds.err
# (c) HBM traffic of the SpMM (A2) kernel: FETCH_SIZE and WRITE_SIZE in separate passes
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/scripts/bench_spmm.py --reps 1 > $O/pmc_fetch.json 2> $O/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/scripts/bench_spmm.py --reps 1 > $O/pmc_write.json 2> $O/pmc_write.err
# (d) HBM traffic of the phase-1 kernel
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_knn -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --check 0 > $O/pmc_fetch_knn.json 2> $O/pmc_fetch_knn.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_knn -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --check 0 > $O/pmc_write_knn.json 2> $O/pmc_write_knn.err
ls -R $O | head -60
# keep the merge small: drop the big per-dispatch trace of torch kernels except ours
find $O -name "*_kernel_trace.csv" -size +20M -delete
