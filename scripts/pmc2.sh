#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r1b; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU --output-format csv -d $O/pmc_wait -- python3 $R/scripts/bench_knn.py --nq 750000 --nref 1500000 --d 50 --k 30 --uniform 0 --reps 1 > $O/pmc_wait.json 2> $O/pmc_wait.err
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SMEM SQ_WAVES --output-format csv -d $O/pmc_wait2 -- python3 $R/scripts/bench_knn.py --nq 750000 --nref 1500000 --d 50 --k 30 --uniform 0 --reps 1 > $O/pmc_wait2.json 2> $O/pmc_wait2.err
python3 - <<'PY'
import csv,glob,collections,os
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/prof_r1b'
for d in ('pmc_wait','pmc_wait2'):
    fs=glob.glob(f'{O}/{d}/runc/*counter_collection.csv')
    if not fs: print(d,'no output'); continue
    a=collections.defaultdict(float)
    for r in csv.DictReader(open(fs[0])):
        if 'knn_cand_f16' in r['Kernel_Name']: a[r['Counter_Name']]+=float(r['Counter_Value'])
    print(d, dict(a))
PY
tail -3 $O/pmc_wait.err
