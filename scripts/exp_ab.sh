#!/bin/bash
mkdir -p /tmp/ab && cp build_ab/A.so build_ab/B.so /tmp/ab/
for rep in 1 2; do for v in A B; do
  CELLMAP_HIP_LIB=/tmp/ab/$v.so CELLMAP_DBG=4 timeout -k 10 100 python bench.py --nq 750000 --nref 1500000 --steps 1 --warmup 1 --cpu-sample 0 --check 0 2>&1 | grep -E "cellmap dbg|phase1" | tail -2 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): print('$v', json.loads(l)['stage_ms_last_step']['phase1_mfma'])
    else: print('$v', l.strip()[60:])"
done; done
