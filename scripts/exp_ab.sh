#!/bin/bash
mkdir -p /tmp/ab && cp ab/*.so /tmp/ab/
CELLMAP_HIP_LIB=/tmp/ab/B.so timeout -k 10 200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -1
for rep in 1 2; do for v in A B; do for lv in 1 3; do
  CELLMAP_FILTER_PRODUCTS=$lv CELLMAP_HIP_LIB=/tmp/ab/$v.so timeout -k 10 100 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --check 0 2>&1 | tail -1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): print('$v level $lv', round(json.loads(l)['stage_ms_last_step']['phase1_mfma'],1))"
done; done; done
