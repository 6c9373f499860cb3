#!/bin/bash
timeout -k 10 200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
for dbg in 0 4; do
  CELLMAP_DBG=$dbg timeout -k 10 100 python bench.py --nq 750000 --nref 1500000 --steps 1 --warmup 1 --cpu-sample 0 --check 0 2>&1 | grep -E "cellmap dbg|phase1" | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): print('dbg', $dbg, json.loads(l)['stage_ms_last_step']['phase1_mfma'])
    else: print(l.strip())"
done
