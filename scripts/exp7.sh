#!/bin/bash
timeout -k 10 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for lv in auto 1 2 3; do
  if [ $lv = auto ]; then unset CELLMAP_FILTER_PRODUCTS; else export CELLMAP_FILTER_PRODUCTS=$lv; fi
  timeout -k 10 200 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --check 512 2>&1 | tail -1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('level $lv', 'ms/step', round(r['ms_per_step'],1), 'p1', round(r['stage_ms_last_step']['phase1_mfma'],1), 'rerank', round(r['stage_ms_last_step']['rerank'],1), 'fb', r['stage_ms_last_step']['fallback'], r['config']['fallback_queries_last_step'], r['config']['candidates_per_query'], r['config']['filter_products'], r['parity_check'])"
done
