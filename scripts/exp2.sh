#!/bin/bash
CELLMAP_DBG=4 timeout -k 10 100 python bench.py --nq 750000 --nref 1500000 --steps 1 --warmup 0 --cpu-sample 0 --check 0 2>&1 | grep -E "cellmap dbg|phase1" | cut -c1-400
