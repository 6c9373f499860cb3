#!/bin/bash
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3 || exit 1
CELLMAP_FILTER_PRODUCTS=4 timeout -k 10 200 python scripts/bench_knn.py --nq 750000 --nref 1500000 --d 50 --k 30 --uniform 0 --reps 1 2>&1 | tail -1 | cut -c1-330
timeout -k 10 200 python scripts/bench_knn.py --nq 4000000 --nref 4000000 --d 2 --k 90 --self --reps 2 2>&1 | tail -1 | cut -c1-330
