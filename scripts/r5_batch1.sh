#!/bin/bash
# round 5, first measurement batch: the driver's bench command (first command of the lease), the rank shares, the new full-size floors,
# the accumulation-bound A/B
set -o pipefail
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5b1; mkdir -p $O
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c2.json 2> $O/bench_c2.err || { tail -20 $O/bench_c2.err; exit 1; }
tail -c 1500 $O/bench_c2.json
rm -f $O/shares.jsonl
timeout -k 10 600 python scripts/rank_shares.py --out $O/shares.jsonl > $O/shares.log 2>&1 || { tail -20 $O/shares.log; exit 1; }
timeout -k 10 500 bash scripts/accumulation_ab.sh > $O/acc_ab.md 2>&1 || { tail -20 $O/acc_ab.md; exit 1; }
cat $O/acc_ab.md
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q > $O/pytest_fullsize.log 2>&1; tail -5 $O/pytest_fullsize.log
