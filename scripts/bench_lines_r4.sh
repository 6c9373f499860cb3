#!/bin/bash
# Builder-run bench lines of round 4 (one MI355X): the driver's command for config 2, then configs 3, 4, 5 at full size.
set -o pipefail
cd $GRAFT_REPO_ROOT; O=gpurun_out/bench_r4; mkdir -p $O
timeout -k 10 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/config2.json 2> $O/config2.err && tail -c 400 $O/config2.json
timeout -k 10 900 python bench.py --config 3 --steps 3 --warmup 1 > $O/config3.json 2> $O/config3.err && tail -c 300 $O/config3.json
timeout -k 10 900 python bench.py --config 5 --steps 3 --warmup 1 > $O/config5.json 2> $O/config5.err && tail -c 300 $O/config5.json
timeout -k 10 900 python bench.py --config 4 --steps 2 --warmup 1 > $O/config4.json 2> $O/config4.err && tail -c 300 $O/config4.json
