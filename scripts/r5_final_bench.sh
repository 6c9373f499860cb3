#!/bin/bash
# round 5, final library: the bench line of every configuration (-> profiles/r05_bench_config{2,3,4,5}.json)
set -o pipefail
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_final; mkdir -p $O
timeout -k 10 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/config2.json 2> $O/config2.err || { tail -20 $O/config2.err; exit 1; }
tail -c 200 $O/config2.json; echo
timeout -k 10 400 python bench.py --config 3 --steps 3 --warmup 1 > $O/config3.json 2> $O/config3.err || { tail -20 $O/config3.err; exit 1; }
tail -c 200 $O/config3.json; echo
timeout -k 10 400 python bench.py --config 5 --steps 3 --warmup 1 > $O/config5.json 2> $O/config5.err || { tail -20 $O/config5.err; exit 1; }
tail -c 200 $O/config5.json; echo
timeout -k 10 500 python bench.py --config 4 --steps 2 --warmup 1 > $O/config4.json 2> $O/config4.err || { tail -20 $O/config4.err; exit 1; }
tail -c 200 $O/config4.json; echo
