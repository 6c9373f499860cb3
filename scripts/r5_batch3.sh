#!/bin/bash
# round 5, third batch: bench lines of every config on the current library, the rocprofv3 passes, the host-tier probes, the fuzzers
set -o pipefail
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5b3; mkdir -p $O
timeout -k 10 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/config2.json 2> $O/config2.err || { tail -20 $O/config2.err; exit 1; }
tail -c 300 $O/config2.json; echo
timeout -k 10 900 python bench.py --config 3 --steps 3 --warmup 1 > $O/config3.json 2> $O/config3.err && tail -c 300 $O/config3.json
timeout -k 10 900 python bench.py --config 5 --steps 3 --warmup 1 > $O/config5.json 2> $O/config5.err && tail -c 300 $O/config5.json
timeout -k 10 900 python bench.py --config 4 --steps 2 --warmup 1 > $O/config4.json 2> $O/config4.err && tail -c 300 $O/config4.json
for x in default 0; do
  if [ $x = default ]; then unset CELLMAP_XFER; else export CELLMAP_XFER=$x; fi
  timeout -k 10 300 python scripts/host_layer_probe.py > $O/host_layer_$x.json 2> $O/host_layer_$x.err; cat $O/host_layer_$x.json
done
unset CELLMAP_XFER
timeout -k 10 400 python scripts/prof_host_api.py > $O/prof_host_api.log 2>&1; grep "^{" $O/prof_host_api.log
bash scripts/profile_r5.sh > $O/profile_r5.log 2>&1; tail -5 $O/profile_r5.log
