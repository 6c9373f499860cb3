#!/bin/bash
# round 5, second batch: the driver's bench command first, the whole GPU suite, the host-tier layer probe with and without the
# staging ring, the N > 1 rehearsal on one GPU
set -o pipefail
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5b2; mkdir -p $O
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c2.json 2> $O/bench_c2.err || { tail -20 $O/bench_c2.err; exit 1; }
tail -c 600 $O/bench_c2.json; echo
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; tail -5 $O/pytest_gpu.log; [ $rc -eq 0 ] || exit 1
for x in default 0; do
  if [ $x = default ]; then unset CELLMAP_XFER; else export CELLMAP_XFER=$x; fi
  timeout -k 10 300 python scripts/host_layer_probe.py > $O/host_layer_$x.json 2> $O/host_layer_$x.err || { tail -5 $O/host_layer_$x.err; exit 1; }
  cat $O/host_layer_$x.json
done
unset CELLMAP_XFER
timeout -k 10 900 bash scripts/rehearse_r5.sh $O/rehearsal.log > $O/rehearsal_summary.txt 2>&1; rc=$?; cat $O/rehearsal_summary.txt | cut -c1-300; exit $rc
