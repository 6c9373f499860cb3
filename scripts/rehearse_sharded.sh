#!/bin/bash
# Rehearsal of the N > 1 bench path on a ONE-GPU box: N ranks share cuda:0, torch.distributed over gloo (all-gather
# exchange; gloo has no device all-to-all), every rank checks its slice of the sharded result against the single-GPU map
# of all queries onto the full reference (identical labels, confidences to 1e-5).  Exercises HipOps, the 2-D layout,
# the exact merge, the all-reduced bandwidth and bench.py's own N > 1 logic; only the RCCL calls themselves are not run.
set -o pipefail
for n in 2 4; do
  timeout -k 10 250 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2953$n \
    bench.py --gpus $n --steps 1 --warmup 1 --backend gloo --share-gpu --exchange allgather --nq 300000 --nref 400000 \
    --cpu-sample 0 --check 0 2>&1 | grep -E "rehearsal|Error" | cut -c1-200 || exit 1
done
