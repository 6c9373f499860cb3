#!/bin/bash
# Rehearsal of the N > 1 bench path on a ONE-GPU box, log kept under gpurun_out/ (copy to profiles/ to commit):
#  (a) RCCL smoke: one rank under torch.distributed.run with --force-dist -- init_process_group("nccl"), the packed
#      all_to_all_single / all_gather_into_tensor on the library's stream, barrier, all_reduce -- then the rank compares its
#      sharded result with the single-GPU pipeline;
#  (b) N = 2 and N = 4 ranks SHARING cuda:0 over gloo (RCCL refuses two ranks on one device; gloo has no device
#      all-to-all, so the all-gather exchange), started by bench.py's OWN launcher (`python bench.py --gpus N`): every rank
#      checks its slice against the single-GPU map of all queries onto the full reference (identical labels,
#      confidences to 1e-5).  Exercises HipOps, the 2-D layout, the packed exchange buffer, the strided exact merge and the
#      all-reduced bandwidth.
set -o pipefail
LOG=${1:-gpurun_out/r02_rehearsal.log}
: > $LOG
for ex in alltoall allgather; do
  timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 \
    bench.py --gpus 1 --steps 1 --warmup 1 --force-dist --exchange $ex --nq 300000 --nref 400000 --cpu-sample 0 --check 0 \
    >> $LOG 2>&1 || { echo "RCCL smoke ($ex) FAILED" >> $LOG; exit 1; }
done
# (a2) the split-size all_to_all_single of the 2-D layout (no sub-communicators), one rank on RCCL
timeout -k 10 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29543 \
    scripts/rccl_split_probe.py >> $LOG 2>&1 || { echo "RCCL split probe FAILED" >> $LOG; exit 1; }
for n in 2 4; do
  timeout -k 10 250 python bench.py --gpus $n --steps 1 --warmup 1 --backend gloo --share-gpu --exchange allgather \
    --nq 300000 --nref 400000 --cpu-sample 0 --check 0 --master-port 2953$n >> $LOG 2>&1 || { echo "rehearsal N=$n FAILED" >> $LOG; exit 1; }
done
timeout -k 10 250 python bench.py --gpus 2 --steps 1 --warmup 1 --backend gloo --share-gpu --exchange allgather --metric cosine \
    --nq 200000 --nref 300000 --cpu-sample 0 --check 0 --master-port 29537 >> $LOG 2>&1 || { echo "rehearsal cosine FAILED" >> $LOG; exit 1; }
# config 4's shape (d=128, K = 128 kernel family, reference sharded over ALL ranks) at a reduced size, 4 shards
timeout -k 10 250 python bench.py --gpus 4 --config 4 --steps 1 --warmup 1 --backend gloo --share-gpu --exchange allgather \
    --nq 120000 --nref 600000 --cpu-sample 0 --check 0 --master-port 29538 >> $LOG 2>&1 || { echo "rehearsal config 4 FAILED" >> $LOG; exit 1; }
grep -E "rehearsal|rccl probe|n_gpus" $LOG | cut -c1-220
