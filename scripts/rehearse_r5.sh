#!/bin/bash
# Round-5 rehearsal (the round-3 script + the new default layout: query groups only, with the reference-sharded secondary leg)
# Round-3 text: rehearsal of the N > 1 bench paths on a ONE-GPU box (ranks share cuda:0 over gloo; RCCL refuses two ranks on one
# device).  Every line carries the parity gate (every rank's sample against the CPU oracle) and a cpu_baseline; the default
# exchange is requested, so the first step's self-check runs: gloo has no device all-to-all, the call raises, and the run
# falls back to the world all-gather -- the fall-back path of the first hardware run, exercised for real.  Configs 3 and 5 run
# at N = 1 and N = 2 / 4 on the same reduced problem: their layer_checksum must agree.
set -o pipefail
LOG=${1:-gpurun_out/r05_rehearsal.log}
: > $LOG
run() { echo "### $*" >> $LOG; timeout -k 10 400 "$@" >> $LOG 2>&1 || { echo "FAILED: $*" >> $LOG; echo "FAILED: $*"; exit 1; }; }
C2="--steps 2 --warmup 1 --nq 300000 --nref 400000 --cpu-sample 512 --check 512 --no-host-leg"
run python bench.py --gpus 1 $C2
run python bench.py --gpus 2 $C2 --backend gloo --share-gpu --master-port 29532
run python bench.py --gpus 4 $C2 --backend gloo --share-gpu --master-port 29534
run python bench.py --gpus 4 $C2 --backend gloo --share-gpu --ref-shards 4 --master-port 29535
run python bench.py --gpus 4 $C2 --backend gloo --share-gpu --ref-shards 2 --master-port 29536
run python bench.py --gpus 3 $C2 --backend gloo --share-gpu --master-port 29531
run python bench.py --gpus 2 --steps 1 --warmup 1 --nq 200000 --nref 300000 --metric cosine --cpu-sample 0 --check 256 --no-host-leg --backend gloo --share-gpu --master-port 29537
C3="--config 3 --steps 2 --warmup 1 --nq 100000 --nref 120000 --genes 2000 --cpu-sample 256 --check 256"
run python bench.py --gpus 1 $C3
run python bench.py --gpus 2 $C3 --backend gloo --share-gpu --master-port 29538
run python bench.py --gpus 4 $C3 --backend gloo --share-gpu --master-port 29539
C5="--config 5 --steps 2 --warmup 1 --nq 200000 --genes 3000 --cpu-sample 256 --check 256"
run python bench.py --gpus 1 $C5
run python bench.py --gpus 2 $C5 --backend gloo --share-gpu --master-port 29540
run python bench.py --gpus 3 $C5 --backend gloo --share-gpu --master-port 29541
run python bench.py --gpus 4 --config 4 --steps 1 --warmup 1 --nq 120000 --nref 600000 --cpu-sample 128 --check 128 --backend gloo --share-gpu --master-port 29542
# one rank on RCCL (init, all-reduce, barrier on the library's stream)
run python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29543 bench.py --gpus 1 --steps 1 --warmup 1 --force-dist --nq 300000 --nref 400000 --cpu-sample 0 --check 256 --no-host-leg
python - "$LOG" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("### "):
        cmd = line[4:].strip()
    if line.startswith("{"):
        r = json.loads(line)
        pc = r.get("parity_check") or {}
        print("n_gpus=%d cfg=%s par=%s value=%.3g ms=%.1f parity=%s exchange=%s checksum=%s cpu=%s" % (
            r["n_gpus"], r["config"].get("baseline_config"), r["config"].get("parallelism"), r["value"], r["ms_per_step"],
            pc.get("ok"), (r["config"].get("exchange") or {}).get("used"), (r.get("layer_checksum") or {}).get("value"),
            ("%.3g on %s cores" % (r["cpu_baseline"]["value"], r["cpu_baseline"]["cores"])) if r.get("cpu_baseline") else None))
PY
grep -E "rehearsal\]|self-check|FAILED" $LOG | cut -c1-250
