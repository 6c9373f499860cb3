#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_a3bank; mkdir -p $O
for round in 1 2; do
for v in 0 1; do
  if [ $v = 1 ]; then export CELLMAP_BENCH_BANKFRIENDLY=1; else unset CELLMAP_BENCH_BANKFRIENDLY; fi
  timeout -k 10 300 python bench.py --config 5 --nq 1000000 --steps 3 --warmup 1 --check 64 --cpu-sample 0 > $O/c5_$v.json 2> $O/c5_$v.err || echo "rc=$?"
  python - <<PY
import json
d = json.loads(open("$O/c5_$v.json").read().strip().splitlines()[-1])
print("bankfriendly=$v round $round: step %.1f ms  spgemm kernel %.1f ms  out nnz %s  parity %s" % (d["ms_per_step"], d["roofline"]["kernel_ms"], d["config"].get("output_nnz", d["config"].get("out_nnz")), d.get("parity_check", {}).get("ok")))
PY
done; done
