#!/bin/bash
# Round 5, item 1: reproduce the driver's slow first call.  bench.py is the FIRST command of the lease, without the
# workspace reservation; then the allocator probe alone and next to a monitoring loop; then bench.py as shipped, alone and
# next to a 5-second rocm-smi sampler (what the driver runs beside its bench).
set -o pipefail
cd $GRAFT_REPO_ROOT; O=gpurun_out/fc_r5; mkdir -p $O
CELLMAP_BENCH_RESERVE=0 timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_noreserve_first.json 2> $O/bench_noreserve_first.err || exit 1
timeout -k 10 120 ./build_ab/alloc_probe quiet > $O/alloc_quiet.json 2>&1 || exit 1
( while true; do rocm-smi --showuse --showmemuse --showpower --showtemp --showclocks --json > /dev/null 2>&1; sleep 0.2; done ) &
SMI=$!
timeout -k 10 120 ./build_ab/alloc_probe smi_loop > $O/alloc_smi.json 2>&1
kill $SMI; wait $SMI 2>/dev/null
( while true; do amd-smi metric --json > /dev/null 2>&1; sleep 0.2; done ) &
SMI=$!
timeout -k 10 120 ./build_ab/alloc_probe amdsmi_loop > $O/alloc_amdsmi.json 2>&1
kill $SMI; wait $SMI 2>/dev/null
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_reserve.json 2> $O/bench_reserve.err || exit 1
( while true; do rocm-smi --showuse --showmemuse --showpower --showtemp --showclocks --json > $O/smi_last.json 2>&1; sleep 5; done ) &
SMI=$!
CELLMAP_BENCH_RESERVE=0 timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_noreserve_smi5s.json 2> $O/bench_noreserve_smi5s.err
kill $SMI; wait $SMI 2>/dev/null
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/fc_r5/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f, "ms/step", round(d["ms_per_step"], 1), "first", round(d["first_call_ms"], 1), json.dumps(d["first_call_breakdown_ms"]))
    print("   host leg", d["host_inclusive"]["seconds"], d["host_inclusive"]["first_pass_seconds"], d["host_inclusive"]["first_pass_allocator_ms"])
    print("   dropin", json.dumps(d.get("dropin")))
PY
