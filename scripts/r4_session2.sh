#!/bin/bash
# Round 4, GPU session 2: host-leg breakdown with and without the pinned staging ring, correctness of the ring at small sizes.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4_s2; mkdir -p $O
cd $R
HL_FETCH=1 timeout -k 10 200 python scripts/host_leg_probe.py > $O/host_leg_ring.json 2> $O/host_leg.err || echo "ring rc=$?"
CELLMAP_NO_XFER=1 HL_FETCH=1 timeout -k 10 200 python scripts/host_leg_probe.py > $O/host_leg_plain.json 2>> $O/host_leg.err || echo "plain rc=$?"
python - <<'PY'
import json, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r4_s2/"
for f in ("host_leg_ring.json", "host_leg_plain.json"):
    d = json.loads(open(O + f).read().strip().splitlines()[-1])
    print(f, "fetch_knn", d.get("fetch_knn_ms"), "fetch_mapping", d.get("fetch_mapping_ms"), d.get("fetch_checksum"))
    for r in d["iterations_ms"]:
        print("  ", {k: round(v, 2) for k, v in r.items()})
PY
CELLMAP_XFER_MIN_KB=1 CELLMAP_XFER_SLOT_KB=64 CELLMAP_XFER_THREADS=3 timeout -k 10 600 python -m pytest tests/test_gpu_boundary.py tests/test_gpu_streaming.py -x -q -m gpu > $O/tests_ring_small.log 2>&1 || echo "tests rc=$?"
tail -5 $O/tests_ring_small.log
