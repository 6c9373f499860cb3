#!/bin/bash
# Round 4, GPU session 3: pinned-allocation cost by size, host leg after the device-side NaN/inf screen, new boundary tests.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4_s3; mkdir -p $O
cd $R
timeout -k 10 200 ./build_ab/h2d_probe 300 > $O/h2d_probe.log 2>&1 || echo "probe rc=$?"
head -12 $O/h2d_probe.log
HL_FETCH=1 timeout -k 10 200 python scripts/host_leg_probe.py > $O/host_leg_ring.json 2> $O/host_leg.err || echo "ring rc=$?"
CELLMAP_NO_XFER=1 HL_FETCH=1 timeout -k 10 200 python scripts/host_leg_probe.py > $O/host_leg_plain.json 2>> $O/host_leg.err || echo "plain rc=$?"
python - <<'PY'
import json, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r4_s3/"
for f in ("host_leg_ring.json", "host_leg_plain.json"):
    d = json.loads(open(O + f).read().strip().splitlines()[-1])
    print(f, "fetch_knn", d.get("fetch_knn_ms"), "fetch_mapping", d.get("fetch_mapping_ms"), d.get("fetch_checksum"))
    for r in d["iterations_ms"]:
        print("  ", {k: round(v, 2) for k, v in r.items()})
PY
timeout -k 10 600 python -m pytest tests/test_gpu_boundary.py -x -q -m gpu -k "non_finite or staging_ring or staged_flow" > $O/tests_new.log 2>&1 || echo "tests rc=$?"
tail -15 $O/tests_new.log
