"""The bench.py lines the driver cannot take on a one-GPU box, run as child processes on the GPU box:

* BASELINE configs[3] AS SPECIFIED in size (1.5 M queries x 12 M reference rows, d = 128, k = 30) on one GPU, with the
  parity gate on rows spread over all queries (round 2 exercised 200 k queries and a 96-row oracle sample);
* the N > 1 code path of configs 2, 3 and 5 with ranks sharing the GPU over gloo (RCCL refuses two ranks per device):
  launcher, 2-D layout, packed exchange with its first-step self-check, exact merge, all-reduced bandwidth, replicated and
  gene-column-sharded aggregation -- every line must carry `parity_check.ok` and the layer checksum must equal the N = 1 run's.
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args, timeout=600):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=timeout)
    assert r.returncode == 0, f"bench.py {' '.join(args)} -> {r.returncode}\n{r.stderr[-3000:]}"
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    return json.loads(line)


def test_config4_at_its_full_size_on_one_gpu():
    import torch

    free_b, _ = torch.cuda.mem_get_info(0)
    if free_b < (40 << 30):
        pytest.skip("needs ~30 GB of device memory")
    r = _bench("--config", "4", "--steps", "1", "--warmup", "1", "--check", "1024", "--cpu-sample", "0", timeout=900)
    assert r["n_gpus"] == 1 and r["config"]["nq"] == 1_500_000 and r["config"]["n_ref"] == 12_000_000 and r["config"]["d"] == 128
    pc = r["parity_check"]
    assert pc["ok"] and pc["indices_bit_exact"] and pc["distances_bit_exact"] and pc["queries"] >= 256, pc
    assert r["roofline"]["frac"] > 0.35, r["roofline"]


@pytest.mark.parametrize("n,ref_shards", [(2, 0), (4, 0), (4, 2), (2, 2)])
def test_sharded_label_transfer_line_is_self_checked(n, ref_shards):
    """ref_shards = 0: the default layout (round 5: query groups only, the reference replicated -- no exchange in the headline)
    with the reference-sharded secondary leg, whose packed exchange checks itself and whose merged lists must equal the
    headline layout's bit for bit; ref_shards > 0: the exchange in the headline path (2-D / pure reference sharding)."""
    args = ["--steps", "1", "--warmup", "1", "--nq", "120000", "--nref", "150000", "--check", "512", "--cpu-sample", "512",
            "--no-host-leg", "--backend", "gloo", "--share-gpu", "--master-port", str(29600 + n + 10 * ref_shards)]
    if ref_shards:
        args += ["--ref-shards", str(ref_shards)]
    r = _bench("--gpus", str(n), *args)
    assert r["n_gpus"] == n and r["parity_check"]["ok"] and r["parity_check"]["ranks_sampled"] == n, r.get("parity_check")
    assert r["cpu_baseline"]["cores"] >= 1
    if ref_shards:
        assert r["config"]["parallelism"] == f"q{n // ref_shards}xr{ref_shards}" and r["config"]["exchange"]["self_check"] == "passed", r["config"]
    else:
        assert r["config"]["parallelism"] == f"q{n}xr1" and r["config"]["exchange"] is None, r["config"]
        rs = r["reference_sharded"]
        assert rs["parallelism"] == f"q1xr{n}" and rs["exchange"]["self_check"] == "passed" and rs["lists_and_labels_equal_headline_layout"] is True, rs


def test_sharded_layer_transfers_equal_the_one_gpu_run():
    c3 = ["--config", "3", "--steps", "1", "--warmup", "1", "--nq", "60000", "--nref", "80000", "--genes", "1500", "--check", "128",
          "--cpu-sample", "128"]
    c5 = ["--config", "5", "--steps", "1", "--warmup", "1", "--nq", "100000", "--genes", "2400", "--check", "128", "--cpu-sample", "128"]
    for base, port in ((c3, 29611), (c5, 29613)):
        one = _bench(*base)
        two = _bench("--gpus", "2", *base, "--backend", "gloo", "--share-gpu", "--master-port", str(port))
        assert one["parity_check"]["ok"] and two["parity_check"]["ok"], (one["parity_check"], two["parity_check"])
        assert two["n_gpus"] == 2 and two["layer_checksum"]["value"] == one["layer_checksum"]["value"], (one["layer_checksum"], two["layer_checksum"])
        assert "cpu_baseline" in one and "cpu_baseline" in two
