"""bench_data.py: the counter-based synthetic streams are pure functions of (stream, row, column) -- any rank can draw its shard,
and a parity gate can re-draw its sample rows.  CPU tests (torch on the host)."""
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch = pytest.importorskip("torch")
import bench_data as bd  # noqa: E402


@pytest.mark.parametrize("ragged", [False, True])
def test_mixture_rows_are_a_function_of_the_row_number(ragged):
    dev = torch.device("cpu")
    full = bd.mixture_rows(torch, bd.STREAM_QRY, 0, 5000, 50, dev, 40, 5.0, 0.5, chunk_rows=1700, ragged=ragged)
    ids = torch.tensor([0, 7, 1699, 1700, 1701, 3400, 4999], dtype=torch.int32)
    again = bd.mixture_rows_at(torch, bd.STREAM_QRY, ids, 50, dev, 40, 5.0, 0.5, ragged)
    assert torch.equal(full[ids.long()], again)
    shard = bd.mixture_rows(torch, bd.STREAM_QRY, 2500, 5000, 50, dev, 40, 5.0, 0.5, ragged=ragged)
    assert torch.equal(full[2500:], shard)


def test_ragged_mixture_has_unequal_clusters_with_their_own_spread():
    dev = torch.device("cpu")
    cw_r, csig, aniso = bd.ragged_params(torch, bd.STREAM_REF, 40, 50, dev)
    cw_q, _, _ = bd.ragged_params(torch, bd.STREAM_QRY, 40, 50, dev)
    w = torch.diff(cw_r, prepend=torch.zeros(1, dtype=cw_r.dtype))
    assert abs(float(cw_r[-1]) - 1.0) < 1e-12 and float(w.max()) > 3.0 * float(w.min())
    assert not torch.allclose(cw_r, cw_q)                       # the two sides weigh the clusters differently
    assert 0.3 <= float(csig.min()) and float(csig.max()) <= 3.01 and float(csig.max()) > 2.0 * float(csig.min())
    assert aniso.shape == (50,) and float(aniso.max()) > 1.5 * float(aniso.min())
