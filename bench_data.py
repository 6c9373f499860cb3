"""Counter-based synthetic inputs for bench.py: every element is a pure function of (stream, row, column).

Why not ``torch.randn(generator=...)``: the benchmark problem must not depend on the number of GPUs, and a rank must be able
to draw ITS shard (reference rows, query rows, gene columns) without drawing everybody else's -- and, for the parity gate,
any rank must be able to re-draw any row.  A sequential generator gives neither (round 2 drew the whole 12 M x 128 stream on
every rank to keep an eighth of it).  Here element (row, col) of stream ``s`` is a 32-bit integer hash of the three numbers
(two rounds of the "lowbias32" finaliser, int32 wrap-around arithmetic in torch ops on the device), turned into

  * ``uniform``  [0, 1) float32 with 24 random bits,
  * ``normal``   standard normal float32 by Box-Muller from two such uniforms (embeddings: SURVEY.md 8d's
                 ``standard_normal((n, d)).astype(float32)``),
  * ``integers`` in [0, m).

The same function on CPU tensors gives the gloo tests their inputs.  Nothing here is on the product path.
"""
from __future__ import annotations

import math
import os

_C1 = 0x7FEB352D
_C2 = 0x846CA68B - (1 << 32)     # as a wrapped int32
_GOLD = 0x9E3779B1 - (1 << 32)

STREAM_REF, STREAM_QRY, STREAM_LABEL, STREAM_OBSM, STREAM_X, STREAM_XY, STREAM_SPCOL, STREAM_SPVAL = range(1, 9)
STREAM_CENTRE = 9          # cluster centres of the mixture generator (shared by both sides)
STREAM_MEMBER = 16         # + the side's stream: which cluster a row belongs to


def _lsr(x, s: int):
    """logical shift right of an int32 tensor (torch's >> is arithmetic)."""
    return (x >> s) & ((1 << (32 - s)) - 1)


def _mix(x):
    x = x ^ _lsr(x, 16)
    x = x * _C1
    x = x ^ _lsr(x, 15)
    x = x * _C2
    return x ^ _lsr(x, 16)


def hash_grid(torch, stream: int, rows, cols):
    """int32 hash of every (row, col) pair: ``rows`` int32 [n] and ``cols`` int32 [m] -> [n, m]."""
    r = _mix(rows ^ ((stream * 0x632BE5AB) & 0x7FFFFFFF))
    return _mix(r[:, None] + cols[None, :] * _GOLD + stream)


def uniform_block(torch, stream: int, r0: int, r1: int, c0: int, c1: int, device):
    """[r1-r0, c1-c0] float32 uniform on [0, 1): element (i, j) depends on (stream, r0 + i, c0 + j) only."""
    rows = torch.arange(r0, r1, dtype=torch.int32, device=device)
    cols = torch.arange(c0, c1, dtype=torch.int32, device=device)
    h = hash_grid(torch, stream, rows, cols)
    return (h & 0xFFFFFF).to(torch.float32) * (1.0 / 16777216.0)


def normal_block(torch, stream: int, r0: int, r1: int, c0: int, c1: int, device):
    """[r1-r0, c1-c0] float32 standard normal (Box-Muller on two hashed uniforms)."""
    rows = torch.arange(r0, r1, dtype=torch.int32, device=device)
    cols = torch.arange(c0, c1, dtype=torch.int32, device=device)
    h1 = hash_grid(torch, stream, rows, cols)
    h2 = _mix(h1 ^ 0x5BD1E995)
    u1 = ((h1 & 0xFFFFFF).to(torch.float32) + 0.5) * (1.0 / 16777216.0)     # (0, 1)
    u2 = (h2 & 0xFFFFFF).to(torch.float32) * (1.0 / 16777216.0)
    return torch.sqrt(-2.0 * torch.log(u1)) * torch.cos((2.0 * math.pi) * u2)


def fill_rows(torch, kind: str, stream: int, r0: int, r1: int, width: int, device, c0: int = 0, scale: float = 1.0,
              chunk_elems: int = 1 << 25, out=None):
    """Rows [r0, r1) x columns [c0, c0 + width) of a stream as one float32 tensor, generated in row chunks (the int32
    temporaries of a chunk stay below ~1 GB)."""
    fn = normal_block if kind == "normal" else uniform_block
    n = r1 - r0
    if out is None:
        out = torch.empty((n, width), dtype=torch.float32, device=device)
    step = max(1, chunk_elems // max(width, 1))
    for a in range(0, n, step):
        b = min(n, a + step)
        blk = fn(torch, stream, r0 + a, r0 + b, c0, c0 + width, device)
        out[a:b] = blk * scale if scale != 1.0 else blk
    return out


def rows_at(torch, kind: str, stream: int, row_ids, width: int, device, c0: int = 0, scale: float = 1.0):
    """The same stream at an arbitrary list of rows (the parity gate re-draws its sample)."""
    fn_rows = torch.as_tensor(row_ids, dtype=torch.int32, device=device)
    cols = torch.arange(c0, c0 + width, dtype=torch.int32, device=device)
    h1 = hash_grid(torch, stream, fn_rows, cols)
    if kind == "normal":
        h2 = _mix(h1 ^ 0x5BD1E995)
        u1 = ((h1 & 0xFFFFFF).to(torch.float32) + 0.5) * (1.0 / 16777216.0)
        u2 = (h2 & 0xFFFFFF).to(torch.float32) * (1.0 / 16777216.0)
        out = torch.sqrt(-2.0 * torch.log(u1)) * torch.cos((2.0 * math.pi) * u2)
    else:
        out = (h1 & 0xFFFFFF).to(torch.float32) * (1.0 / 16777216.0)
    return out * scale if scale != 1.0 else out


def mixture_centres(torch, n_clusters: int, width: int, device, spread: float = 5.0):
    """[n_clusters, width] float32 cluster centres N(0, spread^2) of the Gaussian-mixture embeddings ("a
    50-component Gaussian mixture to mimic cell-type structure"; the 40-cluster shape of DESIGN.md section 6's table)."""
    return normal_block(torch, STREAM_CENTRE, 0, n_clusters, 0, width, device) * spread


STREAM_RAGGED = 40         # ragged mixtures: cluster weights (+ the side's stream), per-cluster spread, per-coordinate scale


def ragged_params(torch, stream: int, n_clusters: int, width: int, device):
    """Ragged mixture (round 5, late): clusters of unequal size -- log-normal weights, DIFFERENT on the two sides (a query set
    over-represents other cell types than the atlas) --, each with its own spread (0.3 .. 3 x sigma, log-uniform) and a random
    per-coordinate scale shared by all clusters (anisotropy).  Returns (cumulative weights of `stream`, spread factors, scales)."""
    w = torch.exp(1.2 * normal_block(torch, STREAM_RAGGED + stream, 0, 1, 0, n_clusters, device)[0].double())
    cw = torch.cumsum(w / w.sum(), 0)
    u = normal_block(torch, STREAM_RAGGED, 1, 2, 0, n_clusters, device)[0]
    csig = torch.exp((0.5 * (1.0 + torch.erf(u / 2 ** 0.5))) * 2.3 - 1.2)          # uniform through the normal CDF
    aniso = torch.exp(0.5 * normal_block(torch, STREAM_RAGGED, 2, 3, 0, width, device)[0])
    return cw, csig, aniso


def mixture_rows_at(torch, stream: int, row_ids, width: int, device, n_clusters: int = 40, spread: float = 5.0, sigma: float = 1.0,
                    ragged: bool = False):
    """Mixture rows at arbitrary row numbers: row i of `stream` = centre[member(i)] + sigma * N(0, 1) -- a pure function of
    (stream, i, column), like every other stream here, so any rank can draw its shard and the parity gate can re-draw its
    sample.  Reference and query share the centres (a query maps onto the atlas' cell types).  ragged: see ragged_params."""
    rows = torch.as_tensor(row_ids, dtype=torch.int32, device=device)
    cen = mixture_centres(torch, n_clusters, width, device, spread)
    zero = torch.zeros(1, dtype=torch.int32, device=device)
    h = _lsr(hash_grid(torch, STREAM_MEMBER + stream, rows, zero)[:, 0], 8)
    if not ragged:
        member = (h % n_clusters).to(torch.int64)
        return cen[member] + rows_at(torch, "normal", stream, rows, width, device) * sigma
    cw, csig, aniso = ragged_params(torch, stream, n_clusters, width, device)
    u = (h.to(torch.float64) + 0.5) / float(1 << 24)
    member = torch.clamp(torch.searchsorted(cw, u), max=n_clusters - 1)
    return cen[member] + rows_at(torch, "normal", stream, rows, width, device) * (sigma * csig[member])[:, None] * aniso[None, :]


def mixture_rows(torch, stream: int, r0: int, r1: int, width: int, device, n_clusters: int = 40, spread: float = 5.0,
                 sigma: float = 1.0, chunk_rows: int = 500_000, ragged: bool = False):
    """Rows [r0, r1) of the mixture as one float32 tensor, in row chunks."""
    out = torch.empty((r1 - r0, width), dtype=torch.float32, device=device)
    for a in range(r0, r1, chunk_rows):
        b = min(r1, a + chunk_rows)
        out[a - r0:b - r0] = mixture_rows_at(torch, stream, torch.arange(a, b, dtype=torch.int32, device=device), width, device,
                                             n_clusters, spread, sigma, ragged)
    return out


def integers(torch, stream: int, r0: int, r1: int, m: int, device):
    """int32 [r1 - r0] in [0, m): label codes of reference rows [r0, r1)."""
    rows = torch.arange(r0, r1, dtype=torch.int32, device=device)
    cols = torch.zeros(1, dtype=torch.int32, device=device)
    h = hash_grid(torch, stream, rows, cols)[:, 0]
    return (_lsr(h, 8) % m).to(torch.int32)


def sparse_rows(torch, r0: int, r1: int, n_genes: int, density: float, device, s0: int = 0, s1: int | None = None,
                chunk_rows: int = 500_000):
    """CSR float32 payload rows [r0, r1) of SURVEY.md 8d's sparse reference.X: ``per = round(density * n_genes)`` entries per
    row, one per stratum of ``width = n_genes // per`` columns (ascending, distinct), values exp(N(0,1)).  ``[s0, s1)``
    restricts to a range of STRATA = a gene-column slab [s0 * width, s1 * width) with columns renumbered from 0 (gene-column
    sharding across GPUs): entry t of a row is the same number whatever the slab.
    Returns (indptr int64 [n+1], cols int32, vals float32, entries per row, first gene column of the slab)."""
    per = max(1, int(round(density * n_genes)))
    width = n_genes // per
    s1 = per if s1 is None else s1
    n, m = r1 - r0, s1 - s0
    cols = torch.empty((n, m), dtype=torch.int32, device=device)
    vals = torch.empty((n, m), dtype=torch.float32, device=device)
    base = (torch.arange(0, m, device=device, dtype=torch.int32) * width)[None, :]
    strata = torch.arange(s0, s1, dtype=torch.int32, device=device)
    for a in range(0, n, chunk_rows):
        b = min(n, a + chunk_rows)
        rows = torch.arange(r0 + a, r0 + b, dtype=torch.int32, device=device)
        h = hash_grid(torch, STREAM_SPCOL, rows, strata)
        cols[a:b] = base + (_lsr(h, 8) % width).to(torch.int32)
        if os.environ.get("CELLMAP_BENCH_BANKFRIENDLY") and width >= 32:
            # EXPERIMENT (profiles/r04_a3_bank_probe.md): what would the sparse product gain if the entries of a (payload row,
            # column bucket) slice sat in LDS banks that never collide within one update instruction?  Columns are re-drawn
            # inside their strata so that slice position p lands in bank (p // EPL) % 32 (EPL = 3 entries per lane at config 5).
            W = ((n_genes + 7) // 8 + 63) // 64 * 64                         # the kernel's bucket width (cm_a3_bucket_width)
            t = strata.to(torch.int64)
            lo_b, hi_b = (t * width) // W, (t * width + width - 1) // W
            bucket = lo_b                                                      # (a stratum that straddles two buckets keeps column t * width)
            first = torch.zeros(int(bucket.max().item()) + 2, dtype=torch.int64, device=device)
            first.scatter_reduce_(0, bucket, t, reduce="amin", include_self=False)
            pos = t - first[bucket]
            rot = (_lsr(hash_grid(torch, STREAM_SPCOL, rows, strata[:1]), 9) % 32).to(torch.int64)   # [rows, 1]: a bank rotation per payload row
            want = ((pos // 3)[None, :] + rot) % 32                           # keeps the rows' supports different (the product stays ~93 % dense)
            r = (want - (t * width)[None, :]) % 32
            r = torch.where((lo_b == hi_b)[None, :], r, torch.zeros_like(r))
            cols[a:b] = ((t * width)[None, :] + r - s0 * width).to(torch.int32)
        h1 = hash_grid(torch, STREAM_SPVAL, rows, strata)
        h2 = _mix(h1 ^ 0x5BD1E995)
        u1 = ((h1 & 0xFFFFFF).to(torch.float32) + 0.5) * (1.0 / 16777216.0)
        u2 = (h2 & 0xFFFFFF).to(torch.float32) * (1.0 / 16777216.0)
        vals[a:b] = torch.exp(torch.sqrt(-2.0 * torch.log(u1)) * torch.cos((2.0 * math.pi) * u2))
    indptr = torch.arange(n + 1, device=device, dtype=torch.int64) * m
    return indptr, cols.view(-1), vals.view(-1), m, s0 * width
