"""Synthetic CSR generators (no matrices ship with the reference: SURVEY 4.1).

Small cases are built with numpy on the host (tests, golden fixtures); the BASELINE.json-sized
ones are built with torch directly in HBM (bench.py, full-size GPU tests) -- a 3.2e8-nnz matrix
never exists on the host.

Value/x fills
    "eighths"  k/8 with k in 0..7  -- the reference harness' trick (test_spmv.c:199-202): every
               product and partial sum is exact in fp32 and fp64, so ANY summation order gives
               the same bits and schedules can be compared bitwise.
    "uniform"  uniform(-1, 1)      -- tolerance tests.
    "ones"     1                   -- x of the reference harness.

All generators are deterministic in (shape, seed).
"""
from __future__ import annotations

import numpy as np

__all__ = [
    "CSR", "fill_values", "fill_x", "banded", "uniform_k", "powerlaw", "skewed_rows",
    "from_row_lengths", "with_empty_rows", "dense_rows", "banded_device", "uniform_k_device", "rmat_columns_device",
    "from_row_lengths_device", "skewed_lengths_device", "powerlaw_lengths_device",
]


class CSR:
    """Plain container: rowptr int32[m+1], colidx int32[nnz], val fp[nnz]; m x n."""

    def __init__(self, m, n, rowptr, colidx, val):
        self.m, self.n = int(m), int(n)
        self.rowptr, self.colidx, self.val = rowptr, colidx, val

    @property
    def nnz(self):
        return int(self.rowptr[-1])

    def astype(self, dtype):
        return CSR(self.m, self.n, self.rowptr, self.colidx, self.val.astype(dtype))


# ----------------------------------------------------------------------------- fills (host)
def fill_values(nnz, kind, dtype, seed):
    rng = np.random.default_rng(seed)
    if kind == "eighths":
        return (rng.integers(0, 8, size=nnz) * 0.125).astype(dtype)
    if kind == "uniform":
        return rng.uniform(-1.0, 1.0, size=nnz).astype(dtype)
    if kind == "ones":
        return np.ones(nnz, dtype=dtype)
    raise ValueError(kind)


def fill_x(n, kind, dtype, seed):
    return fill_values(n, kind, dtype, seed ^ 0x5BD1E995)


# ----------------------------------------------------------------------------- structure (host)
def banded(m, n, below=8, above=7, values="uniform", dtype=np.float64, seed=1):
    """Row i holds columns max(0,i-below) .. min(n-1,i+above): BASELINE config 1."""
    i = np.arange(m, dtype=np.int64)
    lo = np.clip(i - below, 0, n)
    hi = np.clip(i + above + 1, 0, n)
    lens = np.maximum(hi - lo, 0)
    rowptr = np.zeros(m + 1, dtype=np.int64)
    np.cumsum(lens, out=rowptr[1:])
    nnz = int(rowptr[-1])
    row_of = np.repeat(i, lens)
    colidx = (np.arange(nnz, dtype=np.int64) - rowptr[row_of] + lo[row_of]).astype(np.int32)
    return CSR(m, n, rowptr.astype(np.int32), colidx, fill_values(nnz, values, dtype, seed))


def from_row_lengths(lens, n, values="uniform", dtype=np.float64, seed=1, local=0):
    """Random distinct sorted columns for given row lengths.  local>0 keeps a row's columns
    within +-local of its diagonal (banded-ish locality), else uniform over [0, n)."""
    lens = np.minimum(np.asarray(lens, dtype=np.int64), n)
    m = lens.shape[0]
    rng = np.random.default_rng(seed)
    rowptr = np.zeros(m + 1, dtype=np.int64)
    np.cumsum(lens, out=rowptr[1:])
    nnz = int(rowptr[-1])
    colidx = np.empty(nnz, dtype=np.int32)
    for r in np.nonzero(lens)[0]:
        k = int(lens[r])
        if local > 0:
            lo = max(0, min(n - 1, int(r * n // max(m, 1))) - local)
            hi = min(n, lo + max(2 * local, k))
            lo = max(0, hi - max(2 * local, k))
            cols = lo + rng.choice(hi - lo, size=k, replace=False)
        elif k * 4 > n:
            cols = rng.choice(n, size=k, replace=False)
        else:
            cols = np.unique(rng.integers(0, n, size=k))
            while cols.shape[0] < k:
                cols = np.unique(np.concatenate([cols, rng.integers(0, n, size=k - cols.shape[0])]))
        colidx[rowptr[r]:rowptr[r + 1]] = np.sort(cols)
    return CSR(m, n, rowptr.astype(np.int32), colidx, fill_values(nnz, values, dtype, seed + 17))


def uniform_k(m, n, k, values="uniform", dtype=np.float64, seed=1):
    """Exactly k distinct sorted random columns per row."""
    return from_row_lengths(np.full(m, k), n, values, dtype, seed)


def powerlaw_lengths(m, mean_len, max_len, alpha, seed):
    rng = np.random.default_rng(seed)
    raw = rng.pareto(alpha, size=m) + 1.0
    lens = np.minimum(np.floor(raw * mean_len * (alpha - 1.0) / alpha), max_len)
    return lens.astype(np.int64)


def powerlaw(m, n, mean_len=3.1, max_len=4700, alpha=1.6, values="uniform", dtype=np.float64, seed=1):
    """Heavy-tailed row lengths incl. empty rows ("webbase-1M-style" stand-in)."""
    return from_row_lengths(powerlaw_lengths(m, mean_len, max_len, alpha, seed), n, values, dtype, seed + 1)


def skewed_lengths(m, seed):
    """90 % rows 8-24 nnz, 9 % 64-256, 1 % 1k-4k."""
    rng = np.random.default_rng(seed)
    u = rng.random(m)
    lens = rng.integers(8, 25, size=m)
    mid = (u >= 0.90) & (u < 0.99)
    big = u >= 0.99
    lens[mid] = rng.integers(64, 257, size=int(mid.sum()))
    lens[big] = rng.integers(1000, 4001, size=int(big.sum()))
    return lens.astype(np.int64)


def skewed_rows(m, n, values="uniform", dtype=np.float32, seed=1):
    return from_row_lengths(skewed_lengths(m, seed), n, values, dtype, seed + 1)


def with_empty_rows(csr, lead=0, trail=0, every=0):
    """Blank out `lead` leading rows, `trail` trailing rows and every `every`-th row."""
    lens = np.diff(csr.rowptr.astype(np.int64))
    keep = np.ones(csr.m, dtype=bool)
    if lead:
        keep[:lead] = False
    if trail:
        keep[csr.m - trail:] = False
    if every:
        keep[::every] = False
    row_of = np.repeat(np.arange(csr.m), lens)
    sel = keep[row_of]
    new_lens = np.where(keep, lens, 0)
    rowptr = np.zeros(csr.m + 1, dtype=np.int64)
    np.cumsum(new_lens, out=rowptr[1:])
    return CSR(csr.m, csr.n, rowptr.astype(np.int32), csr.colidx[sel].copy(), csr.val[sel].copy())


def dense_rows(m, n, dense_at, other_max=3, values="uniform", dtype=np.float64, seed=1):
    """Rows listed in dense_at hold all n columns; the others 0..other_max (SURVEY 4.3 probe)."""
    rng = np.random.default_rng(seed)
    lens = rng.integers(0, other_max + 1, size=m).astype(np.int64)
    for r in dense_at:
        lens[r] = n
    return from_row_lengths(lens, n, values, dtype, seed + 3)


# ----------------------------------------------------------------------------- device builders
def _torch():
    import torch
    return torch


def _fill_device(nnz, kind, dtype, device, seed):
    torch = _torch()
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    if kind == "eighths":
        return torch.randint(0, 8, (nnz,), generator=g, device=device).to(dtype) * 0.125
    if kind == "uniform":
        return torch.rand(nnz, generator=g, device=device, dtype=dtype) * 2.0 - 1.0
    if kind == "ones":
        return torch.ones(nnz, device=device, dtype=dtype)
    raise ValueError(kind)


def banded_device(m, n, k=32, values="uniform", dtype=None, device="cuda", seed=1, row0=0):
    """Exactly k nnz per row: row i holds columns (i - k/2 + j) mod n, j = 0..k-1 (BASELINE config 2,
    variant (i) of SURVEY 8d).  `row0` shifts the diagonal (row i of the shard is global row row0+i),
    which is how the multi-GPU row-block shards are generated without a monolithic matrix."""
    torch = _torch()
    dtype = dtype or torch.float64
    assert (m + 1) * k < 2**31, "int32 RowPtr: shard too large"
    rowptr = torch.arange(0, (m + 1) * k, k, dtype=torch.int32, device=device)
    colidx = torch.empty(m * k, dtype=torch.int32, device=device)
    step = 1 << 22                                   # bound the int64 temporaries
    offs = torch.arange(k, dtype=torch.int64, device=device) - k // 2
    for r0 in range(0, m, step):
        r1 = min(m, r0 + step)
        rows = torch.arange(r0 + row0, r1 + row0, dtype=torch.int64, device=device)
        colidx[r0 * k:r1 * k] = ((rows[:, None] + offs[None, :]) % n).reshape(-1).to(torch.int32)
    val = _fill_device(m * k, values, dtype, device, seed)
    return m, n, rowptr, colidx, val


def banded_holes_device(m, n, k=32, holes=0.25, values="uniform", dtype=None, device="cuda", seed=1, row0=0):
    """Exactly k nnz per row inside a band of width w = round(k / (1 - holes)) around the diagonal: each row keeps k of the band's w
    columns, chosen at random (sorted) -- BASELINE config 2 with the band no longer one run of consecutive columns per row (mean run
    length 1 / holes): what a stencil or FEM band with missing couplings looks like.  Columns wrap like banded_device's."""
    torch = _torch()
    dtype = dtype or torch.float64
    assert (m + 1) * k < 2**31, "int32 RowPtr: shard too large"
    w = max(k, int(round(k / (1.0 - holes))))
    g = torch.Generator(device=device)
    g.manual_seed(seed + 707)
    rowptr = torch.arange(0, (m + 1) * k, k, dtype=torch.int32, device=device)
    colidx = torch.empty(m * k, dtype=torch.int32, device=device)
    step = 1 << 20
    for r0 in range(0, m, step):
        r1 = min(m, r0 + step)
        keys = torch.rand(r1 - r0, w, generator=g, device=device)
        pick = torch.sort(torch.topk(keys, k, dim=1, sorted=False).indices, dim=1).values      # k distinct offsets in [0, w), ascending
        rows = torch.arange(r0 + row0, r1 + row0, dtype=torch.int64, device=device)
        cols = (rows[:, None] + pick - w // 2) % n
        colidx[r0 * k:r1 * k] = torch.sort(cols, dim=1).values.reshape(-1).to(torch.int32)         # the wrapped rows at both ends stay sorted
    val = _fill_device(m * k, values, dtype, device, seed)
    return m, n, rowptr, colidx, val


def stencil27_device(nx, values="uniform", dtype=None, device="cuda", seed=1):
    """27-point stencil on an nx^3 periodic grid: 27 nnz per row in 9 runs of 3 consecutive columns, three far-apart bands."""
    torch = _torch()
    dtype = dtype or torch.float64
    m = nx ** 3
    assert (m + 1) * 27 < 2**31
    offs = torch.tensor([dz * nx * nx + dy * nx + dx for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1)], device=device)
    rowptr = torch.arange(0, (m + 1) * 27, 27, dtype=torch.int32, device=device)
    colidx = torch.empty(m * 27, dtype=torch.int32, device=device)
    for r0 in range(0, m, 1 << 22):
        r1 = min(m, r0 + (1 << 22))
        rows = torch.arange(r0, r1, device=device)
        colidx[r0 * 27:r1 * 27] = torch.sort((rows[:, None] + offs[None, :]) % m, dim=1).values.reshape(-1).to(torch.int32)
    val = _fill_device(m * 27, values, dtype, device, seed)
    return m, m, rowptr, colidx, val


def uniform_k_device(m, n, k=32, values="uniform", dtype=None, device="cuda", seed=1):
    """Exactly k uniformly random columns per row, sorted within the row (variant (ii) of SURVEY 8d).
    Columns are drawn with replacement; a duplicate column inside a row is legal CSR for SpMV."""
    torch = _torch()
    dtype = dtype or torch.float64
    assert (m + 1) * k < 2**31
    g = torch.Generator(device=device)
    g.manual_seed(seed + 101)
    rowptr = torch.arange(0, (m + 1) * k, k, dtype=torch.int32, device=device)
    colidx = torch.empty(m * k, dtype=torch.int32, device=device)
    step = 1 << 22
    for r0 in range(0, m, step):
        r1 = min(m, r0 + step)
        c = torch.randint(0, n, (r1 - r0, k), generator=g, device=device, dtype=torch.int32)
        colidx[r0 * k:r1 * k] = torch.sort(c, dim=1).values.reshape(-1)
    val = _fill_device(m * k, values, dtype, device, seed)
    return m, n, rowptr, colidx, val


def rmat_columns_device(row_of, m, n, device="cuda", seed=1, a=0.57, b=0.19, c=0.19, d=0.05):
    """R-MAT column generator (Chakrabarti et al.): the column of every entry is drawn bit by bit, each bit
    conditioned on the corresponding bit of the entry's ROW position as in the recursive-matrix model --
    P(col bit = 0 | row bit = 0) = a / (a + b), P(col bit = 1 | row bit = 1) = d / (c + d).  With the Graph500
    parameters this gives what uniform columns lack and real social / web graphs have: hub columns (low
    indices are hot) and community structure (rows of one 2^-k fraction of the matrix prefer the columns
    of the same fraction), which is what decides whether x windows stage and how well gathers merge.
    row_of: int64 row index of every entry.  Returns int32 columns in [0, n)."""
    torch = _torch()
    g = torch.Generator(device=device)
    g.manual_seed(seed + 31)
    scale = max(1, int(n - 1).bit_length())
    rpos = (row_of.to(torch.int64) << scale) // max(int(m), 1)          # row position on the 2^scale grid
    col = torch.zeros_like(rpos)
    p0, p1 = a / (a + b), d / (c + d)
    for level in range(scale - 1, -1, -1):
        rbit = (rpos >> level) & 1
        u = torch.rand(rpos.shape[0], generator=g, device=device)
        cbit = torch.where(rbit == 0, u >= p0, u < p1)
        col |= cbit.to(torch.int64) << level
    return ((col * int(n)) >> scale).to(torch.int32)                    # squeeze the power-of-two grid onto n columns


def from_row_lengths_device(lens, n, values="uniform", dtype=None, device="cuda", seed=1, local=0, cols="uniform", row0=0, m_total=None):
    """Random columns for given per-row lengths (int64 tensor on `device`); columns sorted per row
    when local == 0 is not required by SpMV and is skipped.  local>0: columns within +-local of
    the row's diagonal position.  cols = "rmat": R-MAT columns (rmat_columns_device), sorted within each row.
    row0 / m_total: the rows are rows row0 .. of an m_total-row matrix (a row block generated on its own: the
    position-dependent column models -- local, rmat, web -- then see the GLOBAL row position)."""
    torch = _torch()
    dtype = dtype or torch.float64
    lens = lens.to(device=device, dtype=torch.int64).clamp_(max=n)
    m = lens.shape[0]
    mt = int(m_total) if m_total is not None else m
    rp = torch.zeros(m + 1, dtype=torch.int64, device=device)
    torch.cumsum(lens, 0, out=rp[1:])
    nnz = int(rp[-1].item())
    assert nnz < 2**31, "int32 RowPtr overflow"
    g = torch.Generator(device=device)
    g.manual_seed(seed + 7 + 1000003 * int(row0))
    if local > 0:
        row_of = torch.repeat_interleave(torch.arange(m, device=device, dtype=torch.int64), lens) + int(row0)
        centre = row_of * n // max(mt, 1)
        jitter = torch.randint(-local, local + 1, (nnz,), generator=g, device=device, dtype=torch.int64)
        colidx = (centre + jitter).clamp_(0, n - 1).to(torch.int32)
    elif cols in ("rmat", "web"):
        row_of = torch.repeat_interleave(torch.arange(m, device=device, dtype=torch.int64), lens) + int(row0)
        colidx = rmat_columns_device(row_of, mt, n, device, seed + 1000003 * int(row0))
        if cols == "web":   # web-graph-like: 90 % of a row's links stay near its own position ("same host"), 10 % follow the R-MAT hubs
            centre = row_of * n // max(mt, 1)
            near = (centre + torch.randint(-2000, 2001, (nnz,), generator=g, device=device, dtype=torch.int64)).clamp_(0, n - 1)
            keep = torch.rand(nnz, generator=g, device=device) < 0.1
            colidx = torch.where(keep, colidx.to(torch.int64), near).to(torch.int32)
            del centre, near, keep
        key = (row_of << 32) | colidx.to(torch.int64)                   # sort columns inside every row
        del row_of
        colidx = (torch.sort(key).values & 0xFFFFFFFF).to(torch.int32)
        del key
    else:
        colidx = torch.randint(0, n, (nnz,), generator=g, device=device, dtype=torch.int32)
    val = _fill_device(nnz, values, dtype, device, seed)
    return m, n, rp.to(torch.int32), colidx, val


def skewed_lengths_device(m, device="cuda", seed=1):
    """Device twin of skewed_lengths (config 4)."""
    torch = _torch()
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    u = torch.rand(m, generator=g, device=device)
    lens = torch.randint(8, 25, (m,), generator=g, device=device, dtype=torch.int64)
    mid = torch.randint(64, 257, (m,), generator=g, device=device, dtype=torch.int64)
    big = torch.randint(1000, 4001, (m,), generator=g, device=device, dtype=torch.int64)
    lens = torch.where((u >= 0.90) & (u < 0.99), mid, lens)
    lens = torch.where(u >= 0.99, big, lens)
    return lens


def powerlaw_lengths_device(m, mean_len=3.1, max_len=4700, alpha=1.6, device="cuda", seed=1):
    """Device twin of powerlaw_lengths (config 3 stand-ins)."""
    torch = _torch()
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    u = torch.rand(m, generator=g, device=device, dtype=torch.float64).clamp_(min=1e-12)
    raw = u.pow(-1.0 / alpha)
    lens = torch.floor(raw * mean_len * (alpha - 1.0) / alpha).clamp_(max=max_len)
    return lens.to(torch.int64)
