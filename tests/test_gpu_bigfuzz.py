"""GPU differential fuzz at sizes where the big-matrix-only machinery engages: create-time autotune
(>= 2^24 nnz), tile groups grown to 32 / 64 tiles, the wide x-window form, the long-row sub-matrix with
thousands of rows, the cache-blocked executor with tens of thousands of column slabs.  Exact arithmetic
(multiples of 1/8), so every schedule must reproduce a torch fp64 evaluation of the definition bit for bit."""
import pytest
import torch

from spmv_amd import api, build, synth

pytestmark = pytest.mark.gpu
M = api.SPMV_METHODS
DEV = "cuda:0"
METHODS = [M.Method_Parallel, M.Method_Balanced, M.Method_Balanced2, M.Method_Balanced_Yid, M.Method_SellCSigma, M.Method_CSR5SPMV]


@pytest.fixture(scope="module", autouse=True)
def _lib():
    build.build()
    api.load()


def _definition(rp, ci, va, x):
    """Exact fp64 prefix sums differenced at RowPtr (all partial sums are multiples of 1/64 below 2^40)."""
    prod = va.double() * x[ci.long()].double()
    cs = torch.zeros(ci.numel() + 1, dtype=torch.float64, device=va.device)
    torch.cumsum(prod, 0, out=cs[1:])
    r = rp.long()
    return (cs[r[1:]] - cs[r[:-1]]).to(va.dtype)


def _lengths(kind, m, g):
    u = torch.rand(m, generator=g, device=DEV)
    if kind == "short":
        return torch.randint(0, 12, (m,), generator=g, device=DEV)
    if kind == "equal":
        return torch.full((m,), 27, device=DEV, dtype=torch.int64)
    if kind == "skewed":
        lens = torch.randint(4, 30, (m,), generator=g, device=DEV)
        lens = torch.where(u > 0.93, torch.randint(60, 300, (m,), generator=g, device=DEV), lens)
        return torch.where(u > 0.995, torch.randint(900, 6000, (m,), generator=g, device=DEV), lens)
    lens = torch.randint(1, 40, (m,), generator=g, device=DEV)          # "gaps": runs of empty rows
    lens[(torch.arange(m, device=DEV) // 5000) % 7 == 3] = 0
    return lens


CASES = [  # (rows, columns, row-length kind, column locality, dtype)
    (1_500_000, 1_500_000, "skewed", 4096, torch.float64),    # wide windows: CSR5 groups of 64, CSR-vector wide form
    (1_500_000, 1_500_000, "skewed", 4096, torch.float32),
    (2_000_000, 2_000_000, "equal", 40, torch.float64),       # narrow windows, autotune (5.4e7 nnz)
    (3_000_000, 3_000_000, "short", 0, torch.float64),        # no locality, x = 24 MB: cache-blocked Balanced family
    (4_500_000, 4_500_000, "gaps", 0, torch.float32),         # no locality + runs of empty rows (x = 18 MB)
    (1_200_000, 5_000_000, "gaps", 600, torch.float64),       # n != m
    (4_000_000, 4_000_000, "equal", "web", torch.float64),    # 90 % of a row near the diagonal, 10 % on R-MAT hubs: A = A_near + A_far is built and timed
    (2_500_000, 2_500_000, "skewed", "rmat", torch.float32),  # R-MAT columns, heavy-tailed rows: hub cells of thousands of entries (blk_spread), fp32 groups of 256
    (3_000_000, 3_000_000, "equal", "every7", torch.float64), # banded rows, every seventh row random: no tile stages, the split goes by entries
    (2_000_000, 2_000_000, "gaps", "runs", torch.float64),    # rows = runs of consecutive columns near the diagonal, 1 row in 20 000 broken: RUN tiles / window groups beside ordinary ones
    (1_000_000, 1_000_000, "skewed", "runs", torch.float32),  # the same with 60..6000-entry rows among short ones: CSR5 RUN groups (one row start per lane) beside ordinary ones, long-row sub-matrices of runs
]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_big_shapes_every_schedule_matches_the_definition(case):
    m, n, kind, local, dt = CASES[case]
    g = torch.Generator(device=DEV)
    g.manual_seed(100 + case)
    lens = _lengths(kind, m, g).to(torch.int64)
    if local == "every7":
        _, _, rp, ci, va = synth.from_row_lengths_device(lens, n, "eighths", dt, DEV, seed=200 + case, local=20)
        _, _, _, cr, _ = synth.from_row_lengths_device(lens, n, "eighths", dt, DEV, seed=300 + case, local=0)
        row_of = torch.repeat_interleave(torch.arange(m, device=DEV), lens)
        ci = torch.where(row_of % 7 == 0, cr, ci)
        del cr, row_of
    elif local == "runs":
        _, _, rp, ci, va = synth.from_row_lengths_device(lens, n, "eighths", dt, DEV, seed=200 + case, local=20)
        rows = torch.arange(m, device=DEV)
        start = (rows * n // m + torch.randint(-300, 301, (m,), generator=g, device=DEV)).clamp_(min=0)
        start = torch.minimum(start, (n - lens).clamp_(min=0))
        row_of = torch.repeat_interleave(rows, lens)
        ci = (start[row_of] + torch.arange(ci.numel(), device=DEV) - rp.long()[:-1][row_of]).to(torch.int32)
        pick = torch.nonzero((rows % 20000 == 11) & (lens >= 2)).flatten()
        ci[rp.long()[pick] + 1] = ci[rp.long()[pick]]                       # second entry = first: a duplicate column, no longer a run
        del rows, start, row_of, pick
    elif isinstance(local, str):
        _, _, rp, ci, va = synth.from_row_lengths_device(lens, n, "eighths", dt, DEV, seed=200 + case, cols=local)
    else:
        _, _, rp, ci, va = synth.from_row_lengths_device(lens, n, "eighths", dt, DEV, seed=200 + case, local=local)
    x = (torch.randint(-8, 9, (n,), generator=g, device=DEV).to(dt) * 0.125)
    want = _definition(rp, ci, va, x)
    seen = set()
    runs_seen = 0
    for method in METHODS:
        y = torch.full((m,), float("nan"), dtype=dt, device=DEV)
        with api.Handle(m, n, rp, ci, va, method) as h:
            h.spmv(x, y)
            info = h.info()
            torch.cuda.synchronize()
            seen.add(info["kernel_name"])
            runs_seen += info["run_nnz"] > 0
            if info["far_nnz"] > 0:
                seen.add("split")
            bad = torch.nonzero(y != want)
            assert bad.numel() == 0, (case, method.name, info["kernel_name"], int(bad[0]), float(y[bad[0]]), float(want[bad[0]]))
            if isinstance(local, str) and method in (M.Method_Parallel, M.Method_CSR5SPMV):   # values refreshed in place: blocked streams, both halves of a split
                h.update_values(va * 2)
                h.spmv(x, y)
                torch.cuda.synchronize()
                assert torch.equal(y, 2 * want), (case, method.name, "update_values")
    if local == "runs":
        assert runs_seen, "no schedule found a RUN tile / group"
    if local == "every7":
        assert "split" in seen or "blk_kernel" in seen or "blk_wide_kernel" in seen, seen
    if local == 0:
        assert "blk_kernel" in seen or "blk_wide_kernel" in seen, seen           # the Balanced family found no x window to stage
