"""GPU: locality routing on matrices that are local in PART.  Banded rows and uniformly-random rows in one matrix,
32 nnz/row, exact data (so every executor must equal the definition bit for bit):
  prefix1   the first 1 % of the rows banded, the rest random   -> the blocked executor must take it (round 2: one stageable tile
            group kept the whole matrix on the 6 ms gather-bound kernels) and cost about what the pure random matrix costs;
  tail10    banded, the last 10 % of the rows random            -> A = A_near + A_far: tile schedule + blocked executor (split.hpp);
  every10   banded, every tenth row random: NO 256-row tile stages its x window -> the same split, by entries.
x = 32 MB (4e6 rows): several times an XCD's L2, the regime the routing is for."""
import numpy as np
import pytest
import torch

from spmv_amd import api, build, synth

pytestmark = pytest.mark.gpu
M = api.SPMV_METHODS
DEV = "cuda:0"
ROWS, K = 4_000_000, 32


@pytest.fixture(scope="module", autouse=True)
def _lib():
    build.build()
    api.load()


def _mixed(kind):
    m = ROWS
    _, _, rp, cb, va = synth.banded_device(m, m, K, "eighths", torch.float64, DEV, 1)
    if kind != "banded":
        _, _, _, cr, _ = synth.uniform_k_device(m, m, K, "eighths", torch.float64, DEV, 1)
        rows = torch.arange(m, device=DEV)
        rnd = {"random": rows >= 0, "prefix1": rows >= m // 100, "tail10": rows >= m - m // 10, "every10": rows % 10 == 0}[kind]
        cb = torch.where(rnd.repeat_interleave(K), cr, cb)
    return rp, cb, va


def _run(rp, ci, va, method, x, iters=20, split=1):
    keep = api.get_option("split")
    api.set_option("split", split)
    try:
        h = api.Handle(ROWS, ROWS, rp, ci, va, method)
    finally:
        api.set_option("split", keep)
    y = torch.full((ROWS,), float("nan"), dtype=torch.float64, device=DEV)
    _, ms = api.time_launches(h.h, x, y, 3, iters)
    return h, y, float(ms.min())


def _definition(ci, va, x):
    return (va * x[ci.long()]).view(ROWS, K).sum(1)


def _timing_ok(cond, what):
    """Wall-clock ratios between two separately built, separately tuned handles depend on the box and on who else is on it:
    they are checked only under SPMV_TEST_TIMING=1; otherwise a miss is a warning.  Bit-exactness and the structural asserts stay mandatory."""
    import os, warnings
    if cond:
        return
    if os.environ.get("SPMV_TEST_TIMING") == "1":
        raise AssertionError(what)
    warnings.warn(f"timing expectation missed (not enforced without SPMV_TEST_TIMING=1): {what}")


@pytest.fixture(scope="module")
def x():
    g = torch.Generator(device=DEV); g.manual_seed(4)
    return (torch.randint(-8, 9, (ROWS,), generator=g, device=DEV) * 0.125).to(torch.float64)


@pytest.fixture(scope="module")
def pure_ms(x):
    out = {}
    for kind in ("banded", "random"):
        rp, ci, va = _mixed(kind)
        h, y, t = _run(rp, ci, va, M.Method_Parallel, x)
        assert torch.equal(y, _definition(ci, va, x))
        assert h.info()["cache_blocked"] == (1 if kind == "random" else 0)
        h.close()
        out[kind] = t
    return out


@pytest.mark.parametrize("method", [M.Method_Parallel, M.Method_CSR5SPMV], ids=lambda m: m.name)
def test_random_columns_behind_a_one_percent_banded_prefix(method, x, pure_ms):
    rp, ci, va = _mixed("prefix1")
    h, y, t = _run(rp, ci, va, method, x)
    try:
        info = h.info()
        assert torch.equal(y, _definition(ci, va, x))
        assert info["cache_blocked"] == 1 or info["far_nnz"] > 0.9 * info["nnz"], info      # the random 99 % run on the blocked executor
        # measured 1.24 x at 1e7 rows: the one block that straddles the end of the banded prefix sweeps the column slabs out of step with
        # the others, runs 1.25 x longer, and so do its neighbour on the CU and their two successors (tools/blk_timeline.py)
        _timing_ok(t <= 1.45 * pure_ms["random"], (t, pure_ms, info["split_ms"]))
    finally:
        h.close()


@pytest.mark.parametrize("method", [M.Method_Parallel, M.Method_CSR5SPMV, M.Method_SellCSigma, M.Method_Balanced2], ids=lambda m: m.name)
@pytest.mark.parametrize("kind", ["tail10", "every10"])
def test_banded_matrix_with_ten_percent_random_rows_is_split(kind, method, x, pure_ms):
    rp, ci, va = _mixed(kind)
    want = _definition(ci, va, x)
    h0, y0, t0 = _run(rp, ci, va, method, x, split=0)
    assert torch.equal(y0, want)
    h0.close()
    h, y, t = _run(rp, ci, va, method, x)
    try:
        info = h.info()
        assert torch.equal(y, want)
        assert info["split_ms"][0] > 0 and info["split_ms"][1] > 0, info                      # create() built and timed the pair
        if info["far_nnz"] > 0:                                                                 # ... and kept it: ~10 % of the entries are far
            assert 0.08 * info["nnz"] <= info["far_nnz"] <= 0.13 * info["nnz"], info
            _timing_ok(t <= 1.08 * t0, (t, t0))                                                        # create() kept it for >= 10 % on its own clock; here: not slower (two timings of two builds: margin)
            h.update_values(va * 2)                                                             # both halves refreshed in place
            h.spmv(x, y)
            torch.cuda.synchronize()
            assert torch.equal(y, 2 * want)
        else:
            assert info["split_ms"][1] >= 0.9 * info["split_ms"][0], info                      # rejected only because it was not faster
        _timing_ok(t <= 1.15 * t0, (t, t0, pure_ms))                                                 # not slower than the unsplit handle (rejected: the same schedule built twice, forms tuned separately)
    finally:
        h.close()


def test_split_off_gives_the_unsplit_executors(x):
    rp, ci, va = _mixed("every10")
    h, y, _ = _run(rp, ci, va, M.Method_CSR5SPMV, x, split=0)
    try:
        info = h.info()
        assert info["far_nnz"] == 0 and info["split_ms"] == [0.0, 0.0] and torch.equal(y, _definition(ci, va, x))
    finally:
        h.close()


@pytest.mark.parametrize("family", ["tile", "csr5", "blocked"])
def test_inspector_time_is_bounded_per_nonzero(family):
    """the inspector's cost is reported (spmv_hip_info.inspect_ms) -- and bounded here.  Every inspector is a handful of device passes
    over the matrix plus the create-time timing of a few executor forms: under 0.5 ns per non-zero + 40 ms of fixed cost (allocations, ~30 timed
    launches) at 6.4e7 non-zeros for each executor family -- CSR-vector tiles (windows, slot streams, autotune), CSR5 (transposes, descriptors,
    windows) and row blocks x column slabs (two layouts built and timed).  Measured: 0.06 / 0.09 / 0.25 ns per non-zero at 3.2e8."""
    m, k = 2_000_000, 32
    if family == "blocked":
        _, _, rp, ci, va = synth.uniform_k_device(m, m, k, "eighths", torch.float64, DEV, seed=5)
        method = M.Method_Parallel
    else:
        _, _, rp, ci, va = synth.banded_device(m, m, k, "eighths", torch.float64, DEV, seed=5)
        method = M.Method_Parallel if family == "tile" else M.Method_CSR5SPMV
    with api.Handle(m, m, rp, ci, va, method) as h:      # first create of the process pays allocator warm-up: measure the second
        pass
    with api.Handle(m, m, rp, ci, va, method) as h:
        info = h.info()
    nnz = m * k
    assert info["cache_blocked"] == (1 if family == "blocked" else 0)
    assert info["inspect_ms"] <= 0.5e-6 * nnz + 40.0, (family, info["inspect_ms"], info["kernel_name"])
