"""GPU parity: every HIP schedule against the golden vectors captured from the reference's
Method_Serial and against the oracle, through the C ABI (spmv_amd.api is a ctypes mirror of it).

Bars (north_star): exact-arithmetic ("eighths") inputs -> BIT-EXACT for every schedule;
random inputs -> |y - y_exact| <= tol * sum_j |a_ij x_j| per row with tol = 1e-6 (fp64) /
1e-3 (fp32); additionally a sharper sanity bound of 64 ulp-units of the row magnitude."""
import json
import os

import numpy as np
import pytest

import oracle
from conftest import GOLDEN, load_golden
from spmv_amd import api, build, synth

pytestmark = pytest.mark.gpu

M = api.SPMV_METHODS
with open(os.path.join(GOLDEN, "manifest.json")) as _f:
    NAMES = sorted(json.load(_f)["cases"].keys())

ALL_METHODS = [M.Method_Serial, M.Method_Parallel, M.Method_Balanced, M.Method_Balanced2,
               M.Method_Balanced_Yid, M.Method_SellCSigma, M.Method_CSR5SPMV]
TOL = {np.dtype(np.float64): 1e-6, np.dtype(np.float32): 1e-3}          # north_star
SHARP = {np.dtype(np.float64): 64 * 2.3e-16, np.dtype(np.float32): 64 * 1.2e-7}


@pytest.fixture(scope="module", autouse=True)
def _lib():
    build.build()
    lib = api.load()
    assert lib.spmv_hip_device_count() > 0, "GPU tests need a device"
    return lib


def run_host(csr, x, method, way=api.VECTORIZED_WAY.VECTOR_HIP, nthreads=1):
    """Reference-harness call sequence with HOST arrays (test_spmv.c:88-104)."""
    y = np.full(csr.m, np.nan, dtype=csr.val.dtype)
    h = api.spmv_create_handle_all_in_one(csr.m, csr.n, csr.rowptr, csr.colidx, csr.val, nthreads, method,
                                          csr.val.dtype.itemsize, way, "golden")
    actual = h.contents.spmvMethod
    api.spmv(h, csr.m, csr.rowptr, csr.colidx, csr.val, x, y)
    api.spmv_destory_handle(h)
    return y, actual


def check(y, csr, x, y_ref, exact):
    assert not np.isnan(y).any(), f"{int(np.isnan(y).sum())} rows left unwritten"
    if exact:
        assert np.array_equal(y.view(np.uint8), y_ref.view(np.uint8))
        return
    ye = oracle.spmv_exact(csr, x)
    s = oracle.row_abs_sum(csr, x)
    err = np.abs(y.astype(np.float64) - ye)
    assert (err <= TOL[y.dtype] * s + 1e-300).all(), float((err / np.maximum(s, 1e-300)).max())
    assert (err <= SHARP[y.dtype] * np.maximum(1, np.diff(csr.rowptr)) * s + 1e-300).all()
    # ... and against the bits the REFERENCE's Method_Serial produced for this fixture (tests/golden, y_ref):
    # north_star's bar, |y - y_ref| <= tol * sum_j |a_ij x_j| per row
    err_ref = np.abs(y.astype(np.float64) - y_ref.astype(np.float64))
    assert (err_ref <= TOL[y.dtype] * s + 1e-300).all(), float((err_ref / np.maximum(s, 1e-300)).max())


@pytest.mark.parametrize("method", ALL_METHODS, ids=lambda m: m.name)
@pytest.mark.parametrize("name", NAMES)
def test_golden_host_pointers(name, method):
    csr, x, y_ref = load_golden(name)
    y, _ = run_host(csr, x, method)
    check(y, csr, x, y_ref, exact=name.endswith("eighths"))


@pytest.mark.parametrize("method", ALL_METHODS, ids=lambda m: m.name)
@pytest.mark.parametrize("name", ["banded_f64_eighths", "powerlaw_f32_eighths", "empty_mix_f64_uniform",
                                  "dense_row0_f32_uniform", "skewed_f32_eighths", "nnz0_f64_uniform"])
def test_golden_device_pointers(name, method):
    import torch
    csr, x, y_ref = load_golden(name)
    dev = torch.device("cuda:0")
    rp, ci = torch.from_numpy(csr.rowptr).to(dev), torch.from_numpy(csr.colidx).to(dev)
    va, xd = torch.from_numpy(csr.val).to(dev), torch.from_numpy(x).to(dev)
    yd = torch.full((csr.m,), float("nan"), dtype=va.dtype, device=dev)
    with api.Handle(csr.m, csr.n, rp, ci, va, method) as h:
        h.spmv(xd, yd)
        h.spmv(xd, yd)  # idempotent
    torch.cuda.synchronize()
    check(yd.cpu().numpy(), csr, x, y_ref, exact=name.endswith("eighths"))


@pytest.mark.parametrize("method", [M.Method_Parallel, M.Method_Balanced, M.Method_Balanced2, M.Method_Balanced_Yid, M.Method_SellCSigma,
                                    M.Method_CSR5SPMV], ids=lambda m: m.name)
@pytest.mark.parametrize("name", NAMES)
def test_golden_row_block_column_slab_executor(name, method):
    """Option cache_block = 2 forces the executor that big matrices without column locality get
    automatically, whatever the method (kernels/blocked.hpp).  One wavefront owns a row block and walks a stream
    whose order the inspector fixes, so the result is bit-exact on exact-arithmetic inputs, within the north_star
    tolerance otherwise, and THE SAME BITS from handle to handle."""
    csr, x, y_ref = load_golden(name)
    api.set_option("cache_block", 2)
    try:
        y, actual = run_host(csr, x, method)
        y2, _ = run_host(csr, x, method)
        api.set_option("block_rows", 1024)          # several blocks even on the small fixtures
        y3, _ = run_host(csr, x, method)
    finally:
        api.set_option("cache_block", 1)
        api.set_option("block_rows", 0)
    assert actual in (method, M.Method_Balanced2, M.Method_Balanced)
    if csr.nnz > 0:
        api.set_option("cache_block", 2)
        try:
            with api.Handle(csr.m, csr.n, csr.rowptr, csr.colidx, csr.val, method) as h:
                assert h.info()["kernel_name"] in ("blk_kernel", "blk_wide_kernel")
        finally:
            api.set_option("cache_block", 1)
    check(y, csr, x, y_ref, exact=name.endswith("eighths"))
    check(y3, csr, x, y_ref, exact=name.endswith("eighths"))
    assert np.array_equal(y.view(np.uint8), y2.view(np.uint8))


@pytest.mark.parametrize("way", list(api.VECTORIZED_WAY)[:4], ids=lambda w: w.name)
def test_every_vectorized_way_runs_the_hip_backend(way):
    """The reference stores vectorizedWay and never reads it (common.c:80); here all four values
    select the HIP schedules and the handle records what was asked."""
    csr, x, y_ref = load_golden("banded_f64_eighths")
    y = np.full(csr.m, np.nan)
    h = api.spmv_create_handle_all_in_one(csr.m, csr.n, csr.rowptr, csr.colidx, csr.val, 4, M.Method_Parallel, 8, way)
    assert h.contents.vectorizedWay == int(way) and h.contents.nthreads == 4
    assert h.contents.index is None or not h.contents.index
    api.spmv(h, csr.m, csr.rowptr, csr.colidx, csr.val, x, y)
    api.spmv_destory_handle(h)
    assert np.array_equal(y, y_ref)


def test_out_of_range_method_is_serial():
    csr, x, y_ref = load_golden("tiny_f64_eighths")
    for bad in (-3, 7, 8, 99):
        y, actual = run_host(csr, x, bad)
        assert actual == M.Method_Serial and np.array_equal(y, y_ref)   # common.c:136


def test_balanced_request_is_rewritten_like_the_reference():
    """parallel_balanced2_spmv.c:72-92: Balanced/Balanced2 become Balanced2 when some row is longer
    than one equal-nnz share, Balanced otherwise."""
    short, x1, _ = load_golden("banded_f64_eighths")
    long_, x2, _ = load_golden("single_long_f64_eighths")       # one row of 5000 nnz
    skew, x3, _ = load_golden("skewed_f64_eighths")            # mean 55, rows up to 4000
    for req in (M.Method_Balanced, M.Method_Balanced2):
        assert run_host(short, x1, req)[1] == M.Method_Balanced
        assert run_host(long_, x2, req)[1] == M.Method_Balanced2    # 5000 > 64 steps x 4 x 64 lanes... of one lane group
        assert run_host(skew, x3, req)[1] == M.Method_Balanced2


@pytest.mark.parametrize("method", ALL_METHODS, ids=lambda m: m.name)
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_lifecycle_create_spmv_clear_reuse_destroy(method, dtype):
    csr = synth.powerlaw(2000, 2000, 6.0, 700, 1.5, "eighths", dtype, seed=5)
    x = synth.fill_x(csr.n, "eighths", dtype, 9)
    want = oracle.spmv_serial(csr, x)
    h = api.spmv_create_handle_all_in_one(csr.m, csr.n, csr.rowptr, csr.colidx, csr.val, 8, method, csr.val.dtype.itemsize)
    for _ in range(3):
        y = np.full(csr.m, np.nan, dtype=dtype)
        api.spmv(h, csr.m, csr.rowptr, csr.colidx, csr.val, x, y)
        assert np.array_equal(y, want)
    api.spmv_clear_handle(h)                                    # common.c:69-71
    assert h.contents.spmvMethod == M.Method_Serial and not h.contents.extraHandle
    api.spmv(h, csr.m, csr.rowptr, csr.colidx, csr.val, x, y, check=False)   # cleared handle: reported no-op
    assert api.last_error()[0] == 5
    api.load().spmv_hip_clear_error()
    api.spmv_destory_handle(h)


def test_spmv_with_other_csr_arrays_reinspects(monkeypatch):
    """The reference computes with the arrays passed to spmv() (common.c:286-298)."""
    monkeypatch.setenv("SPMV_HIP_QUIET", "1")
    a = synth.banded(500, 500, 8, 7, "eighths", np.float64, seed=1)
    b = synth.powerlaw(700, 500, 5.0, 300, 1.5, "eighths", np.float64, seed=2)
    x = synth.fill_x(500, "eighths", np.float64, 3)
    h = api.spmv_create_handle_all_in_one(a.m, a.n, a.rowptr, a.colidx, a.val, 1, M.Method_Balanced_Yid, 8)
    y = np.full(b.m, np.nan)
    api.spmv(h, b.m, b.rowptr, b.colidx, b.val, x, y)
    assert np.array_equal(y, oracle.spmv_serial(b, x))
    y = np.full(a.m, np.nan)
    api.spmv(h, a.m, a.rowptr, a.colidx, a.val, x, y)
    assert np.array_equal(y, oracle.spmv_serial(a, x))
    api.spmv_destory_handle(h)


@pytest.mark.parametrize("lanes", [1, 2, 4, 8, 16, 32, 64])
def test_csr_vector_every_lane_width(lanes):
    csr, x, y_ref = load_golden("rowlen_sweep_f64_eighths")
    api.set_option("lanes_per_row", lanes)
    try:
        y, _ = run_host(csr, x, M.Method_Parallel)
    finally:
        api.set_option("lanes_per_row", 0)
    assert np.array_equal(y, y_ref)


def test_nan_in_x_stays_in_its_rows():
    """Padding (SELL) and masked tile slots never multiply x: a NaN in x reaches only rows that
    reference its column (the reference guards padded slots too, inner_spmv.h:466-474)."""
    csr = synth.skewed_rows(1500, 4096, "uniform", np.float64, seed=6)
    x = synth.fill_x(csr.n, "uniform", np.float64, 1)
    x[0] = np.nan
    want = oracle.spmv_serial(csr, x)
    for method in ALL_METHODS:
        y, _ = run_host(csr, x, method)
        assert np.array_equal(np.isnan(y), np.isnan(want)), method


def test_handle_info_and_alg_bytes():
    csr, x, _ = load_golden("banded_f64_uniform")
    with api.Handle(csr.m, csr.n, csr.rowptr, csr.colidx, csr.val, M.Method_Parallel) as h:
        info = h.info()
    assert info["schedule_name"] == "csr-vector" and info["nnz"] == csr.nnz
    assert info["alg_bytes"] == 4 * (csr.m + 1) + csr.nnz * 12 + 8 * csr.n + 8 * csr.m


def test_auto_method_picks_schedule_from_row_statistics():
    """SURVEY 8f f-3 (the reference only has an empty README heading for it, README.md:222)."""
    reg, xr, yr = load_golden("banded_f64_eighths")              # 16 entries per row: fills 4-lane chunks
    odd, xo, yo = load_golden("banded_wide_f64_eighths")         # 78 per row: 61 % of a 128-entry chunk
    skw, xs, ys = load_golden("powerlaw_f64_eighths")
    api.set_option("auto_method", 1)
    try:
        y, actual = run_host(reg, xr, M.Method_Serial)
        assert actual == M.Method_Parallel and np.array_equal(y, yr)
        y, actual = run_host(odd, xo, M.Method_Serial)
        assert actual == M.Method_CSR5SPMV and np.array_equal(y, yo)
        y, actual = run_host(skw, xs, M.Method_Serial)
        assert actual == M.Method_CSR5SPMV and np.array_equal(y, ys)
    finally:
        api.set_option("auto_method", 0)
    assert run_host(reg, xr, M.Method_Serial)[1] == M.Method_Serial


@pytest.mark.parametrize("key,values,method", [
    ("sell_sigma", [64, 256, 4096], M.Method_SellCSigma),
    ("sell_lds_x", [0, 1], M.Method_SellCSigma),
    ("csr5_sigma", [4, 8, 16], M.Method_CSR5SPMV),
    ("rowblock_nnz", [64, 333, 4096, 100000], M.Method_Balanced),
    ("csr5_sigma", [4, 8, 16], M.Method_Balanced_Yid),           # nnz-split = natural-layout tiles of 64 x sigma
    ("x_windows", [0], M.Method_Balanced_Yid),                   # no x windows (global gathers)
    ("vector_form", [4, 5, 6, 10, 11, 12], M.Method_Parallel),   # CSR-vector kernel forms
    ("x_windows", [0], M.Method_SellCSigma),
    ("csr5_two_deep", [1, 2], M.Method_SellCSigma),              # the staged CSR5 group kernel one tile / two tiles deep (long rows)
    ("x_windows", [0], M.Method_CSR5SPMV),
    ("csr5_two_deep", [1, 2], M.Method_CSR5SPMV),
    ("run_tiles", [0], M.Method_CSR5SPMV),                       # no RUN groups (the 16-bit slot stream is read everywhere)
    ("run_tiles", [0], M.Method_Parallel),                       # no RUN / BYTE tiles
    ("xcd_order", [0], M.Method_Balanced_Yid),
])
@pytest.mark.parametrize("name", ["skewed_f64_eighths", "empty_mix_f32_eighths", "banded_wide_f64_eighths"])
def test_tuning_options_do_not_change_results(key, values, method, name):
    """The reference hard-wires C / sigma / CSR5 sigma (common.c:139-140, csr5_spmv.cpp:30); here they
    are options -- results must be bit-identical for every legal value."""
    csr, x, y_ref = load_golden(name)
    default = api.get_option(key)
    try:
        for v in values:
            api.set_option(key, v)
            y, _ = run_host(csr, x, method)
            assert np.array_equal(y.view(np.uint8), y_ref.view(np.uint8)), (key, v)
    finally:
        api.set_option(key, default)


def test_illegal_option_values_are_reported_at_create():
    csr, x, _ = load_golden("tiny_f64_eighths")
    api.set_option("csr5_sigma", 5)
    try:
        with pytest.raises(api.SpmvError, match="csr5_sigma"):
            api.Handle(csr.m, csr.n, csr.rowptr, csr.colidx, csr.val, M.Method_CSR5SPMV)
    finally:
        api.set_option("csr5_sigma", 0)


def test_two_handles_on_two_streams_and_async_mode():
    """One handle = one stream (SURVEY 8b "Threading"): two handles on two non-default streams run
    concurrently and stay correct; async mode returns before completion and is stream-ordered."""
    import torch
    dev = torch.device("cuda:0")
    a, xa, ya = load_golden("skewed_f64_eighths")
    b, xb, yb = load_golden("banded_wide_f64_eighths")
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    handles = []
    for csr, x, stream, method in ((a, xa, s1, M.Method_CSR5SPMV), (b, xb, s2, M.Method_Parallel)):
        rp, ci = torch.from_numpy(csr.rowptr).to(dev), torch.from_numpy(csr.colidx).to(dev)
        va, xd = torch.from_numpy(csr.val).to(dev), torch.from_numpy(x).to(dev)
        h = api.Handle(csr.m, csr.n, rp, ci, va, method)
        h.attach_stream(stream.cuda_stream, async_=True)
        handles.append((h, xd, torch.full((csr.m,), float("nan"), dtype=va.dtype, device=dev)))
    torch.cuda.synchronize()
    for _ in range(20):
        for h, xd, yd in handles:
            h.spmv(xd, yd)
    for h, _, _ in handles:
        assert api.load().spmv_hip_synchronize(h.h) == 0
    assert np.array_equal(handles[0][2].cpu().numpy(), ya) and np.array_equal(handles[1][2].cpu().numpy(), yb)
    for h, _, _ in handles:
        h.close()


def test_m_zero_and_n_zero():
    lib = api.load()
    rp = np.zeros(1, dtype=np.int32)
    h = api.spmv_create_handle_all_in_one(0, 0, rp, np.zeros(0, np.int32), np.zeros(0), 1, M.Method_Parallel, 8)
    api.spmv(h, 0, rp, np.zeros(0, np.int32), np.zeros(0), np.zeros(0), np.zeros(0))
    assert lib.spmv_hip_last_error() == 0
    api.spmv_destory_handle(h)


def test_reorder_option_follows_the_reference_index_protocol():
    """SURVEY 8f f-4.  A band matrix scrambled by a random symmetric permutation: with option
    "reorder" create() RCM-reorders it, publishes the permutation in handle->index, and the caller
    gathers x / scatters y exactly as test_spmv.c:95-101, 130-137 does for OPT_LEVEL 3."""
    rng = np.random.default_rng(3)
    m = 20000                                    # scrambled column spans (~m) must exceed the 48 KiB LDS x budget
    band = synth.banded(m, m, 6, 5, "eighths", np.float64, seed=8)
    sc = rng.permutation(m)                      # scrambled row r = band row sc[r]
    inv = np.empty(m, dtype=np.int64); inv[sc] = np.arange(m)
    lens = np.diff(band.rowptr)[sc]
    rp = np.zeros(m + 1, dtype=np.int32); np.cumsum(lens, out=rp[1:])
    ci = np.empty(band.nnz, dtype=np.int32); va = np.empty(band.nnz)
    for r in range(m):
        s0, s1 = band.rowptr[sc[r]], band.rowptr[sc[r] + 1]
        ci[rp[r]:rp[r + 1]] = inv[band.colidx[s0:s1]]
        va[rp[r]:rp[r + 1]] = band.val[s0:s1]
    A = synth.CSR(m, m, rp, ci, va)
    x = synth.fill_x(m, "eighths", np.float64, 5)
    want = oracle.spmv_serial(A, x)
    api.set_option("reorder", 1)
    try:
        for method in (M.Method_Parallel, M.Method_CSR5SPMV, M.Method_SellCSigma):
            with api.Handle(m, m, A.rowptr, A.colidx, A.val, method) as h:
                index = h.index
                assert index is not None and h.h.contents.Level_3_opt_used == 1
                assert np.array_equal(np.sort(index), np.arange(m))
                xx = x[index]                                   # XX[i] = X[index[i]]
                yy = np.full(m, np.nan)
                h.spmv(xx, yy)
                y = np.empty(m); y[index] = yy                   # Y[index[i]] = YY[i]
                assert np.array_equal(y, want), method
                if method == M.Method_Parallel:
                    assert h.info()["kernel_name"] == "csr_vector_tile_kernel"   # band recovered: x tiles fit LDS
    finally:
        api.set_option("reorder", 0)
    with api.Handle(m, m, A.rowptr, A.colidx, A.val, M.Method_Parallel) as h:
        assert h.index is None and h.h.contents.Level_3_opt_used == 0
        assert h.info()["kernel_name"] == "csr_vector_pipe_kernel"               # scrambled: spans too wide


def _scrambled(csr, rng):
    """rows and columns of a square CSR under one random symmetric permutation (row r of the result = row sc[r] of csr)"""
    m = csr.m
    sc = rng.permutation(m)
    inv = np.empty(m, dtype=np.int64); inv[sc] = np.arange(m)
    lens = np.diff(csr.rowptr)[sc]
    rp = np.zeros(m + 1, dtype=np.int32); np.cumsum(lens, out=rp[1:])
    starts = csr.rowptr[:-1][sc].astype(np.int64)
    src = np.repeat(starts - rp[:-1], lens) + np.arange(int(rp[-1]))
    return synth.CSR(m, m, rp, inv[csr.colidx[src]].astype(np.int32), csr.val[src])


@pytest.mark.parametrize("shape", ["band", "nonsymmetric", "components", "powerlaw"])
def test_device_rcm_orders_like_a_bfs_and_is_reproducible(shape):
    """Round 4 (kernels/rcm.hpp): option reorder = 1 runs reverse Cuthill-McKee on the DEVICE -- A^T's pattern, a queue BFS whose small frontiers are
    walked level after level by one persistent workgroup, a (level, degree, id) sort, the permuted matrix in place of the uploaded one.  Checked:
    the product through the handle->index protocol is exact; the permutation is a permutation and the same from handle to handle; the
    bandwidth of P A P^T is what a BFS ordering gives (band: within 4 x the band's; host RCM, reorder = 2, is the yardstick).
      band          a scrambled band (half-width 6): ~3000 BFS levels, all of them inside one launch of the small-frontier kernel
      nonsymmetric  the band with its lower triangle only: A alone cannot be walked backwards -- the BFS pushes along A^T as well
      components    three scrambled bands side by side + 500 isolated rows/columns + empty rows: component after component, isolated ones at once
      powerlaw      heavy-tailed rows with R-MAT columns: frontiers of 1e4-1e5 vertices, expanded by the whole grid"""
    rng = np.random.default_rng(8)
    if shape in ("band", "nonsymmetric"):
        base = synth.banded(40_000, 40_000, 6, 6, "eighths", np.float64, seed=3)
        if shape == "nonsymmetric":
            keep = base.colidx <= np.repeat(np.arange(base.m), np.diff(base.rowptr))
            rp = np.zeros(base.m + 1, dtype=np.int32); np.cumsum(np.add.reduceat(keep.astype(np.int32), base.rowptr[:-1]), out=rp[1:])
            base = synth.CSR(base.m, base.n, rp, base.colidx[keep], base.val[keep])
        A = _scrambled(base, rng)
    elif shape == "components":
        b = synth.banded(9_000, 9_000, 4, 4, "eighths", np.float64, seed=4)
        m = 3 * b.m + 500
        rp = np.concatenate([b.rowptr[:-1], b.rowptr[:-1] + b.nnz, b.rowptr[:-1] + 2 * b.nnz, np.full(501, 3 * b.nnz)]).astype(np.int32)
        ci = np.concatenate([b.colidx, b.colidx + b.m, b.colidx + 2 * b.m]).astype(np.int32)
        A = _scrambled(synth.CSR(m, m, rp, ci, np.tile(b.val, 3)), rng)
    else:
        A = synth.powerlaw(60_000, 60_000, 6.0, 3000, 1.5, "eighths", np.float64, seed=5)
    m = A.m
    x = synth.fill_x(m, "eighths", np.float64, 6)
    want = oracle.spmv_serial(A, x)

    def bandwidth(index):
        inv = np.empty(m, dtype=np.int64); inv[index] = np.arange(m)
        rows = inv[np.repeat(np.arange(m), np.diff(A.rowptr))]
        return int(np.abs(inv[A.colidx] - rows).max()) if A.nnz else 0
    got = {}
    for mode in (1, 1, 2):
        api.set_thread_option("reorder", mode)
        try:
            h = api.Handle(m, m, A.rowptr, A.colidx, A.val, M.Method_CSR5SPMV)
        finally:
            api.clear_thread_options()
        with h:
            index = h.index
            assert index is not None and h.h.contents.Level_3_opt_used == 1
            assert np.array_equal(np.sort(index), np.arange(m)), mode
            yy = h.spmv(x[index], np.full(m, np.nan))
            y = np.empty(m); y[index] = yy
            assert np.array_equal(y, want), (shape, mode)
            got.setdefault(mode, []).append(index)
    assert np.array_equal(got[1][0], got[1][1]), "the device ordering differs from handle to handle"
    bw_dev, bw_host = bandwidth(got[1][0]), bandwidth(got[2][0])
    if shape in ("band", "nonsymmetric"):
        assert bw_dev <= 4 * 12 and bw_dev <= 2 * bw_host + 16, (bw_dev, bw_host)
    elif shape == "components":
        assert bw_dev <= 4 * 8 + 2, (bw_dev, bw_host)


def test_device_rcm_at_size_is_fast(capsys):
    """2e6 rows x 32 of a band under a random symmetric permutation (62 500 BFS levels): round 1's host RCM added 3.3 s to create(); on the device the
    whole reordering -- transpose, two BFS sweeps, sort, permute -- is bounded here by 1.5 s, and the multiply gets its banded rate back (x tiles staged)."""
    import time
    import torch
    dev = torch.device("cuda:0")
    m, k = 2_000_000, 32
    g = torch.Generator(device=dev); g.manual_seed(5)
    sc = torch.randperm(m, generator=g, device=dev)
    inv = torch.empty_like(sc); inv[sc] = torch.arange(m, device=dev)
    offs = torch.arange(k, device=dev) - k // 2
    ci = inv[(sc[:, None] + offs[None, :]) % m].reshape(-1).to(torch.int32)
    rp = torch.arange(0, (m + 1) * k, k, dtype=torch.int32, device=dev)
    va = (torch.randint(-8, 9, (m * k,), generator=g, device=dev) * 0.125).double()
    x = (torch.randint(-8, 9, (m,), generator=g, device=dev) * 0.125).double()
    want = (va * x[ci.long()]).view(m, k).sum(1)
    api.set_thread_option("reorder", 1)
    try:
        with api.Handle(m, m, rp, ci, va, M.Method_Parallel) as h0:     # the first create of a process pays allocator warm-up
            pass
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        h = api.Handle(m, m, rp, ci, va, M.Method_Parallel)
        create_s = time.perf_counter() - t0
    finally:
        api.clear_thread_options()
    with h:
        index = torch.from_numpy(h.index).to(dev).long()
        info = h.info()
        yy = torch.full((m,), float("nan"), dtype=torch.float64, device=dev)
        h.spmv(x[index].contiguous(), yy)
        y = torch.empty_like(yy); y[index] = yy
        assert torch.equal(y, want)
        assert info["kernel_name"] == "csr_vector_tile_kernel" and info["x_groups_staged"] >= 0.99 * info["x_groups"], info
    print(f"device RCM: create {create_s:.3f} s at {m} rows x {k}")
    assert create_s <= 1.5, create_s


@pytest.mark.parametrize("method", [M.Method_Parallel, M.Method_Balanced, M.Method_Balanced_Yid, M.Method_SellCSigma, M.Method_CSR5SPMV], ids=lambda m: m.name)
def test_column_indices_are_released_when_the_multiply_never_reads_them(method):
    """the resident int32 ColIdx copy (4 B per non-zero) goes back to the pool at the end of create() when the built schedule
    reads only its own slot streams / slabs / tiles (every tile staged) -- spmv_hip_info.device_bytes drops by 4 nnz, the multiply and a values refresh are
    unaffected; option keep_columns = 1 keeps it; a matrix whose tiles do not stage (global gathers) keeps it whatever the option."""
    import torch
    dev = torch.device("cuda:0")
    m, n, rp, ci, va = synth.banded_holes_device(400_000, 400_000, 24, 0.25, "eighths", torch.float64, dev, 7)
    g = torch.Generator(device=dev); g.manual_seed(2)
    x = (torch.randint(-8, 9, (n,), generator=g, device=dev) * 0.125).double()
    want = (va * x[ci.long()]).view(m, 24).sum(1)
    held = {}
    for keep in (1, 0):
        api.set_thread_option("keep_columns", keep)
        try:
            h = api.Handle(m, n, rp, ci, va, method)
        finally:
            api.clear_thread_options()
        with h:
            held[keep] = h.info()["device_bytes"]
            y = torch.full((m,), float("nan"), dtype=torch.float64, device=dev)
            h.spmv(x, y)
            torch.cuda.synchronize()
            assert torch.equal(y, want), (method, keep)
            h.update_values((va * 2).contiguous())
            h.spmv(x, y)
            torch.cuda.synchronize()
            assert torch.equal(y, 2 * want), (method, keep, "update_values")
    assert held[1] - held[0] >= 4 * m * 24, (method, held)          # the int32 column copy (+ its padding)
    # no locality: the tile kernels gather through the global columns -> nothing is released
    _, _, rp2, ci2, va2 = synth.uniform_k_device(60_000, 60_000, 8, "eighths", torch.float64, dev, 9)
    sizes = []
    for keep in (1, 0):
        api.set_thread_option("keep_columns", keep)
        api.set_thread_option("cache_block", 0)
        try:
            with api.Handle(60_000, 60_000, rp2, ci2, va2, M.Method_Parallel) as h:
                sizes.append(h.info()["device_bytes"])
        finally:
            api.clear_thread_options()
    assert sizes[0] == sizes[1], sizes


def test_out_of_range_column_index_is_rejected_at_create(monkeypatch):
    """An index outside [0, n) makes the reference read out of bounds; on a GPU it would fault, so
    create() validates ColIdx and reports SPMV_HIP_E_ARG instead."""
    monkeypatch.setenv("SPMV_HIP_QUIET", "1")
    csr, x, _ = load_golden("banded_f64_eighths")
    bad = csr.colidx.copy()
    bad[17] = csr.n
    with pytest.raises(api.SpmvError, match="ColIdx out of range"):
        api.Handle(csr.m, csr.n, csr.rowptr, bad, csr.val)
    bad[17] = -1
    with pytest.raises(api.SpmvError, match="ColIdx out of range"):
        api.Handle(csr.m, csr.n, csr.rowptr, bad, csr.val)
    rp = csr.rowptr.copy()
    rp[5] = rp[6] + 1                                   # decreasing RowPtr
    with pytest.raises(api.SpmvError, match="RowPtr"):
        api.Handle(csr.m, csr.n, rp, csr.colidx, csr.val)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("method", [M.Method_Parallel, M.Method_Balanced])
def test_multi_window_x_tiles_on_stencil_structure(method, dtype):
    """27-point stencil on a 64^3 periodic grid: the columns of a 256-row tile sit in 9 windows that
    are 4096 apart, so the plain span (> 8192 columns) exceeds the 48 KiB LDS budget in fp64 and
    the tile is staged as several windows (csr_vector_tile.hpp).  Bit-exact against the oracle."""
    nx = 64
    m = nx ** 3
    offs = np.array([dz * nx * nx + dy * nx + dx for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1)])
    rowptr = np.arange(0, (m + 1) * 27, 27, dtype=np.int32)
    colidx = ((np.arange(m)[:, None] + offs[None, :]) % m).astype(np.int32).reshape(-1)
    val = synth.fill_values(m * 27, "eighths", dtype, 3)
    csr = synth.CSR(m, m, rowptr, colidx, val)
    x = synth.fill_x(m, "eighths", dtype, 4)
    want = oracle.spmv_serial(csr, x)
    y = np.full(m, np.nan, dtype=dtype)
    with api.Handle(m, m, csr.rowptr, csr.colidx, csr.val, method) as h:
        h.spmv(x, y)
        info = h.info()
        name = info["kernel_name"]
        assert np.array_equal(y, want)
        assert name in ("csr_vector_tile_kernel", "csr_vector_rows_kernel")
        if method == M.Method_Parallel:
            # round 4: every interior row has the same 27 slot offsets from its first entry -> TEMPLATE tiles (no column stream at all); only the
            # tiles in which the periodic grid wraps keep a stream
            assert info["tmpl_nnz"] >= 0.9 * csr.nnz, (info["tmpl_nnz"], csr.nnz)
        h.update_values(csr.val * 2)
        assert np.array_equal(h.spmv(x, np.full(m, np.nan, dtype=dtype)), 2 * want)


def test_create_destroy_cycles_return_all_device_memory():
    """Every schedule's inspector products (incl. early-freed column copies, long-row sub-matrices, the
    cache-blocked streams and the autotune / inspector scratch) are released by spmv_destory_handle and by
    spmv_clear_handle: free HBM after 3 cycles over all methods equals free HBM after the first."""
    import torch
    csr, x, _ = load_golden("skewed_f64_eighths")
    big = synth.from_row_lengths(np.full(40000, 24), 40000, "eighths", np.float64, seed=3)      # no locality: global gathers
    xb = np.ones(big.n)

    def cycle():
        for mat, xv in ((csr, x), (big, xb)):
            for method in ALL_METHODS:
                for cb in (1, 2):
                    api.set_option("cache_block", cb)
                    y = np.empty(mat.m, dtype=mat.val.dtype)
                    h = api.spmv_create_handle_all_in_one(mat.m, mat.n, mat.rowptr, mat.colidx, mat.val, 1, method,
                                                          mat.val.dtype.itemsize, api.VECTORIZED_WAY.VECTOR_HIP, "leak")
                    api.spmv(h, mat.m, mat.rowptr, mat.colidx, mat.val, xv, y)
                    if method == M.Method_CSR5SPMV:
                        api.spmv_clear_handle(h)          # releases the device state, keeps the handle shell
                    api.spmv_destory_handle(h)
        api.set_option("cache_block", 1)
        torch.cuda.synchronize()
        return torch.cuda.mem_get_info()[0]

    try:
        first = cycle()
        for _ in range(2):
            last = cycle()
    finally:
        api.set_option("cache_block", 1)
    assert last >= first - (1 << 20), (first, last)      # nothing accumulates (1 MiB slack for the runtime)


def test_wide_x_windows_use_the_slot_index_form():
    """fp64 rows whose columns scatter over ~9000 columns: a 256-row tile's window (72 KB) exceeds the 48 KiB
    budget of the byte-offset stream, so CSR-vector and Balanced run the wide form (1024-row blocks / the
    row blocks, 96 KiB budget, slot indices) instead of falling back to global gathers.  Exact inputs -> exact bits."""
    m = n = 40000
    csr = synth.from_row_lengths(np.full(m, 12), n, "eighths", np.float64, seed=21, local=4500)
    rng = np.random.default_rng(5)
    x = (rng.integers(-8, 9, n) * 0.125)
    prod = csr.val * x[csr.colidx]
    cs = np.concatenate([[0.0], np.cumsum(prod)])
    want = cs[csr.rowptr[1:].astype(np.int64)] - cs[csr.rowptr[:-1].astype(np.int64)]
    import torch
    dev = torch.device("cuda:0")
    rp, ci = torch.from_numpy(csr.rowptr).to(dev), torch.from_numpy(csr.colidx).to(dev)
    va, xd = torch.from_numpy(csr.val).to(dev), torch.from_numpy(x).to(dev)
    for method in (M.Method_Parallel, M.Method_Balanced):
        yd = torch.full((m,), float("nan"), dtype=torch.float64, device=dev)
        with api.Handle(m, n, rp, ci, va, method) as h:
            h.spmv(xd, yd)
            info = h.info()
        torch.cuda.synchronize()
        assert info["kernel_name"] == "csr_vector_rows_kernel" and info["x_groups_staged"] * 2 >= info["x_groups"], info
        assert np.array_equal(yd.cpu().numpy(), want), method
    # fp32 of the same matrix fits the narrow budget: the tile kernel
    csr32 = synth.CSR(m, n, csr.rowptr, csr.colidx, csr.val.astype(np.float32))
    y, _ = run_host(csr32, x.astype(np.float32), M.Method_Parallel)
    assert np.array_equal(y, want.astype(np.float32))


@pytest.mark.parametrize("method", ALL_METHODS, ids=lambda m: m.name)
@pytest.mark.parametrize("name", ["skewed_f64_uniform", "powerlaw_f32_uniform", "empty_mix_f64_uniform"])
def test_results_are_bit_reproducible_across_handles(name, method):
    """No racing floating-point atomics anywhere (carries are added in tile order by one lane; a row block of the
    row-block x column-slab executor belongs to one wavefront): two handles built from the same matrix give the
    same bits on inexact data."""
    csr, x, _ = load_golden(name)
    y1, _ = run_host(csr, x, method)
    y2, _ = run_host(csr, x, method)
    assert np.array_equal(y1.view(np.uint8), y2.view(np.uint8))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("method", [M.Method_Parallel, M.Method_Balanced2, M.Method_CSR5SPMV], ids=lambda m: m.name)
def test_matrix_without_column_locality_runs_the_blocked_executor_deterministically(method, dtype):
    """2^21+ non-zeros on uniformly random columns over an x larger than an XCD's L2: no x window fits LDS, so
    every method (not only the Balanced family) is routed to the row-block x column-slab executor by default, and
    inexact data gives the same bits from two different handles and from repeated launches."""
    import torch
    dev = torch.device("cuda:0")
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    m = n = 2_400_000 if dtype == np.float64 else 4_800_000          # x = 19 MB: several times an XCD's 4 MiB L2
    _, _, rp, ci, va = synth.uniform_k_device(m, n, 8, "uniform", tdt, dev, seed=11)
    g = torch.Generator(device=dev); g.manual_seed(5)
    x = torch.rand(n, generator=g, device=dev, dtype=tdt) * 2 - 1
    ys = []
    for _ in range(2):
        y = torch.full((m,), float("nan"), dtype=tdt, device=dev)
        with api.Handle(m, n, rp, ci, va, method) as h:
            info = h.info()
            assert info["cache_blocked"] == 1 and info["kernel_name"] in ("blk_kernel", "blk_wide_kernel"), info
            h.spmv(x, y)
            y2 = torch.empty_like(y)
            h.spmv(x, y2)
        torch.cuda.synchronize()
        assert torch.equal(y, y2)
        ys.append(y)
    assert torch.equal(ys[0], ys[1]), "two handles, same matrix: different bits"
    want = (va.double() * x.double()[ci.long()]).view(m, 8).sum(1)
    scale = (va.double() * x.double()[ci.long()]).abs().view(m, 8).sum(1)
    assert bool(((ys[0].double() - want).abs() <= TOL[np.dtype(dtype)] * scale).all())


@pytest.mark.parametrize("method", [M.Method_Parallel, M.Method_Balanced2, M.Method_SellCSigma], ids=lambda m: m.name)
def test_blocked_executor_equal_work_cut_points_on_a_lopsided_matrix(method):
    """The row blocks of the blocked executor are cut by work (blocked.hpp: blk_partition_kernel).  One row that outweighs
    every share (several cut points collapse onto it: empty blocks are dropped), light rows whose share would span more
    rows than the LDS accumulators hold (blocks are split at the row cap) and a long stretch of empty rows: exact data,
    so the result must equal the definition bit for bit, and spmv_hip_update_values must hit the same positions."""
    import torch
    dev = torch.device("cuda:0")
    m, n = 120_000, 4_000_000
    g = torch.Generator(device=dev); g.manual_seed(21)
    lens = torch.randint(0, 4, (m,), generator=g, device=dev, dtype=torch.int64)
    lens[5] = 3_500_000     # 118 shares of ~31 k work each: ~110 cut points fall on this row, and a share of light rows (work 1-3 each) spans more than the 9984-row cap
    lens[60_000:100_000] = 0
    lens[m - 1] = 17
    _, _, rp, ci, va = synth.from_row_lengths_device(lens, n, "eighths", torch.float64, dev, seed=4)
    x = (torch.randint(-8, 9, (n,), generator=g, device=dev) * 0.125).to(torch.float64)
    prod = va * x[ci.long()]
    cs = torch.cat([torch.zeros(1, dtype=torch.float64, device=dev), torch.cumsum(prod, 0)])
    want = cs[rp[1:].long()] - cs[rp[:-1].long()]
    keep = api.get_option("cache_block")
    try:
        api.set_option("cache_block", 2)
        y = torch.full((m,), float("nan"), dtype=torch.float64, device=dev)
        with api.Handle(m, n, rp, ci, va, method) as h:
            assert h.info()["cache_blocked"] == 1
            h.spmv(x, y)
            torch.cuda.synchronize()
            assert torch.equal(y, want)
            va2 = va * 2
            h.update_values(va2)
            h.spmv(x, y)
            torch.cuda.synchronize()
            assert torch.equal(y, 2 * want)
    finally:
        api.set_option("cache_block", keep)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_blocked_executor_forms_give_the_same_bits_and_create_picks_one_by_timing(dtype):
    """The row-block executor runs 8 or 12 groups per pipeline step (blocked.hpp: blk_kernel<T, UN>); create() times
    the two forms and keeps the faster (info.tuned_choice 100 / 101, info.tune_ms).  They issue the same
    additions in the same order, so on inexact data every form -- and hence whichever one a handle happened to pick --
    gives the same bits; power-law rows with hubs make blocks of very different lengths (partial last steps)."""
    import torch
    dev = torch.device("cuda:0")
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    m = n = 1_300_000
    lens = synth.powerlaw_lengths_device(m, 12.0, 20000, 1.6, dev, 3)
    _, _, rp, ci, va = synth.from_row_lengths_device(lens, n, "uniform", tdt, dev, 3, cols="rmat")
    g = torch.Generator(device=dev); g.manual_seed(9)
    x = torch.rand(n, generator=g, device=dev, dtype=tdt) * 2 - 1
    keep = {k: api.get_option(k) for k in ("cache_block", "blk_groups", "blk_waves")}
    ys = {}
    try:
        api.set_option("cache_block", 2)
        api.set_option("blk_waves", 1)      # the one-wave layout (create() may otherwise keep a wide one: another stored order)
        for variant in (0, 8, 12):
            api.set_option("blk_groups", variant)
            y = torch.full((m,), float("nan"), dtype=tdt, device=dev)
            with api.Handle(m, n, rp, ci, va, M.Method_Balanced2) as h:
                info = h.info()
                assert info["cache_blocked"] == 1 and info["kernel_name"] in ("blk_kernel", "blk_wide_kernel"), info
                if variant == 0:
                    assert info["tuned_choice"] in (100, 101) and min(info["tune_ms"][:2]) > 0, info
                h.spmv(x, y)
            torch.cuda.synchronize()
            assert not bool(torch.isnan(y).any())
            ys[variant] = y
    finally:
        for k, v in keep.items():
            api.set_option(k, v)
    for variant, y in ys.items():
        assert torch.equal(y, ys[0]), f"variant {variant} differs from the tuned default"


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("block_rows", [0, 512])
def test_blocked_executor_hot_cells_and_super_slabs(dtype, block_rows):
    """The blocked layout stores a row block's entries slab after slab with 16-bit column offsets inside super-slabs of 65536
    columns, every super-slab's run padded to whole groups (blocked.hpp).  A matrix with a few hot column ranges (cells of
    thousands of entries, one straddling the super-slab boundary at 131072), a uniform background over four super-slabs, the
    last column range ending in the middle of a slab, empty rows, and one hub row: exact data, so every executor form must
    equal the definition bit for bit, and spmv_hip_update_values must hit the same positions."""
    import torch
    dev = torch.device("cuda:0")
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    m, n = 30_000, 200_037
    g = torch.Generator(device=dev); g.manual_seed(77)
    lens = torch.randint(0, 48, (m,), generator=g, device=dev, dtype=torch.int64)
    lens[1000:1400] = 0
    lens[17] = 150_000
    rp = torch.zeros(m + 1, dtype=torch.int64, device=dev)
    torch.cumsum(lens, 0, out=rp[1:])
    nnz = int(rp[-1].item())
    u = torch.rand(nnz, generator=g, device=dev)
    hot = torch.randint(0, 700, (nnz,), generator=g, device=dev)            # columns 0 .. 699: five and a half slabs
    hot2 = 131_000 + torch.randint(0, 300, (nnz,), generator=g, device=dev)  # straddles the super-slab boundary at 131072
    tail = n - 1 - torch.randint(0, 60, (nnz,), generator=g, device=dev)     # the last, partial slab
    bg = torch.randint(0, n, (nnz,), generator=g, device=dev)
    ci = torch.where(u < 0.45, hot, torch.where(u < 0.6, hot2, torch.where(u < 0.7, tail, bg))).to(torch.int32)
    va = (torch.randint(-8, 9, (nnz,), generator=g, device=dev) * 0.125).to(tdt)
    x = (torch.randint(-8, 9, (n,), generator=g, device=dev) * 0.125).to(tdt)
    prod = va.double() * x.double()[ci.long()]
    cs = torch.cat([torch.zeros(1, dtype=torch.float64, device=dev), torch.cumsum(prod, 0)])
    want = (cs[rp[1:]] - cs[rp[:-1]]).to(tdt)
    rp32 = rp.to(torch.int32)
    keep = {k: api.get_option(k) for k in ("cache_block", "blk_groups", "block_rows")}
    try:
        api.set_option("cache_block", 2)
        api.set_option("block_rows", block_rows)
        stored = {}
        for dense in (1,):
            for variant in (0, 8, 12):
                api.set_option("blk_groups", variant)
                y = torch.full((m,), float("nan"), dtype=tdt, device=dev)
                with api.Handle(m, n, rp32, ci, va, M.Method_Balanced2) as h:
                    info = h.info()
                    assert info["cache_blocked"] == 1 and info["kernel_name"] in ("blk_kernel", "blk_wide_kernel"), info
                    stored[dense] = info["stored_nnz"]
                    h.spmv(x, y)
                    torch.cuda.synchronize()
                    assert torch.equal(y, want), (dense, variant, int((y != want).sum()))
                    if variant == 0:
                        h.update_values(va * 2)
                        h.spmv(x, y)
                        torch.cuda.synchronize()
                        assert torch.equal(y, 2 * want), (dense, "update_values")
        assert stored[1] >= nnz
    finally:
        for k, v in keep.items():
            api.set_option(k, v)


def _hot_cells_matrix(tdt, dev, seed=77):
    """hot column ranges (one straddling a super-slab boundary), a uniform background, a partial last slab, empty rows, a hub row; exact data"""
    import torch
    m, n = 30_000, 200_037
    g = torch.Generator(device=dev); g.manual_seed(seed)
    lens = torch.randint(0, 48, (m,), generator=g, device=dev, dtype=torch.int64)
    lens[1000:1400] = 0
    lens[17] = 150_000
    rp = torch.zeros(m + 1, dtype=torch.int64, device=dev)
    torch.cumsum(lens, 0, out=rp[1:])
    nnz = int(rp[-1].item())
    u = torch.rand(nnz, generator=g, device=dev)
    hot = torch.randint(0, 700, (nnz,), generator=g, device=dev)
    hot2 = 131_000 + torch.randint(0, 300, (nnz,), generator=g, device=dev)
    tail = n - 1 - torch.randint(0, 60, (nnz,), generator=g, device=dev)
    bg = torch.randint(0, n, (nnz,), generator=g, device=dev)
    ci = torch.where(u < 0.45, hot, torch.where(u < 0.6, hot2, torch.where(u < 0.7, tail, bg))).to(torch.int32)
    va = (torch.randint(-8, 9, (nnz,), generator=g, device=dev) * 0.125).to(tdt)
    x = (torch.randint(-8, 9, (n,), generator=g, device=dev) * 0.125).to(tdt)
    prod = va.double() * x.double()[ci.long()]
    cs = torch.cat([torch.zeros(1, dtype=torch.float64, device=dev), torch.cumsum(prod, 0)])
    want = (cs[rp[1:]] - cs[rp[:-1]]).to(tdt)
    return m, n, rp.to(torch.int32), ci, va, x, want


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("waves,det", [(2, 1), (4, 1), (8, 1), (2, 0), (4, 0), (8, 0)])
def test_wide_blocked_forms_match_the_definition_exactly(dtype, waves, det):
    """Round 4: ONE row block per CU whose LDS accumulators are shared by 2 / 4 / 8 wavefronts (blocked.hpp: blk_wide_kernel), the waves taking turns
    at adding (deterministic = 1) or adding in arrival order (0).  Exact data: every form, every groups-per-step value, uniform 512-row blocks and
    equal-work blocks, sparse cells sorted by column or not -- all equal the definition bit for bit, and so does a values refresh."""
    import torch
    dev = torch.device("cuda:0")
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    m, n, rp, ci, va, x, want = _hot_cells_matrix(tdt, dev)
    keys = ("cache_block", "block_rows", "blk_waves", "blk_groups", "blk_subsort", "deterministic")
    keep = {k: api.get_option(k) for k in keys}
    try:
        api.set_option("cache_block", 2)
        api.set_option("blk_waves", waves)
        api.set_option("deterministic", det)
        for block_rows, groups, subsort in ((0, 0, 1), (512, 8, 1), (0, 6 if waves == 8 else 12, 0)):
            api.set_option("block_rows", block_rows)
            api.set_option("blk_groups", groups)
            api.set_option("blk_subsort", subsort)
            y = torch.full((m,), float("nan"), dtype=tdt, device=dev)
            with api.Handle(m, n, rp, ci, va, M.Method_Balanced2) as h:
                info = h.info()
                assert info["cache_blocked"] == 1 and info["kernel_name"] == "blk_wide_kernel" and info["blk_waves"] == waves, info
                assert info["reproducible"] == det, info
                h.spmv(x, y)
                torch.cuda.synchronize()
                assert torch.equal(y, want), (block_rows, groups, subsort, int((y != want).sum()))
                h.update_values(va * 2)
                h.spmv(x, y)
                torch.cuda.synchronize()
                assert torch.equal(y, 2 * want), (block_rows, groups, subsort, "update_values")
    finally:
        for k, v in keep.items():
            api.set_option(k, v)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_wide_blocked_forms_reproducible_when_the_waves_take_turns(dtype):
    """Inexact data on power-law rows with hub columns: the ordered wide forms (option deterministic = 1, the default) add every row's products in
    (phase, wave, instruction, lane) order -- a function of the stored stream -- so two handles give the same bits and so do both groups-per-step
    forms of one width; the arrival-order forms (deterministic = 0) agree with them to rounding (north_star tolerance, scaled by the row's sum of |a x|)."""
    import torch
    dev = torch.device("cuda:0")
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    m = n = 1_300_000
    lens = synth.powerlaw_lengths_device(m, 12.0, 20000, 1.6, dev, 3)
    _, _, rp, ci, va = synth.from_row_lengths_device(lens, n, "uniform", tdt, dev, 3, cols="rmat")
    g = torch.Generator(device=dev); g.manual_seed(9)
    x = torch.rand(n, generator=g, device=dev, dtype=tdt) * 2 - 1
    prod = va.double() * x.double()[ci.long()]
    z = torch.zeros(1, dtype=torch.float64, device=dev)
    cs, ca = torch.cat([z, torch.cumsum(prod, 0)]), torch.cat([z, torch.cumsum(prod.abs(), 0)])
    want, scale = cs[rp[1:].long()] - cs[rp[:-1].long()], ca[rp[1:].long()] - ca[rp[:-1].long()]
    tol = 1e-6 if dtype == np.float64 else 1e-3
    keys = ("cache_block", "blk_waves", "blk_groups", "deterministic")
    keep = {k: api.get_option(k) for k in keys}

    def run(waves, det, groups):
        api.set_option("blk_waves", waves); api.set_option("deterministic", det); api.set_option("blk_groups", groups)
        y = torch.full((m,), float("nan"), dtype=tdt, device=dev)
        with api.Handle(m, n, rp, ci, va, M.Method_Balanced2) as h:
            assert h.info()["blk_waves"] == waves and h.info()["reproducible"] == det, h.info()
            h.spmv(x, y)
            y2 = torch.empty_like(y)
            h.spmv(x, y2)
        torch.cuda.synchronize()
        assert bool(((y.double() - want).abs() <= tol * scale + 1e-300).all()), (waves, det, groups)
        return y, y2
    try:
        api.set_option("cache_block", 2)
        for waves, ga, gb in ((2, 8, 12), (4, 8, 12), (8, 6, 8)):
            ya, ya2 = run(waves, 1, ga)
            yb, _ = run(waves, 1, gb)
            yc, _ = run(waves, 1, 0)        # whichever form create() timed faster
            assert torch.equal(ya, ya2) and torch.equal(ya, yc), f"{waves} waves: not reproducible"
            assert torch.equal(ya, yb), f"{waves} waves: the two groups-per-step forms differ"
            run(waves, 0, 0)
    finally:
        for k, v in keep.items():
            api.set_option(k, v)


@pytest.mark.parametrize("method", ALL_METHODS, ids=lambda m: m.name)
@pytest.mark.parametrize("name", ["skewed_f64_uniform", "skewed_f32_uniform", "empty_mix_f64_uniform", "banded_wide_f32_uniform"])
@pytest.mark.parametrize("blocked", [0, 1])
def test_update_values_refreshes_every_private_layout(name, method, blocked):
    """spmv_hip_update_values: the caller changes Matrix_Val IN PLACE (the reference would simply see it,
    common.c:286-298); one call re-permutes the values into the schedule's layouts (SELL slabs, CSR5 tiles, long-row
    sub-matrix, blocked streams) without re-inspection.  Result = the oracle on the new values."""
    csr, x, _ = load_golden(name)
    val = csr.val.copy()
    api.set_option("cache_block", 2 if blocked else 1)
    api.set_option("check_values", 0)       # this test is about the explicit call: spmv() does not watch the array
    h = None
    try:
        h = api.Handle(csr.m, csr.n, csr.rowptr, csr.colidx, val, method)
        assert h.info()["cache_blocked"] == (1 if blocked and method != M.Method_Serial else 0)
        y0 = h.spmv(x, np.full(csr.m, np.nan, dtype=val.dtype))
        assert np.array_equal(y0, run_host(csr, x, method)[0])
        stale = val.copy()
        val[:] = (val * np.asarray(-1.5, dtype=val.dtype) + np.asarray(0.25, dtype=val.dtype)).astype(val.dtype)  # in place, same pointer
        y_stale = h.spmv(x, np.full(csr.m, np.nan, dtype=val.dtype))
        assert np.array_equal(y_stale, y0), "without a refresh the resident copy is what multiplies (documented)"
        h.update_values(val)
        y1 = h.spmv(x, np.full(csr.m, np.nan, dtype=val.dtype))
        new = synth.CSR(csr.m, csr.n, csr.rowptr, csr.colidx, val)
        assert np.array_equal(y1, run_host(new, x, method)[0]), "refreshed handle != handle created on the new values"
        check(y1, new, x, oracle.spmv_serial(new, x), exact=False)
        h.update_values(stale)                                  # another array of the same pattern
        assert np.array_equal(h.spmv(x, np.empty(csr.m, dtype=val.dtype)), y0)
    finally:
        api.set_option("cache_block", 1)
        api.set_option("check_values", 2)
        if h is not None:
            h.close()


@pytest.mark.parametrize("method", [M.Method_Parallel, M.Method_SellCSigma, M.Method_CSR5SPMV, M.Method_Balanced2], ids=lambda m: m.name)
@pytest.mark.parametrize("gpus", [0, 3])
def test_host_values_changed_in_place_are_seen_with_no_option_set(method, gpus, monkeypatch):
    """the reference multiplies the arrays of THIS call (common.c:286-298).  With HOST arrays -- its only mode -- an in-place
    change of Matrix_Val behind the same pointer is picked up by spmv() by default (sampled checksum, option check_values = 2), on ordinary
    and on multi-GPU handles; nothing to call, no option to set.  A device array is not watched by default (documented)."""
    import torch
    csr, x, _ = load_golden("skewed_f64_uniform")
    val = csr.val.copy()
    if gpus:
        monkeypatch.setenv("SPMV_HIP_GPUS_VIRTUAL", "1")
        api.set_thread_option("gpus", gpus)
    try:
        h = api.Handle(csr.m, csr.n, csr.rowptr, csr.colidx, val, method)
    finally:
        api.clear_thread_options()
    with h:
        assert h.option("check_values") == 2
        y0 = h.spmv(x, np.full(csr.m, np.nan))
        val *= -0.5                                                 # a whole-array update in place (Newton / time step); a power of two: exact
        y1 = h.spmv(x, np.full(csr.m, np.nan))
        assert np.array_equal(y1, -0.5 * y0) and not np.array_equal(y1, y0)
        y2 = h.spmv(x, np.full(csr.m, np.nan))                      # unchanged since: same result, no refresh
        assert np.array_equal(y2, y1)
    if gpus == 0:
        dev = torch.device("cuda:0")
        rp, ci, va = (torch.from_numpy(a).to(dev) for a in (csr.rowptr, csr.colidx, csr.val.copy()))
        xd = torch.from_numpy(x).to(dev)
        with api.Handle(csr.m, csr.n, rp, ci, va, method) as h:
            ya = h.spmv(xd, torch.empty(csr.m, dtype=torch.float64, device=dev)).clone()
            va.mul_(-0.5)
            yb = h.spmv(xd, torch.empty(csr.m, dtype=torch.float64, device=dev))
            assert torch.equal(ya, yb)                              # device arrays: spmv_hip_update_values or check_values = 1


@pytest.mark.parametrize("devptr", [False, True])
def test_check_values_option_makes_in_place_updates_visible(devptr):
    """Option check_values (env SPMV_HIP_CHECK_VALUES=1): spmv() checksums Matrix_Val on every call and refreshes
    by itself -- the pure drop-in for callers that update values in place."""
    import torch
    csr, x, _ = load_golden("skewed_f64_uniform")
    if devptr:
        dev = torch.device("cuda:0")
        rp, ci, va = (torch.from_numpy(a).to(dev) for a in (csr.rowptr, csr.colidx, csr.val.copy()))
        xd = torch.from_numpy(x).to(dev)
    else:
        rp, ci, va, xd = csr.rowptr, csr.colidx, csr.val.copy(), x
    api.set_thread_option("check_values", 1)
    try:
        h = api.Handle(csr.m, csr.n, rp, ci, va, M.Method_CSR5SPMV)
    finally:
        api.clear_thread_options()
    assert h.option("check_values") == 1
    def mul():
        y = torch.empty(csr.m, dtype=torch.float64, device=xd.device) if devptr else np.empty(csr.m)
        h.spmv(xd, y)
        return y.cpu().numpy() if devptr else y
    y0 = mul()
    if devptr:
        va.mul_(3.0)
    else:
        va *= 3.0
    y1 = mul()
    h.close()
    want = run_host(synth.CSR(csr.m, csr.n, csr.rowptr, csr.colidx, csr.val * 3.0), x, M.Method_CSR5SPMV)[0]
    assert not np.array_equal(y0, y1) and np.array_equal(y1, want)


def test_options_are_per_handle_and_thread_local_overrides_do_not_leak():
    csr, x, y_ref = load_golden("rowlen_sweep_f64_eighths")
    api.set_thread_option("lanes_per_row", 16)
    try:
        a = api.Handle(csr.m, csr.n, csr.rowptr, csr.colidx, csr.val, M.Method_Parallel)
    finally:
        api.clear_thread_options()
    b = api.Handle(csr.m, csr.n, csr.rowptr, csr.colidx, csr.val, M.Method_Parallel)
    try:
        assert a.option("lanes_per_row") == 16 and a.info()["lanes_per_row"] == 16
        assert b.option("lanes_per_row") == 0 and api.get_option("lanes_per_row") == 0
        for h in (a, b):
            assert np.array_equal(h.spmv(x, np.empty(csr.m)), y_ref)
    finally:
        a.close(); b.close()


def test_threads_create_differently_tuned_handles_concurrently():
    """Options are resolved per handle (process-wide -> thread-local override -> snapshot): four threads create handles with
    four lane widths at the same time, multiply, refresh values and destroy, without seeing each other's settings."""
    import threading
    csr, x, y_ref = load_golden("rowlen_sweep_f64_eighths")
    out, errs = {}, []

    def work(lanes):
        try:
            for _ in range(5):
                api.set_thread_option("lanes_per_row", lanes)
                val = csr.val.copy()
                h = api.Handle(csr.m, csr.n, csr.rowptr, csr.colidx, val, M.Method_Parallel)
                assert h.option("lanes_per_row") == lanes and h.info()["lanes_per_row"] == lanes
                y = h.spmv(x, np.empty(csr.m))
                val *= 2.0
                h.update_values(val)
                y2 = h.spmv(x, np.empty(csr.m))
                h.close()
                assert np.array_equal(y, y_ref) and np.array_equal(y2, 2.0 * y_ref)
            out[lanes] = True
        except Exception as e:                      # noqa: BLE001
            errs.append((lanes, repr(e)))
        finally:
            api.clear_thread_options()

    threads = [threading.Thread(target=work, args=(l,)) for l in (2, 8, 16, 64)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs, errs
    assert sorted(out) == [2, 8, 16, 64] and api.get_option("lanes_per_row") == 0


def test_stream_bytes_model_of_the_storage_format():
    """spmv_hip_info.stream_bytes: what one launch has to move given the format -- below alg_bytes when the 16-bit
    slot stream replaces ColIdx, above it for SELL's padding; x charged by the windows actually staged."""
    import torch
    dev = torch.device("cuda:0")
    m, n, rp, ci, va = synth.banded_device(400_000, 400_000, 32, "eighths", torch.float64, dev, 3)
    nnz = int(rp[-1].item())
    with api.Handle(m, n, rp, ci, va, M.Method_Parallel) as h:
        i = h.info()
    assert i["kernel_name"] == "csr_vector_tile_kernel" and i["x_groups_staged"] == i["x_groups"]
    # every row one run of 32 consecutive columns except the wrapped rows at both ends: all tiles but the first and the last are RUN tiles
    assert nnz - 2 * 256 * 32 <= i["run_nnz"] <= nnz, i
    fixed = 4 * (m + 1) + nnz * 8 + (nnz - i["run_nnz"]) * 2 + i["run_nnz"] // 32 * 2 + 8 * m   # RowPtr, values, 16-bit slots per entry / per row of a RUN tile, y
    assert fixed < i["stream_bytes"] < fixed + 8 * 2 * n + 300 * i["x_groups"]   # + window tables + staged x (tiles overlap by the band)
    assert 8 * n <= i["x_bytes"] < 8 * 2 * n and i["stream_bytes"] < i["alg_bytes"]
    # a second diagonal 1000 columns away: two runs per row -> no RUN tile, the 16-bit column stream is read again
    ci2 = ci.clone().view(m, 32)
    ci2[:, 16:] = (ci2[:, 16:] + 1000) % n
    ci2 = torch.sort(ci2, dim=1).values.reshape(-1).contiguous()
    with api.Handle(m, n, rp, ci2, va, M.Method_Parallel) as h:
        i2 = h.info()
    # two runs per row: no RUN tile -- every row the same 32 offsets: TEMPLATE tiles (no column stream either, a wider x window)
    assert i2["kernel_name"] == "csr_vector_tile_kernel" and i2["run_nnz"] == 0 and i2["tmpl_nnz"] >= 0.99 * nnz and i2["stream_bytes"] > i["stream_bytes"], i2
    with api.Handle(m, n, rp, ci, va, M.Method_Serial) as h:
        i = h.info()
    assert i["stream_bytes"] == i["alg_bytes"]
    with api.Handle(m, n, rp, ci, va, M.Method_SellCSigma) as h:
        i = h.info()
    # banded: nearly every sigma window is a RUN group (a word per row slot instead of 16 bits per stored entry); padding still counts
    assert i["run_nnz"] >= 0.99 * nnz and i["stored_nnz"] * 8 + 12 * m <= i["stream_bytes"] < i["stored_nnz"] * 10 + 8 * m, i
    with api.Handle(m, n, rp, ci2, va, M.Method_SellCSigma) as h:       # two runs per row: not RUN groups -- but every row the same offsets: TEMPLATE groups (round 4)
        i = h.info()
    assert i["run_nnz"] == 0 and i["tmpl_nnz"] >= 0.99 * nnz and i["stored_nnz"] * 8 + 12 * m <= i["stream_bytes"] < i["stored_nnz"] * 10 + 8 * m, i
    api.set_option("run_tiles", 0)                                      # ... and with the RUN / TEMPLATE checks off the 16-bit slot slabs are read
    try:
        with api.Handle(m, n, rp, ci2, va, M.Method_SellCSigma) as h:
            i = h.info()
    finally:
        api.set_option("run_tiles", 1)
    assert i["run_nnz"] == 0 and i["tmpl_nnz"] == 0 and i["stream_bytes"] >= i["stored_nnz"] * 10 + 8 * m, i


def test_auto_method_measured_mode_builds_times_and_keeps_a_candidate():
    """auto_method = 2: create() builds the candidate schedules, times each on scratch vectors and keeps the
    fastest (spmv_api.c).  Whatever wins, the handle must report it and the result must be exact."""
    import torch
    dev = torch.device("cuda:0")
    m, n, rp, ci, va = synth.banded_device(200_000, 200_000, 32, "eighths", torch.float64, dev, 3)
    g = torch.Generator(device=dev); g.manual_seed(2)
    x = torch.randint(-8, 9, (n,), generator=g, device=dev).double() * 0.125
    want = (va.double() * x[ci.long()]).view(m, 32).sum(1)
    api.set_option("auto_method", 2)
    try:
        y = torch.full((m,), float("nan"), dtype=torch.float64, device=dev)
        with api.Handle(m, n, rp, ci, va, M.Method_Serial) as h:
            h.spmv(x, y)
            chosen = h.method
            sched = h.info()["schedule_name"]
    finally:
        api.set_option("auto_method", 0)
    torch.cuda.synchronize()
    assert chosen in (M.Method_Parallel, M.Method_CSR5SPMV, M.Method_SellCSigma, M.Method_Balanced_Yid, M.Method_Balanced, M.Method_Balanced2)
    assert sched in ("csr-vector", "csr5", "sell-c-sigma", "nnz-split", "row-block")
    assert torch.equal(y, want)


def test_check_values_on_a_reordered_handle_rebuilds_instead_of_refreshing():
    """with options reorder = 1 and check_values = 1 the resident matrix is P A P^T, whose value order is not the
    caller's; a checksum mismatch must not push the caller's CSR-order values into it.  spmv() re-uploads and re-inspects."""
    rng = np.random.default_rng(5)
    m = 6000
    band = synth.banded(m, m, 5, 5, "eighths", np.float64, seed=2)
    sc = rng.permutation(m)
    inv = np.empty(m, dtype=np.int64); inv[sc] = np.arange(m)
    lens = np.diff(band.rowptr)[sc]
    rp = np.zeros(m + 1, dtype=np.int32); np.cumsum(lens, out=rp[1:])
    ci = np.empty(band.nnz, dtype=np.int32); va = np.empty(band.nnz)
    for r in range(m):
        s0, s1 = band.rowptr[sc[r]], band.rowptr[sc[r] + 1]
        ci[rp[r]:rp[r + 1]] = inv[band.colidx[s0:s1]]
        va[rp[r]:rp[r + 1]] = band.val[s0:s1]
    x = synth.fill_x(m, "eighths", np.float64, 5)
    api.set_thread_option("reorder", 1)
    api.set_thread_option("check_values", 1)
    try:
        h = api.Handle(m, m, rp, ci, va, M.Method_CSR5SPMV)
    finally:
        api.clear_thread_options()
    try:
        def mul():
            index = h.index
            assert index is not None
            yy = np.full(m, np.nan)
            h.spmv(x[index], yy)
            y = np.empty(m); y[index] = yy
            return y
        assert np.array_equal(mul(), oracle.spmv_serial(synth.CSR(m, m, rp, ci, va), x))
        va[:] = np.roll(va, 7) * 2.0                                   # in place, behind the same pointer: not a permutation-invariant change
        assert np.array_equal(mul(), oracle.spmv_serial(synth.CSR(m, m, rp, ci, va), x))
    finally:
        h.close()


def test_check_values_sees_swapped_values():
    """The checksum is position-weighted: two values exchanged in place (same multiset, same plain sum) are noticed."""
    csr, x, _ = load_golden("banded_f64_uniform")
    va = csr.val.copy()
    api.set_thread_option("check_values", 1)
    try:
        h = api.Handle(csr.m, csr.n, csr.rowptr, csr.colidx, va, M.Method_Parallel)
    finally:
        api.clear_thread_options()
    try:
        y0 = h.spmv(x, np.empty(csr.m))
        i, j = 3, csr.nnz - 5
        assert va[i] != va[j]
        va[i], va[j] = va[j], va[i]
        y1 = h.spmv(x, np.empty(csr.m))
        want = run_host(synth.CSR(csr.m, csr.n, csr.rowptr, csr.colidx, va), x, M.Method_Parallel)[0]
        assert np.array_equal(y1, want) and not np.array_equal(y0, y1)
    finally:
        h.close()


def test_block_rows_option_is_capped_and_every_row_is_written():
    """the option table once allowed block_rows = 32768, which left half of y unwritten.  The range now ends at 16384."""
    with pytest.raises(ValueError):
        api.set_option("block_rows", 32768)
    csr, x, y_ref = load_golden("powerlaw_f64_eighths")
    keep = {k: api.get_option(k) for k in ("cache_block", "block_rows")}
    try:
        api.set_option("cache_block", 2)
        api.set_option("block_rows", 16384)
        y, _ = run_host(csr, x, M.Method_Balanced2)
        assert not np.isnan(y).any() and np.array_equal(y, y_ref)
    finally:
        for k, v in keep.items():
            api.set_option(k, v)


def _run_rows_matrix(m, n, lens, start, dtype, dev, seed):
    """CSR whose row i holds the run of lens[i] consecutive columns start[i] .. (exact 'eighths' values)"""
    import torch
    rp = torch.zeros(m + 1, dtype=torch.int64, device=dev)
    torch.cumsum(lens, 0, out=rp[1:])
    nnz = int(rp[-1])
    row_of = torch.repeat_interleave(torch.arange(m, device=dev), lens)
    ci = (start[row_of] + torch.arange(nnz, device=dev) - rp[:-1][row_of]).to(torch.int32)
    assert int(ci.min()) >= 0 and int(ci.max()) < n
    g = torch.Generator(device=dev); g.manual_seed(seed)
    va = (torch.randint(-8, 9, (nnz,), generator=g, device=dev) * 0.125).to(dtype)
    return rp.to(torch.int32), ci, va, row_of


def _segment_sums(prod, rp):
    import torch
    cs = torch.cat([torch.zeros(1, dtype=torch.float64, device=prod.device), torch.cumsum(prod.double(), 0)])
    return cs[rp[1:].long()] - cs[rp[:-1].long()]


@pytest.mark.parametrize("method", [M.Method_Parallel, M.Method_Balanced, M.Method_SellCSigma, M.Method_CSR5SPMV, M.Method_Balanced_Yid], ids=lambda m: m.name)
@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("shape", ["ragged", "two_bands", "chunks", "wide", "broken_rows"])
def test_run_tiles_need_no_column_stream(shape, dtype, method):
    """RUN tiles of the row-granular schedules -- the CSR-vector tile kernels (csr_vector_tile.hpp) and the SELL slabs (sell.hpp: RUN window
    groups) --: rows that are one run of consecutive columns get their LDS slots from 16 bits (SELL: a word) per ROW.  Exact data, so the sums
    must equal the definition bit for bit.
      ragged       run lengths 0..40 (empty rows, single entries, rows past one 4L-entry chunk), runs start within +-600 of the diagonal
      two_bands    even rows run near the diagonal, odd rows 200 000 columns away: two x windows per tile, every row still one run
      chunks       64..200 entries per row: several chunks per row, under the long-row threshold
      wide         8-entry runs starting within +-5000 columns of the diagonal: the fp64 tile's x span exceeds 48 KiB -> wide form (slot indices)
      broken_rows  the ragged matrix with one entry of every 3000th row moved: those tiles fall back to the column stream, the others stay RUN"""
    import torch
    dev = torch.device("cuda:0")
    tdt = torch.float64 if dtype == "f64" else torch.float32
    m = n = 300_000
    g = torch.Generator(device=dev); g.manual_seed(11)
    rows = torch.arange(m, device=dev)
    if shape in ("ragged", "broken_rows"):
        lens = torch.randint(0, 41, (m,), generator=g, device=dev)
        start = (rows + torch.randint(-600, 601, (m,), generator=g, device=dev)).clamp_(0, n - 41)
    elif shape == "two_bands":
        lens = torch.full((m,), 12, device=dev)
        start = torch.where(rows % 2 == 0, rows, (rows + 200_000) % n).clamp_(0, n - 12)
    elif shape == "chunks":
        lens = torch.randint(64, 201, (m,), generator=g, device=dev)
        start = (rows - 100).clamp_(0, n - 201)
    else:
        lens = torch.full((m,), 8, device=dev)
        start = (rows + torch.randint(-5000, 5001, (m,), generator=g, device=dev)).clamp_(0, n - 8)
    rp, ci, va, row_of = _run_rows_matrix(m, n, lens, start, tdt, dev, 5)
    nnz = int(rp[-1])
    broken = 0
    if shape == "broken_rows":
        pick = torch.nonzero((rows % 3000 == 7) & (lens >= 2)).flatten()
        pos = rp[pick].long() + 1
        ci[pos] = (ci[pos] + 3).clamp_(max=n - 1)         # second entry of the row: no longer c0 + 1 (a duplicate of a later column is legal CSR)
        broken = int(pick.numel())
    x = (torch.randint(-8, 9, (n,), generator=g, device=dev) * 0.125).to(tdt)
    want = _segment_sums(va.double() * x.double()[ci.long()], rp).to(tdt)
    y = torch.full((m,), float("nan"), dtype=tdt, device=dev)
    if shape == "chunks":   # left alone the planner hands every row above 64 entries to the long-row path (CSR5 sub-matrix): force 4 lanes per row,
        api.set_thread_option("lanes_per_row", 4)   # i.e. 16-entry chunks, up to 13 per row, long-row threshold 256
    try:
        h = api.Handle(m, n, rp, ci, va, method)
    finally:
        api.clear_thread_options()
    with h:
        h.spmv(x, y)
        info = h.info()
        torch.cuda.synchronize()
        assert torch.equal(y, want), (info["kernel_name"], int((y != want).sum()))
        entry_granular = method in (M.Method_CSR5SPMV, M.Method_Balanced_Yid)
        assert info["kernel_name"] in ("csr_vector_tile_kernel", "csr_vector_rows_kernel", "sell_window_kernel", "csr5_group_kernel", "csr5_group_pipe_kernel",
                                       "nat_group_kernel") and info["cache_blocked"] == 0, info
        if entry_granular:
            # CSR5 tiles (csr5.hpp RUN groups: a word per lane and tile) need runs with at most one row start per lane of SIGMA entries: the
            # 64..200-entry rows qualify (all but the matrix's last, partly filled tile); shorter rows keep the 16-bit slot stream, and so does
            # the natural-layout nnz-split kernel (no gain measured there)
            if shape == "chunks" and method == M.Method_CSR5SPMV:
                assert 0.95 * nnz <= info["run_nnz"] <= nnz, (info["run_nnz"], nnz)
            if method == M.Method_Balanced_Yid:
                assert info["run_nnz"] == 0
        elif shape == "broken_rows":
            if method == M.Method_Parallel:   # 256-row tiles: exactly the tiles holding a moved entry read their column stream
                assert nnz - broken * 256 * 40 <= info["run_nnz"] < nnz, (info["run_nnz"], nnz, broken)
            elif method == M.Method_SellCSigma:   # the unit is the window group (1024 rows and up): most groups hold no moved entry
                assert 0 < info["run_nnz"] < nnz, (info["run_nnz"], nnz, broken)
        elif method == M.Method_SellCSigma:           # the slabs hold the rows under the planner's long-row threshold only (the others: CSR5 sub-matrix);
            assert 0 < info["run_nnz"] <= info["stored_nnz"] and info["run_nnz"] >= 0.7 * info["stored_nnz"], info   # stored = slab entries incl. padding
        else:
            assert info["run_nnz"] == nnz, (info["run_nnz"], nnz, info["x_groups"], info["x_groups_staged"])
        va2 = (va * 2).contiguous()                       # values only: the row slots stay
        h.update_values(va2)
        h.spmv(x, y)
        torch.cuda.synchronize()
        assert torch.equal(y, 2 * want)


@pytest.mark.parametrize("method", [M.Method_Parallel, M.Method_Balanced], ids=lambda m: m.name)
@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("shape", ["stencil5", "pattern40", "pattern64", "pattern65", "one_odd_row", "nine_lists", "empty_rows", "far_bands"])
def test_template_tiles_need_no_column_stream(shape, dtype, method):
    """TEMPLATE tiles (round 4, csr_vector_tile.hpp): a staged tile whose rows use at most 8 different lists of slot offsets from their first entry --
    stencils (interior rows + the few edge patterns), rows assembled from a few element patterns -- reads 16 + 8 bits per ROW and the lists once per
    TILE, no column stream (RUN tiles are the one list 0, 1, 2, ...).  Exact data: bit-equal to the definition; info.tmpl_nnz counts the entries.
      stencil5     2-D 5-point stencil on a 600 x 500 grid, NOT periodic: rows at the grid's edges have 3 or 4 entries -- more lists, still template tiles
      nine_lists   rows cycle through nine different lists: one too many -> BYTE / 16-bit tiles
      pattern40    every row: the same 40 offsets within +-300 of the diagonal (two 4L chunks per row at 8 lanes per row)
      pattern64    64 offsets: the largest template            pattern65   65 offsets: one too many -> BYTE / 16-bit tiles
      one_odd_row  pattern40 with one entry of every 5000th row moved: a second list in those tiles
      empty_rows   pattern40 with every 7th row empty: empty rows read nothing and do not break the template
      far_bands    12 offsets in three bands 150 000 columns apart: three x windows per tile, one template"""
    import torch
    dev = torch.device("cuda:0")
    tdt = torch.float64 if dtype == "f64" else torch.float32
    g = torch.Generator(device=dev); g.manual_seed(31)
    if shape == "stencil5":
        nx, ny = 600, 500
        m = n = nx * ny
        rows = torch.arange(m, device=dev)
        ix, iy = rows % nx, rows // nx
        cand = torch.stack([rows - nx, rows - 1, rows, rows + 1, rows + nx], 1)
        ok = torch.stack([iy > 0, ix > 0, torch.ones_like(ix, dtype=torch.bool), ix < nx - 1, iy < ny - 1], 1)
        lens = ok.sum(1)
        ci = cand[ok].to(torch.int32)
    else:
        m = n = 600_000 if shape == "far_bands" else 300_000
        rows = torch.arange(m, device=dev)
        k = {"pattern40": 40, "pattern64": 64, "pattern65": 65, "one_odd_row": 40, "nine_lists": 20, "empty_rows": 40, "far_bands": 12}[shape]
        if shape == "far_bands":
            offs = torch.tensor([-150_001, -150_000, -149_998, -149_990, -3, -1, 0, 2, 149_990, 150_000, 150_003, 150_007], device=dev)
        else:
            offs = torch.sort(torch.randperm(601, generator=g, device=dev)[:k] - 300).values
        lo, hi = int(-offs.min()), int(n - 1 - offs.max()) - 9
        base = rows.clamp(lo, hi)               # the first and last rows repeat a neighbour's columns: still the same offsets
        cand = base[:, None] + offs[None, :]
        if shape == "nine_lists":           # the last column of the list moves with the row number modulo 9
            cand[:, -1] = cand[:, -1] + rows % 9
        lens = torch.full((m,), k, device=dev)
        if shape == "empty_rows":
            lens[rows % 7 == 3] = 0
        keep = torch.arange(k, device=dev)[None, :] < lens[:, None]
        ci = cand[keep].to(torch.int32)
    rp = torch.zeros(m + 1, dtype=torch.int64, device=dev)
    torch.cumsum(lens, 0, out=rp[1:])
    nnz = int(rp[-1])
    odd = 0
    if shape == "one_odd_row":
        pick = torch.nonzero(rows % 5000 == 11).flatten()
        pos = rp[pick] + 5
        ci[pos] = ci[pos] + 1 if False else ci[pos - 1]            # a duplicate of the previous column: legal CSR, another offset list
        odd = int(pick.numel())
    va = (torch.randint(-8, 9, (nnz,), generator=g, device=dev) * 0.125).to(tdt)
    x = (torch.randint(-8, 9, (n,), generator=g, device=dev) * 0.125).to(tdt)
    rp32 = rp.to(torch.int32)
    want = _segment_sums(va.double() * x.double()[ci.long()], rp32).to(tdt)
    y = torch.full((m,), float("nan"), dtype=tdt, device=dev)
    api.set_thread_option("lanes_per_row", 8)
    try:
        h = api.Handle(m, n, rp32, ci, va, method)
    finally:
        api.clear_thread_options()
    with h:
        h.spmv(x, y)
        info = h.info()
        torch.cuda.synchronize()
        assert torch.equal(y, want), (info["kernel_name"], int((y != want).sum()))
        assert info["kernel_name"] in ("csr_vector_tile_kernel", "csr_vector_rows_kernel") and info["cache_blocked"] == 0, info
        t = info["tmpl_nnz"]
        if shape == "pattern65":
            assert t == 0 and info["byte_nnz"] + info["run_nnz"] <= nnz, info
        elif shape == "nine_lists":
            assert t == 0, info
        elif shape == "far_bands" and dtype == "f64" and method == M.Method_Balanced:
            assert t <= nnz                                             # block spans decide what stages: exactness is the test
        else:
            assert t >= 0.99 * nnz, (t, nnz, info["x_groups"], info["x_groups_staged"])
        h.update_values((va * 2).contiguous())
        h.spmv(x, y)
        torch.cuda.synchronize()
        assert torch.equal(y, 2 * want)


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("shape", ["stencil27", "pattern40", "nine_lists", "edges"])
def test_sell_template_groups_need_no_slot_slab(shape, dtype):
    """TEMPLATE groups of the SELL slabs (sell.hpp, round 4): when the rows of a staged window group use at most 8 lists of slot offsets, a word per row
    slot (first slot | length | list number) replaces the 16-bit slot slab -- the SELL counterpart of the CSR-vector TEMPLATE tiles.  Exact data.
      stencil27   27-point stencil on a periodic 48^3 grid              pattern40  every row the same 40 offsets within +-300 of the diagonal
      nine_lists  rows cycle through nine lists: one too many -> the slot slab is read          edges  2-D 5-point stencil, NOT periodic: rows of 3-5 entries"""
    import torch
    dev = torch.device("cuda:0")
    tdt = torch.float64 if dtype == "f64" else torch.float32
    g = torch.Generator(device=dev); g.manual_seed(41)
    if shape == "stencil27":
        m, n, rp, ci, va = synth.stencil27_device(48, "eighths", tdt, dev, 5)
    elif shape == "edges":
        nx, ny = 300, 400
        m = n = nx * ny
        rows = torch.arange(m, device=dev)
        ix, iy = rows % nx, rows // nx
        cand = torch.stack([rows - nx, rows - 1, rows, rows + 1, rows + nx], 1)
        ok = torch.stack([iy > 0, ix > 0, torch.ones_like(ix, dtype=torch.bool), ix < nx - 1, iy < ny - 1], 1)
        rp = torch.zeros(m + 1, dtype=torch.int64, device=dev); torch.cumsum(ok.sum(1), 0, out=rp[1:])
        ci = cand[ok].to(torch.int32); rp = rp.to(torch.int32)
        va = (torch.randint(-8, 9, (ci.numel(),), generator=g, device=dev) * 0.125).to(tdt)
    else:
        m = n = 200_000
        rows = torch.arange(m, device=dev)
        k = 40 if shape == "pattern40" else 20
        offs = torch.sort(torch.randperm(601, generator=g, device=dev)[:k] - 300).values
        base = rows.clamp(300, n - 1 - 300 - 9)
        cand = base[:, None] + offs[None, :]
        if shape == "nine_lists":
            cand[:, -1] = cand[:, -1] + rows % 9
        ci = cand.reshape(-1).to(torch.int32)
        rp = torch.arange(0, (m + 1) * k, k, dtype=torch.int32, device=dev)
        va = (torch.randint(-8, 9, (m * k,), generator=g, device=dev) * 0.125).to(tdt)
    nnz = int(rp[-1])
    x = (torch.randint(-8, 9, (n,), generator=g, device=dev) * 0.125).to(tdt)
    want = _segment_sums(va.double() * x.double()[ci.long()], rp).to(tdt)
    y = torch.full((m,), float("nan"), dtype=tdt, device=dev)
    with api.Handle(m, n, rp, ci, va, M.Method_SellCSigma) as h:
        h.spmv(x, y)
        info = h.info()
        torch.cuda.synchronize()
        assert torch.equal(y, want), (info["kernel_name"], int((y != want).sum()))
        assert info["kernel_name"] == "sell_window_kernel", info
        if shape == "nine_lists":
            assert info["tmpl_nnz"] == 0, info
        elif shape == "stencil27":
            assert info["tmpl_nnz"] >= 0.8 * nnz, (info["tmpl_nnz"], nnz)       # groups in which the periodic grid wraps hold more lists
        else:
            assert info["tmpl_nnz"] >= 0.95 * nnz, (info["tmpl_nnz"], nnz)
        h.update_values((va * 2).contiguous())
        h.spmv(x, y)
        torch.cuda.synchronize()
        assert torch.equal(y, 2 * want)


def _holes_rows_matrix(m, n, lens, start, span, dtype, dev, seed, reverse=False):
    """CSR whose row i holds lens[i] DISTINCT columns drawn from [start[i], start[i] + span[i]) -- first and last column of the range always present
    when lens[i] >= 2, so the row's column span is exactly span[i] --, ascending (or descending: unsorted rows are legal CSR); exact values"""
    import torch
    g = torch.Generator(device=dev); g.manual_seed(seed)
    W = int(span.max())
    keys = torch.rand(m, W, generator=g, device=dev)
    cols = torch.arange(W, device=dev)[None, :]
    keys = torch.where(cols < span[:, None], keys, torch.full_like(keys, 2.0))         # outside the row's range: never picked
    keys[:, 0] = -1.0                                                                  # the range's first column: always
    keys[torch.arange(m, device=dev), (span - 1).clamp_(min=0)] = -0.5                 # ... and its last
    order = torch.argsort(keys, dim=1)                                                 # row i keeps the lens[i] smallest keys
    keep = torch.arange(W, device=dev)[None, :] < lens[:, None]
    picked = torch.where(keep, order, torch.full_like(order, W + 1))
    picked = torch.sort(picked, dim=1, descending=reverse).values
    rp = torch.zeros(m + 1, dtype=torch.int64, device=dev)
    torch.cumsum(lens, 0, out=rp[1:])
    mask = picked <= W
    ci = (start[:, None] + picked)[mask].to(torch.int32)
    nnz = int(rp[-1])
    assert ci.numel() == nnz and int(ci.min()) >= 0 and int(ci.max()) < n
    va = (torch.randint(-8, 9, (nnz,), generator=g, device=dev) * 0.125).to(dtype)
    return rp.to(torch.int32), ci, va


@pytest.mark.parametrize("method", [M.Method_Parallel, M.Method_Balanced], ids=lambda m: m.name)
@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("shape", ["holes", "ragged", "unsorted", "two_bands", "chunks", "span_edge", "wide"])
def test_byte_tiles_read_one_byte_per_entry(shape, dtype, method):
    """BYTE tiles (round 4, csr_vector_tile.hpp): a staged tile in which every row's LDS slots lie within 255 of the row's smallest slot -- bands with
    holes, block rows, anything whose rows span under 256 columns of one window -- reads a BYTE per entry + 16 bits per row instead of 16 bits per
    entry.  Exact data: the sums equal the definition bit for bit; info.byte_nnz says how many entries took the byte stream.
      holes      32 of the 43 columns of a band (BASELINE config 2 with 25 % holes)
      ragged     0..40 entries per row out of ranges of 1..200 columns that start within +-600 of the diagonal (empty rows, single entries)
      unsorted   the same with every row's columns in DESCENDING order: the row's smallest slot is not its first entry's
      two_bands  odd rows 200 000 columns away: two x windows per tile, every row inside one of them
      chunks     64..200 entries per row from ranges of up to 255 columns: several chunks per row
      span_edge  ranges of exactly 256 columns (span 255: the largest byte) everywhere but in every 40th 256-row tile, where one row spans 257
                 columns: those tiles keep the 16-bit stream, the others are BYTE tiles
      wide       8 entries from ranges of 20 columns starting within +-5000 columns of the diagonal: fp64 -> wide form (slot indices, 1024-row blocks)"""
    import torch
    dev = torch.device("cuda:0")
    tdt = torch.float64 if dtype == "f64" else torch.float32
    m, n, rp, ci, va, g = _byte_shape_matrix(shape, tdt, dev)
    _byte_tiles_body(shape, method, m, n, rp, ci, va, g, tdt, dev)


def _byte_shape_matrix(shape, tdt, dev):
    import torch
    m = n = 300_000
    g = torch.Generator(device=dev); g.manual_seed(21)
    rows = torch.arange(m, device=dev)
    reverse = shape == "unsorted"
    if shape == "holes":
        m, n, rp, ci, va = synth.banded_holes_device(m, n, 32, 0.25, "eighths", tdt, dev, 3)
    else:
        if shape in ("ragged", "unsorted"):
            span = torch.randint(1, 201, (m,), generator=g, device=dev)
            lens = torch.minimum(torch.randint(0, 41, (m,), generator=g, device=dev), span)
            start = (rows + torch.randint(-600, 601, (m,), generator=g, device=dev)).clamp_(0, n - 201)
        elif shape == "two_bands":
            span = torch.full((m,), 40, device=dev)
            lens = torch.full((m,), 12, device=dev)
            start = torch.where(rows % 2 == 0, rows, (rows + 200_000) % n).clamp_(0, n - 41)
        elif shape == "chunks":
            span = torch.randint(200, 256, (m,), generator=g, device=dev)
            lens = torch.randint(64, 201, (m,), generator=g, device=dev)
            start = (rows - 100).clamp_(0, n - 257)
        elif shape == "span_edge":
            span = torch.full((m,), 256, device=dev)
            span[(rows % (40 * 256)) == 77] = 257
            lens = torch.full((m,), 10, device=dev)
            start = (rows - 100).clamp_(0, n - 258)
        else:
            span = torch.full((m,), 20, device=dev)
            lens = torch.full((m,), 8, device=dev)
            start = (rows + torch.randint(-5000, 5001, (m,), generator=g, device=dev)).clamp_(0, n - 21)
        rp, ci, va = _holes_rows_matrix(m, n, lens, start, span, tdt, dev, 6, reverse)
    return m, n, rp, ci, va, g


def _byte_tiles_body(shape, method, m, n, rp, ci, va, g, tdt, dev):
    import torch
    nnz = int(rp[-1])
    x = (torch.randint(-8, 9, (n,), generator=g, device=dev) * 0.125).to(tdt)
    want = _segment_sums(va.double() * x.double()[ci.long()], rp).to(tdt)
    y = torch.full((m,), float("nan"), dtype=tdt, device=dev)
    if shape == "chunks":
        api.set_thread_option("lanes_per_row", 4)
    try:
        h = api.Handle(m, n, rp, ci, va, method)
    finally:
        api.clear_thread_options()
    with h:
        h.spmv(x, y)
        info = h.info()
        torch.cuda.synchronize()
        assert torch.equal(y, want), (info["kernel_name"], int((y != want).sum()))
        assert info["kernel_name"] in ("csr_vector_tile_kernel", "csr_vector_rows_kernel") and info["cache_blocked"] == 0, info
        assert info["run_nnz"] + info["byte_nnz"] <= nnz
        if shape == "span_edge":
            if method == M.Method_Parallel:   # every 40th tile holds a 257-column row
                assert 0.96 * nnz <= info["byte_nnz"] <= 0.985 * nnz, (info["byte_nnz"], nnz)
            else:
                assert 0 < info["byte_nnz"] < nnz, (info["byte_nnz"], nnz)
        elif shape == "holes":     # the band wraps at both ends of the matrix: the first and the last tile see columns at both ends of x (two windows)
            assert nnz - 4 * 256 * 32 <= info["byte_nnz"] < nnz, (info["byte_nnz"], nnz)
        else:
            assert info["byte_nnz"] + info["run_nnz"] == nnz, (info["byte_nnz"], info["run_nnz"], nnz, info["x_groups"], info["x_groups_staged"])
            assert info["byte_nnz"] >= 0.9 * nnz
        va2 = (va * 2).contiguous()
        h.update_values(va2)
        h.spmv(x, y)
        torch.cuda.synchronize()
        assert torch.equal(y, 2 * want)


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("shape", ["holes", "ragged", "unsorted", "two_bands", "span_edge"])
def test_sell_byte_groups_read_one_byte_per_entry(shape, dtype):
    """BYTE window groups of the SELL slabs (round 4, sell.hpp): when every row of a staged group keeps its slots within 255 of its FIRST one, the group reads
    an 8-bit slab + a word per row slot instead of the 16-bit slab.  Shapes as in test_byte_tiles_read_one_byte_per_entry; rows with descending columns
    ('unsorted') fail the test by construction and keep the 16-bit slab.  Exact data: the definition's bits, with the forms on and off (run_tiles = 0)."""
    import torch
    dev = torch.device("cuda:0")
    tdt = torch.float64 if dtype == "f64" else torch.float32
    m, n, rp, ci, va, g = _byte_shape_matrix(shape, tdt, dev)
    nnz = int(rp[-1])
    x = (torch.randint(-8, 9, (n,), generator=g, device=dev) * 0.125).to(tdt)
    want = _segment_sums(va.double() * x.double()[ci.long()], rp).to(tdt)
    for on in (1, 0):
        api.set_option("run_tiles", on)
        try:
            h = api.Handle(m, n, rp, ci, va, M.Method_SellCSigma)
        finally:
            api.set_option("run_tiles", 1)
        with h:
            y = torch.full((m,), float("nan"), dtype=tdt, device=dev)
            h.spmv(x, y)
            info = h.info()
            torch.cuda.synchronize()
            assert torch.equal(y, want), (info["kernel_name"], int((y != want).sum()))
            assert info["kernel_name"] == "sell_window_kernel" and info["x_groups_staged"] == info["x_groups"], info
            if not on or shape == "unsorted":
                assert info["byte_nnz"] == 0, info
            elif shape == "span_edge":          # one row in 10 240 spans 257 columns: its window group keeps the 16-bit slab
                assert 0.4 * nnz <= info["byte_nnz"] < nnz, (info["byte_nnz"], nnz)
            elif shape == "holes":              # the band wraps at both ends of the matrix
                assert info["byte_nnz"] >= 0.98 * nnz, (info["byte_nnz"], nnz)
            else:
                assert info["byte_nnz"] + info["run_nnz"] + info["tmpl_nnz"] == nnz and info["byte_nnz"] >= 0.9 * nnz, info
            h.update_values((va * 2).contiguous())
            h.spmv(x, y)
            torch.cuda.synchronize()
            assert torch.equal(y, 2 * want)


@pytest.mark.parametrize("method", [M.Method_CSR5SPMV, M.Method_SellCSigma, M.Method_Balanced_Yid, M.Method_Parallel], ids=lambda m: m.name)
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_wide_x_windows_are_staged_from_any_element_alignment(dtype, method):
    """stage_windows reads wide x windows with 16-byte loads from the first 16-byte boundary on: x (and y) handed over one, two and
    three elements past an aligned address must give the bits of the aligned call.  Rows scattered +-3000 columns around the diagonal:
    windows of 6000+ columns, staged by every tile schedule."""
    import torch
    dev = torch.device("cuda:0")
    tdt = torch.float64 if dtype == "f64" else torch.float32
    m = n = 200_000
    lens = torch.full((m,), 20, dtype=torch.int64, device=dev)
    _, _, rp, ci, va = synth.from_row_lengths_device(lens, n, "eighths", tdt, dev, seed=8, local=3000)
    g = torch.Generator(device=dev); g.manual_seed(12)
    xbig = (torch.randint(-8, 9, (n + 8,), generator=g, device=dev) * 0.125).to(tdt)
    ybig = torch.empty(m + 8, dtype=tdt, device=dev)
    want = _segment_sums(va.double() * xbig[:n].double()[ci.long()], rp).to(tdt)
    with api.Handle(m, n, rp, ci, va, method) as h:
        info = h.info()
        assert info["cache_blocked"] == 0 and info["x_groups_staged"] > 0, info
        for off in (0, 1, 2, 3):
            x = xbig[off: off + n]
            x.copy_(xbig[:n].clone())                    # the same values at the shifted address
            y = ybig[off: off + m]
            y.fill_(float("nan"))
            h.spmv(x, y)
            torch.cuda.synchronize()
            assert torch.equal(y, want), (off, info["kernel_name"], int((y != want).sum()))
            xbig[:n].copy_(x.clone())


@pytest.mark.parametrize("shape", ["short_rows_many_empty", "long_rows", "mixed"])
def test_two_deep_csr5_group_kernel(shape):
    """fp32 CSR5 plans whose groups are all staged run csr5_group_pipe_kernel (two tiles in flight per wave, csr5.hpp).  Wide x windows
    (rows scattered +-3000 columns) make the inspector grow the groups to 32 / 64 tiles, i.e. 8 / 16 tiles per wave:
      short_rows_many_empty  0..6 entries per row, 1 in 7 rows empty: the row-mapped form with >= 64 row starts per tile (the rest of
                             the row map is fetched when the tile is computed)
      long_rows              2000..3000 entries per row: tiles inside one row, carries through the fix-up
      mixed                  both, interleaved
    fp64 plans keep the one-deep kernel.  Exact data: bit-equal to the definition; `variant` 61 (one deep) must give the same bits."""
    import torch
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev); g.manual_seed(21)
    if shape == "short_rows_many_empty":
        m = 600_000
        lens = torch.randint(0, 7, (m,), generator=g, device=dev)
    elif shape == "long_rows":
        m = 2_000
        lens = torch.randint(2000, 3001, (m,), generator=g, device=dev)
    else:
        m = 200_000
        lens = torch.randint(0, 9, (m,), generator=g, device=dev)
        lens[::500] = 2500
    n = max(m, 100_000)
    _, _, rp, ci, va = synth.from_row_lengths_device(lens, n, "eighths", torch.float32, dev, seed=3, local=3000)
    x = (torch.randint(-8, 9, (n,), generator=g, device=dev) * 0.125).to(torch.float32)
    want = _segment_sums(va.double() * x.double()[ci.long()], rp).to(torch.float32)
    ys = {}
    keep = api.get_option("csr5_two_deep")
    try:
        for variant in (0, 1):
            api.set_option("csr5_two_deep", variant)
            with api.Handle(m, n, rp, ci, va, M.Method_CSR5SPMV) as h:
                info = h.info()
                y = torch.full((m,), float("nan"), dtype=torch.float32, device=dev)
                h.spmv(x, y)
                torch.cuda.synchronize()
                assert info["kernel_name"] == ("csr5_group_pipe_kernel" if variant == 0 else "csr5_group_kernel"), info
                assert torch.equal(y, want), (shape, variant, int((y != want).sum()))
                ys[variant] = y
    finally:
        api.set_option("csr5_two_deep", keep)
    with api.Handle(m, n, rp, ci, va.double(), M.Method_CSR5SPMV) as h:
        assert h.info()["kernel_name"] == "csr5_group_kernel"


def _edge_lengths(tn, rng, filler):
    """Row lengths that put every forward-completion case on a tile boundary (tiles of tn entries): a row ending exactly on a cut and one starting on it,
    rows with 1 .. tn - 1 entries behind the cut (more than 64: the in-place loop), a row of exactly tn entries started mid-tile, rows of tn + 1 and
    3 tn + 5 entries (long: a workgroup each), short rows in between."""
    lens, pos = [], 0

    def short_until(target):          # 1-3-entry rows up to position `target`
        nonlocal pos
        while pos < target:
            k = int(min(rng.integers(1, 4), target - pos))
            lens.append(k); pos += k

    def row(k):
        nonlocal pos
        lens.append(int(k)); pos += int(k)

    cut = tn
    short_until(cut)                                   # a row ends exactly on the first cut, the next starts on it
    for back, fwd in [(1, 1), (3, 63), (2, 64), (5, 65), (7, 130), (1, tn - 1), (tn // 2, tn // 2), (tn - 1, 1), (10, tn - 10 + 1), (4, 3 * tn + 1), (tn - 3, 8)]:
        cut = (pos // tn + 2) * tn
        short_until(cut - back)
        row(back + fwd)                                # starts `back` entries before the cut, `fwd` behind it
    short_until(pos + filler)
    return np.asarray(lens, dtype=np.int64)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("sigma", [0, 4, 8, 16])
@pytest.mark.parametrize("shape", ["edges", "edges_empty_rows", "powerlaw", "many_long_rows"])
def test_short_rows_in_one_launch_forward_completion(shape, sigma, dtype):
    """nnz-split tiles that gather through L2 (short heavy-tailed rows, no x window stages) run ONE kernel -- the tile that holds a row's
    start finishes it from the next tile's entries, rows longer than a tile get a workgroup each -- and option row_forward = 0 brings back carries +
    fix-up launch.  Exact data: the reference's bits either way (parallel_balanced_Yid_spmv.c:151-156 is the serial fix-up this replaces); random
    data: the row-relative bar.  More long rows than the list holds: the fix-up form stays."""
    rng = np.random.default_rng(1234 + sigma)
    tn = 64 * (sigma if sigma else 4)                   # auto: sigma = 4 below 2^19 entries
    n = 1 << 21                                         # columns scattered over 2 M: no window stages
    if shape.startswith("edges"):
        lens = _edge_lengths(tn, rng, 40 * tn)
    elif shape == "powerlaw":
        lens = synth.powerlaw_lengths(60000, 3.1, 4700, 1.6, seed=9)
    else:
        lens = rng.integers(1, 4, 40000)
        lens[rng.choice(lens.shape[0], 1500, replace=False)] = tn + 1 + rng.integers(0, 40, 1500)
    if shape == "edges_empty_rows":                     # empty rows in front, behind and in between (the tiles then run over the compacted row space)
        lens = np.concatenate([np.zeros(3, np.int64), np.insert(lens, rng.choice(lens.shape[0], lens.shape[0] // 9), 0), np.zeros(5, np.int64)])
    for kind in ("eighths", "uniform"):
        csr = synth.from_row_lengths(lens, n, kind, dtype, seed=31)
        x = synth.fill_x(n, kind, dtype, 77)
        y_ref = oracle.spmv_serial(csr, x)
        got = {}
        for fwd in (1, 0):
            api.set_option("row_forward", fwd)
            api.set_option("csr5_sigma", sigma)
            try:
                y = np.full(csr.m, np.nan, dtype=dtype)
                with api.Handle(csr.m, csr.n, csr.rowptr, csr.colidx, csr.val, M.Method_Balanced2) as h:
                    h.spmv(x, y)
                    info = h.info()
                    y2 = np.full(csr.m, np.nan, dtype=dtype)
                    h.spmv(x, y2)
            finally:
                api.set_option("row_forward", 1)
                api.set_option("csr5_sigma", 0)
            one_launch = fwd == 1 and shape != "many_long_rows"
            assert info["launch_kernels"] == (["nat_kernel"] if one_launch else ["nat_kernel", "csr5_fixup_kernel"]), (info["launch_kernels"], fwd)
            check(y, csr, x, y_ref, exact=kind == "eighths")
            assert np.array_equal(y.view(np.uint8), y2.view(np.uint8))        # bit-reproducible call to call
            got[fwd] = y
        if kind == "eighths":
            assert np.array_equal(got[0].view(np.uint8), got[1].view(np.uint8))
