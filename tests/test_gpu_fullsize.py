"""GPU, BASELINE.json's full sizes: size-independent properties instead of a CPU oracle pass.

  * exact-arithmetic inputs (values and x are multiples of 1/8): every schedule must give the SAME
    BITS, and those bits must equal a torch fp64 evaluation of the definition of y = A x;
  * linearity A(a x + b z) = a A x + b A z on random data (tolerance scaled by |A||x|);
  * checksum of checksums: sum(y) for x = 1 equals sum(Val);
  * int32 edge: nnz a few thousand below 2^31 on one GPU (288 GB of HBM makes that a normal size).
"""
import numpy as np
import pytest
import torch

from spmv_amd import api, build, synth

pytestmark = pytest.mark.gpu
M = api.SPMV_METHODS
DEV = "cuda:0"


@pytest.fixture(scope="module", autouse=True)
def _lib():
    build.build()
    api.load()


def _run(m, n, rp, ci, va, x, method):
    y = torch.full((m,), float("nan"), dtype=va.dtype, device=va.device)
    with api.Handle(m, n, rp, ci, va, method) as h:
        h.spmv(x, y)
        sched = h.info()["schedule_name"]
    torch.cuda.synchronize()
    return y, sched


def _definition_regular(m, k, ci, va, x):
    """y from the definition for exactly k nnz per row (fp64 accumulate), chunked."""
    out = torch.empty(m, dtype=torch.float64, device=va.device)
    step = 1 << 21
    for r0 in range(0, m, step):
        r1 = min(m, r0 + step)
        c = ci[r0 * k:r1 * k].long()
        out[r0:r1] = (va[r0 * k:r1 * k].double() * x[c].double()).view(r1 - r0, k).sum(1)
    return out


def _definition_segments(rp, ci, va, x):
    """y from the definition for arbitrary row lengths: exact fp64 prefix sums, differenced at RowPtr."""
    nnz = ci.numel()
    cs = torch.zeros(nnz + 1, dtype=torch.float64, device=va.device)
    step = 1 << 26
    carry = torch.zeros((), dtype=torch.float64, device=va.device)
    for p0 in range(0, nnz, step):
        p1 = min(nnz, p0 + step)
        prod = va[p0:p1].double() * x[ci[p0:p1].long()].double()
        torch.cumsum(prod, 0, out=cs[p0 + 1:p1 + 1])
        cs[p0 + 1:p1 + 1] += carry
        carry = cs[p1].clone()
    r = rp.long()
    return cs[r[1:]] - cs[r[:-1]]


def test_config2_all_schedules_bit_identical_and_equal_definition():
    m = n = 10_000_000
    k = 32
    _, _, rp, ci, va = synth.banded_device(m, n, k, "eighths", torch.float64, DEV, seed=3)
    g = torch.Generator(device=DEV); g.manual_seed(5)
    x = torch.randint(0, 8, (n,), generator=g, device=DEV).double() * 0.125
    want = _definition_regular(m, k, ci, va, x)
    seen = {}
    for method in (M.Method_Parallel, M.Method_Balanced, M.Method_Balanced_Yid, M.Method_SellCSigma, M.Method_CSR5SPMV):
        y, sched = _run(m, n, rp, ci, va, x, method)
        assert not torch.isnan(y).any(), sched
        assert torch.equal(y, want), (sched, float((y - want).abs().max()))
        seen[sched] = True
    assert {"csr-vector", "row-block", "nnz-split", "sell-c-sigma", "csr5"} <= set(seen)


def test_config2_random_columns_and_linearity():
    m = n = 10_000_000
    k = 32
    _, _, rp, ci, va = synth.uniform_k_device(m, n, k, "uniform", torch.float64, DEV, seed=7)
    g = torch.Generator(device=DEV); g.manual_seed(11)
    x = torch.rand(n, generator=g, device=DEV, dtype=torch.float64) * 2 - 1
    z = torch.rand(n, generator=g, device=DEV, dtype=torch.float64) * 2 - 1
    a, b = 0.75, -1.5
    with api.Handle(m, n, rp, ci, va, M.Method_Parallel) as h:
        yx = torch.empty(m, dtype=torch.float64, device=DEV); h.spmv(x, yx)
        yz = torch.empty(m, dtype=torch.float64, device=DEV); h.spmv(z, yz)
        yc = torch.empty(m, dtype=torch.float64, device=DEV); h.spmv(a * x + b * z, yc)
        ones = torch.ones(n, dtype=torch.float64, device=DEV)
        y1 = torch.empty(m, dtype=torch.float64, device=DEV); h.spmv(ones, y1)
    torch.cuda.synchronize()
    scale = k * 2.5  # |a||A||x| + |b||A||z| <= 32 * (0.75 + 1.5)
    assert float((yc - (a * yx + b * yz)).abs().max()) <= 1e-6 * scale     # north_star fp64 tolerance
    assert float((yc - (a * yx + b * yz)).abs().max()) <= 64 * 2.3e-16 * scale
    # checksum of checksums: sum_i (A 1)_i = sum of all values
    assert abs(float(y1.sum()) - float(va.sum())) <= 1e-9 * float(va.abs().sum())
    want = _definition_regular(m, k, ci, va, x)
    assert float((yx - want).abs().max()) <= 64 * 2.3e-16 * k


def test_random_columns_every_method_runs_cache_blocked_and_auto_finds_it():
    """Config 2 with uniformly random columns: no x window can be staged, x (80 MB) dwarfs an L2 -> whatever the
    method (Method_Parallel is what BASELINE names for config 2), the multiply runs the row-block x column-slab
    executor (kernels/blocked.hpp), and so does the automatic choice.  Exact inputs -> exact bits; random inputs ->
    the north_star tolerance AND the same bits from two handles (one wavefront owns a row block)."""
    m = n = 10_000_000
    k = 32
    _, _, rp, ci, va = synth.uniform_k_device(m, n, k, "eighths", torch.float64, DEV, seed=13)
    g = torch.Generator(device=DEV); g.manual_seed(15)
    x = torch.randint(0, 8, (n,), generator=g, device=DEV).double() * 0.125
    want = _definition_regular(m, k, ci, va, x)
    for method in (M.Method_Parallel, M.Method_Balanced, M.Method_Balanced_Yid, M.Method_SellCSigma, M.Method_CSR5SPMV):
        y = torch.full((m,), float("nan"), dtype=torch.float64, device=DEV)
        with api.Handle(m, n, rp, ci, va, method) as h:
            h.spmv(x, y)
            info = h.info()
            assert h.method == method
        torch.cuda.synchronize()
        assert info["cache_blocked"] == 1 and info["kernel_name"] in ("blk_kernel", "blk_wide_kernel") and info["x_groups_staged"] == 0, info
        assert torch.equal(y, want), method
    api.set_option("auto_method", 1)
    try:
        y = torch.full((m,), float("nan"), dtype=torch.float64, device=DEV)
        with api.Handle(m, n, rp, ci, va, M.Method_Serial) as h:
            h.spmv(x, y)
            info = h.info()
            chosen = h.method
    finally:
        api.set_option("auto_method", 0)
    torch.cuda.synchronize()
    assert chosen == M.Method_Parallel and info["cache_blocked"] == 1
    assert torch.equal(y, want)
    # random values: tolerance, and bit-reproducible across handles
    _, _, rp, ci, va = synth.uniform_k_device(m, n, k, "uniform", torch.float64, DEV, seed=17)
    xr = torch.rand(n, generator=g, device=DEV, dtype=torch.float64) * 2 - 1
    ys = []
    for method in (M.Method_Balanced_Yid, M.Method_Parallel):
        y = torch.empty(m, dtype=torch.float64, device=DEV)
        with api.Handle(m, n, rp, ci, va, method) as h:
            h.spmv(xr, y)
        torch.cuda.synchronize()
        ys.append(y)
    assert float((ys[0] - _definition_regular(m, k, ci, va, xr)).abs().max()) <= 64 * 2.3e-16 * k
    assert torch.equal(ys[0], ys[1])
    # option 0 switches it off
    api.set_option("cache_block", 0)
    try:
        with api.Handle(m, n, rp, ci, va, M.Method_Balanced_Yid) as h:
            assert h.info()["cache_blocked"] == 0
    finally:
        api.set_option("cache_block", 1)


def _exact_x(n, seed):
    g = torch.Generator(device=DEV); g.manual_seed(seed)
    return torch.randint(0, 8, (n,), generator=g, device=DEV).double() * 0.125


@pytest.mark.parametrize("cols", ["uniform", "rmat"])
def test_config3_webbase_style_powerlaw_at_full_size(cols):
    """BASELINE config 3, stand-in for webbase-1M (1e6 rows, mean 3.1, longest row 4.7k; SuiteSparse files
    are not available offline): Method_Balanced2 (the named schedule), Method_Balanced_Yid and CSR5 at full size,
    exact-arithmetic fill, compared bit for bit with a torch fp64 evaluation of the definition.  Columns uniform and
    R-MAT (hub columns + community structure)."""
    m = n = 1_000_000
    lens = synth.powerlaw_lengths_device(m, 3.1, 4700, 1.6, DEV, 1)
    _, _, rp, ci, va = synth.from_row_lengths_device(lens, n, "eighths", torch.float64, DEV, 1, cols=cols)
    assert int((rp[1:] - rp[:-1]).max()) >= 4000 and 2_000_000 < int(rp[-1]) < 4_000_000
    x = _exact_x(n, 21)
    want = _definition_segments(rp, ci, va, x)
    for method in (M.Method_Balanced2, M.Method_Balanced_Yid, M.Method_CSR5SPMV, M.Method_Parallel):
        y, sched = _run(m, n, rp, ci, va, x, method)
        assert not torch.isnan(y).any(), (method, sched)
        assert torch.equal(y, want), (method, sched, float((y - want).abs().max()))


def test_config3_orkut_style_powerlaw_at_full_size():
    """BASELINE config 3, stand-in for com-Orkut (3.07e6 rows, ~2.3e8 nnz, mean ~76, longest row 33k), R-MAT columns:
    the named schedule Method_Balanced2 and CSR5 at full size against the definition, exact arithmetic."""
    m = n = 3_070_000
    lens = synth.powerlaw_lengths_device(m, 76, 33000, 1.5, DEV, 1)
    _, _, rp, ci, va = synth.from_row_lengths_device(lens, n, "eighths", torch.float64, DEV, 1, cols="rmat")
    assert int(rp[-1]) > 200_000_000 and int((rp[1:] - rp[:-1]).max()) >= 30000
    x = _exact_x(n, 22)
    want = _definition_segments(rp, ci, va, x)
    for method in (M.Method_Balanced2, M.Method_CSR5SPMV):
        y, sched = _run(m, n, rp, ci, va, x, method)
        assert torch.equal(y, want), (method, sched, float((y - want).abs().max()))
        del y
    # uniform columns (no structure at all): the cache-blocked executor
    _, _, rp, ci, va = synth.from_row_lengths_device(lens, n, "eighths", torch.float64, DEV, 2)
    want = _definition_segments(rp, ci, va, x)
    y = torch.full((m,), float("nan"), dtype=torch.float64, device=DEV)
    with api.Handle(m, n, rp, ci, va, M.Method_Balanced2) as h:
        h.spmv(x, y)
        assert h.info()["cache_blocked"] == 1
    torch.cuda.synchronize()
    assert torch.equal(y, want)


def test_config3_through_the_matrix_market_loader(tmp_path, monkeypatch):
    """.mtx file -> spmv_io_load (csrc/io/mtx_io.c; reference mmio_highlevel.h:325-491: text parse, then the
    mtx_cache/*.bin cache on the second load) -> create(Method_Balanced2) -> spmv, on the 1e6-row power-law
    stand-in with R-MAT columns.  The loader must hand back exactly the CSR that was written, and the product
    must equal the definition."""
    import ctypes as C
    m = n = 1_000_000
    lens = synth.powerlaw_lengths_device(m, 3.1, 4700, 1.6, DEV, 3)
    _, _, rp, ci, va = synth.from_row_lengths_device(lens, n, "eighths", torch.float64, DEV, 3, cols="rmat")
    rp_h, ci_h, va_h = rp.cpu().numpy(), ci.cpu().numpy(), va.cpu().numpy()
    nnz = int(rp_h[-1])
    rows = np.repeat(np.arange(1, m + 1, dtype=np.int64), np.diff(rp_h))
    monkeypatch.chdir(tmp_path)
    with open("pl.mtx", "w") as f:
        f.write("%%%%MatrixMarket matrix coordinate real general\n%d %d %d\n" % (m, n, nnz))
        np.savetxt(f, np.column_stack([rows, ci_h.astype(np.int64) + 1, va_h]), fmt="%d %d %.6g")
    lib = api.load()
    I = C.POINTER(C.c_int)
    for expect_cache in (0, 1):
        mm, nn, nz, sym, fc = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int()
        prp, pci, pva = I(), I(), C.c_void_p()
        assert lib.spmv_io_load(b"pl.mtx", 8, C.byref(mm), C.byref(nn), C.byref(nz), C.byref(sym), C.byref(prp), C.byref(pci), C.byref(pva), C.byref(fc)) == 0
        assert (mm.value, nn.value, nz.value, fc.value) == (m, n, nnz, expect_cache)
        csr = api._take_csr(mm, nn, nz, prp, pci, pva, np.dtype(np.float64))
        assert np.array_equal(csr.rowptr, rp_h) and np.array_equal(csr.colidx, ci_h) and np.array_equal(csr.val, va_h)
    x = _exact_x(n, 23)
    want = _definition_segments(rp, ci, va, x).cpu().numpy()
    xh = x.cpu().numpy()
    for method in (M.Method_Balanced2, M.Method_Balanced_Yid):
        y = np.full(m, np.nan)
        with api.Handle(m, n, csr.rowptr, csr.colidx, csr.val, method) as h:      # host arrays, as the harness passes them
            h.spmv(xh, y)
        assert np.array_equal(y, want), method


def test_config4_skewed_fp32_schedules_bit_identical():
    m = n = 10_000_000
    lens = synth.skewed_lengths_device(m, DEV, seed=2)
    _, _, rp, ci, va = synth.from_row_lengths_device(lens, n, "eighths", torch.float32, DEV, seed=4, local=4096)
    g = torch.Generator(device=DEV); g.manual_seed(6)
    x = (torch.randint(0, 8, (n,), generator=g, device=DEV).float() * 0.125)
    want = _definition_segments(rp, ci, va, x).float()   # row sums < 2^12 in units of 1/64: exact in fp32
    info = None
    for method in (M.Method_SellCSigma, M.Method_CSR5SPMV, M.Method_Balanced2, M.Method_Parallel):
        y, sched = _run(m, n, rp, ci, va, x, method)
        assert torch.equal(y, want), (sched, float((y - want).abs().max()))
    with api.Handle(m, n, rp, ci, va, M.Method_SellCSigma) as h:
        info = h.info()
    assert info["sell_c"] == 64 and info["sell_sigma"] == 1024
    assert info["stored_nnz"] < 1.6 * info["nnz"]          # padding stays bounded on skewed rows


def test_config5_shard_shape_global_columns():
    """One rank's shard of the 8e7 x 8e7 matrix: 1e7 local rows, global columns, n = 8e7."""
    m, n, k = 10_000_000, 80_000_000, 32
    _, _, rp, ci, va = synth.banded_device(m, n, k, "eighths", torch.float64, DEV, seed=9, row0=3 * m)
    assert int(ci.min()) >= 3 * m - 16 and int(ci.max()) <= 4 * m + 16
    g = torch.Generator(device=DEV); g.manual_seed(8)
    x = torch.randint(0, 8, (n,), generator=g, device=DEV).double() * 0.125
    y, _ = _run(m, n, rp, ci, va, x, M.Method_Parallel)
    assert torch.equal(y, _definition_regular(m, k, ci, va, x))


def test_nnz_just_below_int32_limit():
    k = 32
    m = 67_108_000                     # nnz = 2 147 456 000 = 2^31 - 27 648
    n = m
    _, _, rp, ci, va = synth.banded_device(m, n, k, "eighths", torch.float32, DEV, seed=1)
    assert int(rp[-1].item()) == m * k < 2**31
    x = torch.ones(n, dtype=torch.float32, device=DEV)
    want = va.view(m, k).double().sum(1).float()
    for method in (M.Method_Parallel, M.Method_Balanced_Yid, M.Method_CSR5SPMV):
        y, sched = _run(m, n, rp, ci, va, x, method)
        assert torch.equal(y, want), sched
        del y
        torch.cuda.empty_cache()


def _abs_segments(rp, ci, va, x):
    """sum of |a x| per row: what the north_star tolerance scales with"""
    nnz = ci.numel()
    cs = torch.zeros(nnz + 1, dtype=torch.float64, device=va.device)
    step = 1 << 26
    carry = torch.zeros((), dtype=torch.float64, device=va.device)
    for p0 in range(0, nnz, step):
        p1 = min(nnz, p0 + step)
        prod = (va[p0:p1].double() * x[ci[p0:p1].long()].double()).abs()
        torch.cumsum(prod, 0, out=cs[p0 + 1:p1 + 1])
        cs[p0 + 1:p1 + 1] += carry
        carry = cs[p1].clone()
    r = rp.long()
    return cs[r[1:]] - cs[r[:-1]]


def _check_rounding(m, n, rp, ci, va, methods, tol):
    g = torch.Generator(device=DEV); g.manual_seed(99)
    x = torch.rand(n, generator=g, device=DEV, dtype=va.dtype) * 2 - 1
    want = _definition_segments(rp, ci, va, x)
    scale = _abs_segments(rp, ci, va, x)
    # the fp64 prefix sums themselves carry ~1e-16 x |prefix| of error: far below tol x scale for every non-empty row
    for method in methods:
        y, sched = _run(m, n, rp, ci, va, x, method)
        assert not bool(torch.isnan(y).any()), (method, sched)
        err = (y.double() - want).abs()
        bad = err > tol * scale + 1e-9 * (scale == 0)
        assert not bool(bad.any()), (method.name, sched, int(bad.sum()), float((err / scale.clamp(min=1e-300)).max()))


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-6), (torch.float32, 1e-3)])
@pytest.mark.parametrize("cols", ["rmat", "uniform"])
def test_rounding_at_size_orkut_style_named_schedule(cols, dtype, tol):
    """Config 3's large stand-in at full size on INEXACT data (uniform(-1, 1) values and x): the named schedule
    (Method_Balanced2 -> row blocks x column slabs: products in the value type, double accumulators in LDS, a summation order
    of its own) and CSR5 must stay within the north_star tolerance of the fp64 definition, scaled by the row's sum of |a x|.
    The exact-arithmetic tests above prove the indexing at this size; this proves the rounding."""
    m = 3_070_000
    lens = synth.powerlaw_lengths_device(m, 76, 33000, 1.5, DEV, 1)
    _, _, rp, ci, va = synth.from_row_lengths_device(lens, m, "uniform", dtype, DEV, 1, cols=cols)
    _check_rounding(m, m, rp, ci, va, [M.Method_Balanced2, M.Method_CSR5SPMV], tol)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.float64, 1e-6)])
def test_rounding_at_size_config4_named_schedule(dtype, tol):
    """Config 4 at full size (1e7 skewed rows, 5.4e8 non-zeros) on inexact data: SELL-C-sigma (slabs + long-row CSR5 sub-matrix
    + carry fix-up), CSR5 and nnz-split within the north_star tolerance of the fp64 definition."""
    m = 10_000_000
    lens = synth.skewed_lengths_device(m, DEV, 1)
    _, _, rp, ci, va = synth.from_row_lengths_device(lens, m, "uniform", dtype, DEV, 1, local=4096)
    _check_rounding(m, m, rp, ci, va, [M.Method_SellCSigma, M.Method_CSR5SPMV, M.Method_Balanced_Yid], tol)


def test_round4_column_stream_forms_at_full_size():
    """BASELINE config 2's partners of round 4 at full size, exact arithmetic, against the torch fp64 definition: the band with 25 % holes (BYTE
    tiles: a byte of column stream per entry) and the 27-point stencil 215^3 (TEMPLATE tiles: none), under the schedule BASELINE names for config 2
    and one other; plus linearity of the holes matrix on inexact data (size-independent property)."""
    m, n, rp, ci, va = synth.banded_holes_device(10_000_000, 10_000_000, 32, 0.25, "eighths", torch.float64, DEV, 3)
    x = _exact_x(n, 31)
    want = _definition_regular(m, 32, ci, va, x)
    for method in (M.Method_Parallel, M.Method_Balanced):
        y = torch.full((m,), float("nan"), dtype=torch.float64, device=DEV)
        with api.Handle(m, n, rp, ci, va, method) as h:
            h.spmv(x, y)
            info = h.info()
        torch.cuda.synchronize()
        assert torch.equal(y, want), method
        assert info["byte_nnz"] >= 0.999 * info["nnz"] and info["run_nnz"] == 0, info
    g = torch.Generator(device=DEV); g.manual_seed(5)
    va2 = torch.rand(m * 32, generator=g, device=DEV, dtype=torch.float64) * 2 - 1
    xa, xb = (torch.rand(n, generator=g, device=DEV, dtype=torch.float64) * 2 - 1 for _ in range(2))
    with api.Handle(m, n, rp, ci, va2, M.Method_Parallel) as h:
        ya, yb, yc = (torch.empty(m, dtype=torch.float64, device=DEV) for _ in range(3))
        h.spmv(xa, ya); h.spmv(xb, yb); h.spmv((0.5 * xa - 2.0 * xb).contiguous(), yc)
    torch.cuda.synchronize()
    assert float((yc - (0.5 * ya - 2.0 * yb)).abs().max()) <= 64 * 2.3e-16 * 32 * 3
    del rp, ci, va, va2, want
    torch.cuda.empty_cache()
    m, n, rp, ci, va = synth.stencil27_device(215, "eighths", torch.float64, DEV, 4)
    x = _exact_x(n, 32)
    want = _definition_regular(m, 27, ci, va, x)
    for method in (M.Method_Parallel, M.Method_SellCSigma):
        y = torch.full((m,), float("nan"), dtype=torch.float64, device=DEV)
        with api.Handle(m, n, rp, ci, va, method) as h:
            h.spmv(x, y)
            info = h.info()
        torch.cuda.synchronize()
        assert torch.equal(y, want), method
        if method == M.Method_Parallel:
            assert info["tmpl_nnz"] >= 0.97 * info["nnz"], info      # every tile but those where the periodic grid wraps


def test_round4_wide_blocked_forms_at_full_size():
    """Config 2-ii at full size under the wide forms of the row-block x column-slab executor (one ~19.5 k-row block per CU): exact inputs -> the bits of
    the definition for two ordered waves (what create() picks by itself), four ordered waves and four waves in arrival order; inexact inputs -> the
    ordered forms agree bit for bit with each other and from handle to handle, the arrival-order form within the north_star tolerance."""
    m = n = 10_000_000
    k = 32
    _, _, rp, ci, va = synth.uniform_k_device(m, n, k, "eighths", torch.float64, DEV, seed=23)
    x = _exact_x(n, 33)
    want = _definition_regular(m, k, ci, va, x)
    keep = {key: api.get_option(key) for key in ("blk_waves", "deterministic")}
    try:
        for waves, det in ((0, 1), (4, 1), (4, 0)):
            api.set_option("blk_waves", waves); api.set_option("deterministic", det)
            y = torch.full((m,), float("nan"), dtype=torch.float64, device=DEV)
            with api.Handle(m, n, rp, ci, va, M.Method_Parallel) as h:
                h.spmv(x, y)
                info = h.info()
            torch.cuda.synchronize()
            assert info["cache_blocked"] == 1 and info["reproducible"] == det, info
            if waves:
                assert info["kernel_name"] == "blk_wide_kernel" and info["blk_waves"] == waves, info
            assert torch.equal(y, want), (waves, det)
        _, _, rp, ci, va = synth.uniform_k_device(m, n, k, "uniform", torch.float64, DEV, seed=29)
        g = torch.Generator(device=DEV); g.manual_seed(6)
        xr = torch.rand(n, generator=g, device=DEV, dtype=torch.float64) * 2 - 1
        ref = _definition_regular(m, k, ci, va, xr)
        ys = {}
        for waves, det in ((2, 1), (2, 1), (4, 1), (4, 0)):
            api.set_option("blk_waves", waves); api.set_option("deterministic", det)
            y = torch.empty(m, dtype=torch.float64, device=DEV)
            with api.Handle(m, n, rp, ci, va, M.Method_Parallel) as h:
                h.spmv(xr, y)
            torch.cuda.synchronize()
            assert float((y - ref).abs().max()) <= 64 * 2.3e-16 * k
            ys.setdefault((waves, det), []).append(y)
        assert torch.equal(ys[(2, 1)][0], ys[(2, 1)][1]) and torch.equal(ys[(2, 1)][0], ys[(4, 1)][0])
    finally:
        for key, v in keep.items():
            api.set_option(key, v)
