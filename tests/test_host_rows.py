"""CPU: BASELINE config 1 through the PRODUCT -- "Method_Serial fp64, VECTOR_NONE, 100k x 100k banded CSR
~16 nnz/row on CPU (reference plumbing, no GPU)".  With option "host_rows" a handle created with VECTOR_NONE and
Method_Serial / Method_Parallel runs the library's own plain-C row loop (spmv_amd/csrc/host_rows.c; reference
serial_spmv.c:9-55, parallel_spmv.c:5-51) on the caller's arrays.  Checked against the reference-held golden
outputs and the oracle; and the switch is explicit: without it VECTOR_NONE still needs a GPU and fails loudly."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle
from conftest import GOLDEN, load_golden
from spmv_amd import api, build, synth

with open(os.path.join(GOLDEN, "manifest.json")) as _f:
    NAMES = sorted(json.load(_f)["cases"].keys())

M, V = api.SPMV_METHODS, api.VECTORIZED_WAY


@pytest.fixture(scope="module", autouse=True)
def lib():
    build.build()
    L = api.load()
    L.spmv_hip_clear_error()
    return L


@pytest.fixture()
def host_rows():
    api.set_thread_option("host_rows", 1)
    yield
    api.clear_thread_options()


def _run(csr, x, method, nthreads=1):
    y = np.full(csr.m, np.nan, dtype=csr.val.dtype)
    h = api.Handle(csr.m, csr.n, csr.rowptr, csr.colidx, csr.val, method, nthreads=nthreads, way=V.VECTOR_NONE)
    try:
        assert h.option("host_rows") == 1
        info = h.info()
        assert info["schedule_name"] == "host-rows" and info["device"] == -1 and info["nnz"] == csr.nnz
        h.spmv(x, y)
        assert h.method == method
    finally:
        h.close()
    return y


@pytest.mark.parametrize("name", NAMES)
@pytest.mark.parametrize("method,threads", [(M.Method_Serial, 1), (M.Method_Parallel, 3)])
def test_golden_cases_on_the_host_loop(host_rows, name, method, threads):
    csr, x, y_ref = load_golden(name)
    y = _run(csr, x, method, threads)
    assert not np.isnan(y).any(), "every row is written, empty rows get 0 (serial_spmv.c:16-21)"
    if name.endswith("eighths"):
        assert np.array_equal(y.view(np.uint8), y_ref.view(np.uint8))   # exact arithmetic: bit for bit the reference's output
    else:
        tol = 1e-6 if csr.val.dtype == np.float64 else 1e-3             # north_star; another summation order than the AVX2 dot
        err = np.abs(y.astype(np.float64) - y_ref.astype(np.float64))
        assert (err <= tol * oracle.row_abs_sum(csr, x) + 1e-300).all()


def test_config1_runs_through_the_product(host_rows):
    """100k x 100k banded, ~16 nnz/row, fp64, Method_Serial, VECTOR_NONE: create -> spmv -> destroy."""
    csr = synth.banded(100_000, 100_000, 8, 7, "uniform", np.float64, seed=1)
    x = synth.fill_x(csr.n, "uniform", np.float64, 2)
    assert abs(csr.nnz / csr.m - 16) < 0.01
    y = _run(csr, x, M.Method_Serial)
    want = oracle.spmv_serial(csr, x)
    err = np.abs(y - want)
    assert (err <= 1e-12 * np.maximum(oracle.row_abs_sum(csr, x), 1e-300)).all()
    y4 = _run(csr, x, M.Method_Parallel, nthreads=4)
    assert np.array_equal(y4.view(np.uint8), y.view(np.uint8))          # rows are independent: the team size changes nothing
    exact = synth.banded(100_000, 100_000, 8, 7, "eighths", np.float64, seed=3)
    xe = synth.fill_x(exact.n, "eighths", np.float64, 4)
    assert np.array_equal(_run(exact, xe, M.Method_Serial), oracle.spmv_serial(exact, xe))


def test_values_changed_in_place_are_seen_like_in_the_reference(host_rows):
    """The host loop multiplies the arrays of THIS call (common.c:286-298): in-place updates need no refresh."""
    csr = synth.banded(500, 500, 4, 3, "uniform", np.float64, seed=5)
    x = synth.fill_x(csr.n, "uniform", np.float64, 6)
    h = api.Handle(csr.m, csr.n, csr.rowptr, csr.colidx, csr.val, M.Method_Serial, way=V.VECTOR_NONE)
    y0 = h.spmv(x, np.empty(csr.m))
    csr.val *= 2.0
    y1 = h.spmv(x, np.empty(csr.m))
    h.close()
    assert np.array_equal(y1, 2.0 * y0)


def test_the_switch_is_explicit(lib, monkeypatch):
    """Without option "host_rows", VECTOR_NONE runs the HIP schedules like every other value -- no device, loud failure;
    with it, only Method_Serial / Method_Parallel take the host loop."""
    if lib.spmv_hip_device_count() > 0:
        pytest.skip("a GPU is visible")
    monkeypatch.setenv("SPMV_HIP_QUIET", "1")
    csr = synth.banded(64, 64)
    assert api.get_option("host_rows") == 0
    with pytest.raises(api.SpmvError, match="no HIP device"):
        api.Handle(csr.m, csr.n, csr.rowptr, csr.colidx, csr.val, M.Method_Serial, way=V.VECTOR_NONE)
    api.set_thread_option("host_rows", 1)
    try:
        with pytest.raises(api.SpmvError, match="no HIP device"):
            api.Handle(csr.m, csr.n, csr.rowptr, csr.colidx, csr.val, M.Method_CSR5SPMV, way=V.VECTOR_NONE)
        with pytest.raises(api.SpmvError, match="no HIP device"):
            api.Handle(csr.m, csr.n, csr.rowptr, csr.colidx, csr.val, M.Method_Serial, way=V.VECTOR_HIP)
    finally:
        api.clear_thread_options()
    lib.spmv_hip_clear_error()


def test_published_dot_product_tables(lib):
    """Dot_{d,s}_Products[VECTORIZED_WAY] (reference spmv_Defines.h:84-91) are exported and compute the row dot."""
    dd = C.CFUNCTYPE(C.c_double, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double))
    ds = C.CFUNCTYPE(C.c_float, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_float))
    td = (C.c_void_p * 4).in_dll(lib, "Dot_d_Products")
    ts = (C.c_void_p * 4).in_dll(lib, "Dot_s_Products")
    idx = np.array([4, 0, 2, 7, 1, 3, 6], dtype=np.int32)
    for way in range(4):
        for tab, proto, dt in ((td, dd, np.float64), (ts, ds, np.float32)):
            val = (np.arange(1, 8) / 8).astype(dt)
            x = (np.arange(8, 0, -1) / 4).astype(dt)
            f = proto(tab[way])
            ct = C.c_double if dt == np.float64 else C.c_float
            got = f(7, idx.ctypes.data_as(C.POINTER(C.c_int)), val.ctypes.data_as(C.POINTER(ct)), x.ctypes.data_as(C.POINTER(ct)))
            assert got == float((val.astype(np.float64) * x[idx].astype(np.float64)).sum())
