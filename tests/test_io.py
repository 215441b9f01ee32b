"""CPU: Matrix Market loader + binary CSR cache (include/spmv_io.h) against
scipy's reader and against the documented layout of the reference's cache
(src/samples/mmio_highlevel.h:325-584)."""
import os
import struct

import numpy as np
import pytest
import scipy.io
import scipy.sparse as sp

from spmv_amd import api, build, synth


@pytest.fixture(scope="module", autouse=True)
def _lib():
    build.build()
    api.load()


def _dense(csr):
    a = np.zeros((csr.m, csr.n))
    rows = np.repeat(np.arange(csr.m), np.diff(csr.rowptr))
    np.add.at(a, (rows, csr.colidx), csr.val)
    return a


def _write(path, header, size, lines):
    with open(path, "w") as f:
        f.write(header + "\n% a comment line\n%\n" + size + "\n" + "\n".join(lines) + "\n")


def test_general_real_matches_scipy_and_keeps_file_order(tmp_path):
    rng = np.random.default_rng(1)
    m, n, k = 40, 55, 300
    r, c, v = rng.integers(1, m + 1, k), rng.integers(1, n + 1, k), rng.uniform(-1, 1, k)
    p = str(tmp_path / "g.mtx")
    _write(p, "%%MatrixMarket matrix coordinate real general", f"{m} {n} {k}", [f"{a} {b} {x:.17g}" for a, b, x in zip(r, c, v)])
    csr, sym = api.read_mtx(p)
    assert not sym and (csr.m, csr.n, csr.nnz) == (m, n, k)            # duplicates kept
    assert np.allclose(_dense(csr), scipy.io.mmread(p).toarray())      # scipy sums duplicates too
    for row in range(m):                                               # file order inside a row
        want = [(b - 1, x) for a, b, x in zip(r, c, v) if a - 1 == row]
        got = list(zip(csr.colidx[csr.rowptr[row]:csr.rowptr[row + 1]], csr.val[csr.rowptr[row]:csr.rowptr[row + 1]]))
        assert [w[0] for w in want] == [g[0] for g in got] and np.allclose([w[1] for w in want], [g[1] for g in got])


@pytest.mark.parametrize("sym", ["symmetric", "hermitian"])
def test_symmetric_expansion(tmp_path, sym):
    rng = np.random.default_rng(2)
    n = 30
    ents = {(int(a), int(b)) for a, b in zip(rng.integers(1, n + 1, 120), rng.integers(1, n + 1, 120)) if a >= b}
    ents = sorted(ents)
    vals = rng.uniform(-1, 1, len(ents))
    field = "complex" if sym == "hermitian" else "real"
    lines = [f"{a} {b} {x:.17g}" + (" 0.25" if field == "complex" else "") for (a, b), x in zip(ents, vals)]
    p = str(tmp_path / "s.mtx")
    _write(p, f"%%MatrixMarket matrix coordinate {field} {sym}", f"{n} {n} {len(ents)}", lines)
    csr, is_sym = api.read_mtx(p)
    assert is_sym
    offdiag = sum(1 for a, b in ents if a != b)
    assert csr.nnz == len(ents) + offdiag                               # mmio_highlevel.h:420-427
    want = np.zeros((n, n))
    for (a, b), x in zip(ents, vals):
        want[a - 1, b - 1] = x
        want[b - 1, a - 1] = x                                          # value copied unchanged (real part only)
    assert np.allclose(_dense(csr), want)


def test_pattern_integer_skew_and_float(tmp_path):
    p = str(tmp_path / "p.mtx")
    _write(p, "%%MatrixMarket matrix coordinate pattern general", "3 4 4", ["1 1", "3 4", "2 2", "3 1"])
    csr, _ = api.read_mtx(p)
    assert np.array_equal(csr.rowptr, [0, 1, 2, 4]) and np.array_equal(csr.colidx, [0, 1, 3, 0]) and (csr.val == 1).all()
    _write(p, "%%MatrixMarket matrix coordinate integer general", "2 2 2", ["1 2 7", "2 1 -3"])
    csr, _ = api.read_mtx(p, np.float32)
    assert csr.val.dtype == np.float32 and np.array_equal(csr.val, [7, -3])
    _write(p, "%%MatrixMarket matrix coordinate real skew-symmetric", "3 3 1", ["3 1 2.5"])
    csr, sym = api.read_mtx(p)
    assert not sym and csr.nnz == 1                                      # read as stored, like the reference


def test_errors(tmp_path):
    with pytest.raises(OSError, match="-1"):
        api.read_mtx(str(tmp_path / "missing.mtx"))
    p = str(tmp_path / "bad.mtx")
    _write(p, "%%MatrixMarket matrix array real general", "2 2", ["1", "2", "3", "4"])
    with pytest.raises(OSError, match="-2"):
        api.read_mtx(p)
    _write(p, "%%MatrixMarket matrix coordinate real general", "2 2 1", ["3 1 1.0"])
    with pytest.raises(OSError, match="-5"):
        api.read_mtx(p)


def test_bin_cache_layout_and_roundtrip(tmp_path):
    csr = synth.powerlaw(200, 300, 5.0, 80, 1.5, "uniform", np.float64, seed=3)
    p = str(tmp_path / "c.bin")
    api.write_bin(p, csr)
    raw = open(p, "rb").read()
    m, n, nnz = struct.unpack("<3i", raw[:12])                           # mmio_highlevel.h:546-551
    assert (m, n, nnz) == (csr.m, csr.n, csr.nnz)
    off = 12
    assert np.array_equal(np.frombuffer(raw, np.int32, m + 1, off), csr.rowptr); off += 4 * (m + 1)
    assert np.array_equal(np.frombuffer(raw, np.int32, nnz, off), csr.colidx); off += 4 * nnz
    assert np.array_equal(np.frombuffer(raw, np.float64, nnz, off), csr.val)
    assert len(raw) == off + 8 * nnz
    back = api.read_bin(p)
    assert np.array_equal(back.rowptr, csr.rowptr) and np.array_equal(back.colidx, csr.colidx) and np.array_equal(back.val, csr.val)
    assert api.cache_path("data/my dir\\a.mtx") == "mtx_cache/data_my_dir_a.mtx.bin"   # mmio_highlevel.h:533-541


def test_cache_written_in_one_value_width_is_not_misread_in_the_other(tmp_path, monkeypatch):
    """The cache format has no dtype field (mmio_highlevel.h:531-584): reading an fp64 cache as fp32 must be
    refused (file longer than nnz floats), and the other way round (short file), so that spmv_io_load re-parses the
    .mtx file instead of returning halves of doubles as floats."""
    import ctypes as C
    csr = synth.powerlaw(120, 120, 4.0, 40, 1.5, "uniform", np.float64, seed=8)
    p64, p32 = str(tmp_path / "d.bin"), str(tmp_path / "s.bin")
    api.write_bin(p64, csr)
    with pytest.raises(OSError):
        api.read_bin(p64, np.float32)
    from spmv_amd.synth import CSR
    api.write_bin(p32, CSR(csr.m, csr.n, csr.rowptr, csr.colidx, csr.val.astype(np.float32)))
    with pytest.raises(OSError):
        api.read_bin(p32, np.float64)
    assert np.array_equal(api.read_bin(p32, np.float32).val, csr.val.astype(np.float32))
    # spmv_io_load: load as f64 (writes the cache), then as f32 -> falls back to the .mtx text and gets real floats
    monkeypatch.chdir(tmp_path)
    a = sp.coo_matrix((csr.val, (np.repeat(np.arange(csr.m), np.diff(csr.rowptr)), csr.colidx)), shape=(csr.m, csr.n))
    scipy.io.mmwrite("w.mtx", a, precision=17)
    lib = api.load()
    I = C.POINTER(C.c_int)
    def load(vsize):
        m, n, nnz, sym, fc = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int()
        rp, ci, va = I(), I(), C.c_void_p()
        assert lib.spmv_io_load(b"w.mtx", vsize, C.byref(m), C.byref(n), C.byref(nnz), C.byref(sym), C.byref(rp), C.byref(ci), C.byref(va), C.byref(fc)) == 0
        dt = np.float64 if vsize == 8 else np.float32
        return api._take_csr(m, n, nnz, rp, ci, va, np.dtype(dt)), fc.value
    d64, from_cache = load(8)
    assert from_cache == 0 and os.path.exists(api.cache_path("w.mtx"))
    d32, from_cache = load(4)
    assert from_cache == 0, "an fp64 cache must not satisfy an fp32 load"
    assert np.array_equal(d32.val, d64.val.astype(np.float32))
    assert load(4)[1] == 1                          # ... and the cache now holds floats


def test_spmv_of_loaded_matrix_matches_scipy(tmp_path):
    """End to end on the CPU side: loader -> oracle SpMV == scipy's A @ x."""
    import oracle
    a = sp.random(150, 170, density=0.05, random_state=4, format="coo")
    p = str(tmp_path / "r.mtx")
    scipy.io.mmwrite(p, a)
    csr, _ = api.read_mtx(p)
    x = np.random.default_rng(5).uniform(-1, 1, 170)
    assert np.allclose(oracle.spmv_serial(csr, x), a.tocsr() @ x, rtol=1e-13, atol=1e-14)
