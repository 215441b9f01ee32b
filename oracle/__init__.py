"""ctypes front end of the CHECKER (oracle/oracle_spmv.c and, when built, the real reference in
oracle/_ref/libmv_l2.so).

TEST INFRASTRUCTURE, NOT PRODUCT: imported only by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under spmv_amd/ imports this package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "liboracle_spmv.so")
REF_SO = os.path.join(HERE, "_ref", "libmv_l2.so")

_I = C.POINTER(C.c_int)
_V = C.c_void_p


HARNESS = os.path.join(HERE, "_ref", "test_spmv_hip")


def build(quiet=True):
    """make -C oracle  (restatement always; _ref and the reference harness only when
    /root/reference is present)."""
    out = subprocess.run(["make", "-C", HERE, "all", "harness"], capture_output=True, text=True)
    if out.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + out.stdout + out.stderr)
    if not quiet:
        print(out.stdout)


_oracle = None
_ref = None


def lib():
    global _oracle
    if _oracle is None:
        if not os.path.exists(ORACLE_SO):
            build()
        L = C.CDLL(ORACLE_SO)
        for name in ("oracle_spmv_serial", "oracle_spmv_omp"):
            f = getattr(L, name)
            f.argtypes = [C.c_int, _I, _I, _V, _V, _V, C.c_ulong]
            f.restype = None
        for name in ("oracle_spmv_exact", "oracle_row_abs_sum"):
            f = getattr(L, name)
            f.argtypes = [C.c_int, _I, _I, _V, _V, C.POINTER(C.c_double), C.c_ulong]
            f.restype = None
        _oracle = L
    return _oracle


def have_ref():
    return os.path.exists(REF_SO)


class RefHandle(C.Structure):
    """Mirror of the reference's struct spmv_Handle (include/spmv_Defines.h:44-68)."""
    _fields_ = [("spmvMethod", C.c_int), ("data_size", C.c_ulong), ("nthreads", C.c_ulong),
                ("vectorizedWay", C.c_int), ("Level_3_opt_used", C.c_int), ("RowPtr", _I),
                ("ColIdx", _I), ("index", _I), ("Matrix_Val", _V), ("Y_temp", _V),
                ("extraHandle", _V)]


def ref_lib():
    global _ref
    if _ref is None:
        if not have_ref():
            raise FileNotFoundError(REF_SO + " (run `make -C oracle` where /root/reference exists)")
        L = C.CDLL(REF_SO)
        H = C.POINTER(RefHandle)
        L.spmv_create_handle_all_in_one.argtypes = [C.POINTER(H), C.c_int, C.c_int, _I, _I, _V,
                                                    C.c_ulong, C.c_int, C.c_ulong, C.c_int, C.c_char_p]
        L.spmv_create_handle_all_in_one.restype = None
        L.spmv.argtypes = [H, C.c_int, _I, _I, _V, _V, _V]
        L.spmv.restype = None
        L.spmv_destory_handle.argtypes = [H]
        L.spmv_destory_handle.restype = None
        _ref = L
    return _ref


def _p(a, t=_V):
    return a.ctypes.data_as(t)


def _prep(csr, x):
    rp = np.ascontiguousarray(csr.rowptr, dtype=np.int32)
    ci = np.ascontiguousarray(csr.colidx, dtype=np.int32)
    va = np.ascontiguousarray(csr.val)
    assert va.dtype in (np.float64, np.float32)
    xx = np.ascontiguousarray(x, dtype=va.dtype)
    return rp, ci, va, xx


def spmv_serial(csr, x):
    """Restatement of Method_Serial (serial_spmv.c:9-55); returns y in the value dtype."""
    rp, ci, va, xx = _prep(csr, x)
    y = np.full(csr.m, np.nan, dtype=va.dtype)
    lib().oracle_spmv_serial(csr.m, _p(rp, _I), _p(ci, _I), _p(va), _p(xx), _p(y), va.dtype.itemsize)
    return y


def spmv_omp(csr, x):
    rp, ci, va, xx = _prep(csr, x)
    y = np.full(csr.m, np.nan, dtype=va.dtype)
    lib().oracle_spmv_omp(csr.m, _p(rp, _I), _p(ci, _I), _p(va), _p(xx), _p(y), va.dtype.itemsize)
    return y


def spmv_exact(csr, x):
    """Double-accumulated y (independent of the value dtype's rounding)."""
    rp, ci, va, xx = _prep(csr, x)
    y = np.empty(csr.m, dtype=np.float64)
    lib().oracle_spmv_exact(csr.m, _p(rp, _I), _p(ci, _I), _p(va), _p(xx),
                            y.ctypes.data_as(C.POINTER(C.c_double)), va.dtype.itemsize)
    return y


def row_abs_sum(csr, x):
    rp, ci, va, xx = _prep(csr, x)
    s = np.empty(csr.m, dtype=np.float64)
    lib().oracle_row_abs_sum(csr.m, _p(rp, _I), _p(ci, _I), _p(va), _p(xx),
                             s.ctypes.data_as(C.POINTER(C.c_double)), va.dtype.itemsize)
    return s


def ref_spmv(csr, x, method=0, nthreads=1, sentinel=None, repeat=1):
    """Run the REAL reference: create(method) -> spmv -> destroy.  Returns (y, actual_method).
    y is pre-filled with `sentinel` (default NaN) so unwritten rows show."""
    L = ref_lib()
    rp, ci, va, xx = _prep(csr, x)
    y = np.full(csr.m, np.nan if sentinel is None else sentinel, dtype=va.dtype)
    h = C.POINTER(RefHandle)()
    L.spmv_create_handle_all_in_one(C.byref(h), csr.m, csr.n, _p(rp, _I), _p(ci, _I), _p(va),
                                    nthreads, method, va.dtype.itemsize, 0, None)
    actual = h.contents.spmvMethod
    for _ in range(repeat):
        L.spmv(h, csr.m, _p(rp, _I), _p(ci, _I), _p(va), _p(xx), _p(y))
    L.spmv_destory_handle(h)
    return y, actual
