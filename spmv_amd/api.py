"""Python host side above the C ABI of libspmv_hip.so (include/spmv.h, include/spmv_hip.h).

It mirrors the reference's operator interface one to one -- same function names, same argument
order and meaning, same void returns (reference: include/spmv.h:19-71, common.c:123-190,
278-304) -- so the parity tests read like the reference's own harness (test_spmv.c:62-156).
Arrays may be numpy arrays (host) or torch tensors (host or cuda); only their raw pointers cross
the boundary.  torch is used for device memory, streams and torch.distributed only.

There is no fallback: if libspmv_hip.so is missing or cannot be loaded `load()` raises, and without
a GPU every call reports SPMV_HIP_E_NODEVICE.  (The library's one host loop -- VECTOR_NONE with option
"host_rows", BASELINE config 1 -- is a configuration the caller switches on, csrc/host_rows.c.)
"""
from __future__ import annotations

import ctypes as C
import enum
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "lib", "libspmv_hip.so")


class VECTORIZED_WAY(enum.IntEnum):  # include/spmv_Defines.h
    VECTOR_NONE = 0
    VECTOR_AVX2 = 1
    VECTOR_AVX512 = 2
    VECTOR_HIP = 3
    VECTOR_TOTAL_SIZE = 4


class SPMV_METHODS(enum.IntEnum):  # include/spmv_Defines.h
    Method_Serial = 0
    Method_Parallel = 1
    Method_Balanced = 2
    Method_Balanced2 = 3
    Method_Balanced_Yid = 4
    Method_SellCSigma = 5
    Method_CSR5SPMV = 6
    Method_Total_Size = 7
    Method_Numa = 8


_I = C.POINTER(C.c_int)
_V = C.c_void_p


class spmv_Handle(C.Structure):
    """struct spmv_Handle (include/spmv_Defines.h; field order is the reference's)."""
    _fields_ = [("spmvMethod", C.c_int), ("data_size", C.c_ulong), ("nthreads", C.c_ulong),
                ("vectorizedWay", C.c_int), ("Level_3_opt_used", C.c_int), ("RowPtr", _I),
                ("ColIdx", _I), ("index", _I), ("Matrix_Val", _V), ("Y_temp", _V),
                ("extraHandle", _V)]


spmv_Handle_t = C.POINTER(spmv_Handle)


class spmv_hip_info(C.Structure):
    _fields_ = [("device", C.c_int), ("schedule", C.c_int), ("lanes_per_row", C.c_int),
                ("sell_c", C.c_int), ("sell_sigma", C.c_int), ("tile_nnz", C.c_int),
                ("m", C.c_int), ("n", C.c_int), ("nnz", C.c_longlong), ("stored_nnz", C.c_longlong),
                ("max_row_len", C.c_int), ("min_row_len", C.c_int), ("empty_rows", C.c_int),
                ("mean_row_len", C.c_double), ("device_bytes", C.c_longlong),
                ("alg_bytes", C.c_longlong), ("inspect_ms", C.c_double),
                ("schedule_name", C.c_char_p), ("kernel_name", C.c_char_p),
                ("tuned_choice", C.c_int), ("tune_ms", C.c_float * 3),
                ("x_groups", C.c_int), ("x_groups_staged", C.c_int), ("cache_blocked", C.c_int),
                ("stream_bytes", C.c_longlong), ("x_bytes", C.c_longlong), ("route_ms", C.c_float * 2), ("split_ms", C.c_float * 2), ("far_nnz", C.c_longlong), ("run_nnz", C.c_longlong), ("byte_nnz", C.c_longlong), ("tmpl_nnz", C.c_longlong),
                ("blk_waves", C.c_int), ("launch_kernels", C.c_char * 160), ("reproducible", C.c_int)]


# Every symbol include/*.h declares: functions with their prototypes, then data symbols.
FUNCTIONS = {
    "spmv_create_handle_all_in_one": (None, [C.POINTER(spmv_Handle_t), C.c_int, C.c_int, _V, _V, _V,
                                             C.c_ulong, C.c_int, C.c_ulong, C.c_int, C.c_char_p]),
    "spmv": (None, [spmv_Handle_t, C.c_int, _V, _V, _V, _V, _V]),
    "spmv_destory_handle": (None, [spmv_Handle_t]),
    "spmv_clear_handle": (None, [spmv_Handle_t]),
    "spmv_hip_last_error": (C.c_int, []),
    "spmv_hip_last_error_string": (C.c_char_p, []),
    "spmv_hip_clear_error": (None, []),
    "spmv_hip_device_count": (C.c_int, []),
    "spmv_hip_trim_pool": (None, []),
    "spmv_hip_set_stream": (C.c_int, [spmv_Handle_t, _V]),
    "spmv_hip_set_async": (C.c_int, [spmv_Handle_t, C.c_int]),
    "spmv_hip_synchronize": (C.c_int, [spmv_Handle_t]),
    "spmv_hip_set_option": (C.c_int, [C.c_char_p, C.c_long]),
    "spmv_hip_get_option": (C.c_long, [C.c_char_p]),
    "spmv_hip_set_thread_option": (C.c_int, [C.c_char_p, C.c_long]),
    "spmv_hip_clear_thread_options": (None, []),
    "spmv_hip_get_handle_option": (C.c_long, [spmv_Handle_t, C.c_char_p]),
    "spmv_hip_update_values": (C.c_int, [spmv_Handle_t, _V]),
    "spmv_hip_multi_gpus": (C.c_int, [spmv_Handle_t]),
    "spmv_hip_multi_uses_rccl": (C.c_int, [spmv_Handle_t]),
    "spmv_hip_multi_slices": (C.c_int, [spmv_Handle_t, C.c_int, C.POINTER(_V), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong),
                                        C.POINTER(_V), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), _I]),
    "spmv_hip_multi_step": (C.c_int, [spmv_Handle_t]),
    "spmv_hip_multi_step_async": (C.c_int, [spmv_Handle_t]),
    "spmv_hip_multi_synchronize": (C.c_int, [spmv_Handle_t]),
    "spmv_hip_create_handle_from_blocks": (None, [C.POINTER(spmv_Handle_t), C.c_int, _I, C.c_int, C.POINTER(_V), C.POINTER(_V), C.POINTER(_V),
                                                  C.c_int, C.c_ulong]),
    "spmv_hip_get_info": (C.c_int, [spmv_Handle_t, C.POINTER(spmv_hip_info)]),
    "spmv_hip_time_launches": (C.c_double, [spmv_Handle_t, _V, _V, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    # include/spmv_io.h (host only)
    "spmv_io_read_mtx": (C.c_int, [C.c_char_p, C.c_size_t, _I, _I, _I, _I, C.POINTER(_I), C.POINTER(_I), C.POINTER(_V)]),
    "spmv_io_cache_path": (C.c_int, [C.c_char_p, C.c_char_p, C.c_size_t]),
    "spmv_io_write_bin": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int, _V, _V, _V, C.c_size_t]),
    "spmv_io_read_bin": (C.c_int, [C.c_char_p, C.c_size_t, _I, _I, _I, C.POINTER(_I), C.POINTER(_I), C.POINTER(_V)]),
    "spmv_io_load": (C.c_int, [C.c_char_p, C.c_size_t, _I, _I, _I, _I, C.POINTER(_I), C.POINTER(_I), C.POINTER(_V), _I]),
    "spmv_io_free": (None, [_V]),
}
DATA_SYMBOLS = ("Methods_names", "Vectorized_names", "funcNames", "Dot_s_Products", "Dot_d_Products")

_lib = None


def load():
    """dlopen libspmv_hip.so (building it is spmv_amd.build / __graft_entry__.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not found: run `python -m spmv_amd.build` "
                               "(there is no CPU fallback for the HIP library)")
        try:  # make sure one HIP runtime serves torch and this library (same SONAME)
            import torch  # noqa: F401
        except Exception:  # pragma: no cover - torch is optional for pure-numpy callers
            pass
        lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in FUNCTIONS.items():
            f = getattr(lib, name)
            f.restype, f.argtypes = res, args
        _lib = lib
    return _lib


def methods_names():
    lib = load()
    arr = (C.c_char_p * int(SPMV_METHODS.Method_Total_Size)).in_dll(lib, "Methods_names")
    return [s.decode() for s in arr]


def vectorized_names():
    lib = load()
    arr = (C.c_char_p * int(VECTORIZED_WAY.VECTOR_TOTAL_SIZE)).in_dll(lib, "Vectorized_names")
    return [s.decode() for s in arr]


class SpmvError(RuntimeError):
    pass


def last_error():
    lib = load()
    return lib.spmv_hip_last_error(), lib.spmv_hip_last_error_string().decode()


def _raise_if_error(where):
    code, text = last_error()
    if code:
        load().spmv_hip_clear_error()
        raise SpmvError(f"{where}: [{code}] {text}")


def _ptr(a):
    """Raw address of a numpy array / torch tensor / None / int."""
    if a is None:
        return None
    if isinstance(a, int):
        return a
    if isinstance(a, np.ndarray):
        if not a.flags["C_CONTIGUOUS"]:
            raise ValueError("array must be contiguous")
        return a.ctypes.data
    if hasattr(a, "data_ptr"):
        if not a.is_contiguous():
            raise ValueError("tensor must be contiguous")
        return a.data_ptr()
    raise TypeError(type(a))


def _itemsize(a):
    return a.dtype.itemsize if isinstance(a, np.ndarray) else a.element_size()


# ----------------------------------------------------------------------------- the four functions
def spmv_create_handle_all_in_one(m, n, RowPtr, ColIdx, Matrix_Val, nthreads, Function, size,
                                  vectorizedWay=VECTORIZED_WAY.VECTOR_HIP, MtxToken=None, check=True):
    """-> spmv_Handle_t.  Same arguments as the C function; the handle is the return value instead
    of an out-parameter.  The arrays must stay alive as long as spmv() is called with them."""
    lib = load()
    lib.spmv_hip_clear_error()
    h = spmv_Handle_t()
    tok = MtxToken.encode() if isinstance(MtxToken, str) else MtxToken
    lib.spmv_create_handle_all_in_one(C.byref(h), int(m), int(n), _ptr(RowPtr), _ptr(ColIdx), _ptr(Matrix_Val),
                                      int(nthreads), int(Function), int(size), int(vectorizedWay), tok)
    if check:
        _raise_if_error("spmv_create_handle_all_in_one")
    return h


def spmv(handle, m, RowPtr, ColIdx, Matrix_Val, Vector_Val_X, Vector_Val_Y, check=True):
    lib = load()
    lib.spmv(handle, int(m), _ptr(RowPtr), _ptr(ColIdx), _ptr(Matrix_Val), _ptr(Vector_Val_X), _ptr(Vector_Val_Y))
    if check:
        _raise_if_error("spmv")


def spmv_destory_handle(handle):
    load().spmv_destory_handle(handle)


def spmv_clear_handle(handle):
    load().spmv_clear_handle(handle)


# ----------------------------------------------------------------------------- extensions
def set_option(key, value):
    if load().spmv_hip_set_option(key.encode(), int(value)) != 0:
        load().spmv_hip_clear_error()
        raise ValueError(f"bad option {key}={value}")


def get_option(key):
    return load().spmv_hip_get_option(key.encode())


def set_thread_option(key, value):
    """Override for handles created by the calling thread (spmv_hip_set_thread_option)."""
    if load().spmv_hip_set_thread_option(key.encode(), int(value)) != 0:
        load().spmv_hip_clear_error()
        raise ValueError(f"bad option {key}={value}")


def clear_thread_options():
    load().spmv_hip_clear_thread_options()


def update_values(handle, Matrix_Val):
    """New values behind the same pattern (spmv_hip_update_values): no re-inspection."""
    if load().spmv_hip_update_values(handle, _ptr(Matrix_Val)) != 0:
        _raise_if_error("spmv_hip_update_values")


def get_info(handle):
    info = spmv_hip_info()
    if load().spmv_hip_get_info(handle, C.byref(info)) != 0:
        _raise_if_error("spmv_hip_get_info")
    out = {k: getattr(info, k) for k, _ in spmv_hip_info._fields_}
    out["schedule_name"] = (out["schedule_name"] or b"").decode()
    out["kernel_name"] = (out["kernel_name"] or b"").decode()
    out["launch_kernels"] = [k for k in (out["launch_kernels"] or b"").decode().split("+") if k]
    out["tune_ms"] = [float(v) for v in out["tune_ms"]]
    out["route_ms"] = [float(v) for v in out["route_ms"]]
    out["split_ms"] = [float(v) for v in out["split_ms"]]
    return out


def set_stream(handle, stream_ptr, async_=True):
    lib = load()
    lib.spmv_hip_set_stream(handle, stream_ptr)
    lib.spmv_hip_set_async(handle, 1 if async_ else 0)
    _raise_if_error("spmv_hip_set_stream")


def time_launches(handle, x, y, warmup=10, iters=100):
    """-> (mean_ms, per-launch ms array): hipEvents on the handle's stream around each launch."""
    ms = (C.c_float * iters)()
    mean = load().spmv_hip_time_launches(handle, _ptr(x), _ptr(y), warmup, iters, ms)
    if mean < 0:
        _raise_if_error("spmv_hip_time_launches")
    return mean, np.frombuffer(ms, dtype=np.float32).copy()


def _take_csr(m, n, nnz, rp, ci, va, dtype):
    """Copy malloc'ed C arrays into numpy arrays and free the C side."""
    from .synth import CSR
    lib = load()
    rowptr = np.ctypeslib.as_array(rp, shape=(m.value + 1,)).copy()
    colidx = np.ctypeslib.as_array(ci, shape=(max(nnz.value, 1),))[: nnz.value].copy()
    vt = C.c_double if dtype == np.float64 else C.c_float
    val = np.ctypeslib.as_array(C.cast(va, C.POINTER(vt)), shape=(max(nnz.value, 1),))[: nnz.value].copy()
    for p in (rp, ci, va):
        lib.spmv_io_free(C.cast(p, _V))
    return CSR(m.value, n.value, rowptr, colidx, val)


def read_mtx(path, dtype=np.float64):
    """Matrix Market coordinate file -> (CSR, is_symmetric)   [spmv_io_read_mtx]."""
    lib = load()
    m, n, nnz, sym = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    rp, ci, va = _I(), _I(), _V()
    rc = lib.spmv_io_read_mtx(os.fsencode(path), np.dtype(dtype).itemsize, C.byref(m), C.byref(n), C.byref(nnz), C.byref(sym),
                              C.byref(rp), C.byref(ci), C.byref(va))
    if rc != 0:
        raise OSError(f"spmv_io_read_mtx({path!r}) failed with {rc}")
    return _take_csr(m, n, nnz, rp, ci, va, np.dtype(dtype)), bool(sym.value)


def write_bin(path, csr):
    rc = load().spmv_io_write_bin(os.fsencode(path), csr.m, csr.n, csr.nnz, _ptr(np.ascontiguousarray(csr.rowptr, np.int32)),
                                  _ptr(np.ascontiguousarray(csr.colidx, np.int32)), _ptr(np.ascontiguousarray(csr.val)),
                                  csr.val.dtype.itemsize)
    if rc != 0:
        raise OSError(f"spmv_io_write_bin({path!r}) failed with {rc}")


def read_bin(path, dtype=np.float64):
    lib = load()
    m, n, nnz = C.c_int(), C.c_int(), C.c_int()
    rp, ci, va = _I(), _I(), _V()
    rc = lib.spmv_io_read_bin(os.fsencode(path), np.dtype(dtype).itemsize, C.byref(m), C.byref(n), C.byref(nnz),
                              C.byref(rp), C.byref(ci), C.byref(va))
    if rc != 0:
        raise OSError(f"spmv_io_read_bin({path!r}) failed with {rc}")
    return _take_csr(m, n, nnz, rp, ci, va, np.dtype(dtype))


def cache_path(mtx_path):
    buf = C.create_string_buffer(4096)
    if load().spmv_io_cache_path(os.fsencode(mtx_path), buf, len(buf)) != 0:
        raise ValueError("path too long")
    return buf.value.decode()


class Handle:
    """RAII convenience around the four functions (create in __init__, destroy in close())."""

    def __init__(self, m, n, rowptr, colidx, val, method=SPMV_METHODS.Method_Parallel, nthreads=1,
                 way=VECTORIZED_WAY.VECTOR_HIP, token=None):
        self.m, self.n = int(m), int(n)
        self._keep = (rowptr, colidx, val)  # the C side keeps the caller's pointers for identity checks
        self.h = spmv_create_handle_all_in_one(m, n, rowptr, colidx, val, nthreads, method,
                                               _itemsize(val), way, token)

    @property
    def method(self):
        return SPMV_METHODS(self.h.contents.spmvMethod)

    def info(self):
        return get_info(self.h)

    @property
    def index(self):
        """handle->index as a numpy array (the RCM permutation when option "reorder" is on), else None."""
        p = self.h.contents.index
        if not p:
            return None
        return np.ctypeslib.as_array(p, shape=(self.m,)).copy()

    def spmv(self, x, y):
        rp, ci, va = self._keep
        spmv(self.h, self.m, rp, ci, va, x, y)
        return y

    def update_values(self, val):
        """The caller changed the values (in place or in a new array of the same pattern)."""
        update_values(self.h, val)
        self._keep = (self._keep[0], self._keep[1], val)

    def option(self, key):
        return load().spmv_hip_get_handle_option(self.h, key.encode())

    # multi-GPU handles (option "gpus")
    def multi_gpus(self):
        return load().spmv_hip_multi_gpus(self.h)

    def multi_slices(self, gpu):
        """-> dict(x_ptr, x_first, x_count, y_ptr, y_first, y_count, device) of device `gpu`'s slice of x and block of y."""
        xs, ys = _V(), _V()
        xf, xc, yf, yc, dv = C.c_longlong(), C.c_longlong(), C.c_longlong(), C.c_longlong(), C.c_int()
        if load().spmv_hip_multi_slices(self.h, gpu, C.byref(xs), C.byref(xf), C.byref(xc), C.byref(ys), C.byref(yf), C.byref(yc), C.byref(dv)) != 0:
            _raise_if_error("spmv_hip_multi_slices")
        return dict(x_ptr=xs.value, x_first=xf.value, x_count=xc.value, y_ptr=ys.value, y_first=yf.value, y_count=yc.value, device=dv.value)

    def multi_step(self):
        if load().spmv_hip_multi_step(self.h) != 0:
            _raise_if_error("spmv_hip_multi_step")

    def multi_step_async(self):
        if load().spmv_hip_multi_step_async(self.h) != 0:
            _raise_if_error("spmv_hip_multi_step_async")

    def multi_synchronize(self):
        if load().spmv_hip_multi_synchronize(self.h) != 0:
            _raise_if_error("spmv_hip_multi_synchronize")

    @classmethod
    def from_blocks(cls, blocks, n, method=SPMV_METHODS.Method_Parallel):
        """Multi-GPU handle from separate row blocks [(rowptr, colidx, val), ...]: local int32 RowPtr, GLOBAL columns
        (spmv_hip_create_handle_from_blocks).  spmv() takes full-length x / y; the CSR arguments are ignored."""
        lib = load()
        lib.spmv_hip_clear_error()
        G = len(blocks)
        rows = (C.c_int * G)(*[int(b[0].shape[0]) - 1 for b in blocks])
        rps = (_V * G)(*[_ptr(b[0]) for b in blocks])
        cis = (_V * G)(*[_ptr(b[1]) for b in blocks])
        vas = (_V * G)(*[_ptr(b[2]) for b in blocks])
        self = cls.__new__(cls)
        self.m, self.n = int(sum(rows)), int(n)
        self._keep = (None, None, None)
        self._blocks = blocks
        self.h = spmv_Handle_t()
        lib.spmv_hip_create_handle_from_blocks(C.byref(self.h), G, rows, int(n), rps, cis, vas, int(method), _itemsize(blocks[0][2]))
        _raise_if_error("spmv_hip_create_handle_from_blocks")
        return self

    def attach_stream(self, stream_ptr, async_=True):
        set_stream(self.h, stream_ptr, async_)

    def close(self):
        if self.h:
            spmv_destory_handle(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
