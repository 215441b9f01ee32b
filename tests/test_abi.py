"""CPU: the C-ABI library loads and exports every symbol include/*.h declares (no compute calls:
there is no GPU here), the enum/struct mirrors match the headers, and the host-side argument
rules of the reference hold (NULL handle no-op, loud failure without a device)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from spmv_amd import api, build, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build.build()
    return api.load()


def _declared_functions():
    names = set()
    for hdr in ("spmv.h", "spmv_hip.h", "spmv_io.h", "spmv_hip_tools.h"):
        text = open(os.path.join(ROOT, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(spmv\w*)\s*\(", text))
    return names


def test_every_declared_function_is_exported_and_bound(lib):
    declared = _declared_functions()
    assert {"spmv_create_handle_all_in_one", "spmv", "spmv_destory_handle", "spmv_clear_handle"} <= declared
    assert declared == set(api.FUNCTIONS), (declared ^ set(api.FUNCTIONS))
    for name in declared:
        assert getattr(lib, name) is not None


def test_data_symbols(lib):
    for sym in api.DATA_SYMBOLS:
        C.c_void_p.in_dll(lib, sym)
    # reference spelling (common.c:322-339) + the new VECTOR_HIP entry
    assert api.methods_names() == ["Method_Serial", "Method_Parallel", "Method_Balanced", "Method_Balanced2",
                                   "Method_BalancedYid", "Method_SellCSigma", "Method_Csr5Spmv"]
    assert api.vectorized_names() == ["VECTOR_NONE", "VECTOR_AVX2", "VECTOR_AVX512", "VECTOR_HIP"]
    fn = (C.c_char_p * 28).in_dll(lib, "funcNames")
    assert fn[0] == b"Method_Serial_VECTOR_NONE" and fn[27] == b"Method_Csr5Spmv_VECTOR_HIP"


def test_enum_values_match_header():
    text = open(os.path.join(ROOT, "include", "spmv_Defines.h")).read()
    for e in (api.SPMV_METHODS, api.VECTORIZED_WAY):
        for member in e:
            m = re.search(rf"\b{member.name}\s*=\s*(\d+)", text)
            if m:
                assert int(m.group(1)) == member.value, member
    assert api.VECTORIZED_WAY.VECTOR_HIP == 3 and api.VECTORIZED_WAY.VECTOR_TOTAL_SIZE == 4


def test_handle_struct_layout_matches_reference_order():
    names = [f[0] for f in api.spmv_Handle._fields_]
    assert names == ["spmvMethod", "data_size", "nthreads", "vectorizedWay", "Level_3_opt_used", "RowPtr",
                     "ColIdx", "index", "Matrix_Val", "Y_temp", "extraHandle"]
    assert C.sizeof(api.spmv_Handle) == 80 and api.spmv_Handle.extraHandle.offset == 72


def test_null_handle_and_options(lib):
    lib.spmv_hip_clear_error()
    api.spmv_destory_handle(None)          # common.c:54-61: NULL is a no-op
    api.spmv_clear_handle(None)
    api.spmv(None, 3, None, None, None, None, None)   # common.c:285
    assert api.last_error()[0] == 0
    assert api.get_option("sell_c") == 64 and api.get_option("sell_sigma") == 1024
    with pytest.raises(ValueError):
        api.set_option("lanes_per_row", 3)
    with pytest.raises(ValueError):
        api.set_option("no_such_key", 1)
    api.set_option("lanes_per_row", 16)
    assert api.get_option("lanes_per_row") == 16
    api.set_option("lanes_per_row", 0)


def test_fails_loudly_without_a_device(lib, monkeypatch):
    """No GPU in this container: create must report NODEVICE, keep a valid handle, and spmv must
    not touch y -- there is no CPU fallback to fall into."""
    if lib.spmv_hip_device_count() > 0:
        pytest.skip("a GPU is visible")
    monkeypatch.setenv("SPMV_HIP_QUIET", "1")
    csr = synth.banded(64, 64)
    with pytest.raises(api.SpmvError, match="no HIP device"):
        api.Handle(csr.m, csr.n, csr.rowptr, csr.colidx, csr.val)
    h = api.spmv_create_handle_all_in_one(csr.m, csr.n, csr.rowptr, csr.colidx, csr.val, 1,
                                          api.SPMV_METHODS.Method_Parallel, 8, check=False)
    assert h and not h.contents.extraHandle
    y = np.full(csr.m, -7.0)
    api.spmv(h, csr.m, csr.rowptr, csr.colidx, csr.val, np.ones(csr.n), y, check=False)
    assert api.last_error()[0] == 5 and (y == -7.0).all()
    lib.spmv_hip_clear_error()
    api.spmv_destory_handle(h)


def test_multi_gpu_option_fails_loudly_without_a_device(lib, monkeypatch):
    if lib.spmv_hip_device_count() > 0:
        pytest.skip("a GPU is visible")
    monkeypatch.setenv("SPMV_HIP_QUIET", "1")
    csr = synth.banded(64, 64)
    api.set_thread_option("gpus", 4)
    try:
        with pytest.raises(api.SpmvError, match="no HIP device"):
            api.Handle(csr.m, csr.n, csr.rowptr, csr.colidx, csr.val)
    finally:
        api.clear_thread_options()
    lib.spmv_hip_clear_error()


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "spmv_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".c", ".h", ".hip", ".hpp")):
                text = open(os.path.join(base, f)).read()
                assert "import oracle" not in text and "from oracle" not in text and "liboracle" not in text, f
