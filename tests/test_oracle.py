"""CPU: the oracle (oracle/oracle_spmv.c) against the golden vectors captured from the REAL
reference's Method_Serial (oracle/pin_oracle.py), and -- when oracle/_ref is present -- against
the reference itself, live."""
import json
import os

import numpy as np
import pytest

import oracle
from oracle import cases
from conftest import GOLDEN, load_golden

with open(os.path.join(GOLDEN, "manifest.json")) as _f:
    NAMES = sorted(json.load(_f)["cases"].keys())


def test_manifest_covers_case_table():
    assert NAMES == sorted(cases.case_names())


@pytest.mark.parametrize("name", NAMES)
def test_oracle_bitwise_equals_reference_golden(name):
    csr, x, y_ref = load_golden(name)
    y = oracle.spmv_serial(csr, x)
    assert y.dtype == y_ref.dtype
    assert not np.isnan(y_ref).any(), "reference left rows unwritten?"
    assert np.array_equal(y.view(np.uint8), y_ref.view(np.uint8))


@pytest.mark.parametrize("name", NAMES)
def test_fixture_inputs_regenerate(name):
    """The committed inputs are what oracle/cases.py builds (fixtures are reproducible data)."""
    csr, x, _ = load_golden(name)
    c2, x2 = cases.build_case(name)
    assert (csr.m, csr.n) == (c2.m, c2.n)
    assert np.array_equal(csr.rowptr, c2.rowptr) and np.array_equal(csr.colidx, c2.colidx)
    assert np.array_equal(csr.val, c2.val) and np.array_equal(x, x2)


@pytest.mark.parametrize("name", [n for n in NAMES if n.endswith("eighths")])
def test_eighths_are_exact(name):
    """k/8 operands: every summation order gives the same bits (test_spmv.c:199-207 trick)."""
    csr, x, y_ref = load_golden(name)
    assert np.array_equal(y_ref.astype(np.float64), oracle.spmv_exact(csr, x))


@pytest.mark.parametrize("name", NAMES)
def test_omp_variant_matches_serial(name):
    csr, x, y_ref = load_golden(name)
    assert np.array_equal(oracle.spmv_omp(csr, x).view(np.uint8), y_ref.view(np.uint8))


@pytest.mark.skipif(not oracle.have_ref(), reason="oracle/_ref not built (no /root/reference here)")
@pytest.mark.parametrize("name", ["banded_f64_uniform", "powerlaw_f32_uniform", "rowlen_sweep_f64_uniform",
                                  "dense_row0_f32_uniform", "empty_mix_f64_uniform"])
def test_live_reference_agrees(name):
    csr, x, y_ref = load_golden(name)
    y, actual = oracle.ref_spmv(csr, x, method=0, nthreads=1)
    assert actual == 0
    assert np.array_equal(y.view(np.uint8), y_ref.view(np.uint8))


def test_config1_shape_oracle_vs_reference_or_exact():
    """BASELINE config 1: Method_Serial fp64, 100k x 100k banded ~16 nnz/row, on the CPU."""
    from spmv_amd import synth
    csr = synth.banded(100_000, 100_000, 8, 7, "uniform", np.float64, seed=1)
    x = synth.fill_x(csr.n, "uniform", np.float64, 2)
    assert abs(csr.nnz / csr.m - 16) < 0.01
    y = oracle.spmv_serial(csr, x)
    if oracle.have_ref():
        y_ref, _ = oracle.ref_spmv(csr, x, method=0, nthreads=1)
        assert np.array_equal(y.view(np.uint8), y_ref.view(np.uint8))
    err = np.abs(y - oracle.spmv_exact(csr, x))
    assert (err <= 1e-12 * np.maximum(oracle.row_abs_sum(csr, x), 1e-300)).all()
