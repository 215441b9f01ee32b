"""The golden-vector case table (SURVEY 8c "Golden vectors to commit", 4.4 item 4).

Shared by oracle/pin_oracle.py (which runs the REAL reference on each case and writes
tests/golden/<name>.npz) and by the tests (which read the fixtures back).  Each entry builds a
small CSR matrix + x deterministically; the shapes cover what the reference's other methods get
wrong: empty rows (leading / interior / trailing), a long row 0, one row far longer
than nnz/workers, m < workers, n != m, nnz = 0, and every remainder branch of the dot kernel.
"""
from __future__ import annotations

import numpy as np

from spmv_amd import synth


def _rowlen_sweep(dtype, values):
    m, n = 410, 977
    lens = np.arange(m) % 41
    return synth.from_row_lengths(lens, n, values, dtype, seed=11)


def _single_long_row(dtype, values):
    return synth.from_row_lengths(np.array([5000]), 5000, values, dtype, seed=12)


def _nnz0(dtype, values):
    return synth.from_row_lengths(np.zeros(17, dtype=np.int64), 9, values, dtype, seed=13)


def _tiny(dtype, values):
    return synth.from_row_lengths(np.array([2, 0, 5]), 5, values, dtype, seed=14)


def _empty_mix(dtype, values):
    base = synth.banded(1000, 1000, 8, 7, values, dtype, seed=15)
    return synth.with_empty_rows(base, lead=10, trail=10, every=7)


def _powerlaw(dtype, values):
    return synth.powerlaw(3000, 3000, mean_len=3.1, max_len=800, alpha=1.6, values=values, dtype=dtype, seed=16)


# name -> (builder(dtype, values) -> CSR)
STRUCTURES = {
    "banded":        lambda dt, v: synth.banded(1200, 1200, 8, 7, v, dt, seed=1),
    "banded_wide":   lambda dt, v: synth.banded(600, 900, 40, 37, v, dt, seed=2),          # n != m
    "uniformk32":    lambda dt, v: synth.uniform_k(900, 1800, 32, v, dt, seed=3),          # n != m
    "powerlaw":      _powerlaw,
    "dense_row0":    lambda dt, v: synth.dense_rows(500, 700, [0], 3, v, dt, seed=4),
    "dense_mid":     lambda dt, v: synth.dense_rows(600, 640, [300, 301], 3, v, dt, seed=5),
    "empty_mix":     _empty_mix,
    "rowlen_sweep":  _rowlen_sweep,
    "single_long":   _single_long_row,
    "tiny":          _tiny,
    "nnz0":          _nnz0,
    "skewed":        lambda dt, v: synth.skewed_rows(1500, 4096, v, dt, seed=6),
}

DTYPES = {"f64": np.float64, "f32": np.float32}
FILLS = ("uniform", "eighths")


def case_names():
    return [f"{s}_{d}_{f}" for s in STRUCTURES for d in DTYPES for f in FILLS]


def build_case(name):
    """-> (CSR, x) for a case name such as 'banded_f64_uniform'."""
    struct, d, fill = name.rsplit("_", 2)
    dtype = DTYPES[d]
    csr = STRUCTURES[struct](dtype, fill)
    seed = (sum(ord(c) for c in name) * 2654435761) & 0x7FFFFFFF
    x = synth.fill_x(csr.n, fill, dtype, seed)
    return csr, x
