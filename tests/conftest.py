"""pytest configuration: the `gpu` marker and shared fixtures.

`-m "not gpu"` : oracle vs golden vectors, host logic, C-ABI export check (no GPU needed).
`-m gpu`       : parity tests proper -- HIP path vs oracle / golden vectors through the C-ABI.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def load_golden(name):
    """-> (CSR, x, y_ref) from tests/golden/<name>.npz (written by oracle/pin_oracle.py)."""
    from spmv_amd.synth import CSR
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        csr = CSR(int(z["m"]), int(z["n"]), z["rowptr"].copy(), z["colidx"].copy(), z["val"].copy())
        return csr, z["x"].copy(), z["y_ref"].copy()


@pytest.fixture(scope="session")
def golden_names():
    import json
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        return sorted(json.load(f)["cases"].keys())
