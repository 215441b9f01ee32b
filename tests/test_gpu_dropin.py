"""GPU: the reference's OWN harness (src/samples/test_spmv.c, compiled unmodified by oracle/Makefile
against libspmv_hip.so -> oracle/_ref/test_spmv_hip) run on a Matrix Market file.  Its inline golden
uses values rand()%8/8 and x = 1 (test_spmv.c:199-207), i.e. exact arithmetic, and it prints the
RMSE per method in CSV column 6 (test_spmv.c:147-149): every method must print 0."""
import os
import subprocess

import numpy as np
import pytest

import oracle
from spmv_amd import synth

pytestmark = pytest.mark.gpu


def _write_mtx(path, csr):
    rows = np.repeat(np.arange(csr.m), np.diff(csr.rowptr))
    with open(path, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n")
        f.write(f"{csr.m} {csr.n} {csr.nnz}\n")
        for r, c, v in zip(rows, csr.colidx, csr.val):
            f.write(f"{r + 1} {c + 1} {v:.17g}\n")


@pytest.mark.skipif(not os.path.exists(oracle.HARNESS), reason="oracle/_ref/test_spmv_hip not built (needs /root/reference at build time)")
@pytest.mark.parametrize("kind", ["banded", "powerlaw"])
def test_reference_harness_runs_on_the_hip_library(tmp_path, kind):
    if kind == "banded":
        csr = synth.banded(3000, 3000, 8, 7, "uniform", np.float64, seed=1)
    else:
        csr = synth.powerlaw(4000, 4000, 6.0, 900, 1.5, "uniform", np.float64, seed=2)
        # the harness mallocs nothing for empty matrices but handles empty rows; keep them
    mtx = tmp_path / f"{kind}.mtx"
    _write_mtx(str(mtx), csr)
    env = dict(os.environ, SPMV_HIP_QUIET="1")
    out = subprocess.run([oracle.HARNESS, str(mtx), "1", "1"], capture_output=True, text=True, cwd=str(tmp_path),
                         timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.count(",") >= 9]
    assert len(lines) == 6, out.stdout            # methods 1..6 (test_spmv.c:238-244)
    seen = set()
    for l in lines:
        f = l.split(",")
        seen.add(f[1])
        assert int(f[4]) == csr.nnz
        assert float(f[5]) == 0.0, l               # RMSE vs the harness' own golden
    assert {"Method_Parallel", "Method_SellCSigma", "Method_Csr5Spmv"} <= seen
