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


def test_csv_driver_and_cache_interchange_with_the_reference_harness(tmp_path):
    """spmv_amd/bin/test_spmv (this repo's C driver): same CLI and CSV as the
    reference harness; the binary cache it writes (mtx_cache/*.bin, f-1) is then read by the
    REFERENCE's harness binary, and both print RMSE 0 for the same six methods."""
    from spmv_amd import build
    build.build()
    ours = os.path.join(os.path.dirname(build.LIB), "..", "bin", "test_spmv")
    csr = synth.powerlaw(5000, 5000, 7.0, 1500, 1.5, "uniform", np.float64, seed=9)
    mtx = tmp_path / "m.mtx"
    _write_mtx(str(mtx), csr)
    os.mkdir(tmp_path / "mtx_cache")
    env = dict(os.environ, SPMV_HIP_QUIET="1")

    def run(exe, extra_env=None):
        out = subprocess.run([exe, "m.mtx", "1", "2"], capture_output=True, text=True, cwd=str(tmp_path), timeout=300,
                             env=dict(env, **(extra_env or {})))
        assert out.returncode == 0, out.stderr[-2000:]
        return [l.split(",") for l in out.stdout.splitlines() if l.count(",") >= 9]

    a = run(ours)
    assert os.path.exists(tmp_path / "mtx_cache" / "m.mtx.bin")
    assert len(a) == 12 and all(float(f[5]) == 0.0 and int(f[4]) == csr.nnz for f in a)      # 6 methods x threads {1, 2}
    assert {f[2] for f in a} == {"VECTOR_HIP"}
    h = run(ours, {"SPMV_HOST_VECTORS": "1", "TEST_METHOD": "6"})                              # host pointers, one method
    assert len(h) == 2 and all(float(f[5]) == 0.0 and f[1] == "Method_Csr5Spmv" for f in h)
    f32 = run(ours, {"VALUE_TYPE": "float", "TEST_METHOD": "5"})
    assert all(float(f[5]) == 0.0 for f in f32)
    if os.path.exists(oracle.HARNESS):
        os.remove(tmp_path / "mtx_cache" / "m.mtx.bin")
        a2 = run(ours)                      # rewrite the fp64 cache, then let the reference harness read it
        b = run(oracle.HARNESS)
        assert [f[:2] + f[4:6] for f in b] == [f[:2] + f[4:6] for f in a2]                     # same names, nnz, rmse
