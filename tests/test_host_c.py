"""CPU: the library's host-side C (Matrix Market reader, cache I/O, RCM reordering) built with
AddressSanitizer + UBSan and driven over valid, malformed and truncated inputs (the
reference has no sanitizer configuration at all; GPU sanitizers are not available on this pool, so
the host C is where they run)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "spmv_amd", "csrc")


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("hostc") / "host_c_check")
    cmd = ["gcc", "-std=c11", "-O1", "-g", "-Wall", "-Wextra", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           f"-I{os.path.join(ROOT, 'include')}", f"-I{CSRC}", os.path.join(ROOT, "tests", "c", "host_c_check.c"),
           os.path.join(CSRC, "io", "mtx_io.c"), os.path.join(CSRC, "reorder", "rcm.c"), os.path.join(CSRC, "spmv_plan.c"), "-lpthread", "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return out


def _run(exe, *args):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    return subprocess.run([exe, *args], capture_output=True, text=True, env=env, timeout=120)


def test_planner_maps_methods_and_shapes_csr_vector_from_histograms(exe):
    """spmv_plan.c under ASan/UBSan: method -> schedule map, the Balanced/Balanced2 rewrite of the reference
    (parallel_balanced2_spmv.c:72-92), the histogram cost model for CSR-vector (L, long-row threshold), auto_method."""
    r = _run(exe, "plan")
    assert r.returncode == 0, r.stdout + r.stderr
    assert "equal32: sched=1 L=8 thr=0" in r.stdout and "plan rc=0" in r.stdout


def test_rcm_recovers_a_scrambled_band_without_sanitizer_findings(exe):
    r = _run(exe, "rcm")
    assert r.returncode == 0, r.stdout + r.stderr
    assert "after=5 " in r.stdout and "ERROR" not in r.stderr


def test_mtx_reader_on_valid_malformed_and_truncated_files(exe, tmp_path):
    cases = {
        "ok.mtx": "%%MatrixMarket matrix coordinate real general\n% c\n3 3 2\n1 1 1.5\n3 2 -2\n",
        "sym.mtx": "%%MatrixMarket matrix coordinate real symmetric\n3 3 2\n2 1 1.5\n3 3 4\n",
        "trunc_entries.mtx": "%%MatrixMarket matrix coordinate real general\n3 3 5\n1 1 1.5\n",
        "trunc_value.mtx": "%%MatrixMarket matrix coordinate real general\n3 3 1\n1 1",
        "no_size.mtx": "%%MatrixMarket matrix coordinate real general\n% only comments\n",
        "oob.mtx": "%%MatrixMarket matrix coordinate real general\n2 2 1\n0 1 1.0\n",
        "neg_size.mtx": "%%MatrixMarket matrix coordinate real general\n-2 2 1\n1 1 1.0\n",
        "huge_count.mtx": "%%MatrixMarket matrix coordinate real general\n2 2 99999999999\n1 1 1.0\n",
        "sym_rect.mtx": "%%MatrixMarket matrix coordinate real symmetric\n2 5 1\n1 4 1.0\n",
        "banner.mtx": "%MatrixMarket matrix coordinate real general\n2 2 1\n1 1 1.0\n",
        "binary.mtx": "%%MatrixMarket matrix coordinate real general\n2 2 1\n\x00\xff\xfe 1 1\n",
        "empty.mtx": "",
    }
    paths = []
    for name, text in cases.items():
        p = tmp_path / name
        p.write_bytes(text.encode("latin-1"))
        paths.append(str(p))
    r = _run(exe, "mtx", *paths, str(tmp_path / "missing.mtx"))
    assert r.returncode == 0 and "ERROR" not in r.stderr, r.stdout + r.stderr
    rc = {line.split()[0].split("/")[-1]: int(line.split("rc=")[1].split()[0]) for line in r.stdout.splitlines() if " rc=" in line}
    assert rc["ok.mtx"] == 0 and rc["sym.mtx"] == 0
    assert rc["trunc_entries.mtx"] == -5 and rc["trunc_value.mtx"] == -5 and rc["oob.mtx"] == -5 and rc["sym_rect.mtx"] == -5
    assert rc["no_size.mtx"] == -4 and rc["neg_size.mtx"] == -4 and rc["huge_count.mtx"] == -4
    assert rc["banner.mtx"] == -2 and rc["empty.mtx"] == -2 and rc["binary.mtx"] == -5 and rc["missing.mtx"] == -1


def test_bin_cache_roundtrip_and_short_file(exe, tmp_path):
    r = _run(exe, "bin", str(tmp_path / "c.bin"))
    assert r.returncode == 0 and "ERROR" not in r.stderr, r.stdout + r.stderr
