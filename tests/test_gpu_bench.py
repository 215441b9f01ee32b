"""GPU: the driver contract of bench.py, run as the driver runs it (a subprocess, one JSON line on stdout), on a small matrix:
the keys the judge reads must be there and sane -- so that a change to the library or to bench.py cannot silently break the
round-end measurement."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_has_the_contract_keys_and_sane_values():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--rows", "200000",
                        "--cpu-rows", "20000", "--cpu-seconds", "0.2", "--config-iters", "3"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "configs", "multi_gpu_c_entry"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert d["value"] > 0 and d["ms_per_step"] > 0 and "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert rf["frac"] > 0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    # the three fractions side by side; `frac` is the counter one when the committed profile matches this run, else the model's -- and says which
    assert rf["frac_source"] in ("counter", "model") and rf["frac_model"] > 0 and rf["frac_alg"] > 0
    assert (rf["frac_counter"] is None) == (rf["traffic"] is None) and rf["frac"] == (rf["frac_counter"] if rf["frac_counter"] is not None else rf["frac_model"])
    assert rf["traffic"] is None or rf["traffic"] > 0
    assert [k["kernel"] for k in rf["kernels"]] == ["csr_vector_tile_kernel"]
    assert rf["read_calibration"] is None or rf["read_calibration"].get("read_gbps", 1) > 0
    cb = d["cpu_baseline"]
    assert cb["parity_ok"] is True and cb["kind"] in ("reference", "port") and cb["cores"] >= 1 and cb["value"] > 0
    assert set(d["configs"]) == {"2-ii", "2-holes", "stencil27", "3-orkut-style", "3-webbase-style", "4"}
    for name, c in d["configs"].items():
        assert c["parity_ok"] is True and c["rows_unwritten"] == 0, (name, c)
        assert c["ms_min"] > 0 and c["ms_min"] <= c["ms_mean"] and c["frac"] > 0 and c["launches"] == 3, (name, c)
        assert c["frac_source"] in ("counter", "model") and c["frac_model"] > 0 and c["frac_alg"] > 0 and len(c["kernels"]) >= 1, (name, c)
        if c["cache_blocked"]:
            assert c["reproducible"] is True and c["deterministic_0"]["parity_ok"] is True, (name, c)
    assert d["configs"]["4"]["dtype"] == "f32" and d["configs"]["4"]["schedule"] == "sell-c-sigma"
    assert d["configs"]["3-orkut-style"]["method"] == "Method_Balanced2"
    mg = d["multi_gpu_c_entry"]
    assert mg["gpus"] >= 1 and mg["matches_single_handle"] is True and mg["ms_per_step"] > 0


def test_c_level_leg_of_the_multi_gpu_bench_rehearsed_with_virtual_shards():
    """At N > 1 rank 0 starts `bench.py --c-leg` as a child: one process, N devices, row blocks handed over separately, the distributed step timed
    for range / allgather / bcast.  Rehearsed here with 3 shards on the one device (SPMV_HIP_GPUS_VIRTUAL=1): the line must parse and block 0 must
    match the definition for every exchange."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", SPMV_HIP_GPUS_VIRTUAL="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--c-leg", "--gpus", "3", "--rows", "100000", "--config-iters", "3"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["gpus_requested"] == 3 and set(d["exchanges"]) == {"range", "allgather", "bcast"}, d
    for name, e in d["exchanges"].items():
        assert e["gpus"] == 3 and e["ms_per_step"] > 0 and e["block0_parity_ok"] is True and e["nnz_total"] == 3 * 100000 * 32, (name, e)
