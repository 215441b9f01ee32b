"""CPU: the pure helpers of bench.py -- the three roofline fractions, the traffic lookup's keying (kernel, shape, dtype, the model's byte count and the
source hash), the column-stream accounting and the per-kernel shares.  No GPU, no library call: fake spmv_hip_info dictionaries."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _info(**kw):
    base = dict(kernel_name="csr_vector_tile_kernel", m=1000, n=1000, nnz=32000, alg_bytes=400_000, stream_bytes=300_000, x_bytes=9000, cache_blocked=0,
                x_groups=4, x_groups_staged=4, run_nnz=0, byte_nnz=0, tmpl_nnz=0, launch_kernels=["csr_vector_tile_kernel"])
    base.update(kw)
    return base


@pytest.fixture()
def fake_profiles(monkeypatch, tmp_path):
    from spmv_amd import srchash
    (tmp_path / "profiles").mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "_TRAFFIC", None)

    def write(sha, entries):
        with open(tmp_path / "profiles" / "traffic_r99.json", "w") as f:
            json.dump({"csrc_sha": sha, "entries": entries}, f)
        bench._TRAFFIC = None
    return write, srchash.csrc_sha()


def test_fractions_side_by_side_and_frac_is_the_counter_one_only_when_the_profile_matches(fake_profiles):
    write, sha = fake_profiles
    entry = {"kernel_short": "csr_vector_tile_kernel", "m": 1000, "nnz": 32000, "dtype": "f64", "model_stream_bytes": 300_000, "hbm_bytes_per_launch": 290_000.0,
             "source": "profiles/x", "kernels": [{"kernel": "csr_vector_tile_kernel", "avg_ms": 0.5, "hbm_bytes": 290_000.0}]}
    write(sha, [entry])
    rf = bench.roofline_fields(_info(), 0.001, "f64")          # 1 us launch
    assert rf["frac_source"] == "counter" and rf["traffic"] == 290_000.0
    assert rf["frac"] == rf["frac_counter"] == round(290_000 / 1e-6 / 1e9 / 8000.0, 4)
    assert rf["frac_model"] == round(300_000 / 1e-6 / 1e9 / 8000.0, 4) and rf["frac_alg"] == round(400_000 / 1e-6 / 1e9 / 8000.0, 4)
    assert rf["kernels"] == [{"kernel": "csr_vector_tile_kernel", "avg_ms_rocprof": 0.5, "share": 1.0, "hbm_bytes": 290_000.0}]
    # another matrix of the same shape (config 2 vs config 2 with holes): the model's byte count differs -> no match
    rf = bench.roofline_fields(_info(stream_bytes=330_000), 0.001, "f64")
    assert rf["frac_source"] == "model" and rf["traffic"] is None and rf["frac"] == rf["frac_model"] and rf["frac_counter"] is None
    # two matrices whose byte counts differ by less than the tolerance (the Orkut-style stand-in with R-MAT and with uniform columns): the closest entry wins
    near = dict(entry, model_stream_bytes=300_020, hbm_bytes_per_launch=310_000.0, source="profiles/near")
    write(sha, [near, entry])
    assert bench.roofline_fields(_info(), 0.001, "f64")["traffic"] == 290_000.0
    assert bench.roofline_fields(_info(stream_bytes=300_019), 0.001, "f64")["traffic"] == 310_000.0
    # a profile taken on another source tree says nothing about this run
    write("0" * 16, [entry])
    rf = bench.roofline_fields(_info(), 0.001, "f64")
    assert rf["frac_source"] == "model" and rf["traffic"] is None


def test_cache_resident_shapes_get_the_cache_level_bound(fake_profiles):
    write, sha = fake_profiles
    write(sha, [])
    small = bench.roofline_fields(_info(stream_bytes=48 << 20, alg_bytes=51 << 20), 0.027, "f64")
    assert small["cache_resident"]["level"] == "infinity_cache" and small["cache_resident"]["peak"] == bench.IC_READ_GBPS
    tiny = bench.roofline_fields(_info(stream_bytes=8 << 20, alg_bytes=9 << 20), 0.01, "f64")
    assert tiny["cache_resident"]["level"] == "l2"
    big = bench.roofline_fields(_info(stream_bytes=3 << 30, alg_bytes=4 << 30), 0.41, "f64")
    assert "cache_resident" not in big


def test_column_stream_bytes_per_nonzero():
    assert bench.column_stream_bytes(_info(run_nnz=32000)) == 0.0                       # RUN tiles: 2 B per row, nothing per entry
    assert bench.column_stream_bytes(_info(tmpl_nnz=32000)) == 0.0                      # TEMPLATE tiles likewise
    assert bench.column_stream_bytes(_info(byte_nnz=32000)) == 1.0                      # BYTE tiles
    assert bench.column_stream_bytes(_info()) == 2.0                                    # staged: 16-bit slots
    assert bench.column_stream_bytes(_info(x_groups_staged=0)) == 4.0                   # unstaged: int32 ColIdx
    assert bench.column_stream_bytes(_info(cache_blocked=1)) == 4.0                     # blocked streams: a 32-bit word per entry
    assert bench.column_stream_bytes(_info(run_nnz=16000, byte_nnz=8000)) == round((8000 * 1.0 + 8000 * 2.0) / 32000, 3)


def test_kernel_shares_list_every_launch():
    parts = [{"kernel": "csr5_group_pipe_kernel", "avg_ms": 0.41, "hbm_bytes": 2.7e9}, {"kernel": "sell_window_kernel", "avg_ms": 0.17, "hbm_bytes": 1.1e9},
             {"kernel": "csr5_fixup_kernel", "avg_ms": 0.008, "hbm_bytes": 1e7}]
    out = bench.kernel_shares(["sell_window_kernel", "csr5_group_pipe_kernel", "csr5_fixup_kernel"], parts)
    assert [k["kernel"] for k in out] == ["sell_window_kernel", "csr5_group_pipe_kernel", "csr5_fixup_kernel"]
    assert abs(sum(k["share"] for k in out) - 1.0) < 2e-3 and out[1]["share"] > 0.69
    assert bench.kernel_shares(["nat_kernel", "csr5_fixup_kernel"], None) == [{"kernel": "nat_kernel"}, {"kernel": "csr5_fixup_kernel"}]
