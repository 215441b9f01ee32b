#!/usr/bin/env python3
"""One BASELINE config (or SURVEY 8d stand-in) under its named schedule: build the workload, create the
handle, launch the multiply --iters times, print one JSON line (shape, kernel, byte counts, ms).  This is
the program rocprofv3 runs in tools/profile_configs.sh (kernel stats and the separate PMC passes).

    python3 tools/run_config.py --config 2|2r|3w|3o|3o-uniform|4|5shard [--method ID] [--iters K] [--opt key=value ...]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from spmv_amd import api, build, synth  # noqa: E402
from spmv_amd.srchash import csrc_sha  # noqa: E402

M = api.SPMV_METHODS
CONFIGS = {
    "2": ("config 2: 1e7 x 1e7, 32 nnz/row banded, fp64", M.Method_Parallel),
    "2r": ("config 2 variant (ii): uniformly random columns, fp64", M.Method_Parallel),
    "2h": ("config 2 with holes: 32 of the 43 columns of a band per row (mean run length 4), fp64", M.Method_Parallel),
    "s27": ("27-point stencil 215^3 (9.9e6 rows, 2.7e8 nnz), fp64", M.Method_Parallel),
    "3w": ("config 3 stand-in webbase-1M-style: 1e6 rows, mean 3.1, max 4.7k, R-MAT columns, fp64", M.Method_Balanced2),
    "3w-uniform": ("config 3 stand-in webbase-1M-style, uniform columns, fp64", M.Method_Balanced2),
    "3w-web": ("config 3 stand-in webbase-1M-style, web-like columns (90 % within +-2000 of the row, 10 % R-MAT hubs), fp64", M.Method_Balanced2),
    "3o": ("config 3 stand-in com-Orkut-style: 3.07e6 rows, ~2.3e8 nnz, R-MAT columns, fp64", M.Method_Balanced2),
    "3o-uniform": ("config 3 stand-in com-Orkut-style, uniform columns, fp64", M.Method_Balanced2),
    "3o-scrambled": ("config 3 stand-in com-Orkut-style (R-MAT) with its vertices renumbered at random: hubs scattered over the index range, fp64", M.Method_Balanced2),
    "4": ("config 4: 1e7 rows skewed nnz, fp32, columns within +-4096", M.Method_SellCSigma),
    "5shard": ("config 5 shard: 1e7 of 8e7 rows, global columns, fp64", M.Method_Parallel),
}


def make(config, dev):
    f64 = torch.float64
    if config == "2":
        return synth.banded_device(10_000_000, 10_000_000, 32, "uniform", f64, dev, 1)
    if config == "2h":
        return synth.banded_holes_device(10_000_000, 10_000_000, 32, 0.25, "uniform", f64, dev, 1)
    if config == "s27":
        return synth.stencil27_device(215, "uniform", f64, dev, 1)
    if config == "2r":
        return synth.uniform_k_device(10_000_000, 10_000_000, 32, "uniform", f64, dev, 1)
    if config in ("3w", "3w-uniform", "3w-web"):
        lens = synth.powerlaw_lengths_device(1_000_000, 3.1, 4700, 1.6, dev, 1)
        return synth.from_row_lengths_device(lens, 1_000_000, "uniform", f64, dev, 1, cols={"3w": "rmat", "3w-uniform": "uniform", "3w-web": "web"}[config])
    if config in ("3o", "3o-uniform"):
        lens = synth.powerlaw_lengths_device(3_070_000, 76, 33000, 1.5, dev, 1)
        return synth.from_row_lengths_device(lens, 3_070_000, "uniform", f64, dev, 1, cols="rmat" if config == "3o" else "uniform")
    if config == "3o-scrambled":
        lens = synth.powerlaw_lengths_device(3_070_000, 76, 33000, 1.5, dev, 1)
        m, n, rp, ci, va = synth.from_row_lengths_device(lens, 3_070_000, "uniform", f64, dev, 1, cols="rmat")
        g = torch.Generator(device=dev); g.manual_seed(99)
        sc = torch.randperm(m, generator=g, device=dev)
        inv = torch.empty_like(sc); inv[sc] = torch.arange(m, device=dev)
        lens2 = lens[sc]
        rp2 = torch.zeros(m + 1, dtype=torch.int64, device=dev)
        torch.cumsum(lens2, 0, out=rp2[1:])
        src = torch.repeat_interleave(rp.long()[:-1][sc] - rp2[:-1], lens2) + torch.arange(int(rp2[-1]), device=dev)
        ci2 = inv[ci.long()[src]].to(torch.int32)
        va2 = va[src]
        del src, ci, va
        return m, n, rp2.to(torch.int32), ci2, va2
    if config == "4":
        lens = synth.skewed_lengths_device(10_000_000, dev, 1)
        return synth.from_row_lengths_device(lens, 10_000_000, "uniform", torch.float32, dev, 1, local=4096)
    if config == "5shard":
        return synth.banded_device(10_000_000, 80_000_000, 32, "uniform", f64, dev, 1, row0=30_000_000)
    raise SystemExit(f"unknown config {config}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", required=True, choices=sorted(CONFIGS))
    ap.add_argument("--method", type=int, default=-1)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--opt", action="append", default=[], help="library option key=value")
    a = ap.parse_args()
    if os.environ.get("SPMV_LIB"):   # same-box A/B against another build of the library (e.g. the previous commit's)
        api.LIB_PATH = os.environ["SPMV_LIB"]
    else:
        build.build()
    api.load()
    dev = "cuda:0"
    for kv in a.opt:
        k, v = kv.split("=")
        api.set_option(k, int(v))
    name, method = CONFIGS[a.config]
    if a.method >= 0:
        method = M(a.method)
    m, n, rp, ci, va = make(a.config, dev)
    x = torch.rand(n, dtype=va.dtype, device=dev) * 2 - 1
    y = torch.empty(m, dtype=va.dtype, device=dev)
    t0 = time.time()
    h = api.Handle(m, n, rp, ci, va, method)
    create_s = time.time() - t0
    info = h.info()
    mean, ms = api.time_launches(h.h, x, y, a.warmup, a.iters)
    actual = h.method.name
    h.close()
    out = {"config": a.config, "name": name, "method": M(method).name, "method_used": actual, "schedule": info["schedule_name"],
           "kernel": info["kernel_name"], "m": m, "n": n, "nnz": info["nnz"], "dtype": "f64" if va.dtype == torch.float64 else "f32",
           "stored_nnz": info["stored_nnz"], "cache_blocked": info["cache_blocked"], "x_groups": info["x_groups"],
           "x_groups_staged": info["x_groups_staged"], "alg_bytes": info["alg_bytes"], "stream_bytes": info["stream_bytes"],
           "x_bytes": info["x_bytes"], "ms_min": round(float(ms.min()), 5), "ms_mean": round(float(mean), 5),
           "create_s": round(create_s, 3), "inspect_ms": round(info["inspect_ms"], 2), "options": a.opt,
           "launch_kernels": info["launch_kernels"], "blk_waves": info["blk_waves"], "reproducible": info["reproducible"], "run_nnz": info["run_nnz"],
           "byte_nnz": info["byte_nnz"], "tmpl_nnz": info["tmpl_nnz"], "csrc_sha": csrc_sha(),
           "gflops": round(2 * info["nnz"] / float(ms.min()) / 1e6, 1),
           "moved_gbps": round(info["stream_bytes"] / float(ms.min()) / 1e6, 1), "frac_moved": round(info["stream_bytes"] / float(ms.min()) / 1e6 / 8000, 4),
           "alg_gbps": round(info["alg_bytes"] / float(ms.min()) / 1e6, 1), "frac_alg": round(info["alg_bytes"] / float(ms.min()) / 1e6 / 8000, 4)}
    print("RUNCONFIG " + json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
