#!/usr/bin/env python3
"""Locality routing on mixed matrices: banded rows and random-column rows in one matrix, 32 nnz/row, fp64.
    python tools/route_bench.py [rows]      prints per shape and method: ms, executor, create()'s measured choice (route_ms)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from spmv_amd import api, build, synth
build.build(); api.load()
dev = "cuda:0"
m = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
k = 32


import route_bench_lib


def mixed(kind):
    return route_bench_lib.mixed(kind, m, k, dev)


for kind in sys.argv[2:] or ["banded", "random", "prefix1", "tail10", "every10"]:
    rp, ci, va = mixed(kind)
    x = torch.rand(m, dtype=torch.float64, device=dev) * 2 - 1
    y = torch.empty(m, dtype=torch.float64, device=dev)
    for method in (1, 6):
        h = api.Handle(m, m, rp, ci, va, method)
        info = h.info()
        mean, ms = api.time_launches(h.h, x, y, 5, 20)
        print("ROUTE " + json.dumps({"kind": kind, "method": api.SPMV_METHODS(method).name, "kernel": info["kernel_name"], "blocked": info["cache_blocked"],
                                     "ms_min": round(float(ms.min()), 4), "x_groups": info["x_groups"], "staged": info["x_groups_staged"],
                                     "route_ms": [round(v, 4) for v in info["route_ms"]], "split_ms": [round(v, 4) for v in info["split_ms"]],
                                     "far_share": round(info["far_nnz"] / max(info["nnz"], 1), 4), "tuned": info["tuned_choice"], "tune_ms": [round(v, 4) for v in info["tune_ms"]], "stored": info["stored_nnz"], "inspect_ms": round(info["inspect_ms"], 1)}), flush=True)
        h.close()
    del rp, ci, va, x, y
    torch.cuda.empty_cache()
