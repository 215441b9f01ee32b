#!/usr/bin/env python3
"""Blocked executor: rows per block (option block_rows; 0 = default rule; BLOCK_ROWS=...) x executor form (VARIANTS=0,29,37: tuned / two-stage / three-stage x 12) on the
run_config shapes:  python tools/ab_blockrows.py 3o 3o-uniform 2r"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from spmv_amd import api, build
import run_config as rc
build.build(); api.load()
dev = "cuda:0"
api.set_option("cache_block", 2)
for cfg in sys.argv[1:] or ["3o"]:
    if cfg.startswith("web24"):   # 4e6 x 24 web-like, fp64 / fp32 (web24f)
        from spmv_amd import synth
        dt = torch.float32 if cfg.endswith("f") else torch.float64
        m = n = 4_000_000
        lens = torch.full((m,), 24, dtype=torch.int64, device=dev)
        _, _, rp, ci, va = synth.from_row_lengths_device(lens, m, "uniform", dt, dev, 1, cols="web")
    else:
        m, n, rp, ci, va = rc.make(cfg, dev)
    x = torch.rand(n, dtype=va.dtype, device=dev); y = torch.empty(m, dtype=va.dtype, device=dev)
    for br in [int(v) for v in os.environ.get("BLOCK_ROWS", "0,2048,4096,8192,16384").split(",")]:
        row = []
        for var in [int(v) for v in os.environ.get("VARIANTS", "0,29,37").split(",")]:
            api.set_option("variant", var); api.set_option("block_rows", br)
            h = api.Handle(m, n, rp, ci, va, 4)
            _, ms = api.time_launches(h.h, x, y, 5, 20)
            row.append("v%d %.4f" % (var, float(ms.min())))
            h.close()
        print(cfg, "block_rows", br, " | ".join(row), flush=True)
    api.set_option("variant", 0); api.set_option("block_rows", 0)
    del rp, ci, va, x, y
    torch.cuda.empty_cache()
