#!/usr/bin/env python3
"""A/B of executor variants (option "variant") on the run_config shapes, blocked executor forced:
    VARIANTS=0,26,23 python tools/ab_variants.py 2r 3o 3o-uniform web24"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from spmv_amd import api, build, synth
import run_config as rc
build.build(); api.load()
dev = "cuda:0"
api.set_option("cache_block", int(os.environ.get("CACHE_BLOCK", "2")))
variants = [int(v) for v in os.environ.get("VARIANTS", "0").split(",")]
for cfg in sys.argv[1:] or ["2r", "3o", "3o-uniform"]:
    if cfg.startswith("web24"):   # 4e6 x 24 web-like, fp64 / fp32 (web24f)
        dt = torch.float32 if cfg.endswith("f") else torch.float64
        m = n = 4_000_000
        lens = torch.full((m,), 24, dtype=torch.int64, device=dev)
        _, _, rp, ci, va = synth.from_row_lengths_device(lens, m, "uniform", dt, dev, 1, cols="web")
    else:
        m, n, rp, ci, va = rc.make(cfg, dev)
    x = torch.rand(n, dtype=va.dtype, device=dev); y = torch.empty(m, dtype=va.dtype, device=dev)
    ref = None
    for var in variants:
        api.set_option("variant", var)
        h = api.Handle(m, n, rp, ci, va, 4)
        mean, ms = api.time_launches(h.h, x, y, 5, 20)
        if ref is None:
            ref = y.clone()
        err = float((y - ref).abs().max())
        print(cfg, "variant", var, h.info()["kernel_name"], "ms_min", round(float(ms.min()), 4), "max|dy| vs first", err, flush=True)
        h.close()
    del rp, ci, va, x, y, ref
    torch.cuda.empty_cache()
