#!/usr/bin/env python3
"""A/B config 4 (and the webbase-style stand-in) under two library builds: SPMV_LIB selects the file."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from spmv_amd import api
if os.environ.get("SPMV_LIB"):
    api.LIB_PATH = os.environ["SPMV_LIB"]
api.load()
import run_config as rc
dev = "cuda:0"
for cfg, meths in (("4", (5, 6, 4)), ("3w", (4, 6))):
    m, n, rp, ci, va = rc.make(cfg, dev)
    x = torch.rand(n, dtype=va.dtype, device=dev); y = torch.empty(m, dtype=va.dtype, device=dev)
    for meth in meths:
        h = api.Handle(m, n, rp, ci, va, meth)
        mean, ms = api.time_launches(h.h, x, y, 5, 30)
        print(os.environ.get("SPMV_LIB", "new")[-12:], cfg, "method", meth, h.info()["kernel_name"], "ms_min", round(float(ms.min()), 4), flush=True)
        h.close()
    del rp, ci, va, x, y
    torch.cuda.empty_cache()
