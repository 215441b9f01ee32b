#!/usr/bin/env python3
"""A/B of two library builds (SPMV_LIB=path selects the file; default: the in-tree build) on one method over several shapes:
    METHOD=5 python tools/ab_lib.py 2 4 short      (run_config shapes; "short" = 6e6 rows of 3..19 entries, +-300 columns, fp64 and fp32)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from spmv_amd import api, synth
if os.environ.get("SPMV_LIB"):
    api.LIB_PATH = os.environ["SPMV_LIB"]
api.load()
import run_config as rc
dev = "cuda:0"
tag = os.path.basename(os.environ.get("SPMV_LIB", "in-tree"))
meth = int(os.environ.get("METHOD", "5"))
def run(name, m, n, rp, ci, va):
    x = torch.rand(n, dtype=va.dtype, device=dev); y = torch.empty(m, dtype=va.dtype, device=dev)
    h = api.Handle(m, n, rp, ci, va, meth)
    _, ms = api.time_launches(h.h, x, y, 5, 30)
    print(tag, name, str(va.dtype)[6:], h.info()["kernel_name"], "ms_min", round(float(ms.min()), 4), flush=True)
    h.close()
for cfg in sys.argv[1:] or ["2", "4", "short"]:
    if cfg == "short":
        for dt in (torch.float64, torch.float32):
            g = torch.Generator(device=dev); g.manual_seed(2)
            m = n = 6_000_000
            lens = torch.randint(3, 20, (m,), generator=g, device=dev, dtype=torch.int64)
            _, _, rp, ci, va = synth.from_row_lengths_device(lens, n, "uniform", dt, dev, 3, local=300)
            run("short rows 3..19, local 300", m, n, rp, ci, va)
            del rp, ci, va
    else:
        m, n, rp, ci, va = rc.make(cfg, dev)
        run("config " + cfg, m, n, rp, ci, va)
        del rp, ci, va
    torch.cuda.empty_cache()
