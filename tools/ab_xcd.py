#!/usr/bin/env python3
"""A/B of the XCD-aware block mapping (the default; option xcd_order = 0 = dispatch order) on the tile kernels that gather x through L2:
webbase-style power-law matrices with the three column models, 3e5 ... 4e6 rows (tile executors forced)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from spmv_amd import api, build, synth
build.build(); api.load()
dev = "cuda:0"
M = api.SPMV_METHODS
api.set_option("cache_block", 0)
for cols in ("web", "rmat", "uniform"):
    for m in (300_000, 1_000_000, 2_000_000, 4_000_000):
        lens = synth.powerlaw_lengths_device(m, 3.1, 4700, 1.6, dev, 1)
        _, _, rp, ci, va = synth.from_row_lengths_device(lens, m, "uniform", torch.float64, dev, 1, cols=cols)
        x = torch.rand(m, dtype=torch.float64, device=dev); y = torch.empty(m, dtype=torch.float64, device=dev)
        row = []
        for meth in (M.Method_Balanced2, M.Method_CSR5SPMV):
            for var in (0, 30):
                api.set_option("xcd_order", 0 if var == 30 else 1)
                h = api.Handle(m, m, rp, ci, va, meth)
                _, ms = api.time_launches(h.h, x, y, 5, 30)
                row.append("%s v%d %s %.4f" % (M(meth).name[7:], var, h.info()["kernel_name"], float(ms.min())))
                h.close()
        api.set_option("xcd_order", 1)
        print(cols, m, int(rp[-1]), " | ".join(row), flush=True)
