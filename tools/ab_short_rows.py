#!/usr/bin/env python3
"""Short rows (webbase-style, 2.6 nnz/row): tile kernel (cache_block = 0) against the blocked executor (cache_block = 2) at
1e6 / 2e6 / 4e6 rows, three column models -- where should the blocked executor take over below 8 entries per row?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from spmv_amd import api, build, synth
build.build(); api.load()
dev = "cuda:0"
M = api.SPMV_METHODS
for cols in ("rmat", "uniform", "web"):
    for m in (1_000_000, 2_000_000, 4_000_000):
        lens = synth.powerlaw_lengths_device(m, 3.1, 4700, 1.6, dev, 1)
        _, _, rp, ci, va = synth.from_row_lengths_device(lens, m, "uniform", torch.float64, dev, 1, cols=cols)
        x = torch.rand(m, dtype=torch.float64, device=dev); y = torch.empty(m, dtype=torch.float64, device=dev)
        row = []
        for cb in (0, 2):
            api.set_option("cache_block", cb)
            h = api.Handle(m, m, rp, ci, va, M.Method_Balanced2)
            _, ms = api.time_launches(h.h, x, y, 5, 30)
            row.append("cb%d %s %.4f" % (cb, h.info()["kernel_name"], float(ms.min())))
            h.close()
        api.set_option("cache_block", 1)
        print(cols, m, "x MB", m * 8 / 1e6, " | ".join(row), flush=True)
