#!/usr/bin/env python3
"""Launch floor: webbase-style power-law matrices (mean 3.1 nnz/row) from 1e3 to 4e6 rows, every tile schedule, the
three column models; min launch time (hipEvents) -- how much of the 1e6-row time is fixed cost?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from spmv_amd import api, build, synth
build.build(); api.load()
dev = "cuda:0"
M = api.SPMV_METHODS
for cols in ("rmat", "uniform", "web"):
    for m in (1000, 30_000, 100_000, 300_000, 1_000_000, 4_000_000):
        lens = synth.powerlaw_lengths_device(m, 3.1, min(4700, m), 1.6, dev, 1)
        _, _, rp, ci, va = synth.from_row_lengths_device(lens, m, "uniform", torch.float64, dev, 1, cols=cols)
        x = torch.rand(m, dtype=torch.float64, device=dev); y = torch.empty(m, dtype=torch.float64, device=dev)
        row = []
        for meth in (M.Method_Balanced2, M.Method_CSR5SPMV, M.Method_Parallel, M.Method_SellCSigma):
            for sg in ((0, 4, 16) if meth == M.Method_Balanced2 else (0,)):
                api.set_option("csr5_sigma", sg)
                h = api.Handle(m, m, rp, ci, va, meth)
                _, ms = api.time_launches(h.h, x, y, 5, 30)
                row.append("%s%s/%s %.4f" % (M(meth).name[7:], "(s%d)" % sg if sg else "", h.info()["kernel_name"], float(ms.min())))
                h.close()
            api.set_option("csr5_sigma", 0)
        print(cols, m, int(rp[-1]), " | ".join(row), flush=True)
