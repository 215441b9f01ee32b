#!/usr/bin/env python3
"""Where should the blocked executor take over?  Uniformly random columns, 16 nnz/row, x from 4 to 32 MB:
tile schedule (cache_block=0) vs blocked executor (cache_block=2), Method_Balanced2 and Method_Parallel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from spmv_amd import api, build, synth
build.build(); api.load()
dev = "cuda:0"
for dt in (torch.float64, torch.float32):
    for n in ((250_000, 500_000, 1_000_000, 2_000_000, 4_000_000) if dt == torch.float64 else (500_000, 1_000_000, 2_000_000, 4_000_000, 8_000_000)):
        m, _, rp, ci, va = synth.uniform_k_device(n, n, 16, "uniform", dt, dev, seed=3)
        x = torch.rand(n, dtype=dt, device=dev); y = torch.empty(m, dtype=dt, device=dev)
        row = []
        for meth in (4, 6):
            for cb in (0, 2):
                api.set_option("cache_block", cb)
                h = api.Handle(m, n, rp, ci, va, meth)
                mean, ms = api.time_launches(h.h, x, y, 5, 30)
                row.append((meth, cb, h.info()["kernel_name"], round(float(ms.min()) * 1e3, 1)))
                h.close()
        api.set_option("cache_block", 1)
        print(str(dt)[6:], "n", n, "x MB", n * va.element_size() / 1e6, row, flush=True)
