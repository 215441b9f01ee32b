#!/usr/bin/env python3
"""A/B: a large matrix WITH empty rows (the tiles run over the compacted row space and write y[row_map[r]]) under CSR5 and
nnz-split; SPMV_LIB selects the library file (old vs new build)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from spmv_amd import api, synth
if os.environ.get("SPMV_LIB"):
    api.LIB_PATH = os.environ["SPMV_LIB"]
api.load()
dev = "cuda:0"
g = torch.Generator(device=dev); g.manual_seed(1)
for dt in (torch.float64, torch.float32):
    m = n = 4_000_000
    lens = torch.randint(1, 40, (m,), generator=g, device=dev)
    lens[(torch.arange(m, device=dev) // 5000) % 7 == 3] = 0
    lens[torch.rand(m, generator=g, device=dev) < 0.2] = 0
    _, _, rp, ci, va = synth.from_row_lengths_device(lens.to(torch.int64), n, "uniform", dt, dev, seed=5, local=600)
    x = torch.rand(n, dtype=dt, device=dev); y = torch.empty(m, dtype=dt, device=dev)
    for meth in (6, 4):
        h = api.Handle(m, n, rp, ci, va, meth)
        mean, ms = api.time_launches(h.h, x, y, 5, 30)
        print(os.environ.get("SPMV_LIB", "new")[-12:], str(dt)[6:], "method", meth, h.info()["kernel_name"], "nnz", int(rp[-1]), "ms_min", round(float(ms.min()), 4), flush=True)
        h.close()
