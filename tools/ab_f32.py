#!/usr/bin/env python3
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from spmv_amd import api, build, synth
build.build(); api.load()
dev = "cuda:0"
api.set_option("cache_block", 2)
for dt in (torch.float64, torch.float32):
    m = n = 4_000_000
    lens = torch.full((m,), 24, dtype=torch.int64, device=dev)
    _, _, rp, ci, va = synth.from_row_lengths_device(lens, m, "uniform", dt, dev, 1, cols="web")
    x = torch.rand(n, dtype=dt, device=dev); y = torch.empty(m, dtype=dt, device=dev)
    for var in (0, 19, 21):
        for br in (0, 2048):
            for sk in (0, 4):
                api.set_option("variant", var); api.set_option("block_rows", br); api.set_option("slab_kib", sk)
                h = api.Handle(m, n, rp, ci, va, 4)
                mean, ms = api.time_launches(h.h, x, y, 5, 20)
                print(str(dt)[6:], "variant", var, "block_rows", br, "slab_kib", sk, h.info()["kernel_name"], "ms_min", round(float(ms.min()), 4), flush=True)
                h.close()
