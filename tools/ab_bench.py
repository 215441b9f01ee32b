#!/usr/bin/env python3
"""Developer tool: interleaved A/B of (method, vector_form) pairs on one matrix in ONE process
(cdna_hip_programming.md 5.4 rule 24): all handles are created first, then R rounds visit them in
turn; min and median over rounds are reported, which removes the thermal drift a sequential run shows."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from spmv_amd import api, synth, build

ap = argparse.ArgumentParser()
ap.add_argument("--m", type=int, default=10_000_000)
ap.add_argument("--k", type=int, default=32)
ap.add_argument("--dtype", default="f64")
ap.add_argument("--kind", default="banded")
ap.add_argument("--pairs", default="1:10,1:11,1:5,1:4,6:0,5:0,2:0", help="method:vector_form,...")
ap.add_argument("--rounds", type=int, default=8)
ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()
build.build()
dt = torch.float64 if a.dtype == "f64" else torch.float32
dev = "cuda:0"
if a.kind == "banded":
    m, n, rp, ci, va = synth.banded_device(a.m, a.m, a.k, "uniform", dt, dev, 1)
elif a.kind == "skewed":
    m, n, rp, ci, va = synth.from_row_lengths_device(synth.skewed_lengths_device(a.m, dev, 1), a.m, "uniform", dt, dev, 1, local=4096)
x = torch.rand(n, dtype=dt, device=dev) * 2 - 1
y = torch.empty(m, dtype=dt, device=dev)
hs = []
for pr in a.pairs.split(","):
    meth, var = (int(v) for v in pr.split(":"))
    api.set_option("vector_form", var)
    api.set_option("autotune", 0)
    h = api.Handle(m, n, rp, ci, va, meth)
    hs.append((f"{pr}#{len(hs)}", h, h.info()))
times = {pr: [] for pr, _, _ in hs}
for r in range(a.rounds):
    for pr, h, info in hs:
        mean, ms = api.time_launches(h.h, x, y, 1, a.iters)
        times[pr].append(float(ms.min()))
for pr, h, info in hs:
    t = np.array(times[pr]); gb = info["alg_bytes"] / 1e9
    print(f"{pr:8s} {info['kernel_name']:26s} min {t.min():.4f} med {np.median(t):.4f} max {t.max():.4f} ms   frac(min) {gb / t.min() / 8:.3f}", flush=True)
    h.close()
