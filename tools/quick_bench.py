#!/usr/bin/env python3
"""Developer tool: time every schedule on a config-2-shaped matrix (not the judged bench)."""
import argparse, json, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spmv_amd import api, synth, build

ap = argparse.ArgumentParser()
ap.add_argument("--m", type=int, default=10_000_000)
ap.add_argument("--k", type=int, default=32)
ap.add_argument("--dtype", default="f64")
ap.add_argument("--kind", default="banded")
ap.add_argument("--methods", default="1,2,3,4,5,0")
ap.add_argument("--lanes", default="0")
ap.add_argument("--variants", default="0")
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--alpha", type=float, default=1.6)
ap.add_argument("--maxlen", type=int, default=100000)
ap.add_argument("--hostptr", action="store_true")
ap.add_argument("--local", type=int, default=4096)
ap.add_argument("--sortcols", action="store_true")
ap.add_argument("--reorder", default="0")
a = ap.parse_args()
build.build()
dt = torch.float64 if a.dtype == "f64" else torch.float32
dev = "cuda:0"
t0 = time.time()
if a.kind == "banded":
    m, n, rp, ci, va = synth.banded_device(a.m, a.m, a.k, "uniform", dt, dev, 1)
elif a.kind == "scrambled":   # banded, then a random symmetric permutation of rows and columns
    m = n = a.m
    g = torch.Generator(device=dev); g.manual_seed(5)
    sc = torch.randperm(m, generator=g, device=dev)
    inv = torch.empty_like(sc); inv[sc] = torch.arange(m, device=dev)
    offs = torch.arange(a.k, device=dev) - a.k // 2
    ci = inv[(sc[:, None] + offs[None, :]) % n].reshape(-1).to(torch.int32)
    rp = torch.arange(0, (m + 1) * a.k, a.k, dtype=torch.int32, device=dev)
    va = torch.rand(m * a.k, generator=g, device=dev, dtype=dt) * 2 - 1
elif a.kind == "stencil27":   # 3-D 27-point stencil on an nx^3 grid (periodic): three far-apart bands
    nx = round(a.m ** (1 / 3)); m = n = nx ** 3
    offs = torch.tensor([dz * nx * nx + dy * nx + dx for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1)], device=dev)
    rp = torch.arange(0, (m + 1) * 27, 27, dtype=torch.int32, device=dev)
    ci = torch.empty(m * 27, dtype=torch.int32, device=dev)
    for r0 in range(0, m, 1 << 22):
        r1 = min(m, r0 + (1 << 22))
        rows = torch.arange(r0, r1, device=dev)
        ci[r0 * 27:r1 * 27] = ((rows[:, None] + offs[None, :]) % n).reshape(-1).to(torch.int32)
    va = torch.rand(m * 27, device=dev, dtype=dt) * 2 - 1
elif a.kind == "stencil5":    # 2-D 5-point stencil on an nx^2 periodic grid: 5 nnz per row, three bands
    nx = round(a.m ** 0.5); m = n = nx * nx
    offs = torch.tensor([-nx, -1, 0, 1, nx], device=dev)
    rp = torch.arange(0, (m + 1) * 5, 5, dtype=torch.int32, device=dev)
    rows = torch.arange(m, device=dev)
    ci = ((rows[:, None] + offs[None, :]) % n).reshape(-1).to(torch.int32)
    va = torch.rand(m * 5, device=dev, dtype=dt) * 2 - 1
elif a.kind == "random":
    m, n, rp, ci, va = synth.uniform_k_device(a.m, a.m, a.k, "uniform", dt, dev, 1)
elif a.kind == "skewed":
    m, n, rp, ci, va = synth.from_row_lengths_device(synth.skewed_lengths_device(a.m, dev, 1), a.m, "uniform", dt, dev, 1, local=a.local)
elif a.kind == "powerlaw":
    m, n, rp, ci, va = synth.from_row_lengths_device(synth.powerlaw_lengths_device(a.m, a.k, a.maxlen, a.alpha, dev, 1), a.m, "uniform", dt, dev, 1)
if a.sortcols:   # sort columns inside each row (key = row * n + col)
    lens = (rp[1:] - rp[:-1]).long()
    rows = torch.repeat_interleave(torch.arange(m, device=dev), lens)
    key = rows * n + ci.long()
    ci = (torch.sort(key).values % n).to(torch.int32)
    del rows, key
torch.cuda.synchronize()
x = torch.rand(n, dtype=dt, device=dev) * 2 - 1
y = torch.empty(m, dtype=dt, device=dev)
nnz = int(rp[-1].item())
print(f"# {a.kind} m={m} nnz={nnz} gen {time.time()-t0:.1f}s", flush=True)
yref = None
for meth in [int(s) for s in a.methods.split(",")]:
    for lanes in [int(s) for s in a.lanes.split(",")]:
      for var in [int(s) for s in a.variants.split(",")]:
       for reo in [int(s) for s in a.reorder.split(",")]:
        api.set_option("reorder", reo)
        api.set_option("lanes_per_row", lanes)
        api.set_option("vector_form", var)
        tcr = time.time()
        try:
            h = api.Handle(m, n, rp, ci, va, meth)
        except api.SpmvError as e:
            print("skip", meth, e); continue
        tcreate = time.time() - tcr
        info = h.info()
        y.fill_(float("nan"))
        mean, ms = api.time_launches(h.h, x, y, 5, a.iters)
        torch.cuda.synchronize()
        yc = y
        if h.index is not None:     # reordered handle: y came out permuted (and x was not gathered: timing only)
            yc = None
        if yref is None and yc is not None:
            yref = y.clone()
        err = (y - yref).abs().max().item() if (yc is not None and yref is not None) else float("nan")
        gb = info["alg_bytes"] / 1e9
        print(json.dumps(dict(method=api.SPMV_METHODS(meth).name, sched=info["schedule_name"], lanes=info["lanes_per_row"], variant=var,
              ms_mean=round(mean, 4), ms_min=round(float(ms.min()), 4), gbps_alg=round(float(gb / (ms.min() / 1e3)), 1),
              frac_8TBs=round(float(gb / (ms.min() / 1e3) / 8000), 3), gflops=round(float(2 * nnz / ms.min() / 1e6), 1),
              inspect_ms=round(info["inspect_ms"], 2), reorder=reo, tc0=round(tcreate, 2), kernel=info["kernel_name"], tuned=info["tuned_choice"], tune_ms=[round(v, 4) for v in info["tune_ms"]], stored=info["stored_nnz"], maxdiff_vs_first=err, nan=int(torch.isnan(y).sum().item()))), flush=True)
        if a.hostptr:
            xh, yh = x.cpu().numpy(), y.cpu().numpy().copy()
            api.set_stream(h.h, None, async_=False)
            t0 = time.time()
            for _ in range(5):
                api.spmv(h.h, m, rp, ci, va, xh, yh)
            print(json.dumps(dict(hostptr_ms=round((time.time() - t0) / 5 * 1e3, 3), ok=bool(abs(yh - y.cpu().numpy()).max() == 0))), flush=True)
        h.close()
