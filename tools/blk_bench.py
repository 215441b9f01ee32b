#!/usr/bin/env python3
"""Row-block x column-slab executor on the shapes without column locality: ms (min of 20), B_alg fraction, inspector time,
stored entries and the dense share, per option set.
    python tools/blk_bench.py 2r 3o 3o-uniform web24 [--sets "blk_groups=8;blk_waves=4,deterministic=0"]      (SPMV_LIB=<another build>: same-box A/B)"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from spmv_amd import api, build, synth
import run_config as rc
ap = argparse.ArgumentParser()
ap.add_argument("cfgs", nargs="*", default=["2r", "3o", "3o-uniform", "web24"])
ap.add_argument("--sets", default="", help="option sets separated by ';', each 'k=v,k=v'")
ap.add_argument("--method", type=int, default=4)
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
if os.environ.get("SPMV_LIB"):   # an A/B build (python -m spmv_amd.build --debug ...)
    api.LIB_PATH = os.environ["SPMV_LIB"]
else:
    build.build()
api.load()
dev = "cuda:0"
sets = [dict(kv.split("=") for kv in s.split(",") if kv) for s in (a.sets.split(";") if a.sets else [""])]
for cfg in a.cfgs:
    if cfg.startswith("web24"):   # 4e6 x 24 web-like, fp64 / fp32 (web24f)
        dt = torch.float32 if cfg.endswith("f") else torch.float64
        m = n = 4_000_000
        lens = torch.full((m,), 24, dtype=torch.int64, device=dev)
        _, _, rp, ci, va = synth.from_row_lengths_device(lens, m, "uniform", dt, dev, 1, cols="web")
    else:
        m, n, rp, ci, va = rc.make(cfg, dev)
    x = torch.rand(n, dtype=va.dtype, device=dev) * 2 - 1
    y = torch.empty(m, dtype=va.dtype, device=dev)
    want = None
    for opts in sets:
        keep = {k: api.get_option(k) for k in opts}
        for k, v in opts.items():
            api.set_option(k, int(v))
        h = api.Handle(m, n, rp, ci, va, a.method)
        info = h.info()
        mean, ms = api.time_launches(h.h, x, y, 5, a.iters)
        if want is None:   # definition in fp64, by segment sums (exactness not expected: report the scaled error)
            prod = va.double() * x.double()[ci.long()]
            cs = torch.cat([torch.zeros(1, dtype=torch.float64, device=dev), torch.cumsum(prod, 0)])
            want = cs[rp[1:].long()] - cs[rp[:-1].long()]
            ap_ = torch.cat([torch.zeros(1, dtype=torch.float64, device=dev), torch.cumsum(prod.abs(), 0)])
            scale = (ap_[rp[1:].long()] - ap_[rp[:-1].long()]).clamp_(min=1e-300)
            del prod, cs, ap_
        err = float(((y.double() - want).abs() / scale).max())
        t = float(ms.min())
        print("BLK " + json.dumps({"cfg": cfg, "opts": opts, "kernel": info["kernel_name"], "blocked": info["cache_blocked"], "ms_min": round(t, 4),
                                   "ms_mean": round(float(mean), 4), "frac_alg": round(info["alg_bytes"] / t / 1e6 / 8000, 4),
                                   "frac_moved": round(info["stream_bytes"] / t / 1e6 / 8000, 4), "inspect_ms": round(info["inspect_ms"], 1),
                                   "stored_over_nnz": round(info["stored_nnz"] / max(info["nnz"], 1), 4), "tuned": info["tuned_choice"],
                                   "tune_ms": [round(v, 4) for v in info["tune_ms"]], "split_ms": [round(v, 4) for v in info["split_ms"]],
                                   "far_share": round(info["far_nnz"] / max(info["nnz"], 1), 4), "rel_err": err}), flush=True)
        h.close()
        for k, v in keep.items():
            api.set_option(k, v)
    del rp, ci, va, x, y, want, scale
    torch.cuda.empty_cache()
