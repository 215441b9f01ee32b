#!/usr/bin/env python3
"""Measure every BASELINE.json config shape (and the stand-ins SURVEY 8d lists) with the schedule the
config names, plus the other schedules for comparison; write gpurun_out/configs_<tag>.json.
Run on the GPU box:  python tools/measure_configs.py r01   (then copy the file to profiles/)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from spmv_amd import api, build, synth  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
build.build()
dev = "cuda:0"
M = api.SPMV_METHODS
out = {"tag": tag, "device": torch.cuda.get_device_name(0), "note": "min over 20 launches after 5 warm-up, hipEvents on the launch stream; "
       "GB/s = B_alg / t with B_alg = 4(m+1) + nnz(4+s) + s n + s m; frac = GB/s / 8000", "configs": []}


def run(name, m, n, rp, ci, va, methods, iters=20):
    x = torch.rand(n, dtype=va.dtype, device=dev) * 2 - 1
    y = torch.empty(m, dtype=va.dtype, device=dev)
    rows = []
    for meth in methods:
        t0 = time.time()
        h = api.Handle(m, n, rp, ci, va, meth)
        create_s = time.time() - t0
        info = h.info()
        mean, ms = api.time_launches(h.h, x, y, 5, iters)
        h.close()
        gb = info["alg_bytes"] / 1e9
        rows.append({"method": M(meth).name, "schedule": info["schedule_name"], "kernel": info["kernel_name"],
                     "ms_min": round(float(ms.min()), 4), "ms_mean": round(float(mean), 4),
                     "gbps_alg": round(gb / (float(ms.min()) / 1e3), 1), "frac_of_8TBs": round(gb / (float(ms.min()) / 1e3) / 8000, 3),
                     "gflops": round(2 * info["nnz"] / float(ms.min()) / 1e6, 1), "create_s": round(create_s, 3),
                     "stored_over_nnz": round(info["stored_nnz"] / max(info["nnz"], 1), 3)})
        print(name, rows[-1], flush=True)
    out["configs"].append({"name": name, "m": m, "n": n, "nnz": int(rp[-1].item()), "dtype": str(va.dtype), "results": rows})


ALL = [M.Method_Parallel, M.Method_Balanced, M.Method_Balanced_Yid, M.Method_SellCSigma, M.Method_CSR5SPMV]
_, _, rp, ci, va = synth.banded_device(10_000_000, 10_000_000, 32, "uniform", torch.float64, dev, 1)
run("config 2: 1e7 x 1e7, 32 nnz/row banded, fp64 (named schedule: CSR-vector)", 10_000_000, 10_000_000, rp, ci, va, ALL)
_, _, rp, ci, va = synth.uniform_k_device(10_000_000, 10_000_000, 32, "uniform", torch.float64, dev, 1)
run("config 2 variant (ii): uniformly random columns", 10_000_000, 10_000_000, rp, ci, va, [M.Method_Balanced_Yid, M.Method_Balanced, M.Method_Parallel, M.Method_CSR5SPMV], 5)
lens = synth.powerlaw_lengths_device(1_000_000, 3.1, 4700, 1.6, dev, 1)
_, _, rp, ci, va = synth.from_row_lengths_device(lens, 1_000_000, "uniform", torch.float64, dev, 1)
run("config 3 stand-in webbase-1M-style (named schedule: Balanced2 nnz-split)", 1_000_000, 1_000_000, rp, ci, va,
    [M.Method_Balanced2, M.Method_CSR5SPMV, M.Method_Parallel, M.Method_SellCSigma])
lens = synth.powerlaw_lengths_device(3_070_000, 76, 33000, 1.5, dev, 1)
_, _, rp, ci, va = synth.from_row_lengths_device(lens, 3_070_000, "uniform", torch.float64, dev, 1)
run("config 3 stand-in com-Orkut-style (random columns)", 3_070_000, 3_070_000, rp, ci, va, [M.Method_Balanced2, M.Method_CSR5SPMV], 5)
lens = synth.skewed_lengths_device(10_000_000, dev, 1)
_, _, rp, ci, va = synth.from_row_lengths_device(lens, 10_000_000, "uniform", torch.float32, dev, 1, local=4096)
run("config 4: 1e7 rows skewed nnz, fp32, columns within +-4096 (named schedule: SELL C=64 sigma=1024)", 10_000_000, 10_000_000, rp, ci, va,
    [M.Method_SellCSigma, M.Method_CSR5SPMV, M.Method_Balanced2, M.Method_Parallel], 10)
_, _, rp, ci, va = synth.banded_device(10_000_000, 80_000_000, 32, "uniform", torch.float64, dev, 1, row0=30_000_000)
run("config 5 shard: 1e7 of 8e7 rows, global columns (one rank of the 8-GPU case, no exchange)", 10_000_000, 80_000_000, rp, ci, va,
    [M.Method_Parallel])
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", f"configs_{tag}.json"), "w") as f:
    json.dump(out, f, indent=1)
