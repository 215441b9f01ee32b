#!/usr/bin/env python3
"""Measure every BASELINE.json config shape (and the stand-ins SURVEY 8d lists) with the schedule the config names,
plus the other schedules for comparison; write gpurun_out/configs_<tag>.json (copy it to profiles/).

    python tools/measure_configs.py r02 [config ...]

Per (config, method): min / mean launch time over 20 launches after 5 warm-ups (hipEvents on the launch stream), create
time, and three byte counts over the min time as fractions of 8 TB/s:
  frac_moved   spmv_hip_info.stream_bytes -- what the schedule's storage format makes one launch move (physical)
  frac_alg     SURVEY 8d's algorithmic bytes B_alg = 4(m+1) + nnz(4+s) + s n + s m (effective rate; can exceed the
               physical one when a 2 B/nnz slot stream replaces the 4 B/nnz ColIdx)
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
from spmv_amd import api, build  # noqa: E402
import run_config as rc  # noqa: E402

M = api.SPMV_METHODS
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
wanted = sys.argv[2:] or ["2", "2h", "s27", "2r", "3w", "3w-uniform", "3w-web", "3o", "3o-uniform", "4", "5shard"]
METHODS = {
    "2": [M.Method_Parallel, M.Method_Balanced, M.Method_Balanced_Yid, M.Method_SellCSigma, M.Method_CSR5SPMV],
    "2h": [M.Method_Parallel, M.Method_Balanced, M.Method_Balanced_Yid, M.Method_SellCSigma, M.Method_CSR5SPMV],
    "s27": [M.Method_Parallel, M.Method_Balanced, M.Method_Balanced_Yid, M.Method_SellCSigma, M.Method_CSR5SPMV],
    "2r": [M.Method_Parallel, M.Method_Balanced2, M.Method_CSR5SPMV, M.Method_SellCSigma],
    "3w": [M.Method_Balanced2, M.Method_CSR5SPMV, M.Method_Parallel, M.Method_SellCSigma],
    "3w-uniform": [M.Method_Balanced2, M.Method_CSR5SPMV, M.Method_Parallel],
    "3w-web": [M.Method_Balanced2, M.Method_CSR5SPMV, M.Method_Parallel, M.Method_SellCSigma],
    "3o": [M.Method_Balanced2, M.Method_CSR5SPMV, M.Method_Parallel],
    "3o-uniform": [M.Method_Balanced2, M.Method_CSR5SPMV],
    "4": [M.Method_SellCSigma, M.Method_CSR5SPMV, M.Method_Balanced2, M.Method_Parallel],
    "5shard": [M.Method_Parallel],
}
build.build()
api.load()
dev = "cuda:0"
out = {"tag": tag, "device": torch.cuda.get_device_name(0),
       "note": "min over 20 launches after 5 warm-up, hipEvents on the launch stream; frac_moved = stream_bytes / t / 8 TB/s (physical), "
               "frac_alg = B_alg / t / 8 TB/s (SURVEY 8d algorithmic bytes, effective)", "configs": []}
for cfg in wanted:
    name, _ = rc.CONFIGS[cfg]
    m, n, rp, ci, va = rc.make(cfg, dev)
    x = torch.rand(n, dtype=va.dtype, device=dev) * 2 - 1
    y = torch.empty(m, dtype=va.dtype, device=dev)
    rows = []
    runs = [(meth, 1) for meth in METHODS[cfg]]
    if cfg in ("2r", "3o", "3o-uniform"):   # the blocked executor with reproducibility waived (wide form, arrival order)
        runs.append((METHODS[cfg][0], 0))
    for meth, det in runs:
        api.set_option("deterministic", det)
        t0 = time.time()
        h = api.Handle(m, n, rp, ci, va, meth)
        create_s = time.time() - t0
        info = h.info()
        mean, ms = api.time_launches(h.h, x, y, 5, 20)
        used = h.method.name
        h.close()
        t = float(ms.min()) / 1e3
        api.set_option("deterministic", 1)
        rows.append({"method": M(meth).name, "deterministic": det, "blk_waves": info["blk_waves"], "reproducible": info["reproducible"],
                     "launch_kernels": info["launch_kernels"], "run_nnz": info["run_nnz"], "tmpl_nnz": info["tmpl_nnz"], "byte_nnz": info["byte_nnz"],
                     "device_bytes": info["device_bytes"], "method_used": used, "schedule": info["schedule_name"], "kernel": info["kernel_name"],
                     "cache_blocked": info["cache_blocked"], "ms_min": round(t * 1e3, 4), "ms_mean": round(float(mean), 4),
                     "gflops": round(2 * info["nnz"] / t / 1e9, 1),
                     "stream_bytes": info["stream_bytes"], "alg_bytes": info["alg_bytes"],
                     "frac_moved": round(info["stream_bytes"] / t / 8e12, 4), "frac_alg": round(info["alg_bytes"] / t / 8e12, 4),
                     "create_s": round(create_s, 3), "inspect_ms": round(info["inspect_ms"], 1),
                     "stored_over_nnz": round(info["stored_nnz"] / max(info["nnz"], 1), 3)})
        print(cfg, rows[-1], flush=True)
    out["configs"].append({"config": cfg, "name": name, "m": m, "n": n, "nnz": int(rp[-1].item()), "dtype": str(va.dtype), "results": rows})
    del rp, ci, va, x, y
    torch.cuda.empty_cache()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", f"configs_{tag}.json"), "w") as f:
    json.dump(out, f, indent=1)
