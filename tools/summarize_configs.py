#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>/ (tools/profile_configs.sh) into tracked files under profiles/:

  profiles/<tag>_configs_pmc.json     per config: the run's own JSON (shape, kernel, model bytes, ms), rocprofv3 kernel stats
                                      (calls, avg / min ns) and per-launch FETCH_SIZE / WRITE_SIZE / TCC_HIT / TCC_MISS of every spmv::
                                      kernel of ONE multiply (SELL = slab kernel + long-row CSR5 kernel + carry fix-up), their sum
                                      ("multiply"), the corrected HBM bytes and the fractions of 8 TB/s that follow
  profiles/<tag>_<config>_kernel_stats.csv   the --stats table (our kernels + the top of the rest)
  profiles/<tag>_bench_kernel_stats.csv      the same for `python3 bench.py`
  profiles/traffic_<tag>.json         what bench.py reads for roofline.traffic: entries keyed by (kernel, m, nnz, dtype)

Counter units and corrections follow MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports
HALF the bytes of wide coalesced streaming reads, so hbm_bytes = 2 x FETCH_SIZE + WRITE_SIZE for STREAMING kernels (checked on
known-bytes stream kernels: ratio 1.992 / 2.000, profiles/r01_pmc.json and the calibration entry here).  For the gather-bound kernels
(64-byte line fetches of scattered x reads) the factor is not calibrated: both the raw and the doubled figure are given and the
doubled one is an upper bound."""
import csv
import glob
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


def find(base, suffix):
    """newest match: rocprofv3 names its files by process id, and gpurun MERGES a run's output into gpurun_out/ without
    deleting what an earlier run left there"""
    hits = glob.glob(os.path.join(base, "**", f"*{suffix}"), recursive=True)
    return max(hits, key=os.path.getmtime) if hits else None


def counters(base):
    """-> {kernel: {counter: [values per dispatch]}}"""
    path = find(base, "counter_collection.csv")
    out = {}
    if not path:
        return out
    with open(path) as f:
        for row in csv.DictReader(f):
            out.setdefault(row["Kernel_Name"], {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    return out


def short(name):
    n = name.split("(")[0]
    n = n.replace("void ", "").replace("spmv::", "").split("<")[0]
    return n


def stats_table(base, out_csv):
    path = find(base, "kernel_stats.csv")
    if not path:
        return []
    rows = list(csv.DictReader(open(path)))
    ours = [r for r in rows if "spmv::" in r["Name"]]
    rest = [r for r in rows if "spmv::" not in r["Name"]][:5]
    with open(out_csv, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=rows[0].keys())
        w.writeheader()
        for r in ours + rest:
            w.writerow(r)
    return [{"name": r["Name"][:110], "short": short(r["Name"]), "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
             "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"])} for r in ours]


STATS_ITERS, STATS_WARMUP = 20, 3   # tools/profile_configs.sh: run_config.py --iters 20 (warm-up 3) in the stats pass
MULTIPLY = ("csr_vector_tile_kernel", "csr_vector_pipe_kernel", "csr_vector_rows_kernel", "csr_scalar_kernel", "nat_group_kernel", "nat_kernel",
            "csr5_group_kernel", "csr5_group_pipe_kernel", "csr5_kernel", "csr5_fixup_kernel", "sell_window_kernel", "sell_kernel", "sell_fused_kernel", "blk_kernel", "blk_wide_kernel", "fill_zero_kernel")
summary = {"tag": tag, "units": "FETCH_SIZE / WRITE_SIZE in KiB (rocprofv3) -> bytes = KiB x 1024; hbm_bytes = 2 x FETCH + WRITE (gfx950 wide-read "
           "under-count, MI355X_MICROARCH.md 'HBM'); fractions are of 8.0 TB/s", "configs": {}}
entries = []
shas = set()
for log in sorted(glob.glob(os.path.join(src, "*.stats.log"))):
    cfg = os.path.basename(log)[: -len(".stats.log")]
    if cfg == "bench":
        continue
    run = None
    for line in open(log, errors="replace"):
        if line.startswith("RUNCONFIG "):
            run = json.loads(line[len("RUNCONFIG "):])
    if run is None:
        summary["configs"][cfg] = {"error": "run_config.py printed no result (see gpurun_out)"}
        continue
    ks = stats_table(os.path.join(src, cfg, "stats"), os.path.join(dst, f"{tag}_{cfg}_kernel_stats.csv"))
    fetch, write, hit = (counters(os.path.join(src, cfg, sub)) for sub in ("fetch", "write", "hit"))
    # The kernels of ONE multiply: those launched at least once per timed launch of the stats pass (20 timed + 3 warm-up;
    # autotune candidates and inspector kernels run fewer times).  Matched by their full (template-instantiated) name.
    n_mul = STATS_ITERS + STATS_WARMUP
    avg = lambda v: (sum(v) / len(v)) if v else None   # noqa: E731
    kernels = {}
    tot_ns = tot_bytes = tot_raw = 0.0
    for st in ks:
        if st["short"] not in MULTIPLY or st["calls"] < n_mul:
            continue
        name = next((nm for nm in set(fetch) | set(write) | set(hit) if nm[:110] == st["name"]), None)
        fv = fetch.get(name, {}).get("FETCH_SIZE", []) if name else []
        wv = write.get(name, {}).get("WRITE_SIZE", []) if name else []
        hv = hit.get(name, {}).get("TCC_HIT_sum", []) if name else []
        mv = hit.get(name, {}).get("TCC_MISS_sum", []) if name else []
        f_raw = avg(fv) * 1024.0 if fv else None
        w_raw = avg(wv) * 1024.0 if wv else None
        per_mul = max(1, st["calls"] // n_mul)   # create-time autotune launches the chosen form a few extra times
        k = {"full_name": st["name"], "calls": st["calls"], "launches_per_multiply": per_mul, "avg_ns": st["avg_ns"], "min_ns": st["min_ns"],
             "launches_counted": len(fv) or len(wv) or len(hv), "fetch_bytes_raw": f_raw, "write_bytes": w_raw,
             "hbm_bytes_per_launch": (2.0 * f_raw if f_raw is not None else 0.0) + (w_raw or 0.0),
             "hbm_bytes_per_launch_raw_fetch": (f_raw or 0.0) + (w_raw or 0.0),
             "tcc_hit": avg(hv), "tcc_miss": avg(mv),
             "l2_hit_rate": (avg(hv) / (avg(hv) + avg(mv))) if hv and mv and (avg(hv) + avg(mv)) > 0 else None}
        k["hbm_gbps_at_avg"] = k["hbm_bytes_per_launch"] / st["avg_ns"]
        k["frac_of_8TBs_at_avg"] = k["hbm_bytes_per_launch"] / st["avg_ns"] / 8000.0
        kernels[st["short"] + ("" if st["short"] not in kernels else "#" + str(len(kernels)))] = k
        tot_ns += st["avg_ns"] * per_mul
        tot_bytes += k["hbm_bytes_per_launch"] * per_mul
        tot_raw += k["hbm_bytes_per_launch_raw_fetch"] * per_mul
    dom = run["kernel"]
    d = kernels.get(dom)
    cfg_out = {"run": run, "kernels": kernels}
    if tot_ns > 0:
        cfg_out["multiply"] = {
            "kernels": sorted(kernels), "ms_rocprof_sum_of_kernels": tot_ns / 1e6, "ms_hip_events_min": run["ms_min"],
            "model_stream_bytes": run["stream_bytes"], "alg_bytes": run["alg_bytes"],
            "pmc_hbm_bytes_2xfetch_plus_write": tot_bytes, "pmc_hbm_bytes_1xfetch_plus_write": tot_raw,
            "model_over_pmc": run["stream_bytes"] / tot_bytes if tot_bytes else None,
            "frac_pmc_bytes": tot_bytes / tot_ns / 8000.0, "frac_model_bytes": run["stream_bytes"] / tot_ns / 8000.0,
            "frac_alg_bytes": run["alg_bytes"] / tot_ns / 8000.0,
            "l2_hit_rate_dominant": d["l2_hit_rate"] if d else None}
    if d and tot_bytes > 0:   # what bench.py reads: the whole multiply (every kernel of one spmv(), summed) and its parts
        shas.add(run.get("csrc_sha"))
        entries.append({"config": cfg, "kernel_short": dom, "m": run["m"], "nnz": run["nnz"], "dtype": run["dtype"], "model_stream_bytes": run["stream_bytes"],
                        "hbm_bytes_per_launch": tot_bytes, "fetch_bytes_raw": d["fetch_bytes_raw"], "write_bytes": d["write_bytes"],
                        "kernels": [{"kernel": kn.split("#")[0], "avg_ms": kv["avg_ns"] * kv["launches_per_multiply"] / 1e6, "hbm_bytes": kv["hbm_bytes_per_launch"] * kv["launches_per_multiply"]}
                                    for kn, kv in kernels.items()],
                        "ms_rocprof_sum_of_kernels": tot_ns / 1e6, "csrc_sha": run.get("csrc_sha"),
                        "source": f"profiles/{tag}_configs_pmc.json#{cfg}"})
    summary["configs"][cfg] = cfg_out

bench_ks = stats_table(os.path.join(src, "bench", "stats"), os.path.join(dst, f"{tag}_bench_kernel_stats.csv"))
if bench_ks:
    summary["bench_py_kernel_stats"] = bench_ks[:8]
    for line in open(os.path.join(src, "bench.stats.log"), errors="replace"):
        if line.startswith("{") and '"metric"' in line:
            summary["bench_py_line_under_profiler"] = json.loads(line)
cal = counters(os.path.join(src, "calib", "fetch"))
known = None
try:
    for line in open(os.path.join(src, "calib.log"), errors="replace"):
        if line.startswith("CALIB ") and "bytes=" in line:
            known = float(line.split("bytes=")[1].split()[0])
except OSError:
    pass
for name, c in cal.items():
    vals = c.get("FETCH_SIZE", [])
    if vals and known and "calib_read" in name:
        raw = sum(vals) / len(vals) * 1024.0
        summary["calibration"] = {"kernel": name[:90], "known_bytes": known, "fetch_size_bytes_raw": raw, "ratio_known_over_raw": known / raw,
                                  "note": "16-byte-per-lane nt loads of a 3.75 GiB array (tools/gbench.hip calib)"}
    elif vals and "stream_read<true>" in name:   # round-1 harness (tools/kbench.hip, removed in round 2)
        raw = sum(vals) / len(vals) * 1024.0
        summary["calibration"] = {"kernel": name[:90], "known_bytes": 10_000_000 * 32 * 12.0, "fetch_size_bytes_raw": raw,
                                  "ratio_known_over_raw": 10_000_000 * 32 * 12.0 / raw}
with open(os.path.join(dst, f"{tag}_configs_pmc.json"), "w") as f:
    json.dump(summary, f, indent=1)
with open(os.path.join(dst, f"traffic_{tag}.json"), "w") as f:
    json.dump({"tag": tag, "correction": "2 x FETCH_SIZE (gfx950 wide-read under-count) + WRITE_SIZE, KiB -> bytes; separate --pmc passes",
               "csrc_sha": (shas.pop() if len(shas) == 1 else None),   # the source tree the passes ran on (spmv_amd/srchash.py); bench.py ignores the file when its tree differs
               "entries": entries}, f, indent=1)
for cfg, c in summary["configs"].items():
    print(cfg, json.dumps(c.get("multiply", c.get("error")), indent=None)[:700])
