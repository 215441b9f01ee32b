#!/usr/bin/env python3
"""bench.py -- the reference's headline benchmark on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

`--gpus N` with N > 1 and no WORLD_SIZE in the environment: this process -- before it touches any GPU --
starts N rank processes itself (one per device, LOCAL_RANK = device) and relays rank 0's JSON line; it fails
loudly when the box has fewer than N devices.

A "step" is one pass of the hot path: y = A x through the C ABI (spmv(), include/spmv.h) on a
matrix that is already resident in HBM; x and y are device-resident (DESIGN.md gives the
PCIe-inclusive host-pointer figure, it is never `value`).

Workload at N = 1 = BASELINE config 2: CSR fp64, 1e7 x 1e7, exactly 32 nnz/row, banded
(SURVEY 8d variant (i)), schedule Method_Parallel = CSR-vector.  At N > 1 = config 5 scaled
weakly: every rank holds a 1e7-row block of the (N*1e7)^2 banded matrix (local int32 RowPtr,
global columns) and the matching slice of x; each step first exchanges x over RCCL
(spmv_amd.dist) and then multiplies; y stays distributed.  All three exchanges are run and reported
in the one JSON line ("exchanges"): "halo" (the library's default: only the referenced entries move,
point to point; `value` is this one), "allgather" (every rank gathers the whole x each step: the
solver-style loop) and "bcast" (north_star's literal broadcast of x from rank 0), each with the
time of the exchange alone and of the multiply alone.

Timing follows the driver contract: W untimed steps, then exactly K steps between
barrier + torch.cuda.synchronize() pairs, MAX over ranks, rank 0 prints ONE JSON line.
`value` = 2 * nnz(all ranks) * K / time (FLOPs per SpMV = 2 nnz: test_spmv.c:126).

roofline (the dominant kernel; measured live): HIP events on the stream the kernel is launched on
bracket the launches (N = 1: the K timed steps themselves; N > 1: a multiply-only pass after them).  Three fractions of the
8 TB/s HBM peak are reported side by side, over the same mean launch time:
  frac_alg     = SURVEY 8d's ALGORITHMIC bytes B_alg = 4(m+1) + nnz(4+s) + s n + s m ("achieved_alg").  The accounting figure of the
                 metric; an EFFECTIVE rate -- it charges 4 B/nnz of ColIdx, which the staged schedules replace by a 2 B (BYTE tiles: 1 B,
                 RUN tiles: 2 B per ROW) LDS-slot stream, so it can exceed 1;
  frac_model   = bytes the schedule's storage format has to move (spmv_hip_info.stream_bytes: streams as stored, the x elements
                 staged, y, window tables), the library's own traffic model;
  frac_counter = HBM bytes per launch measured by rocprofv3 (2 x FETCH_SIZE + WRITE_SIZE, separate --pmc passes,
                 MI355X_MICROARCH.md "HBM"), from the committed profiles/traffic_rNN.json -- used only when that file was measured on
                 exactly this kernel, shape, dtype AND source tree (csrc hash), else null.
  frac / achieved = frac_counter when there is one, else frac_model ("frac_source" says which): the physical HBM rate.
  traffic      = the counter bytes (or null);  read_calibration = a pure 16 B/lane read of the same byte count, same box, same run
                 (spmv_amd/bin/gbench read): the ceiling the rates sit under.
`cpu_baseline` (rank 0, N = 1): the real reference's OpenMP path (oracle/_ref, Method_Parallel,
kind "reference") or, if that library is absent, the oracle's OpenMP port, on the first
--cpu-rows rows of the same matrix, with the GPU result checked against it.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_PROC_BIND", "true")   # unbound Method_Parallel is 4x slower
os.environ.setdefault("OMP_PLACES", "cores")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md "HBM3E peak BW 8.0 TB/s spec"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)   # test_spmv.c:103-124: 10 warm + 100 timed
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--rows", type=int, default=10_000_000, help="rows per GPU")
    ap.add_argument("--nnz-per-row", type=int, default=32)
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--workload", default="banded", choices=["banded", "random", "powerlaw"],
                    help="headline matrix; powerlaw (N > 1): heavy-tailed rows with R-MAT columns, rows cut into equal-nnz blocks")
    ap.add_argument("--method", type=int, default=1, help="SPMV_METHODS id (1 = Method_Parallel = CSR-vector)")
    ap.add_argument("--xchg", default="all", choices=["all", "halo", "allgather", "bcast", "none"],
                    help="N > 1: which x exchange(s) to run; 'all' = halo (headline), allgather, bcast")
    ap.add_argument("--cpu-rows", type=int, default=1_000_000)
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU baseline: keep timing calls for about this long (at least 100 calls)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="N = 1: skip the other BASELINE configs and the C-level multi-GPU step")
    ap.add_argument("--config-iters", type=int, default=20, help="timed launches per extra config")
    ap.add_argument("--no-overlap", action="store_true", help="halo mode: do not split interior / boundary rows")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the judged path); gloo + SPMV_BENCH_ONE_DEVICE=1 rehearses N>1 on a 1-GPU box")
    ap.add_argument("--scaling", default="both", choices=["weak", "strong", "both"],
                    help="N > 1: weak = --rows per GPU (the headline `value`), strong = --rows in total split over the N ranks; both: the strong leg is "
                         "reported under 'strong' in the same line")
    ap.add_argument("--no-c-leg", action="store_true", help="N > 1: skip the C-level multi-GPU host (one process, N devices) that rank 0 starts as a child after the ranks are done")
    ap.add_argument("--c-leg", action="store_true", help=argparse.SUPPRESS)   # the child process of that leg
    ap.add_argument("--c-leg-timeout", type=float, default=240.0)
    return ap.parse_args()


# ---------------------------------------------------------------------------------------- launcher
def spawn_ranks(args):
    """--gpus N without a launcher: start the N ranks from here.  Nothing in this process has touched a GPU
    (torch.cuda.device_count() does not initialise one on this image), and the ranks are fresh child
    processes -- never an exec of a process that holds a GPU context."""
    import torch
    n = args.gpus
    one_device = bool(os.environ.get("SPMV_BENCH_ONE_DEVICE"))
    ndev = torch.cuda.device_count()
    if ndev < n and not one_device:
        sys.stderr.write(f"bench.py: --gpus {n} but this box exposes {ndev} GPU(s); refusing to run a {n}-GPU benchmark on fewer devices "
                         "(SPMV_BENCH_ONE_DEVICE=1 --backend gloo rehearses the multi-rank path on one device)\n")
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out, _ = procs[0].communicate()
    rc = procs[0].returncode
    for p in procs[1:]:
        rc = max(rc, abs(p.wait()))
    sys.stdout.write(out)
    sys.stdout.flush()
    return rc


# ---------------------------------------------------------------------------------------- CPU baseline
def cpu_baseline(args, rp, ci, va, x, y_gpu, n_cols):
    """Time the reference's OpenMP CSR path on a bounded sample of the same workload (host cores
    of this box) and check the GPU result against it.  The oracle package is used here only as the
    thing being timed/checked against -- never by the product path."""
    import ctypes as C
    import numpy as np
    import oracle
    rows = min(args.cpu_rows, rp.numel() - 1)
    p1 = int(rp[rows].item())
    from spmv_amd.synth import CSR
    csr = CSR(rows, n_cols, rp[: rows + 1].cpu().numpy(), ci[:p1].cpu().numpy(), va[:p1].cpu().numpy())
    xh = x.cpu().numpy()
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        gomp = C.CDLL("libgomp.so.1")
        gomp.omp_set_num_threads(int(cores))
    except OSError:
        pass
    kind = "reference" if oracle.have_ref() else "port"
    I = C.POINTER(C.c_int)
    y = np.full(rows, np.nan, dtype=csr.val.dtype)
    if kind == "reference":
        L = oracle.ref_lib()
        h = C.POINTER(oracle.RefHandle)()
        L.spmv_create_handle_all_in_one(C.byref(h), rows, n_cols, csr.rowptr.ctypes.data_as(I), csr.colidx.ctypes.data_as(I),
                                        csr.val.ctypes.data_as(C.c_void_p), cores, 1, csr.val.dtype.itemsize, 0, None)
        def run():
            L.spmv(h, rows, csr.rowptr.ctypes.data_as(I), csr.colidx.ctypes.data_as(I), csr.val.ctypes.data_as(C.c_void_p),
                   xh.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p))
    else:
        L = oracle.lib()
        def run():
            L.oracle_spmv_omp(rows, csr.rowptr.ctypes.data_as(I), csr.colidx.ctypes.data_as(I), csr.val.ctypes.data_as(C.c_void_p),
                              xh.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), csr.val.dtype.itemsize)
    for _ in range(3):
        run()
    times = []
    t_end = time.perf_counter() + args.cpu_seconds
    while len(times) < 4000 and (time.perf_counter() < t_end or len(times) < 100):   # >= the harness' 100 calls, ~cpu_seconds of work
        t0 = time.perf_counter()
        run()
        times.append(time.perf_counter() - t0)
    if kind == "reference":
        L.spmv_destory_handle(h)
    mean = float(np.mean(times))
    # parity of the GPU result on the sample rows (tolerance of north_star, scaled per row)
    s = oracle.row_abs_sum(csr, xh)
    err = np.abs(y_gpu[:rows].cpu().numpy().astype(np.float64) - y.astype(np.float64))
    tol = 1e-6 if csr.val.dtype == np.float64 else 1e-3
    rel = float((err / np.maximum(s, 1e-300)).max())
    return {
        "value": round(2.0 * p1 / mean / 1e9, 3), "unit": "GFLOP/s", "cores": int(cores), "kind": kind,
        "sample": f"first {rows} rows ({p1} nnz) of the same matrix, {len(times)} timed calls of "
                  f"{'reference Method_Parallel' if kind == 'reference' else 'oracle OpenMP row loop'}, mean {mean * 1e3:.2f} ms",
        "ms": round(mean * 1e3, 3), "gpu_vs_cpu_max_rel_err": rel, "parity_ok": bool(rel <= tol),
    }



# ---------------------------------------------------------------------------------------- the other BASELINE configs
def definition(rp, ci, va, x):
    """y = A x by segment sums in fp64 (torch), and the per-row sum of |a x| the tolerance scales with."""
    import torch
    prod = va.double() * x.double()[ci.long()]
    z = torch.zeros(1, dtype=torch.float64, device=va.device)
    cs = torch.cat([z, torch.cumsum(prod, 0)])
    ca = torch.cat([z, torch.cumsum(prod.abs(), 0)])
    r0, r1 = rp[:-1].long(), rp[1:].long()
    return cs[r1] - cs[r0], ca[r1] - ca[r0]


IC_BYTES, L2_BYTES = 256 << 20, 32 << 20
IC_READ_GBPS, L2_READ_GBPS = 8600.0, 34500.0   # MI355X_MICROARCH.md: tables resident in the Infinity Cache read at 8.6 TB/s chip-wide; L2 ~34.5 TB/s


def roofline_fields(info, ms, dtype, nnz=None, alg_bytes=None):
    """The three fractions of the HBM peak over one mean launch time (module docstring) + the cache-level bound for shapes that never leave the caches."""
    alg = int(alg_bytes if alg_bytes is not None else info["alg_bytes"])
    moved = int(info["stream_bytes"])
    sec = float(ms) * 1e-3
    traffic, src, parts = traffic_from_profiles(info["kernel_name"], info["m"], info["nnz"], dtype, moved)
    frac_alg = alg / sec / 1e9 / HBM_PEAK_GBPS
    frac_model = moved / sec / 1e9 / HBM_PEAK_GBPS
    frac_counter = (traffic / sec / 1e9 / HBM_PEAK_GBPS) if traffic else None
    frac = frac_counter if frac_counter is not None else frac_model
    out = {
        "bound": "hbm", "achieved": round(frac * HBM_PEAK_GBPS, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(frac, 4),
        "frac_source": "counter" if frac_counter is not None else "model",
        "frac_counter": None if frac_counter is None else round(frac_counter, 4), "frac_model": round(frac_model, 4), "frac_alg": round(frac_alg, 4),
        "achieved_alg": round(frac_alg * HBM_PEAK_GBPS, 1), "traffic": traffic, "traffic_source": src,
        "bytes_moved_per_launch": moved, "alg_bytes_per_launch": alg, "x_bytes_per_launch": int(info["x_bytes"]),
        "kernel": info["kernel_name"], "kernels": kernel_shares(info["launch_kernels"], parts),
        "column_stream_bytes_per_nnz": column_stream_bytes(info),
    }
    if moved < IC_BYTES:   # a problem that stays in the caches between launches: HBM is the wrong bar, say which level is the right one
        level, peak = ("l2", L2_READ_GBPS) if moved < L2_BYTES else ("infinity_cache", IC_READ_GBPS)
        out["cache_resident"] = {"level": level, "peak": peak, "unit": "GB/s", "frac": round(moved / sec / 1e9 / peak, 4),
                                 "note": f"one multiply moves {moved / 2**20:.0f} MiB < the {'32 MiB of L2' if level == 'l2' else '256 MiB Infinity Cache'}: "
                                         "launch to launch it is served on-die (MI355X_MICROARCH.md), so the HBM fractions understate nothing and bound nothing"}
    return out


def column_stream_bytes(info):
    """Bytes of column information the multiply reads per non-zero: 4 (int32 ColIdx), 2 (16-bit LDS slots), 1 (BYTE tiles), ~0 (RUN tiles: 2 B per row)."""
    nnz = max(int(info["nnz"]), 1)
    if info["cache_blocked"]:
        return 4.0
    staged = info["x_groups_staged"] / max(info["x_groups"], 1) if info["x_groups"] else 0.0
    run, byte = int(info.get("run_nnz", 0)) + int(info.get("tmpl_nnz", 0)), int(info.get("byte_nnz", 0))
    rest = max(nnz - run - byte, 0)
    return round((run * 0.0 + byte * 1.0 + rest * (2.0 * staged + 4.0 * (1.0 - staged))) / nnz, 3)


def kernel_shares(names, parts):
    """Every kernel of one multiply (live, spmv_hip_info.launch_kernels) with its rocprofv3 average and share when the committed profile matches."""
    out = []
    tot = sum(p["avg_ms"] for p in parts) if parts else 0.0
    for n in names:
        e = {"kernel": n}
        for p in parts or []:
            if p["kernel"] == n:
                e.update(avg_ms_rocprof=round(p["avg_ms"], 5), share=round(p["avg_ms"] / tot, 3) if tot else None, hbm_bytes=p.get("hbm_bytes"))
        out.append(e)
    return out


def extra_configs(args, dev):
    """BASELINE configs 2-ii, 2 with holes, a 27-point stencil, 3 (both stand-ins) and 4 under their named schedules, on this run's clock (the reference's harness
    times every method on the matrix it is given, test_spmv.c:103-127, 238-244).  Sizes scale with --rows / 1e7 so that a small
    --rows run stays small.  Per config: >= --config-iters launches timed one by one with HIP events on the launch stream
    (spmv_hip_time_launches), min and mean; the three fractions of roofline_fields; every kernel of one multiply; parity against the fp64 torch
    evaluation of the definition, tolerance of north_star scaled by the row's sum of |a x|.  Configs whose multiply runs the row-block x
    column-slab executor are measured twice: as the library does it by default (bit-reproducible) and with option deterministic = 0."""
    import torch
    from spmv_amd import api, synth
    M = api.SPMV_METHODS
    scale = args.rows / 1e7
    f64, f32 = torch.float64, torch.float32

    def cfg_2ii():
        m = max(1024, int(10_000_000 * scale))
        return synth.uniform_k_device(m, m, 32, "uniform", f64, dev, 1)

    def cfg_2h():
        m = max(1024, int(10_000_000 * scale))
        return synth.banded_holes_device(m, m, 32, 0.25, "uniform", f64, dev, 1)

    def cfg_s27():
        nx = max(16, int(round(215 * scale ** (1.0 / 3.0))))
        return synth.stencil27_device(nx, "uniform", f64, dev, 1)

    def cfg_3o():
        m = max(1024, int(3_070_000 * scale))
        lens = synth.powerlaw_lengths_device(m, 76, min(33000, m), 1.5, dev, 1)
        return synth.from_row_lengths_device(lens, m, "uniform", f64, dev, 1, cols="rmat")

    def cfg_3w():
        m = max(1024, int(1_000_000 * scale))
        lens = synth.powerlaw_lengths_device(m, 3.1, min(4700, m), 1.6, dev, 1)
        return synth.from_row_lengths_device(lens, m, "uniform", f64, dev, 1, cols="rmat")

    def cfg_4():
        m = max(1024, int(10_000_000 * scale))
        lens = synth.skewed_lengths_device(m, dev, 1)
        return synth.from_row_lengths_device(lens, m, "uniform", f32, dev, 1, local=4096)

    table = [
        ("2-ii", "config 2 variant (ii): uniformly random columns, 32 nnz/row, fp64", M.Method_Parallel, cfg_2ii),
        ("2-holes", "config 2 with holes: 32 of the 43 columns of a band per row (mean run length 4: no row is one run of consecutive columns), fp64", M.Method_Parallel, cfg_2h),
        ("stencil27", "27-point stencil on a periodic 215^3 grid (9.9e6 rows, 2.7e8 nnz: nine runs of three per row in three far-apart bands), fp64 -- the other "
                      "regular shape beside config 2: rows are not runs, but every row uses the same slot offsets (TEMPLATE tiles)", M.Method_Parallel, cfg_s27),
        ("3-orkut-style", "config 3 stand-in com-Orkut-style: power-law rows (mean 76, max 33 k), R-MAT columns, fp64", M.Method_Balanced2, cfg_3o),
        ("3-webbase-style", "config 3 stand-in webbase-1M-style: power-law rows (mean 3.1, max 4.7 k), R-MAT columns, fp64", M.Method_Balanced2, cfg_3w),
        ("4", "config 4: skewed rows (90 % 8-24, 9 % 64-256, 1 % 1 k-4 k), columns within +-4096, fp32, SELL C=64 sigma=1024", M.Method_SellCSigma, cfg_4),
    ]

    def measure(m, n, rp, ci, va, method, x, want, scale_row, dtype):
        y = torch.full((m,), float("nan"), dtype=va.dtype, device=dev)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        h = api.Handle(m, n, rp, ci, va, method)
        create_s = time.perf_counter() - t1
        info = h.info()
        mean, ms = api.time_launches(h.h, x, y, 3, max(args.config_iters, 1))
        retimed = 0
        if mean > 1.5 * float(ms.min()):     # a launch or two took many times the others (seen once: one 30 ms launch among twenty of 0.5 ms): time the set again, keep the calmer one, say so
            mean2, ms2 = api.time_launches(h.h, x, y, 3, max(args.config_iters, 1))
            retimed = 1
            if mean2 < mean:
                mean, ms = mean2, ms2
        used = h.method.name
        h.close()
        tol = 1e-6 if dtype == "f64" else 1e-3
        bad = ~((y.double() - want).abs() <= tol * scale_row + 1e-300)
        rel = float(((y.double() - want).abs() / scale_row.clamp(min=1e-300)).max())
        rf = roofline_fields(info, float(mean), dtype)
        r = {"method_used": used, "schedule": info["schedule_name"], "kernel": info["kernel_name"], "kernels": rf["kernels"],
             "cache_blocked": int(info["cache_blocked"]), "blk_waves": int(info["blk_waves"]), "reproducible": bool(info["reproducible"]),
             "stored_nnz": int(info["stored_nnz"]), "launches": int(ms.size), "ms_min": round(float(ms.min()), 5), "ms_mean": round(float(mean), 5), "ms_max": round(float(ms.max()), 5), "retimed": retimed,
             "gflops": round(2.0 * info["nnz"] / (float(mean) * 1e-3) / 1e9, 1),
             "bytes_moved_per_launch": rf["bytes_moved_per_launch"], "alg_bytes_per_launch": rf["alg_bytes_per_launch"],
             "frac": rf["frac"], "frac_source": rf["frac_source"], "frac_counter": rf["frac_counter"], "frac_model": rf["frac_model"], "frac_alg": rf["frac_alg"],
             "traffic": rf["traffic"], "traffic_source": rf["traffic_source"], "column_stream_bytes_per_nnz": rf["column_stream_bytes_per_nnz"],
             "parity_ok": bool(not bad.any()), "max_rel_err": rel, "rows_unwritten": int(torch.isnan(y).sum()),
             "inspect_ms": round(info["inspect_ms"], 2), "create_s": round(create_s, 3), "device_bytes": int(info["device_bytes"])}
        if "cache_resident" in rf:
            r["cache_resident"] = rf["cache_resident"]
        return r, info

    out = {}
    for key, name, method, make in table:
        t0 = time.perf_counter()
        m, n, rp, ci, va = make()
        dtype = "f64" if va.dtype == f64 else "f32"
        g = torch.Generator(device=dev)
        g.manual_seed(77)
        x = torch.rand(n, generator=g, device=dev, dtype=va.dtype) * 2 - 1
        want, scale_row = definition(rp, ci, va, x)
        r, info = measure(m, n, rp, ci, va, method, x, want, scale_row, dtype)
        out[key] = {"workload": name, "method": method.name, "m": m, "n": n, "nnz": int(info["nnz"]), "dtype": dtype, **r}
        if info["cache_blocked"]:   # the same matrix with reproducibility waived: the wide form's waves add in arrival order
            api.set_thread_option("deterministic", 0)
            try:
                r0, _ = measure(m, n, rp, ci, va, method, x, want, scale_row, dtype)
            finally:
                api.clear_thread_options()
            out[key]["deterministic_0"] = {k: r0[k] for k in ("kernel", "blk_waves", "reproducible", "ms_min", "ms_mean", "gflops", "frac", "frac_source", "frac_model", "frac_alg",
                                                               "parity_ok", "max_rel_err", "inspect_ms")}
        out[key]["total_s"] = round(time.perf_counter() - t0, 2)
        del rp, ci, va, x, want, scale_row
        torch.cuda.empty_cache()
    return out


def read_calibration(nbytes):
    """A pure 16-byte-per-lane in-order read of `nbytes` on this box, now (spmv_amd/bin/gbench read, a child process): the ceiling the
    multiply's rates sit under.  None if the tool is not built."""
    exe = os.path.join(ROOT, "spmv_amd", "bin", "gbench")
    if not os.path.exists(exe):
        return None
    try:
        r = subprocess.run([exe, "read", str(int(nbytes)), "10"], capture_output=True, text=True, timeout=120)
        line = next(l for l in r.stdout.splitlines() if l.startswith("{"))
        d = json.loads(line)
        d["frac_of_peak"] = round(d["read_gbps"] / HBM_PEAK_GBPS, 4)
        d["note"] = "min of 10 launches of tools/gbench.hip calib_read over the same byte count as bytes_moved_per_launch (no stores, no gathers)"
        return d
    except Exception as e:      # noqa: BLE001 -- a missing calibration must not cost the line
        return {"error": f"{type(e).__name__}: {str(e)[:100]}"}


def multi_step_leg(args, dev, rp, ci, va, n, x_full, y_ref):
    """The C-level multi-GPU handle (option "gpus", csrc/shim/multi.hpp) on the headline matrix at G = devices present: the
    distributed step spmv_hip_multi_step_async / _synchronize (x in the devices' slices, "range" exchange beside the multiply),
    timed over --config-iters steps between device synchronisations."""
    import torch
    from spmv_amd import api
    G = torch.cuda.device_count()
    api.set_thread_option("gpus", G)
    api.set_thread_option("x_exchange", 1)
    try:
        t0 = time.perf_counter()
        h = api.Handle(int(rp.numel() - 1), n, rp, ci, va, args.method)
        create_s = time.perf_counter() - t0
    finally:
        api.clear_thread_options()
    try:
        return _time_multi_handle(args, h, va.dtype, va.element_size(), x_full, y_ref, dev, create_s, "range")
    finally:
        h.close()


def _time_multi_handle(args, h, tdt, isz, x_full, y_ref, dev, create_s, xchg_name, x_slices=None):
    import ctypes as C
    import torch
    from spmv_amd import api
    g_used = h.multi_gpus()
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    if xchg_name == "bcast" and x_slices is not None:   # north_star's literal form: the whole x on device 0, broadcast from there every step
        s0 = h.multi_slices(0)
        off = 0
        for g in range(len(x_slices)):
            assert hip.hipMemcpy(s0["x_ptr"] + isz * (off - s0["x_first"]), x_slices[g].data_ptr(), isz * x_slices[g].numel(), 4) == 0
            off += x_slices[g].numel()
    else:
        for g in range(g_used):     # x into the devices' slices (device-to-device)
            s = h.multi_slices(g)
            src = x_slices[g].data_ptr() if x_slices is not None else x_full.data_ptr() + isz * s["x_first"]
            assert hip.hipMemcpy(s["x_ptr"], src, isz * s["x_count"], 4) == 0
    for _ in range(3):
        h.multi_step()
    iters = max(args.config_iters, 1)
    for g in range(torch.cuda.device_count()):
        torch.cuda.synchronize(g)
    t0 = time.perf_counter()
    for _ in range(iters):
        h.multi_step_async()
    h.multi_synchronize()
    ms = (time.perf_counter() - t0) / iters * 1e3
    ok = None
    if y_ref is not None:       # y blocks against the single-handle result of the headline run
        ok = True
        for g in range(g_used):
            s = h.multi_slices(g)
            blk = torch.empty(s["y_count"], dtype=tdt, device=dev)
            assert hip.hipMemcpy(blk.data_ptr(), s["y_ptr"], isz * s["y_count"], 3) == 0
            ref = y_ref[s["y_first"]: s["y_first"] + s["y_count"]]
            ok = ok and bool(((blk - ref).abs() <= 1e-9 * ref.abs() + 1e-12).all())   # another kernel form may have been tuned in: rounding only
    info = h.info()
    return {"entry": f"spmv_hip_multi_step_async + spmv_hip_multi_synchronize (x_exchange = {xchg_name})", "gpus": g_used,
            "devices_present": torch.cuda.device_count(), "uses_rccl": bool(api.load().spmv_hip_multi_uses_rccl(h.h)), "steps": iters, "ms_per_step": round(ms, 5),
            "gflops": round(2.0 * info["nnz"] / (ms * 1e-3) / 1e9, 1), "nnz_total": int(info["nnz"]), "matches_single_handle": ok, "create_s": round(create_s, 3),
            "note": "wall clock around enqueue + synchronize; at G = 1 there is no exchange, the step is the multiply through the multi-GPU entry"}


def c_leg_child(args):
    """N > 1, a process of its own (started by rank 0 once the per-process ranks are done): ONE process drives all N devices through the C-level
    host (csrc/shim/multi.hpp).  The row blocks are generated on their devices and handed over separately (spmv_hip_create_handle_from_blocks:
    no monolithic 2.56e9-nnz array), then the distributed step is timed for each x exchange: range (peer copies of the referenced columns,
    overlapped), allgather and bcast (RCCL, dlopen'ed by the library).  y block 0 is checked against the definition.  One JSON line."""
    import torch
    from spmv_amd import api, build, synth
    G = args.gpus
    virtual = bool(os.environ.get("SPMV_HIP_GPUS_VIRTUAL"))     # rehearsal on a one-GPU box: every shard on device 0 (the library's own test switch)
    if torch.cuda.device_count() < G and not virtual:
        print(json.dumps({"error": f"{torch.cuda.device_count()} device(s) visible, {G} needed"}))
        return 2
    build.build()
    api.load()
    dt = torch.float64 if args.dtype == "f64" else torch.float32
    isz = 8 if args.dtype == "f64" else 4
    m_loc, k = args.rows, args.nnz_per_row
    n_glob = m_loc * G
    blocks, xs = [], []
    g0 = torch.Generator(device="cuda:0")
    for g in range(G):
        dev = torch.device("cuda", 0 if virtual else g)
        torch.cuda.set_device(dev)
        _, _, rp, ci, va = synth.banded_device(m_loc, n_glob, k, "uniform", dt, dev, seed=1 + g, row0=g * m_loc)
        blocks.append((rp, ci, va))
        gg = torch.Generator(device=dev)
        gg.manual_seed(1234 + g)
        xs.append(torch.rand(m_loc, generator=gg, device=dev, dtype=dt) * 2 - 1)
    torch.cuda.set_device(0)
    out = {"gpus_requested": G, "rows_per_gpu": m_loc, "virtual_shards_on_one_device": virtual, "exchanges": {}}
    for code, name in ((1, "range"), (0, "allgather"), (2, "bcast")):
        api.set_thread_option("x_exchange", code)
        try:
            t0 = time.perf_counter()
            h = api.Handle.from_blocks(blocks, n_glob, args.method)
            create_s = time.perf_counter() - t0
        finally:
            api.clear_thread_options()
        try:
            if True:
                r = _time_multi_handle(args, h, dt, isz, None, None, torch.device("cuda", 0), create_s, name, x_slices=xs)
                # parity of block 0: rows of device 0 only reference columns of slices 0 and (wrap) G - 1
                import ctypes as C
                hip = C.CDLL("libamdhip64.so")
                hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
                s0 = h.multi_slices(0)
                yb = torch.empty(s0["y_count"], dtype=dt, device="cuda:0")
                assert hip.hipMemcpy(yb.data_ptr(), s0["y_ptr"], isz * s0["y_count"], 3) == 0
                xfull = torch.zeros(n_glob, dtype=dt, device="cuda:0")
                xfull[:m_loc] = xs[0]
                xfull[(G - 1) * m_loc:] = xs[G - 1].to("cuda:0")
                if G > 2:
                    xfull[m_loc:2 * m_loc] = xs[1].to("cuda:0")
                rp, ci, va = blocks[0]
                want, scale_row = definition(rp, ci, va, xfull)
                tol = 1e-6 if args.dtype == "f64" else 1e-3
                r["block0_parity_ok"] = bool(((yb.double() - want).abs() <= tol * scale_row + 1e-300).all())
                del xfull, want, scale_row, yb
            out["exchanges"][name] = r
        finally:
            h.close()
    print(json.dumps(out), flush=True)
    return 0


def run_c_leg(args, world):
    """rank 0, after the ranks have released RCCL: the C-level host on all N devices, in a child process with a time limit (a hang or a crash of a
    path that has never met real multi-GPU hardware must not cost the scaling line)."""
    cmd = [sys.executable, os.path.abspath(__file__), "--c-leg", "--gpus", str(world), "--rows", str(args.rows), "--nnz-per-row", str(args.nnz_per_row),
           "--dtype", args.dtype, "--method", str(args.method), "--config-iters", str(args.config_iters)]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK",
                                                             "ROLE_RANK", "ROLE_WORLD_SIZE", "TORCHELASTIC_RUN_ID")}
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=args.c_leg_timeout, env=env)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if lines:
            d = json.loads(lines[-1])
            d["returncode"] = r.returncode
            return d
        return {"error": f"no result line (rc {r.returncode})", "stderr_tail": r.stderr[-400:]}
    except subprocess.TimeoutExpired:
        return {"error": f"timed out after {args.c_leg_timeout:.0f} s"}
    except Exception as e:      # noqa: BLE001
        return {"error": f"{type(e).__name__}: {str(e)[:200]}"}


_TRAFFIC = None


def traffic_from_profiles(kernel_name, m, nnz, dtype, model_bytes):
    """HBM bytes per launch measured by rocprofv3 --pmc (separate FETCH_SIZE / WRITE_SIZE passes, tools/profile_configs.sh ->
    profiles/traffic_rNN.json) for THIS kernel on THIS shape, built from THIS source tree (csrc hash): (bytes, source, per-kernel parts);
    (None, None, None) when no committed entry matches -- a profile of other code says nothing about this run."""
    global _TRAFFIC
    if _TRAFFIC is None:
        from spmv_amd.srchash import csrc_sha
        sha = csrc_sha()
        _TRAFFIC = []
        for name in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.startswith("traffic_r") and f.endswith(".json")), reverse=True):
            try:
                with open(os.path.join(ROOT, "profiles", name)) as f:
                    d = json.load(f)
            except (OSError, ValueError):
                continue
            if d.get("csrc_sha") == sha:
                _TRAFFIC += d.get("entries", [])
    best = None   # two matrices of one shape (config 2 and config 2 with holes; the Orkut-style stand-in with R-MAT and with uniform columns) differ in what their
    for e in _TRAFFIC:   # schedule has to move: the model's byte count is part of the key, and the closest entry within 0.5 % wins
        if e.get("kernel_short") == kernel_name and e.get("m") == m and e.get("nnz") == nnz and e.get("dtype") == dtype:
            diff = abs(int(e.get("model_stream_bytes", -1)) - int(model_bytes))
            if diff <= 0.005 * int(model_bytes) and (best is None or diff < best[0]):
                best = (diff, e)
    if best is not None:
        return best[1].get("hbm_bytes_per_launch"), best[1].get("source"), best[1].get("kernels")
    return None, None, None


# ---------------------------------------------------------------------------------------- one rank
NCCL_FAIL_RC = 3


def run_rank(args):
    import numpy as np  # noqa: F401
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    one_device = bool(os.environ.get("SPMV_BENCH_ONE_DEVICE"))      # rehearsal only: every rank on cuda:0
    dev_index = 0 if (one_device or world == 1) else local_rank
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE {world}"
    if world > 1:
        if args.backend == "nccl":
            # RCCL over xGMI is the judged path.  If it cannot come up there is NO number to report: a host-staged (gloo) figure printed under
            # the same keys could be mistaken for it.  Every rank sees the same failure, says so on stderr and exits non-zero; nothing goes to stdout.
            try:
                if os.environ.get("SPMV_BENCH_FORCE_NCCL_FAIL"):        # tests/test_bench_launcher.py: the failure path without a broken node
                    raise RuntimeError("forced by SPMV_BENCH_FORCE_NCCL_FAIL")
                torch.cuda.set_device(dev_index)
                dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
                dist.barrier()
            except Exception as e:      # noqa: BLE001
                sys.stderr.write(f"bench.py: rank {rank}: nccl (RCCL) initialisation failed ({type(e).__name__}: {str(e)[:200]}); no line is printed -- "
                                 "--backend gloo with SPMV_BENCH_ONE_DEVICE=1 rehearses the ranks on one device and labels its line as such\n")
                return NCCL_FAIL_RC
        else:
            torch.cuda.set_device(dev_index)
            dist.init_process_group("gloo")
        assert dist.get_world_size() == world
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)

    from spmv_amd import api, build, synth
    from spmv_amd.dist import ShardedSpMV
    if rank == 0:
        build.build()
    if world > 1:
        dist.barrier()
    api.load()

    dt = torch.float64 if args.dtype == "f64" else torch.float32
    s = 8 if args.dtype == "f64" else 4
    k = args.nnz_per_row
    rdev = dev if (world == 1 or args.backend == "nccl") else torch.device("cpu")
    K = args.steps

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(v):
        if world == 1:
            return v
        t = torch.tensor([v], dtype=torch.float64, device=rdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def timed(fn, iters):
        """iters calls of fn between barrier + synchronize pairs; MAX over ranks; seconds.  HIP events on the launch
        stream bracket the same calls (torch's current stream is the handle's stream: attach_stream)."""
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        sync_all()
        t0 = time.perf_counter()
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        sync_all()
        return max_over_ranks(time.perf_counter() - t0), e0.elapsed_time(e1) / iters

    def run_workload(m_loc, modes, keep_arrays):
        """One workload (rows per rank = m_loc): build this rank's row block, then for every x exchange in `modes` create the sharded handle, warm
        up, time exactly K steps, and (N > 1) the pieces of a step alone.  -> dict(results, head, nnz_all, m_loc, n_glob [, arrays])"""
        n_glob = m_loc * world
        if args.workload == "banded":
            _, _, rp, ci, va = synth.banded_device(m_loc, n_glob, k, "uniform", dt, dev, seed=1 + rank, row0=rank * m_loc)
        elif args.workload == "powerlaw":
            # heavy-tailed rows (mean k, max 33 k) over the whole (N * rows)^2 matrix, R-MAT columns; every rank draws the same row
            # lengths, cuts them into EQUAL-NNZ row blocks (the reference's splitter, parallel_balanced2_spmv.c:41-53) and builds only
            # its own block -- the monolithic matrix never exists
            from spmv_amd.dist import equal_nnz_cuts
            lens = synth.powerlaw_lengths_device(n_glob, float(k), min(33000, n_glob), 1.5, dev, 1)
            rp_all = torch.zeros(n_glob + 1, dtype=torch.int64, device=dev)
            torch.cumsum(lens, 0, out=rp_all[1:])
            cuts = equal_nnz_cuts(rp_all, world)
            a, b = cuts[rank], cuts[rank + 1]
            _, _, rp, ci, va = synth.from_row_lengths_device(lens[a:b], n_glob, "uniform", dt, dev, seed=1, cols="rmat", row0=a, m_total=n_glob)
            m_loc = b - a
            del lens, rp_all
        else:
            _, _, rp, ci, va = synth.uniform_k_device(m_loc, n_glob, k, "uniform", dt, dev, seed=1 + rank)
        nnz_loc = int(rp[-1].item())
        g = torch.Generator(device=dev)
        g.manual_seed(1234)                     # same full x on every rank, sliced by ownership
        x_full = torch.rand(n_glob, generator=g, device=dev, dtype=dt) * 2 - 1
        results, head = {}, None
        for mode in modes:
            t0 = time.perf_counter()
            sh = ShardedSpMV(rp, ci, va, n_glob, xchg=mode, method=args.method, overlap=not args.no_overlap)
            if sh.xchg in ("none", "bcast") and (sh.xchg == "none" or rank == 0):
                sh.set_full_x(x_full)
            x_loc = sh.x_local_view()            # x lives where the kernel reads it: exchange() is copy-free
            x_loc.copy_(x_full[sh.c0:sh.c1])
            y = torch.full((m_loc,), float("nan"), dtype=dt, device=dev)
            torch.cuda.synchronize()
            create_s = time.perf_counter() - t0
            info = sh.handle.info()                 # split mode: the interior handle (dominant kernel)
            for _ in range(args.warmup):
                sh.step(x_loc, y)
            elapsed, ev_ms = timed(lambda: sh.step(x_loc, y), K)
            r = {"ms_per_step": elapsed / K * 1e3, "create_s": create_s, "ghost_columns_rank0": sh.n_ghost, "overlap_split": bool(sh.split),
                 "boundary_rows_rank0": int(sh.bnd_rows.numel()) if sh.split else 0}
            if world > 1:   # the pieces of a step, each alone (same barrier / MAX-over-ranks protocol, a quarter of the steps)
                kk = max(5, K // 4)
                t_x, _ = timed(lambda: sh.exchange_only(x_loc), kk)
                t_m, mul_ms = timed(lambda: sh.multiply(y), kk)
                r.update(exchange_only_ms=t_x / kk * 1e3, multiply_only_ms=t_m / kk * 1e3,
                         exposed_comm_ms=max(0.0, (elapsed / K - t_m / kk) * 1e3))
            else:
                mul_ms = ev_ms
            if head is None:
                nnz_k = int(sh._A_int[1].numel()) if sh.split else nnz_loc
                head = dict(info=info, mul_ms=mul_ms, nnz_k=nnz_k, n_x=(sh.n_local if sh.split else sh.n_x), xchg=sh.xchg,
                            x_ext=sh.x_ext.clone() if world == 1 else None, y=y.clone() if world == 1 else None)
            results[mode] = r
            sh.close()
            del sh, y
            torch.cuda.empty_cache()
        nnz_all = nnz_loc
        if world > 1:
            nn = torch.tensor([nnz_loc], dtype=torch.int64, device=rdev)
            dist.all_reduce(nn, op=dist.ReduceOp.SUM)
            nnz_all = int(nn.item())
        w = dict(results=results, head=head, nnz_all=nnz_all, m_loc=m_loc, n_glob=n_glob)
        if keep_arrays:
            w.update(rp=rp, ci=ci, va=va, x_full=x_full)
        return w

    modes = ["none"] if world == 1 else (["halo", "allgather", "bcast"] if args.xchg == "all" else [args.xchg])
    weak = run_workload(args.rows, modes, keep_arrays=(world == 1)) if (world == 1 or args.scaling in ("weak", "both")) else None
    strong = None
    if world > 1 and args.scaling in ("strong", "both"):
        strong = run_workload(max(1024, args.rows // world), modes if args.scaling == "strong" else modes[:1], keep_arrays=False)
    main_w, scaling = (weak, "weak") if weak is not None else (strong, "strong")

    def exchanges_of(w):
        return {mode: {"gflops": round(2.0 * w["nnz_all"] / (r["ms_per_step"] * 1e-3) / 1e9, 2), "ms_per_step": round(r["ms_per_step"], 5),
                       "exchange_only_ms": round(r["exchange_only_ms"], 5), "multiply_only_ms": round(r["multiply_only_ms"], 5),
                       "exposed_comm_ms": round(r["exposed_comm_ms"], 5), "ghost_columns_rank0": r["ghost_columns_rank0"],
                       "overlap_split": r["overlap_split"]}
                for mode, r in w["results"].items()}

    out = None
    if rank == 0:
        head, results, nnz_all, m_loc, n_glob = main_w["head"], main_w["results"], main_w["nnz_all"], main_w["m_loc"], main_w["n_glob"]
        info, mul_ms = head["info"], head["mul_ms"]
        first = results[modes[0]]
        ms_step = first["ms_per_step"]
        gflops = 2.0 * nnz_all / (ms_step * 1e-3) / 1e9
        # the dominant kernel of THIS rank: SURVEY 8d's algorithmic bytes for what it multiplies, the model's bytes and the counters'
        alg_bytes = 4 * (m_loc + 1) + head["nnz_k"] * (4 + s) + s * head["n_x"] + s * m_loc
        rf = roofline_fields(info, mul_ms, args.dtype, alg_bytes=alg_bytes)
        rf.update({
            # the reference's own bytes model, x counted once per non-zero (csr5_avx2/utils.h:10-14, numa.c:247-248): comparability only
            "reference_model_gbps": round(((m_loc + 1 + head["nnz_k"]) * 4 + (2 * head["nnz_k"] + m_loc) * s) / (mul_ms * 1e-3) / 1e9, 1),
            "launch_ms_mean": round(mul_ms, 5), "launches_timed": K if world == 1 else max(5, K // 4),
            "run_nnz": int(info.get("run_nnz", 0)), "byte_nnz": int(info.get("byte_nnz", 0)),
            "note": "frac = counter bytes / time / 8 TB/s when profiles/ holds a rocprofv3 measurement of exactly this kernel, shape, dtype and source "
                    "tree (frac_source), else the storage-format model's bytes; frac_alg divides SURVEY 8d's algorithmic bytes by the same time -- an "
                    "EFFECTIVE rate that charges 4 B/nnz of ColIdx: run_nnz entries sit in tiles whose rows are single runs of consecutive columns and "
                    "read 2 B per ROW instead (generator-specific: true for this banded matrix, not for config 2 with holes or 2-ii under 'configs'), "
                    "byte_nnz entries read 1 B, the rest of the staged entries 2 B",
        })
        out = {
            "metric": "SpMV GFLOP/s (fp64 CSR, y = A x through spmv())" if s == 8 else "SpMV GFLOP/s (fp32 CSR)",
            "value": round(gflops, 2), "unit": "GFLOP/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 5), "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {
                "workload": f"config {'2' if world == 1 else '5 (' + scaling + ')'}: {args.workload} CSR {n_glob}x{n_glob}, "
                            f"{k} nnz/row{' (mean; equal-nnz row blocks)' if args.workload == 'powerlaw' else ''}, {m_loc} rows on rank 0",
                "schedule": f"{api.SPMV_METHODS(args.method).name} -> {info['schedule_name']}"
                            + (f" L={info['lanes_per_row']}" if info['lanes_per_row'] else ""),
                "x_exchange": head["xchg"], "rccl_ranks": (dist.get_world_size() if world > 1 else 1),
                "backend": (args.backend if world > 1 else "none"),
                "backend_note": ("gloo: host-staged exchange on one device -- a rehearsal of the rank logic, NOT RCCL over xGMI" if world > 1 and args.backend != "nccl" else None),
                "ghost_columns_rank0": first["ghost_columns_rank0"], "overlap_split": first["overlap_split"],
                "boundary_rows_rank0": first["boundary_rows_rank0"], "vectors": "device-resident x, y",
                "nnz_total": nnz_all, "create_seconds": round(first["create_s"], 3), "inspect_ms": round(info["inspect_ms"], 3),
                "what_was_exchanged": ("nothing: one GPU" if world == 1 else
                                       f"'{head['xchg']}' in the timed steps of `value` ({first['ghost_columns_rank0']} remote x entries on rank 0 per step); "
                                       "every other exchange and its cost alone: 'exchanges'"),
            },
            "hbm_gbps": rf["achieved"],
            "roofline": rf,
        }
        if world > 1:
            out["exchanges"] = exchanges_of(main_w)
            out["scaling_claim"] = {"exchange": "halo", "note": "the >= 6x-at-8-GPUs target (north_star) is claimed on the value above = the "
                                    "'halo' exchange (only the referenced x entries move, point to point, overlapped with the interior rows); "
                                    "'bcast' -- north_star's literal broadcast of x from rank 0 -- and 'allgather' are measured in the same run "
                                    "and reported under 'exchanges'"}
            if strong is not None and strong is not main_w:
                f0 = strong["results"][modes[0]]
                out["strong"] = {"scaling": "strong", "workload": f"{args.workload} CSR {strong['n_glob']}x{strong['n_glob']} in total, {strong['m_loc']} rows per rank",
                                 "value": round(2.0 * strong["nnz_all"] / (f0["ms_per_step"] * 1e-3) / 1e9, 2), "unit": "GFLOP/s", "ms_per_step": round(f0["ms_per_step"], 5),
                                 "nnz_total": strong["nnz_all"], "exchanges": exchanges_of(strong),
                                 "note": "total work fixed at --rows rows (SURVEY 8d config 5: strong beside weak); same protocol, same K steps; efficiency is the driver's to compute "
                                         "against the N = 1 line"}
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(args, weak["rp"], weak["ci"], weak["va"], head["x_ext"], head["y"], head["n_x"])
        if world == 1 and not args.no_configs:
            t0 = time.perf_counter()
            out["multi_gpu_c_entry"] = multi_step_leg(args, dev, weak["rp"], weak["ci"], weak["va"], n_glob, weak["x_full"], head["y"])
            for kk in ("rp", "ci", "va", "x_full"):
                weak.pop(kk, None)
            torch.cuda.empty_cache()
            out["configs"] = extra_configs(args, dev)
            out["configs_seconds"] = round(time.perf_counter() - t0, 1)
        if world == 1:
            out["roofline"]["read_calibration"] = read_calibration(rf["bytes_moved_per_launch"])
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        if world > 1 and not args.no_c_leg and args.backend == "nccl":
            # the C-level host (one process, N devices, RCCL by dlopen): a child process of rank 0, started once the ranks have released RCCL
            for kk in ("rp", "ci", "va", "x_full"):
                main_w.pop(kk, None)
            torch.cuda.empty_cache()
            out["multi_gpu_c_entry"] = run_c_leg(args, world)
        print(json.dumps(out), flush=True)
    return 0


def main():
    args = parse()
    if args.c_leg:
        return c_leg_child(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args)
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
