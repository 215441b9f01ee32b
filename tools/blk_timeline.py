#!/usr/bin/env python3
"""Per-block timeline of one blk_kernel launch (debug build only: python -m spmv_amd.build --debug SPMV_BLK_DEBUG_FORMS;
SPMV_LIB=spmv_amd/lib/libspmv_hip_dbg.so python tools/blk_timeline.py random prefix1): start / duration statistics per launch-order decile,
the slowest blocks, concurrency."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
from spmv_amd import api
api.LIB_PATH = os.environ.get("SPMV_LIB", api.LIB_PATH)
lib = api.load()
import route_bench_lib as rb
import run_config as rc
dev = "cuda:0"
m = int(os.environ.get("ROWS", "10000000"))
api.set_option("split", 0)
for kind in sys.argv[1:] or ["random", "prefix1"]:
    if kind in ("2r", "3o", "3o-uniform", "3w", "4"):      # a BASELINE config (tools/run_config.py)
        m, _, rp, ci, va = rc.make(kind, dev)
    else:
        m = int(os.environ.get("ROWS", "10000000"))
        rp, ci, va = rb.mixed(kind, m, 32, dev)
    x = torch.rand(m, dtype=va.dtype, device=dev); y = torch.empty(m, dtype=va.dtype, device=dev)
    h = api.Handle(m, m, rp, ci, va, 4)
    print("   form:", h.info()["kernel_name"], "waves", h.info()["blk_waves"], "reproducible", h.info()["reproducible"])
    for _ in range(3):
        h.spmv(x, y)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (4 * 8192))()
    assert lib.spmv_shim_debug_blk_times(buf) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 4).astype(np.int64)
    a = a[a[:, 1] > 0]
    a = a[a[:, 1] >= a[:, 1].max() - 300_000]     # the LAST launch only (100 MHz ticks: 3 ms): create() timed other forms before, whose stamps linger in higher slots
    t0 = a[:, 0].min()
    start, dur = (a[:, 0] - t0) / 100.0, (a[:, 1] - a[:, 0]) / 100.0      # microseconds
    blk, ns = a[:, 3] & 0xffffffff, a[:, 3] >> 32
    end = start + dur
    slots = (2 if h.info()["blk_waves"] <= 1 else 1) * 256
    print(f"== {kind}: {len(a)} workgroups, makespan {end.max():.0f} us, sum of durations / {slots} slots {dur.sum() / slots:.0f} us, "
          f"dur min/median/max {dur.min():.0f}/{np.median(dur):.0f}/{dur.max():.0f}, info {h.info()['tuned_choice']} {h.info()['kernel_name']}")
    for lo in range(0, len(a), max(1, len(a) // 8)):
        s = slice(lo, lo + max(1, len(a) // 8))
        print(f"   wg {lo:5d}..: start {start[s].min():7.0f}-{start[s].max():7.0f} us  dur mean {dur[s].mean():7.0f} min {dur[s].min():7.0f} max {dur[s].max():7.0f}  groups {ns[s].mean():.0f}")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.save(os.path.join(ROOT, "gpurun_out", f"timeline_{kind}.npy"), a)
    slow = np.argsort(-dur)[:8]
    print("   slowest:", [(int(i), int(blk[i]), round(float(start[i])), round(float(dur[i])), int(a[i, 2] & 0xf)) for i in slow])
    h.close()
    del rp, ci, va, x, y
    torch.cuda.empty_cache()
