import sys, time, os
sys.path.insert(0, os.getcwd())
import torch
from spmv_amd import api, synth, build
build.build(); api.load()
dev = "cuda:0"
def run(name, m, n, rp, ci, va):
    for mode in (1, 2):
        api.set_option("auto_method", mode)
        t0 = time.time()
        h = api.Handle(m, n, rp, ci, va, api.SPMV_METHODS.Method_Serial)
        dt = time.time() - t0
        x = torch.ones(n, dtype=va.dtype, device=dev); y = torch.empty(m, dtype=va.dtype, device=dev)
        mean, ms = api.time_launches(h.h, x, y, 3, 10)
        print(name, "auto", mode, h.method.name, h.info()["kernel_name"], "create %.3f s" % dt, "ms %.4f" % float(ms.min()), flush=True)
        h.close()
    api.set_option("auto_method", 0)
m, n, rp, ci, va = synth.banded_device(10_000_000, 10_000_000, 32, "uniform", torch.float64, dev, 1)
run("banded", m, n, rp, ci, va)
m, n, rp, ci, va = synth.from_row_lengths_device(synth.skewed_lengths_device(10_000_000, dev, 1), 10_000_000, "uniform", torch.float32, dev, 1, local=4096)
run("skewed", m, n, rp, ci, va)
m, n, rp, ci, va = synth.uniform_k_device(10_000_000, 10_000_000, 32, "uniform", torch.float64, dev, 1)
run("random", m, n, rp, ci, va)
nx = 215; m = n = nx ** 3
offs = torch.tensor([dz * nx * nx + dy * nx + dx for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1)], device=dev)
rp = torch.arange(0, (m + 1) * 27, 27, dtype=torch.int32, device=dev)
ci = torch.empty(m * 27, dtype=torch.int32, device=dev)
for r0 in range(0, m, 1 << 22):
    r1 = min(m, r0 + (1 << 22)); rows = torch.arange(r0, r1, device=dev)
    ci[r0 * 27:r1 * 27] = ((rows[:, None] + offs[None, :]) % n).reshape(-1).to(torch.int32)
va = torch.rand(m * 27, device=dev, dtype=torch.float64)
run("stencil", m, n, rp, ci, va)
