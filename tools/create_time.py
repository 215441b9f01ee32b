import sys, time, torch
sys.path.insert(0, "/root/repo")
from spmv_amd import api, synth
dev = "cuda:0"
lens = synth.skewed_lengths_device(10_000_000, dev, 1)
m, n, rp, ci, va = synth.from_row_lengths_device(lens, 10_000_000, "uniform", torch.float32, dev, 1, local=4096)
torch.cuda.synchronize()
for meth in (6, 6, 5, 5, 4, 4, 1, 1):
    t0 = time.time(); h = api.Handle(m, n, rp, ci, va, meth); torch.cuda.synchronize(); t1 = time.time()
    print(meth, "create", round(t1 - t0, 4), "inspect_ms", round(h.info()["inspect_ms"], 2)); h.close()
