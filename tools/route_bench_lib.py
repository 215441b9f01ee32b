"""Mixed banded / random-column test matrices shared by tools/route_bench.py and tools/blk_timeline.py."""
import torch
from spmv_amd import synth


def mixed(kind, m, k, dev):
    """rows are banded or random by `kind`: 'banded', 'random', 'prefix1' (first 1 % banded), 'tail10' (last 10 % random), 'every10' (every 10th row
    random), 'emptyprefix1' (the first 1 % of the rows empty, the rest random: what the far half of prefix1 looks like)"""
    _, _, rp, cb, va = synth.banded_device(m, m, k, "uniform", torch.float64, dev, 1)
    if kind == "banded":
        return rp, cb, va
    _, _, _, cr, _ = synth.uniform_k_device(m, m, k, "uniform", torch.float64, dev, 1)
    rows = torch.arange(m, device=dev)
    if kind == "emptyprefix1":
        keep = (rows >= m // 100)
        lens = keep.to(torch.int64) * k
        rp2 = torch.zeros(m + 1, dtype=torch.int64, device=dev)
        torch.cumsum(lens, 0, out=rp2[1:])
        sel = keep.repeat_interleave(k)
        return rp2.to(torch.int32), cr[sel].contiguous(), va[sel].contiguous()
    rnd = {"random": rows >= 0, "prefix1": rows >= m // 100, "tail10": rows >= m - m // 10, "every10": rows % 10 == 0}[kind]
    ci = torch.where(rnd.repeat_interleave(k), cr, cb)
    return rp, ci, va
