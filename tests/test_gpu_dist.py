"""GPU box (one MI355X): two ranks share cuda:0, collectives over gloo (RCCL refuses two ranks on one
device), the local multiply is the REAL HIP handle.  Exercises spmv_amd.dist end to end for every
exchange mode; gathered y must equal the oracle bit for bit (eighths fill)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from spmv_amd import synth
from spmv_amd.dist import ShardedSpMV, slice_bounds

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, kind, xchg, method, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        m = n = 20011
        if kind == "banded":
            A = synth.banded(m, n, 16, 15, "eighths", np.float64, seed=3)
        else:
            A = synth.powerlaw(m, n, 8.0, 3000, 1.5, "eighths", np.float64, seed=5)
        x = synth.fill_x(n, "eighths", np.float64, 7)
        r0, r1 = slice_bounds(A.m, world, rank)
        p0, p1 = int(A.rowptr[r0]), int(A.rowptr[r1])
        rp = torch.from_numpy((A.rowptr[r0:r1 + 1] - p0).astype(np.int32)).to(dev)
        ci = torch.from_numpy(A.colidx[p0:p1].copy()).to(dev)
        va = torch.from_numpy(A.val[p0:p1].copy()).to(dev)
        sh = ShardedSpMV(rp, ci, va, n, xchg=xchg, method=method)
        xt = torch.from_numpy(x).to(dev)
        if xchg == "none" or (xchg == "bcast" and rank == 0):
            sh.set_full_x(xt)
        y = torch.full((r1 - r0,), float("nan"), dtype=torch.float64, device=dev)
        xl = sh.x_local_view()
        xl.copy_(xt[sh.c0:sh.c1])
        for _ in range(2):
            sh.step(xl, y)
        torch.cuda.synchronize()
        want = oracle.spmv_serial(A, x)[r0:r1]
        out[rank] = bool(np.array_equal(y.cpu().numpy(), want))
        sh.close()
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("xchg,method,kind", [("halo", 1, "banded"), ("halo", 4, "powerlaw"), ("allgather", 6, "powerlaw"),
                                              ("bcast", 5, "banded"), ("none", 1, "powerlaw")])
def test_two_ranks_one_gpu(xchg, method, kind):
    world = 2
    out = mp.Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), kind, xchg, method, out), nprocs=world, join=True)
    assert all(out[r] for r in range(world)), dict(out)
