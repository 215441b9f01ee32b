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


def _rccl_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    try:
        dev = torch.device("cuda:0")
        m = n = 20011
        A = synth.powerlaw(m, n, 8.0, 3000, 1.5, "eighths", np.float64, seed=5)
        x = synth.fill_x(n, "eighths", np.float64, 7)
        want = oracle.spmv_serial(A, x)
        rp, ci, va = (torch.from_numpy(a).to(dev) for a in (A.rowptr, A.colidx, A.val))
        xt = torch.from_numpy(x).to(dev)
        got = {}
        for xchg in ("halo", "allgather", "bcast"):
            sh = ShardedSpMV(rp, ci, va, n, xchg=xchg, method=4, force_exchange=True)
            assert sh.xchg == xchg and dist.get_backend() == "nccl"
            if xchg == "bcast":
                sh.set_full_x(xt)
            xl = sh.x_local_view()
            xl.copy_(xt)
            y = torch.full((m,), float("nan"), dtype=torch.float64, device=dev)
            for _ in range(3):
                sh.step(xl, y)
            sh.exchange_only(xl)
            torch.cuda.synchronize()
            got[xchg] = bool(np.array_equal(y.cpu().numpy(), want))
            sh.close()
        # the reductions bench.py makes over the group, on device tensors
        t = torch.tensor([1.5], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
        got["all_reduce"] = float(t.item()) == 1.5
        out.update(got)
    finally:
        dist.destroy_process_group()


def test_rccl_collectives_of_every_exchange_mode_with_one_rank():
    """RCCL refuses two ranks on one device, so the one-GPU box cannot run world 2 over it; but every collective
    spmv_amd.dist issues (all_to_all_single with split sizes, in-place all_gather_into_tensor, broadcast, all_reduce,
    barrier) can run through backend "nccl" (= RCCL) in a group of one, on device tensors and on the handle's stream."""
    out = mp.Manager().dict()
    mp.spawn(_rccl_worker, args=(1, _free_port(), out), nprocs=1, join=True)
    assert dict(out) == {"halo": True, "allgather": True, "bcast": True, "all_reduce": True}, dict(out)
