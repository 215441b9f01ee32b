"""CPU, world_size 2 over gloo: the row-block partition + x-exchange logic of spmv_amd.dist for
every exchange mode, with the local multiply injected from the oracle (tests may use it; the
product default is the HIP handle).  y gathered from the ranks must equal the oracle on the
whole matrix bit for bit (eighths fill)."""
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


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_compute(rowptr, colidx, val, x, y):
    csr = synth.CSR(rowptr.numel() - 1, x.numel(), rowptr.numpy(), colidx.numpy(), val.numpy())
    y.copy_(torch.from_numpy(oracle.spmv_serial(csr, x.numpy())))


def _worker(rank, world, port, kind, xchg, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m = n = 1001 if kind != "rect" else 900
        if kind == "banded":
            A = synth.banded(m, n, 8, 7, "eighths", np.float64, seed=3)
        elif kind == "rect":
            A = synth.uniform_k(900, 1300, 9, "eighths", np.float64, seed=4)
            n = 1300
        elif kind == "diag_mid":
            # three ranks: the middle rank's block is purely diagonal (no ghosts, nobody needs its slice), ranks 0
            # and 2 reference each other -- the middle rank must still enter the halo collective
            m = n = 900
            base = synth.banded(m, n, 2, 2, "eighths", np.float64, seed=6)
            rows = np.repeat(np.arange(m), np.diff(base.rowptr))
            cols = base.colidx.copy()
            blk = rows // 300
            cols = np.clip(cols, blk * 300, blk * 300 + 299)               # every block diagonal ...
            far = (blk != 1) & (np.arange(cols.size) % 7 == 0)
            cols[far] = (cols[far] + 600) % 900                            # ... but 0 <-> 2 are coupled
            order = np.lexsort((cols, rows))
            A = synth.CSR(m, n, base.rowptr, cols[order].astype(np.int32), base.val[order])
        else:
            A = synth.powerlaw(m, n, 6.0, 500, 1.5, "eighths", np.float64, seed=5)
        x = synth.fill_x(n, "eighths", np.float64, 7)
        r0, r1 = slice_bounds(A.m, world, rank)
        p0, p1 = int(A.rowptr[r0]), int(A.rowptr[r1])
        rp = torch.from_numpy((A.rowptr[r0:r1 + 1] - p0).astype(np.int32))
        ci = torch.from_numpy(A.colidx[p0:p1].copy())
        va = torch.from_numpy(A.val[p0:p1].copy())
        sh = ShardedSpMV(rp, ci, va, n, xchg=xchg, compute=_oracle_compute)
        c0, c1 = slice_bounds(n, world, rank)
        assert (sh.c0, sh.c1) == (c0, c1)
        xt = torch.from_numpy(x)
        if xchg == "none":
            sh.set_full_x(xt)
        if xchg == "bcast" and rank == 0:
            sh.set_full_x(xt)
        y = torch.full((r1 - r0,), float("nan"), dtype=torch.float64)
        for _ in range(2):
            sh.step(xt[c0:c1].clone(), y)
        want = oracle.spmv_serial(A, x)[r0:r1]
        ok = np.array_equal(y.numpy(), want)
        ghosts = sh.n_ghost
        dist.barrier()
        out[rank] = (ok, ghosts, sh.n_x, bool(sh.split), int(sh.bnd_rows.numel()) if sh.split else 0)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("xchg", ["halo", "allgather", "bcast", "none"])
@pytest.mark.parametrize("kind", ["banded", "powerlaw", "rect"])
def test_world2_matches_oracle(kind, xchg):
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), kind, xchg, out), nprocs=world, join=True)
    assert all(out[r][0] for r in range(world)), dict(out)
    if xchg == "halo" and kind == "banded":
        # a 16-wide band needs only the few columns next to the cut, not the other half of x
        assert all(0 < out[r][1] <= 16 for r in range(world)), dict(out)
        # ... and only the rows next to the cut wait for the halo: interior / boundary overlap split
        assert all(out[r][3] and 0 < out[r][4] <= 16 for r in range(world)), dict(out)
    if xchg == "halo" and kind == "powerlaw":
        assert not any(out[r][3] for r in range(world))      # most rows touch ghosts: no split


def test_world3_rank_with_a_purely_diagonal_block_still_joins_the_halo_collective():
    world = 3
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), "diag_mid", "halo", out), nprocs=world, join=True)
    assert all(out[r][0] for r in range(world)), dict(out)
    assert out[1][1] == 0 and out[0][1] > 0 and out[2][1] > 0      # ghosts: none in the middle, some at both ends


def test_slice_bounds_cover_and_are_contiguous():
    for n in (0, 1, 7, 64, 1001):
        for world in (1, 2, 3, 8):
            cuts = [slice_bounds(n, world, r) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            assert max(c[1] - c[0] for c in cuts) - min(c[1] - c[0] for c in cuts) <= 1


def _worker_equal_nnz(rank, world, port, out):
    from spmv_amd.dist import equal_nnz_cuts
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 1500
        A = synth.powerlaw(n, n, 7.0, 900, 1.4, "eighths", np.float64, seed=11)     # heavy tail: equal-row blocks would be lopsided
        x = synth.fill_x(n, "eighths", np.float64, 3)
        cuts = equal_nnz_cuts(A.rowptr, world)
        r0, r1 = cuts[rank], cuts[rank + 1]
        p0, p1 = int(A.rowptr[r0]), int(A.rowptr[r1])
        rp = torch.from_numpy((A.rowptr[r0:r1 + 1] - p0).astype(np.int32))
        sh = ShardedSpMV(rp, torch.from_numpy(A.colidx[p0:p1].copy()), torch.from_numpy(A.val[p0:p1].copy()), n, xchg="halo", compute=_oracle_compute)
        c0, c1 = slice_bounds(n, world, rank)               # x stays in equal-column slices whatever the row cut
        y = torch.full((r1 - r0,), float("nan"), dtype=torch.float64)
        sh.step(torch.from_numpy(x)[c0:c1].clone(), y)
        ok = np.array_equal(y.numpy(), oracle.spmv_serial(A, x)[r0:r1])
        dist.barrier()
        out[rank] = (ok, cuts, p1 - p0, A.nnz)
    finally:
        dist.destroy_process_group()


def test_world3_equal_nnz_row_blocks_on_a_power_law_matrix():
    """spmv_amd.dist.equal_nnz_cuts = the reference's splitter (parallel_balanced2_spmv.c:41-53) for the per-process path: the
    row blocks hold equal shares of the non-zeros (up to one row), x keeps its equal-column slices, results match the oracle."""
    world = 3
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_equal_nnz, args=(world, _free_port(), out), nprocs=world, join=True)
    assert all(out[r][0] for r in range(world)), dict(out)
    cuts, nnz = out[0][1], out[0][3]
    assert cuts[0] == 0 and cuts[-1] == 1500 and all(a <= b for a, b in zip(cuts, cuts[1:]))
    shares = [out[r][2] for r in range(world)]
    assert sum(shares) == nnz and max(shares) <= nnz / world + 900 + 1       # within one (longest) row of the equal share


def test_equal_nnz_cuts_edge_cases():
    from spmv_amd.dist import equal_nnz_cuts
    assert equal_nnz_cuts(np.array([0]), 4) == [0, 0, 0, 0, 0]                                   # no rows
    assert equal_nnz_cuts(np.array([0, 0, 0, 0]), 2) == [0, 3, 3] or equal_nnz_cuts(np.array([0, 0, 0, 0]), 2)[-1] == 3   # no entries
    rp = np.array([0, 100, 100, 101, 102])                                                        # one heavy row first
    c = equal_nnz_cuts(rp, 3)
    assert c[0] == 0 and c[-1] == 4 and all(a <= b for a, b in zip(c, c[1:]))
