"""Row-block partitioned SpMV across the GPUs of one node (BASELINE config 5).

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI, "gloo" in CPU tests).
Rank r owns a contiguous block of rows AND the matching slice of x and y -- the GPU analogue of
the reference's only distribution idea, the NUMA experiment's row blocks with x cut into
contiguous slices and owner = col / slice (src/samples/numa.c:277-304, 149-152).  y stays
distributed; the one exchange step of the path is x.

Exchange modes (`xchg`):
  "halo"       (default) inspector: the distinct remote columns a shard references are found once
               (torch.unique on the device), requested from their owners, and ColIdx is renumbered
               into [own slice | ghosts].  Per step each rank packs the entries its peers asked
               for and ONE all_to_all_single delivers every ghost -- point-to-point volumes only,
               which is what xGMI's 7 direct links want (a banded shard needs 2 x 16 values per
               step, a random one degenerates to the full vector).
  "allgather"  every step all ranks gather the whole x (n values per rank): solver-style loop with
               no inspector.
  "bcast"      north_star's literal form: the full x lives on rank 0 and is broadcast every step.
  "none"       x is static and already complete on every rank (the reference harness re-uses one
               x for 110 calls, test_spmv.c:103-124): no per-step communication.

The local multiply is always the HIP library through the C ABI (spmv_amd.api); `compute=` exists
so the CPU gloo tests can drive the partition/exchange logic without a GPU.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import api

__all__ = ["slice_bounds", "equal_nnz_cuts", "ShardedSpMV"]


def slice_bounds(n, world, rank):
    """Contiguous near-equal slices: first (n % world) ranks get one extra element."""
    base, extra = divmod(int(n), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def equal_nnz_cuts(rowptr, world):
    """Row cut points of `world` EQUAL-NNZ row blocks: cuts[t] = upper_bound(RowPtr, min(t * ceil(nnz / world), nnz)) - 1, the
    reference's splitter (init_csrSplitter_balanced2, parallel_balanced2_spmv.c:41-53), with cuts[0] = 0, cuts[world] = m and the
    sequence kept non-decreasing.  A power-law matrix cut into equal-ROW blocks leaves one rank with several times the others'
    work; the x slices stay equal-COLUMN slices (slice_bounds) whatever the rows.  rowptr: 1-D tensor / array of m + 1 entries
    (any integer type, RowPtr[0] = 0); returns a list of world + 1 ints."""
    rp = torch.as_tensor(rowptr).to(torch.int64)
    m = int(rp.numel()) - 1
    nnz = int(rp[-1]) if m >= 0 and rp.numel() else 0
    stride = -(-nnz // max(world, 1))
    keys = torch.tensor([min(t * stride, nnz) for t in range(world + 1)], dtype=torch.int64, device=rp.device)
    cuts = (torch.searchsorted(rp, keys, right=True) - 1).clamp_(0, max(m, 0)).tolist()
    cuts[0], cuts[-1] = 0, max(m, 0)
    for t in range(1, world + 1):
        cuts[t] = max(cuts[t], cuts[t - 1])
    return [int(c) for c in cuts]


def _owner_of(cols, n, world):
    base, extra = divmod(int(n), int(world))
    if base == 0:
        return cols.clone()
    cut = extra * (base + 1)
    return torch.where(cols < cut, cols // (base + 1), extra + (cols - cut) // max(base, 1))


def _staged(fn):
    """RCCL moves device tensors directly.  Under the gloo backend (CPU tests, or a box where RCCL is
    not usable) device tensors are staged through host copies around the same collective."""
    def wrapped(out, inp, *args, **kw):
        if dist.get_backend(kw.get("group")) == "gloo" and (out.is_cuda or inp.is_cuda):
            o, i = out.cpu(), inp.cpu()
            fn(o, i, *args, **kw)
            out.copy_(o)
        else:
            fn(out, inp, *args, **kw)
    return wrapped


_all_to_all_single = _staged(dist.all_to_all_single)
_all_gather_into_tensor = _staged(dist.all_gather_into_tensor)


def _broadcast(t, src, group=None):
    if dist.get_backend(group) == "gloo" and t.is_cuda:
        h = t.cpu()
        dist.broadcast(h, src=src, group=group)
        t.copy_(h)
    else:
        dist.broadcast(t, src=src, group=group)


class ShardedSpMV:
    """y_local = A[rows of this rank, :] @ x   with x distributed like the rows.

    rowptr / colidx / val describe THIS rank's shard: local int32 RowPtr, GLOBAL int32 column
    indices (never a monolithic 2.56e9-nnz array: SURVEY 7 "int32 limits").  n_global is the
    column count; this rank owns x[c0:c1] with (c0, c1) = slice_bounds(n_global, world, rank).
    """

    def __init__(self, rowptr, colidx, val, n_global, xchg="halo", method=api.SPMV_METHODS.Method_Parallel,
                 group=None, compute=None, overlap=True, force_exchange=False):
        """force_exchange: keep the exchange mode in a group of ONE rank (the collectives then run over a single
        member) -- how the GPU test drives every RCCL call of this file on a one-GPU box."""
        assert xchg in ("halo", "allgather", "bcast", "none")
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.xchg = xchg if (self.world > 1 or (force_exchange and dist.is_initialized())) else "none"
        self.n_global = int(n_global)
        self.m_local = int(rowptr.shape[0] - 1)
        self.c0, self.c1 = slice_bounds(self.n_global, self.world, self.rank)
        self.n_local = self.c1 - self.c0
        self.device, self.dtype = val.device, val.dtype
        self.rowptr, self.val = rowptr, val
        self.send_idx = None
        self.any_halo = False
        self.send_counts = self.recv_counts = None
        if self.xchg == "halo":
            colidx = self._plan_halo(colidx)
            self.n_x = self.n_local + self.n_ghost
        else:
            self.n_ghost = 0
            self.n_x = self.n_global
        self.colidx = colidx
        self.x_ext = torch.zeros(self.n_x, dtype=self.dtype, device=self.device)
        self._compute = compute
        self._method = method
        self.handle = None
        # Overlap (halo mode): rows that reference no ghost column ("interior") are multiplied while
        # the halo is in flight, the few boundary rows after it has landed.  Only worth it when the
        # boundary is a small part of the shard.
        self.split = False
        if self.xchg == "halo" and overlap and self.n_ghost > 0:
            self._plan_overlap(rowptr, colidx, val)
        if not self.split:
            self._A = (rowptr, colidx, val)
            if compute is None:
                self.handle = self._make_handle(self.m_local, rowptr, colidx, val)

    def _make_handle(self, m, rowptr, colidx, val):
        h = api.Handle(m, self.n_x, rowptr, colidx, val, self._method)
        if self.device.type == "cuda":
            h.attach_stream(torch.cuda.current_stream(self.device).cuda_stream, async_=True)
        return h

    def _plan_overlap(self, rowptr, colidx, val, max_boundary_fraction=0.1):
        m = self.m_local
        lens = (rowptr[1:] - rowptr[:-1]).to(torch.int64)
        ghost_nnz = torch.nonzero(colidx >= self.n_local).flatten()          # positions of ghost references
        # row of each ghost reference: RowPtr is sorted -> searchsorted (no nnz-sized row index array)
        rows = torch.searchsorted(rowptr.to(torch.int64), ghost_nnz, right=True) - 1
        bnd = torch.unique(rows)
        if bnd.numel() == 0 or bnd.numel() > max_boundary_fraction * max(m, 1):
            return
        is_bnd = torch.zeros(m, dtype=torch.bool, device=self.device)
        is_bnd[bnd] = True
        # interior matrix: same m rows, boundary rows emptied (they come out as 0 and are overwritten)
        lens_i = torch.where(is_bnd, torch.zeros_like(lens), lens)
        rp_i = torch.zeros(m + 1, dtype=torch.int64, device=self.device)
        torch.cumsum(lens_i, 0, out=rp_i[1:])
        keep = torch.ones(colidx.numel(), dtype=torch.bool, device=self.device)
        rp64 = rowptr.to(torch.int64)
        for r in bnd.tolist() if bnd.numel() <= 4096 else []:
            keep[int(rp64[r]):int(rp64[r + 1])] = False
        if bnd.numel() > 4096:                                               # many boundary rows: vectorised mask
            row_of = torch.repeat_interleave(torch.arange(m, device=self.device), lens)
            keep = ~is_bnd[row_of]
            del row_of
        # boundary matrix: the boundary rows only, compacted
        lens_b = lens[bnd]
        rp_b = torch.zeros(bnd.numel() + 1, dtype=torch.int64, device=self.device)
        torch.cumsum(lens_b, 0, out=rp_b[1:])
        self._A_int = (rp_i.to(torch.int32), colidx[keep].contiguous(), val[keep].contiguous())
        self._A_bnd = (rp_b.to(torch.int32), colidx[~keep].contiguous(), val[~keep].contiguous())
        self.bnd_rows = bnd
        self.y_bnd = torch.zeros(bnd.numel(), dtype=self.dtype, device=self.device)
        self.split = True
        if self._compute is None:
            self.handle = self._make_handle(m, *self._A_int)
            self.handle_bnd = self._make_handle(int(bnd.numel()), *self._A_bnd)

    # ------------------------------------------------------------------ inspector (halo)
    def _plan_halo(self, colidx):
        world, rank, dev = self.world, self.rank, self.device
        cols = colidx.to(torch.int64)
        remote_mask = (cols < self.c0) | (cols >= self.c1)
        ghosts = torch.unique(cols[remote_mask])                       # sorted distinct remote columns
        self.n_ghost = int(ghosts.numel())
        owner = _owner_of(ghosts, self.n_global, world)
        recv_counts = torch.bincount(owner, minlength=world).to(torch.int64)   # what I need, per owner
        send_counts = torch.empty_like(recv_counts)
        _all_to_all_single(send_counts, recv_counts, group=self.group)       # what peers need from me
        self.recv_counts = [int(v) for v in recv_counts.tolist()]
        self.send_counts = [int(v) for v in send_counts.tolist()]
        # tell each owner WHICH of its entries I need (ghosts is sorted => grouped by owner)
        want = torch.empty(int(sum(self.send_counts)), dtype=torch.int64, device=dev)
        _all_to_all_single(want, ghosts, output_split_sizes=self.send_counts,
                           input_split_sizes=self.recv_counts, group=self.group)
        self.send_idx = (want - self.c0).contiguous()                 # positions in my slice, peer-major
        # Whether the per-step all_to_all happens at all is decided HERE, once and for the whole group: a rank
        # whose block is purely diagonal (no ghosts, nobody needs its entries) must still enter the collective
        # the other ranks enter -- skipping it on rank-local state deadlocks gloo and desynchronises RCCL's
        # sequence numbers.
        busy = torch.tensor([self.n_ghost + int(self.send_idx.numel())], dtype=torch.int64, device=dev)
        if dist.get_backend(self.group) == "gloo" and busy.is_cuda:
            b = busy.cpu()
            dist.all_reduce(b, op=dist.ReduceOp.MAX, group=self.group)
            busy = b
        else:
            dist.all_reduce(busy, op=dist.ReduceOp.MAX, group=self.group)
        self.any_halo = bool(int(busy.item()) > 0)
        assert self.send_idx.numel() == 0 or (int(self.send_idx.min()) >= 0 and int(self.send_idx.max()) < self.n_local)
        # renumber: own column c -> c - c0 ; remote column -> n_local + rank in `ghosts`
        new = torch.where(remote_mask, torch.zeros_like(cols), cols - self.c0)
        if self.n_ghost:
            new[remote_mask] = self.n_local + torch.searchsorted(ghosts, cols[remote_mask])
        assert self.n_local + self.n_ghost < 2**31
        return new.to(torch.int32).contiguous()

    # ------------------------------------------------------------------ per step
    def x_local_view(self):
        """This rank's slice of x inside the buffer the kernel reads ("halo": the head of
        [own slice | ghosts]).  Writing x here makes exchange() copy-free."""
        if self.xchg == "halo":
            return self.x_ext[: self.n_local]
        return self.x_ext[self.c0:self.c1]

    def exchange(self, x_local):
        """Bring this step's x where the shard's columns point.  x_local: this rank's slice."""
        if self.xchg == "halo":
            if x_local.data_ptr() != self.x_ext.data_ptr():
                self.x_ext[: self.n_local].copy_(x_local)
            if self.any_halo:                       # group-wide decision (_plan_halo)
                send = x_local[self.send_idx] if self.send_idx.numel() else x_local.new_empty(0)
                _all_to_all_single(self.x_ext[self.n_local:], send, output_split_sizes=self.recv_counts,
                                   input_split_sizes=self.send_counts, group=self.group)
        elif self.xchg == "allgather":
            if self.n_global % self.world == 0:
                _all_gather_into_tensor(self.x_ext, x_local.contiguous(), group=self.group)
            else:  # uneven slices: gather equal-sized padded pieces, then compact
                width = -(-self.n_global // self.world)
                piece = x_local.new_zeros(width)
                piece[: self.n_local].copy_(x_local)
                tmp = x_local.new_empty(width * self.world)
                _all_gather_into_tensor(tmp, piece, group=self.group)
                for r in range(self.world):
                    lo, hi = slice_bounds(self.n_global, self.world, r)
                    self.x_ext[lo:hi].copy_(tmp[r * width: r * width + hi - lo])
        elif self.xchg == "bcast":
            # rank 0 holds the whole vector; the slices are first collected there only if the
            # caller passes slices (bench passes rank 0's full x through set_full_x instead)
            _broadcast(self.x_ext, 0, group=self.group)
        else:  # none: x_ext was filled once by set_full_x / the caller
            pass
        return self.x_ext

    def set_full_x(self, x_full):
        """For "none"/"bcast": install a complete x (length n_global) on this rank."""
        assert self.xchg in ("none", "bcast", "allgather")
        self.x_ext.copy_(x_full)

    def _mul(self, which, x, y):
        if self._compute is not None:
            self._compute(*which, x, y)
        else:
            (self.handle if which is not getattr(self, "_A_bnd", None) else self.handle_bnd).spmv(x, y)

    def multiply(self, y_local):
        """Multiply with whatever exchange() left in x_ext (non-overlapped form)."""
        if self.split:
            self._mul(self._A_int, self.x_ext, y_local)
            self._mul(self._A_bnd, self.x_ext, self.y_bnd)
            y_local.index_copy_(0, self.bnd_rows, self.y_bnd)
        else:
            self._mul(self._A, self.x_ext, y_local)
        return y_local

    def step(self, x_local, y_local, events=None):
        """One SpMV step: exchange x, multiply.  In split (overlap) mode the interior rows run while
        the halo all_to_all is in flight; the boundary rows follow once it has landed.
        events = (e0, e1): recorded on the current stream around the dominant multiply."""
        if not self.split:
            self.exchange(x_local)
            if events:
                events[0].record()
            self.multiply(y_local)
            if events:
                events[1].record()
            return y_local
        if x_local.data_ptr() != self.x_ext.data_ptr():
            self.x_ext[: self.n_local].copy_(x_local)
        send = x_local[self.send_idx]
        staged = dist.get_backend(self.group) == "gloo" and send.is_cuda
        if staged:                                    # gloo cannot take device tensors: no async overlap
            _all_to_all_single(self.x_ext[self.n_local:], send, output_split_sizes=self.recv_counts,
                               input_split_sizes=self.send_counts, group=self.group)
            work = None
        else:
            work = dist.all_to_all_single(self.x_ext[self.n_local:], send, output_split_sizes=self.recv_counts,
                                          input_split_sizes=self.send_counts, group=self.group, async_op=True)
        if events:
            events[0].record()
        self._mul(self._A_int, self.x_ext, y_local)   # touches x_ext[:n_local] only
        if events:
            events[1].record()
        if work is not None:
            work.wait()                               # current stream now waits for the halo
        self._mul(self._A_bnd, self.x_ext, self.y_bnd)
        y_local.index_copy_(0, self.bnd_rows, self.y_bnd)
        return y_local

    def exchange_only(self, x_local):
        """The communication of one step without the multiply (bench.py reports it separately)."""
        if self.split:
            send = x_local[self.send_idx]
            _all_to_all_single(self.x_ext[self.n_local:], send, output_split_sizes=self.recv_counts,
                               input_split_sizes=self.send_counts, group=self.group)
            return self.x_ext
        return self.exchange(x_local)

    def close(self):
        for name in ("handle", "handle_bnd"):
            h = getattr(self, name, None)
            if h is not None:
                h.close()
                setattr(self, name, None)
