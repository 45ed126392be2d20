"""Data-parallel training over the GPUs of one node: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI), replicas of U, V and the Adam moments on every rank.

Sharding: a GLOBAL batch of B*R samples per optimiser step; rank r owns the contiguous
slice [r*B, (r+1)*B) of every global batch; the divisor of the mean loss is the global batch size, so the
result equals the single-GPU run with batch_size = B*R up to summation order, and every rank ends each
step with bit-identical replicas (they all apply the same gathered/reduced quantities in the same order).

The product path is `NativeDP`: the whole per-step loop (coefficient kernel, ONE in-place ncclAllGather of the
B {g, term} pairs per rank, fused step over the global batch) runs inside libmfcd_hip.so (mfcd_dp_train_steps),
which binds RCCL at run time; Python only hands over pointers once per epoch.  `train_steps_dp` below is the same
protocol spelled out step by step over torch.distributed (any backend): it is what the CPU tests drive with gloo
and an oracle-backed compute object, and it also carries the north star's dense "allreduce" form.

Two exact forms of the per-step exchange:
  "allreduce"  the north-star form: each rank scatters its samples' row gradients into a dense
               [(n+m), d] fp32 buffer, ONE all-reduce(sum) of that buffer, then dense Adam from it;
  "allgather"  (default) replicas are identical, so only the B per-sample backward coefficients (and
               BCE terms) travel: one all-gather of 2*B floats per rank; every rank then rebuilds the
               row gradients of the whole global batch from its own replica inside the fused step.
               Same arithmetic per sample, 4 orders of magnitude fewer bytes on the wire.

The compute steps go through a small backend object so that the sharding/collective logic can be
exercised on CPU (gloo) by the tests, which inject an oracle-backed backend; the product backend is
`HipCompute` (C-ABI calls, no fallback).
"""
import numpy as np
import torch
import torch.distributed as dist

from . import _lib, engine


class HipCompute:
    """C-ABI backed compute steps of the data-parallel loop (device tensors only)."""

    def __init__(self, binding):
        self.b = binding
        U, V = binding.model.U.data, binding.model.V.data
        self.n, self.d, self.m, self.dev = U.shape[0], U.shape[1], V.shape[0], U.device
        self.L = _lib.load()
        nbytes = self.L.mfcd_train_workspace_bytes(1 << 16, 1 << 16, self.n, self.m, self.d)
        self.ws = torch.empty(int(nbytes), dtype=torch.uint8, device=self.dev)
        self.grad = None

    def coefficients(self, rec_local, divisor, out=None):
        """→ fp32 [2, B_local]: row 0 backward coefficients g_t (divisor = global batch), row 1 BCE terms.
        With `out` (a [2, >=B_local] fp32 buffer) the kernel writes in place and nothing is allocated."""
        U, V = self.b.model.U.data, self.b.model.V.data
        B = rec_local.shape[0]
        if out is None:
            out = torch.zeros((2, max(B, 1)), dtype=torch.float32, device=self.dev)
        _lib.check(self.L.mfcd_batch_coefficients(_lib.ptr(U), _lib.ptr(V), _lib.ptr(rec_local), B, self.n, self.m,
                                                  self.d, divisor, out[0].data_ptr(), out[1].data_ptr(), None,
                                                  _lib.stream_ptr(self.dev)))
        return out[:, :B]

    def apply(self, rec_global, g_global):
        """One Adam step from the coefficients of the whole global batch (in place on the caller's tensors)."""
        U, V, mU, vU, mV, vV = self.b.tensors()
        lr, b1, b2, eps, wd = self.b.hyper()
        _lib.check(self.L.mfcd_apply_step(_lib.ptr(U), _lib.ptr(V), _lib.ptr(mU), _lib.ptr(vU), _lib.ptr(mV),
                                          _lib.ptr(vV), _lib.ptr(rec_global), _lib.ptr(g_global.contiguous()),
                                          rec_global.shape[0], self.b.step + 1, self.n, self.m, self.d, lr, b1, b2,
                                          eps, wd, _lib.ptr(self.ws), self.ws.numel(), _lib.stream_ptr(self.dev)))
        self.b.advance(1)

    def dense_grad(self, rec_local, divisor):
        """→ (flat fp32 gradient buffer [(n+m)*d] holding this rank's share, fp32 [B_local] BCE terms)."""
        U, V = self.b.model.U.data, self.b.model.V.data
        if self.grad is None:
            self.grad = torch.empty((self.n + self.m) * self.d, dtype=torch.float32, device=self.dev)
        gU, gV = self.grad[: self.n * self.d], self.grad[self.n * self.d:]
        B = rec_local.shape[0]
        terms = torch.zeros(max(B, 1), dtype=torch.float32, device=self.dev)
        _lib.check(self.L.mfcd_dense_grad(_lib.ptr(U), _lib.ptr(V), _lib.ptr(rec_local), B, self.n, self.m, self.d,
                                          divisor, _lib.ptr(gU), _lib.ptr(gV), _lib.ptr(terms),
                                          _lib.stream_ptr(self.dev)))
        return self.grad, terms[:B]

    def adam_dense(self, grad):
        U, V, mU, vU, mV, vV = self.b.tensors()
        lr, b1, b2, eps, wd = self.b.hyper()
        gU, gV = grad[: self.n * self.d], grad[self.n * self.d:]
        _lib.check(self.L.mfcd_adam_dense(_lib.ptr(U), _lib.ptr(V), _lib.ptr(mU), _lib.ptr(vU), _lib.ptr(mV),
                                          _lib.ptr(vV), _lib.ptr(gU), _lib.ptr(gV), self.b.step + 1, self.n, self.m,
                                          self.d, lr, b1, b2, eps, wd, _lib.stream_ptr(self.dev)))
        self.b.advance(1)


class NativeDP:
    """Data-parallel optimiser steps through the native loop (include/mfcd.h: mfcd_dp_train_steps).

    With a torch.distributed group, rank 0 draws an ncclUniqueId that is broadcast over that group, and every rank
    creates its own RCCL communicator inside the library (on the current device).  Without a group (`world` given
    explicitly, comm-less) the library computes every rank's shard in this process: the single-process rehearsal
    the GPU tests use to check the sharding at world sizes a one-GPU box cannot host."""

    ID_BYTES = 128

    def __init__(self, binding, group=None, simulate_world=None):
        import ctypes
        self.b = binding
        self.L = _lib.load()
        U, V = binding.model.U.data, binding.model.V.data
        if U.dtype not in (torch.float32, torch.bfloat16):
            raise NotImplementedError("the data-parallel loop takes fp32 or bf16 factor tables")
        self.entry = self.L.mfcd_dp_train_steps if U.dtype == torch.float32 else self.L.mfcd_dp_train_steps_bf16
        self.n, self.d, self.m, self.dev = U.shape[0], U.shape[1], V.shape[0], U.device
        self.comm = None
        if simulate_world is not None:
            self.rank, self.world = 0, int(simulate_world)
        else:
            self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
            self.comm = create_native_comm(self.L, self.dev, self.rank, self.world, group)
        self.ws = None

    def close(self):
        if self.comm is not None:
            torch.cuda.synchronize(self.dev)
            _lib.check(self.L.mfcd_dp_comm_destroy(self.comm))
            self.comm = None

    def train_steps(self, stream, batch_local, loss_out=None):
        """Consume `stream` (the GLOBAL sample order, identical on every rank) in global batches of
        batch_local * world.  Returns the fp32 device tensor of global batch-mean losses.  No host sync."""
        U, V, mU, vU, mV, vV = self.b.tensors()
        N = stream.shape[0]
        Bg = batch_local * self.world
        nsteps = (N + Bg - 1) // Bg
        if loss_out is None:
            loss_out = torch.empty(max(nsteps, 1), dtype=torch.float32, device=self.dev)
        need = self.L.mfcd_dp_workspace_bytes(N, batch_local, self.world, self.n, self.m, self.d)
        if self.ws is None or self.ws.numel() < need:
            self.ws = torch.empty(int(need), dtype=torch.uint8, device=self.dev)
        lr, b1, b2, eps, wd = self.b.hyper()
        _lib.check(self.entry(
            _lib.ptr(U), _lib.ptr(V), _lib.ptr(mU), _lib.ptr(vU), _lib.ptr(mV), _lib.ptr(vV), _lib.ptr(stream), N,
            batch_local, self.rank, self.world, self.b.step, self.n, self.m, self.d, lr, b1, b2, eps, wd,
            _lib.ptr(loss_out), _lib.ptr(self.ws), self.ws.numel(), self.comm, _lib.stream_ptr(self.dev)))
        self.b.advance(nsteps)
        return loss_out[:nsteps]


# ---------------------------------------------------------------------------------------------------------------
# Row-sharded state: strong scaling at the reference's batch size (include/mfcd.h: mfcd_shard_*)
# ---------------------------------------------------------------------------------------------------------------
def shard_rows(rows, rank, world):
    """Rows [lo, hi) of a `rows`-row table that rank `rank` of `world` owns (contiguous, sizes differ by at most 1)."""
    return rows * rank // world, rows * (rank + 1) // world


class RowShard:
    """This rank's rows of U, V and of their Adam moments, cut out of full tensors (and written back by `gather`)."""

    def __init__(self, binding, rank, world):
        self.b, self.rank, self.world = binding, rank, world
        U, V, mU, vU, mV, vV = binding.tensors()
        self.n, self.d, self.m = U.shape[0], U.shape[1], V.shape[0]
        self.u_lo, self.u_hi = shard_rows(self.n, rank, world)
        self.v_lo, self.v_hi = shard_rows(self.m, rank, world)
        cut = lambda t, lo, hi: t[lo:hi].clone().contiguous()          # noqa: E731
        self.U, self.mU, self.vU = (cut(t, self.u_lo, self.u_hi) for t in (U, mU, vU))
        self.V, self.mV, self.vV = (cut(t, self.v_lo, self.v_hi) for t in (V, mV, vV))

    def tensors(self):
        return self.U, self.V, self.mU, self.vU, self.mV, self.vV

    def gather(self, group=None):
        """All-gather the shards back into the binding's full tensors (every rank ends with the whole model)."""
        full = self.b.tensors()
        for part, whole, rows in zip(self.tensors(), full, (self.n, self.m, self.n, self.n, self.m, self.m)):
            sizes = [shard_rows(rows, r, self.world) for r in range(self.world)]
            pieces = [torch.empty((hi - lo, self.d), dtype=whole.dtype, device=whole.device) for lo, hi in sizes]
            dist.all_gather(pieces, part, group=group)
            for (lo, hi), piece in zip(sizes, pieces):
                whole[lo:hi] = piece


class HipShardCompute:
    """C-ABI backed halves of one row-sharded optimiser step (device tensors only)."""

    def __init__(self, shard):
        self.s, self.L = shard, _lib.load()
        self.dev = shard.U.device

    def new_xbuf(self, B):
        return torch.empty(3 * B * self.s.d, dtype=torch.float32, device=self.dev)

    def pack(self, batch, B, xbuf):
        s = self.s
        _lib.check(self.L.mfcd_shard_pack(_lib.ptr(s.U) if s.U.numel() else None, _lib.ptr(s.V) if s.V.numel() else None,
                                          _lib.ptr(batch), batch.shape[0], B, s.d, s.u_lo, s.u_hi, s.v_lo, s.v_hi,
                                          _lib.ptr(xbuf), _lib.stream_ptr(self.dev)))

    def pack_ahead(self, next_batch, B, xbuf, step, hyper):
        """Rows of `next_batch` as they will be after optimiser step `step`, which has not run yet."""
        s = self.s
        p = lambda t: _lib.ptr(t) if t.numel() else None              # noqa: E731
        lr, b1, b2, eps, wd = hyper
        _lib.check(self.L.mfcd_shard_pack_ahead(p(s.U), p(s.V), p(s.mU), p(s.vU), p(s.mV), p(s.vV), _lib.ptr(next_batch),
                                                next_batch.shape[0], B, step, s.d, s.u_lo, s.u_hi, s.v_lo, s.v_hi,
                                                lr, b1, b2, eps, wd, _lib.ptr(xbuf), _lib.stream_ptr(self.dev)))

    def apply(self, batch, B, xbuf, step, hyper, terms):
        s = self.s
        p = lambda t: _lib.ptr(t) if t.numel() else None              # noqa: E731
        lr, b1, b2, eps, wd = hyper
        _lib.check(self.L.mfcd_shard_apply(p(s.U), p(s.V), p(s.mU), p(s.vU), p(s.mV), p(s.vV), _lib.ptr(batch),
                                           batch.shape[0], B, _lib.ptr(xbuf), step, s.d, s.u_lo, s.u_hi, s.v_lo,
                                           s.v_hi, lr, b1, b2, eps, wd, _lib.ptr(terms), _lib.stream_ptr(self.dev)))


def batch_collisions(records, B):
    """flags[k] = True when batch k+1 names a row that batch k names too (same table) — what mfcd_shard_collisions
    marks on the device; host form for callers that own the collective.  `records`: int [N, >=3] array of (u, i, j)."""
    r = np.asarray(records)[:, :3]
    nsteps = (r.shape[0] + B - 1) // B
    flags = np.zeros(nsteps, dtype=bool)
    for k in range(nsteps - 1):
        a, b = r[k * B:(k + 1) * B], r[(k + 1) * B:(k + 2) * B]
        flags[k] = bool(np.intersect1d(a[:, 0], b[:, 0]).size or np.intersect1d(a[:, 1:], b[:, 1:]).size)
    return flags


def train_steps_sharded(compute, stream, B, step0, hyper, group=None, pipelined=True):
    """Consume `stream` (int32 [N,4] records, identical on every rank) in batches of B — the reference's batch, NOT
    B*world — with the state sharded by rows over the group.  Per step: pack the owned rows of the batch, ONE
    all-reduce(sum) of the exchange buffer viewed as int32 (exact: one non-zero contributor per row), apply.
    Returns the fp32 tensor of per-step batch-mean losses (identical on every rank; no collective needed for them:
    every rank holds every sample's rows).  `compute` provides new_xbuf / pack / pack_ahead / apply (HipShardCompute;
    the CPU tests inject an oracle-backed one).

    pipelined (default): where batch k+1 shares no row with batch k its rows are packed AHEAD of step k, rolled
    forward over it (`compute.pack_ahead`: the dense update with a zero sparse gradient is a pure function of a row's
    p, m, v), and their all-reduce is in flight while step k runs; pairs of batches that share a row keep the strict
    pack → all-reduce → step chain.  Same results bit for bit (the native loop mfcd_shard_train_steps does the same
    with a side stream inside the library)."""
    N = stream.shape[0]
    nsteps = (N + B - 1) // B
    losses = torch.empty(nsteps, dtype=torch.float32, device=stream.device)
    xbufs = [compute.new_xbuf(B), compute.new_xbuf(B)]
    terms = torch.empty(B, dtype=torch.float32, device=stream.device)
    collide = batch_collisions(stream.cpu().numpy(), B) if pipelined else np.ones(max(nsteps, 1), dtype=bool)
    batch_of = lambda k: stream[k * B:(k + 1) * B]                     # noqa: E731
    reduce = lambda x: dist.all_reduce(x.view(torch.int32), op=dist.ReduceOp.SUM, group=group, async_op=True)  # noqa: E731
    work = [None, None]
    if nsteps:
        compute.pack(batch_of(0), B, xbufs[0])
        work[0] = reduce(xbufs[0])
    for k in range(nsteps):
        cur, nxt = k & 1, (k & 1) ^ 1
        batch = batch_of(k)
        more = k + 1 < nsteps
        ahead = more and not collide[k]
        if ahead:                                   # batch k+1 goes on the wire before step k runs
            compute.pack_ahead(batch_of(k + 1), B, xbufs[nxt], step0 + k + 1, hyper)
            work[nxt] = reduce(xbufs[nxt])
        work[cur].wait()
        compute.apply(batch, B, xbufs[cur], step0 + k + 1, hyper, terms)
        if more and not ahead:                      # a shared row: batch k+1 is packed from the updated state
            compute.pack(batch_of(k + 1), B, xbufs[nxt])
            work[nxt] = reduce(xbufs[nxt])
        losses[k] = terms[: batch.shape[0]].sum() / batch.shape[0]   # ~1 ulp of the native loop's fixed-order mean
    return losses


class NativeShard:
    """Row-sharded optimiser steps through the native loop (include/mfcd.h: mfcd_shard_train_steps): the per-step
    pack → ncclAllReduce → fused step chain is enqueued inside libmfcd_hip.so.  With `simulate_world` (no group) the
    tensors stay whole and the library plays every rank in this process: the single-process rehearsal."""

    def __init__(self, binding, group=None, simulate_world=None):
        self.b, self.L = binding, _lib.load()
        U, V = binding.model.U.data, binding.model.V.data
        if U.dtype not in (torch.float32, torch.bfloat16):
            raise NotImplementedError("the row-sharded loop takes fp32 or bf16 factor tables")
        self.entry = self.L.mfcd_shard_train_steps if U.dtype == torch.float32 else self.L.mfcd_shard_train_steps_bf16
        self.n, self.d, self.m, self.dev = U.shape[0], U.shape[1], V.shape[0], U.device
        self.comm, self.shard, self.group = None, None, group
        if simulate_world is not None:
            self.rank, self.world = 0, int(simulate_world)
        else:
            self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
            self.comm = create_native_comm(self.L, self.dev, self.rank, self.world, group)
            self.shard = RowShard(binding, self.rank, self.world)
        self.ws = None

    def close(self):
        if self.comm is not None:
            torch.cuda.synchronize(self.dev)
            _lib.check(self.L.mfcd_dp_comm_destroy(self.comm))
            self.comm = None

    def train_steps(self, stream, B, loss_out=None):
        N = stream.shape[0]
        nsteps = (N + B - 1) // B
        if loss_out is None:
            loss_out = torch.empty(max(nsteps, 1), dtype=torch.float32, device=self.dev)
        need = self.L.mfcd_shard_workspace_bytes(N, B, self.d)
        if self.ws is None or self.ws.numel() < need:
            self.ws = torch.empty(int(need), dtype=torch.uint8, device=self.dev)
        tens = self.shard.tensors() if self.shard is not None else self.b.tensors()
        p = lambda t: _lib.ptr(t) if t.numel() else _lib.ptr(self.ws)   # noqa: E731  (an empty shard is never dereferenced)
        lr, b1, b2, eps, wd = self.b.hyper()
        _lib.check(self.entry(*[p(t) for t in tens], _lib.ptr(stream), N, B, self.rank, self.world,
                                                 self.b.step, self.n, self.m, self.d, lr, b1, b2, eps, wd,
                                                 _lib.ptr(loss_out), _lib.ptr(self.ws), self.ws.numel(), self.comm,
                                                 _lib.stream_ptr(self.dev)))
        self.b.advance(nsteps)
        return loss_out[:nsteps]

    def gather(self):
        if self.shard is not None:
            self.shard.gather(self.group)


def create_native_comm(L, dev, rank, world, group=None):
    """RCCL communicator inside libmfcd_hip.so for this rank: rank 0 draws the ncclUniqueId, the caller's
    torch.distributed group carries it."""
    import ctypes
    nbytes = NativeDP.ID_BYTES
    uid = torch.zeros(nbytes, dtype=torch.uint8)
    if rank == 0:
        buf = (ctypes.c_ubyte * nbytes)()
        _lib.check(L.mfcd_dp_unique_id(ctypes.cast(buf, ctypes.c_void_p), nbytes))
        uid = torch.tensor(list(buf), dtype=torch.uint8)
    backend = dist.get_backend(group)
    carrier = uid.to(dev) if backend == "nccl" else uid
    dist.broadcast(carrier, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    raw = bytes(carrier.cpu().tolist())
    handle = ctypes.c_void_p()
    with torch.cuda.device(dev):
        _lib.check(L.mfcd_dp_comm_create(raw, nbytes, rank, world, ctypes.byref(handle)))
    return handle


def shard_bounds(global_lo, global_hi, batch_local, rank):
    """Sample range of `rank` inside the global batch [global_lo, global_hi): contiguous slices of batch_local."""
    lo = min(global_hi, global_lo + rank * batch_local)
    hi = min(global_hi, lo + batch_local)
    return lo, hi


def train_steps_dp(compute, stream, batch_local, mode="allgather", group=None):
    """Consume `stream` (records of the GLOBAL sample order, identical on every rank; [N,4] int32 tensor)
    in global batches of batch_local * world_size.  Returns a fp32 tensor with the global batch-mean loss of
    every step (identical on all ranks).  `compute` provides coefficients/apply/dense_grad/adam_dense.
    Buffers are allocated once per call; per step there is one compute call, one collective, one compute call."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    N = stream.shape[0]
    B = batch_local
    Bg = B * world
    nsteps = (N + Bg - 1) // Bg
    dev = stream.device
    if nsteps == 0:
        return torch.empty(0, dtype=torch.float32, device=dev)
    nglob_all = torch.tensor([min(N, lo + Bg) - lo for lo in range(0, N, Bg)], dtype=torch.float32, device=dev)
    if mode == "allgather":
        mine = torch.zeros((2, B), dtype=torch.float32, device=dev)
        gathered = torch.empty((nsteps, world * 2 * B), dtype=torch.float32, device=dev)   # every step's exchange
        g_all = torch.empty(Bg, dtype=torch.float32, device=dev)
        for k, lo in enumerate(range(0, N, Bg)):
            hi = min(N, lo + Bg)
            nglob = hi - lo                               # divisor of the mean (structure.py:849, short last batch)
            mylo, myhi = shard_bounds(lo, hi, B, rank)
            if myhi - mylo < B:
                mine.zero_()                              # padded slots must carry g = 0, term = 0
            if myhi > mylo:
                compute.coefficients(stream[mylo:myhi], nglob, out=mine)
            dist.all_gather_into_tensor(gathered[k], mine.view(-1), group=group)   # 1-D in, 1-D out: every backend
            # rank-major order == global sample order (contiguous shards of B); slots beyond nglob are padding
            g_all.view(world, B).copy_(gathered[k].view(world, 2, B)[:, 0, :])
            compute.apply(stream[lo:hi], g_all[:nglob])
        terms = gathered.view(nsteps, world, 2, B)[:, :, 1, :].reshape(nsteps, Bg)
        return terms.sum(dim=1) / nglob_all
    if mode == "allreduce":
        tsum = torch.zeros(nsteps, dtype=torch.float32, device=dev)
        for k, lo in enumerate(range(0, N, Bg)):
            hi = min(N, lo + Bg)
            mylo, myhi = shard_bounds(lo, hi, B, rank)
            grad, terms = compute.dense_grad(stream[mylo:myhi], hi - lo)
            if myhi > mylo:
                tsum[k] = terms.sum()
            dist.all_reduce(grad, op=dist.ReduceOp.SUM, group=group)
            compute.adam_dense(grad)
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM, group=group)   # one reduction of all the step losses
        return tsum / nglob_all
    raise ValueError(f"unknown data-parallel mode {mode!r}")


def broadcast_state(binding, src=0, group=None):
    """Make every rank start from rank `src`'s parameters and moments."""
    for t in binding.tensors():
        dist.broadcast(t, src=src, group=group)


def hip_slab_pass(U, V, X_rows, row0, s, what):
    """Rows [row0, row0 + k) of the UV^T metric pass on this GPU (include/mfcd.h: mfcd_uvt_stats_slab): per-row sums
    for those rows and this slab's share of the two global sums."""
    from . import metrics
    L = _lib.load()
    U, V, X_rows = U.contiguous(), V.contiguous(), X_rows.contiguous()
    (n, d), m, k = U.shape, V.shape[0], X_rows.shape[0]
    rs = torch.empty((k, 8), dtype=torch.float64, device=U.device) if what & 1 else None
    share = torch.empty(4, dtype=torch.float64, device=U.device) if what & 2 else None
    ws = metrics._workspace(L.mfcd_uvt_slab_workspace_bytes(n, m, d, k), U.device)
    _lib.check(L.mfcd_uvt_stats_slab(_lib.ptr(U), _lib.ptr(V), _lib.ptr(X_rows), n, m, d, float(s), int(what), int(row0),
                                     k, _lib.ptr(rs), _lib.ptr(share), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(U.device)))
    return rs, share


def uvt_stats_sharded(U, V, X_rows, row0, s=1.0, what=3, group=None, slab_pass=hip_slab_pass):
    """The UV^T metric pass with X sharded by ROW BLOCKS over the ranks ("eval pass shards by row blocks of
    U/X with an all-reduce of a handful of scalars").  Every rank holds the full U, V (the data-parallel replicas; the
    column centring of UV^T needs every row of U, which is (n, d), not (n, m)) and rows [row0, row0 + k) of X, k may
    differ per rank and may be 0.  Per rank: one slab pass over its rows (MFMA work / R).  Exchange: ONE all-gather of
    the 4-double shares (summed in rank order: deterministic, equal to the slab-ordered single-GPU sum) and, when the
    per-row sums are asked for, one all-gather of the [k, 8] blocks, padded to the largest k.
    Returns what metrics.uvt_stats returns — row_stats [n, 8] assembled in row order, scal [4] — on every rank."""
    world = dist.get_world_size(group)
    n, k = U.shape[0], X_rows.shape[0]
    dev = U.device
    rs = share = None
    if k:
        rs, share = slab_pass(U, V, X_rows, row0, s, what)
    meta = torch.tensor([row0, k], dtype=torch.int64, device=dev)
    metas = [torch.empty_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    spans = [(int(t[0]), int(t[1])) for t in metas]
    at = 0
    for a, b in spans:                      # rank order, contiguous, no overlap; a rank may hold no row
        if b and a != at:
            raise ValueError(f"row blocks must tile [0, {n}) in rank order, got {spans}")
        at += b
    if at != n:
        raise ValueError(f"row blocks must tile [0, {n}) in rank order, got {spans}")
    scal = None
    if what & 2:
        mine = share if share is not None else torch.zeros(4, dtype=torch.float64, device=dev)
        shares = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(shares, mine, group=group)
        scal = torch.zeros(4, dtype=torch.float64, device=dev)
        for t in shares:
            scal += t
    row_stats = None
    if what & 1:
        kmax = max(b for _, b in spans)
        pad = torch.zeros((kmax, 8), dtype=torch.float64, device=dev)
        if k:
            pad[:k] = rs
        blocks = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(blocks, pad, group=group)
        row_stats = torch.cat([blk[:b] for blk, (_, b) in zip(blocks, spans)])
    return row_stats, scal


def reconstruction_error_sharded(U, V, X_rows, row0, s, group=None, slab_pass=hip_slab_pass):
    """structure.py:925-955 with X sharded by row blocks: ||(UV^T - colmean) - sX||_F / ||sX||_F, same value on every rank."""
    _, scal = uvt_stats_sharded(U, V, X_rows, row0, s, what=2, group=group, slab_pass=slab_pass)
    import numpy as np
    e2, r2 = scal[:2].cpu().tolist()
    return float(np.sqrt(e2) / np.sqrt(r2)) if r2 > 0 else float("nan") if e2 == 0 else float("inf")   # as metrics.reconstruction_error


def _probe_dp_phases(binding, stream, B, nsteps, sharded):
    """Per-phase split of one multi-GPU optimiser step, measured with HIP event pairs over `nsteps` steps of the SPLIT
    form of the loop (the same kernels and the same collective the native loop enqueues, issued phase by phase from
    Python so that events fit between them): coefficient / pack kernel, collective, fused step.  Untimed region of the
    benchmark; the model advances by nsteps steps on every rank alike.  → dict of mean microseconds per phase."""
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = stream.device
    ev = lambda: torch.cuda.Event(enable_timing=True)                  # noqa: E731
    marks = []
    if sharded:
        shard = RowShard(binding, rank, world)
        comp = HipShardCompute(shard)
        xbuf, terms = comp.new_xbuf(B), torch.empty(B, dtype=torch.float32, device=dev)
        hyper = binding.hyper()
        for k in range(nsteps):
            batch = stream[k * B:(k + 1) * B]
            e = [ev() for _ in range(4)]
            e[0].record(); comp.pack(batch, B, xbuf)
            e[1].record(); dist.all_reduce(xbuf.view(torch.int32), op=dist.ReduceOp.SUM)
            e[2].record(); comp.apply(batch, B, xbuf, binding.step + k + 1, hyper, terms)
            e[3].record(); marks.append(e)
        binding.advance(nsteps)
        shard.gather()
        names = ("pack_us", "collective_us", "step_us")
    else:
        comp = HipCompute(binding)
        Bg = B * world
        mine = torch.zeros((2, B), dtype=torch.float32, device=dev)
        gathered = torch.empty(world * 2 * B, dtype=torch.float32, device=dev)
        g_all = torch.empty(Bg, dtype=torch.float32, device=dev)
        for k in range(nsteps):
            lo, hi = k * Bg, (k + 1) * Bg
            e = [ev() for _ in range(4)]
            e[0].record(); comp.coefficients(stream[lo + rank * B: lo + (rank + 1) * B], Bg, out=mine)
            e[1].record(); dist.all_gather_into_tensor(gathered, mine.view(-1))
            e[2].record()
            g_all.view(world, B).copy_(gathered.view(world, 2, B)[:, 0, :])
            comp.apply(stream[lo:hi], g_all)
            e[3].record(); marks.append(e)
        names = ("coefficients_us", "collective_us", "step_us")
    torch.cuda.synchronize()
    tail = marks[len(marks) // 4:]                                      # first quarter: warm-up of the split path
    out = {nm: round(sum(e[i].elapsed_time(e[i + 1]) for e in tail) * 1e3 / len(tail), 2) for i, nm in enumerate(names)}
    out["sum_us"] = round(sum(out.values()), 2)
    out["steps_probed"] = len(tail)
    out["how"] = ("HIP event pairs around each phase of the per-step loop issued from Python (same kernels, same collective "
                  "as the native loop; gaps between phases include the Python issue time)")
    return out


def _dp_cost_model(cfg, world, sharded):
    """What one optimiser step of the multi-GPU form should cost on one node, from single-GPU measurements and the
    guide's xGMI figures: kernel times measured on one MI355X, collective = RCCL small-message latency
    (ring over xGMI, latency-bound at these sizes: the payload is 512 B per rank, or 3*B*d*4 bytes for the row exchange)."""
    elems = (cfg["n"] + cfg["m"]) * cfg["d"]
    sweep_us = 24.0 * elems / 6.0e6                 # fused streaming step: 24 B/element at ~6 TB/s
    if sharded:
        coll_us = 12.0 + 2.0 * max(0, world.bit_length() - 1) + 3 * cfg["B"] * cfg["d"] * 4 / 50e3   # all-reduce of <= 192 rows
        per_step = {"pack_us": 3.0, "collective_us": round(coll_us, 1), "step_us": round(max(2.5, sweep_us / world), 1)}
        samples = cfg["B"]
        # pipelined exchange: pairs of consecutive batches that share no row (uniform triplets: no common user among
        # B x B draws from n, no common item among 2B x 2B draws from m) overlap the collective with pack + step
        free = float(np.exp(-cfg["B"] ** 2 / cfg["n"]) * np.exp(-(2 * cfg["B"]) ** 2 / cfg["m"]))
        serial = sum(per_step.values())
        hop_us = 7.0          # event record -> wait on the other stream, measured on a one-rank group (profiles/r03_shard_chains_one_rank.txt)
        # a free step's collective leaves the main stream and comes back (two hops); consecutive free steps form two
        # interleaved chains apply(k) -> pack_ahead(k+2) -> hop -> collective -> hop -> apply(k+2), each advancing two steps
        overlapped = max(per_step["pack_us"] + per_step["step_us"],
                         (per_step["pack_us"] + per_step["step_us"] + 2 * hop_us + per_step["collective_us"]) / 2.0)
        tot = free * overlapped + (1.0 - free) * serial
        return {"per_step_us": per_step, "free_pairs": round(free, 3), "strict_chain_us": round(serial, 1),
                "stream_hop_us": hop_us, "overlapped_us": round(overlapped, 1), "sum_us": round(tot, 1),
                "predicted_value": round(samples / tot * 1e6, 1),
                "note": "row-sharded loop with the pipelined exchange: where batch k+1 shares no row with batch k "
                        "(free_pairs of the steps for uniform triplets) its all-reduce runs on a side stream under step "
                        "k (two stream hops per free step; two interleaved chains of free steps) and the other pairs keep "
                        "pack -> collective -> step on the main stream.  The collective's latency stays the floor: at C2 one GPU's resident form (0.55 us per "
                        "step) is far below it; the form pays off where the state does not fit one GPU's registers "
                        "(C4: 32 us per step on one GPU)"}
    else:
        coll_us = 10.0 + 2.0 * max(0, world.bit_length() - 1)                                         # all-gather of 512 B per rank
        per_step = {"coefficients_us": 2.5, "collective_us": round(coll_us, 1), "step_us": round(max(3.0, sweep_us), 1)}
        samples = cfg["B"] * world
    tot = sum(per_step.values())
    return {"per_step_us": per_step, "sum_us": round(tot, 1), "predicted_value": round(samples / tot * 1e6, 1),
            "note": "strictly serial chain kernel -> collective -> kernel per optimiser step (no overlap is possible inside "
                    "a step: the step reads what the collective delivers, and the next step's first kernel reads what this "
                    "step writes); the single-GPU register-resident form takes 0.55 us per 64-sample step at C2, so at C2 no "
                    "per-step collective can beat one GPU — the multi-GPU forms pay off where the state does not fit one "
                    "GPU's registers (C4: 32 us per step on one GPU)"}


def _bench_independent_replicas(cfg, dev, steps, warmup, seed):
    """The form in which the reference's OWN workload uses several GPUs (Runs.ipynb: grids x repetitions through
    parameter_scan; structure.scan_over_ranks here): every rank trains its own model on its own data with the single-GPU
    step form, no data-path collective.  Same timing rule as the headline (barrier + synchronize on both sides, max over
    ranks); value = samples all ranks consumed / that time."""
    import time

    import bench as bench_mod
    world, rank = dist.get_world_size(), dist.get_rank()
    runner = bench_mod.Runner(cfg, dev, seed + 7919 * (rank + 1))
    runner.open_epoch()
    runner.run(max(warmup, runner.steps_per_epoch))
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    consumed = runner.run(steps)
    torch.cuda.synchronize()
    dist.barrier()
    t = torch.tensor([time.perf_counter() - t0, float(consumed)], dtype=torch.float64, device=dev)
    tmax = t.clone()
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    runner.bind.flush()
    engine.check_status()
    dt, total = float(tmax[0].item()), float(t[1].item())
    return {"value": round(total / dt, 1), "unit": "triplet-updates/s", "steps": steps, "ms_per_step": round(dt * 1e3 / steps, 6),
            "scaling": "weak", "collectives_on_the_data_path": 0,
            "step_form": ("big-resident" if getattr(runner.bind, "_big", None) and runner.bind._big.ws is not None else
                          engine.train_plan(min(steps, runner.steps_per_epoch) * cfg["B"], cfg["B"], cfg["n"], cfg["m"],
                                            cfg["d"])["form_name"]),
            "what": "one independent training run per rank (own model, own samples, batch 64, the single-GPU step form): "
                    "the way the reference's parameter scans spread over GPUs (structure.scan_over_ranks)"}


def bench_data_parallel(cfg, dev, steps, warmup, seed, mode="native", extras=True):
    """bench.py's N>1 leg: weak scaling, per-rank batch cfg['B'], global batch B*world; returns the JSON dict.
    extras: also probe the per-phase split, attach the cost model and (C2 headline) a `c4` sub-record — BASELINE
    configs[3], the configuration named for 8-GPU data parallelism — in both multi-GPU forms."""
    import time

    import numpy as np

    import bench as bench_mod
    import structure as S
    world, rank = dist.get_world_size(), dist.get_rank()
    tr, va, U0, V0 = bench_mod.make_workload(cfg, seed)      # same seed -> identical data on every rank
    model = S.MatrixFactorization(cfg["n"], cfg["m"], cfg["d"])
    with torch.no_grad():
        model.U.copy_(torch.from_numpy(U0))
        model.V.copy_(torch.from_numpy(V0))
    model = model.to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=cfg["lr"], weight_decay=cfg["wd"])
    binding = engine.AdamBinding(model, opt)
    broadcast_state(binding)
    native = None
    sharded = mode == "shard"
    if mode == "native":
        # RCCL not loadable inside the library is the same on every rank (same image, same process layout): then the
        # per-step torch.distributed loop over the same HIP kernels takes over, and the JSON line says so
        try:
            native = NativeDP(binding)
        except _lib.MfcdError as e:
            if "RCCL" not in str(e):
                raise
            mode = "allgather"
    elif sharded:
        native = NativeShard(binding)       # row-sharded state, batch = B (the reference's), results = one GPU's
    compute = HipCompute(binding) if native is None else None
    train = engine.SampleStore(tr, cfg["n"], cfg["m"], dev)
    val = engine.SampleStore(va, cfg["n"], cfg["m"], dev)
    gen = torch.Generator().manual_seed(seed + 1)           # same permutations on every rank
    B = cfg["B"]
    Bg = B if sharded else B * world
    steps_per_epoch = (train.N + Bg - 1) // Bg
    state = {"stream": None, "pos": 0}

    def run(nsteps):
        consumed = 0
        while nsteps > 0:
            if state["stream"] is None:
                state["stream"], state["pos"] = train.ordered(torch.randperm(train.N, generator=gen)), 0
            take = min(steps_per_epoch - state["pos"], nsteps)
            lo, hi = state["pos"] * Bg, min(train.N, (state["pos"] + take) * Bg)
            if native is not None:
                native.train_steps(state["stream"][lo:hi], B)
            else:
                train_steps_dp(compute, state["stream"][lo:hi], B, mode=mode)
            consumed += hi - lo
            state["pos"] += take
            nsteps -= take
            if state["pos"] == steps_per_epoch:               # validation pass, sharded over ranks by batch
                if sharded:
                    native.gather()                           # the no-grad pass needs every row: all-gather the shards
                vlo = (val.N * rank) // world
                vhi = (val.N * (rank + 1)) // world
                engine.eval_batches(model.U.data, model.V.data, val.dev[vlo:vhi], B)
                state["stream"] = None
        return consumed

    run(warmup)
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    consumed = run(steps)
    torch.cuda.synchronize()
    dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    dt = float(dt.item())
    if sharded:
        native.gather()
    phases = None
    if extras and mode in ("native", "allgather", "shard"):
        probe_steps = 48
        order = train.ordered(torch.randperm(train.N, generator=gen))
        if order.shape[0] >= probe_steps * Bg:
            if sharded and native is not None and native.shard is not None:
                native.gather()
            phases = _probe_dp_phases(binding, order[: probe_steps * Bg], B, probe_steps, sharded)
            if sharded and native is not None and native.shard is not None:
                native.shard = RowShard(binding, rank, world)          # the probe moved the full tensors on
    # replicas must still agree bit for bit
    chk = torch.stack([model.U.data.double().sum(), model.V.data.double().sum()])
    lo_, hi_ = chk.clone(), chk.clone()
    dist.all_reduce(lo_, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi_, op=dist.ReduceOp.MAX)
    in_sync = bool(torch.equal(lo_, hi_))
    if native is not None:
        native.close()
    what = {"native": "native loop in libmfcd_hip.so: one in-place RCCL all-gather of 64 {g, term} pairs per rank per "
                      "optimiser step", "allgather": "torch.distributed all-gather per optimiser step",
            "allreduce": "dense fp32 gradient all-reduce per optimiser step",
            "shard": "row-sharded state in libmfcd_hip.so: one RCCL all-reduce of the batch's <= 192 rows per optimiser "
                     "step, global batch 64, results equal to one GPU"}[mode]
    abytes = bench_mod.algorithmic_bytes_per_step(dict(cfg, B=Bg))
    out = {
        "metric": "triplet-updates/sec", "value": round(consumed / dt, 1), "unit": "triplet-updates/s",
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(dt * 1e3 / steps, 6),
        "higher_is_better": True, "scaling": "strong" if sharded else "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"{cfg.get('name', 'C2')}: n={cfg['n']} m={cfg['m']} d={cfg['d']} p={cfg['p']} K={cfg.get('K', 1)} "
                               f"random triplets, {'global' if sharded else 'per-GPU'} batch 64, Adam lr=1e-3 wd=1e-5, "
                               "validation pass per epoch",
                   "global_batch": Bg, "train_samples": train.N,
                   "parallelism": f"dp{world} ({what})", "multi_gpu_form": mode,
                   "replicas_in_sync": in_sync},
        "roofline": {"bound": "hbm", "achieved": round(abytes / (dt / steps) / 1e9, 1), "peak": bench_mod.HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(abytes / (dt / steps) / 1e9 / bench_mod.HBM_PEAK_GBS, 4),
                     "traffic": None, "algorithmic_bytes_per_launch": abytes,
                     "note": ("per-GPU step period including the collective; every rank sweeps 1/world of the state"
                              if sharded else
                              "per-GPU step period including the collective; every rank streams the full replicated state")},
    }
    if extras:
        out["phase_split"] = phases
        out["cost_model"] = _dp_cost_model(cfg, world, sharded)
        if cfg.get("name", "C2") == "C2":
            # BASELINE configs[3] (C4: 65536^2, d = 64, K = 4) beside the headline, in both multi-GPU forms: short runs.
            # The headline above is complete at this point; the sub-records run forms no multi-GPU hardware has run
            # before (the pipelined row exchange LAST), so a watchdog keeps a stall in them from costing the line.
            finished = _extras_watchdog(out, rank)
            c4 = dict(bench_mod.C4, name="C4")
            sub = {}
            forms = ("native", "shard_strict", "shard") if mode != "allgather" else ("allgather",)
            try:    # what ONE GPU does with C4 on its own (the big resident form, csrc/big.hip), on every rank at once
                r = _bench_independent_replicas(c4, dev, 600, 300, seed)
                sub["independent_replicas"] = {k: r[k] for k in ("value", "unit", "ms_per_step", "steps", "scaling",
                                                                 "collectives_on_the_data_path", "step_form", "what")}
            except Exception as e:
                sub["independent_replicas"] = {"error": f"{type(e).__name__}: {e}"[:200]}
            for form in forms:
                try:
                    engine.set_tuning(shard_pipeline=0 if form == "shard_strict" else 1)
                    r = bench_data_parallel(c4, dev, 200, 20, seed, mode="shard" if form == "shard_strict" else form,
                                            extras=False)
                    sub[form] = {k: r[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup", "scaling")}
                    sub[form]["global_batch"] = r["config"]["global_batch"]
                    sub[form]["replicas_in_sync"] = r["config"]["replicas_in_sync"]
                    cm = _dp_cost_model(c4, world, form.startswith("shard"))
                    sub[form]["cost_model"] = {k: cm[k] for k in ("per_step_us", "sum_us", "predicted_value")}
                    if form == "shard_strict":      # the strict chain is the plain sum of the phases
                        sub[form]["cost_model"]["sum_us"] = cm.get("strict_chain_us", cm["sum_us"])
                        sub[form]["cost_model"]["predicted_value"] = round(c4["B"] / sub[form]["cost_model"]["sum_us"] * 1e6, 1)
                except Exception as e:       # the headline stands on its own
                    sub[form] = {"error": f"{type(e).__name__}: {e}"[:200]}
            try:
                out["independent_replicas"] = _bench_independent_replicas(cfg, dev, steps, warmup, seed)
            except Exception as e:
                out["independent_replicas"] = {"error": f"{type(e).__name__}: {e}"[:200]}
            engine.set_tuning(shard_pipeline=1)
            out["c4"] = {"workload": "BASELINE configs[3]: n=m=65536, d=64, p=0.0005, K=4 (3.4 M training samples), "
                                     "8-GPU data parallel", "forms": sub,
                         "note": "shard_strict = pack -> all-reduce -> step on one stream; shard = the default chain "
                                 "(pipelined exchange when world > 1)"}
            finished.set()
    return out


def _extras_watchdog(out, rank, seconds=None):
    """If the sub-records after the headline stall (a collective that never completes on some rank), rank 0 prints the
    line as it stands — the headline is complete before they start — and every rank leaves.  Returns the event to set
    when the sub-records are done."""
    import json
    import os
    import threading
    seconds = float(os.environ.get("MFCD_BENCH_EXTRAS_TIMEOUT", seconds or 300.0))
    finished = threading.Event()

    def watch():
        if finished.wait(seconds):
            return
        if rank == 0:
            line = {k: v for k, v in list(out.items())}
            line["extras"] = (f"abandoned after {seconds:.0f} s: a sub-record of the multi-GPU leg did not finish; the "
                              "headline fields were complete before it started")
            import sys                         # bench.py (run as __main__) routes fd 1 to stderr while it runs and keeps
            fd = getattr(sys.modules.get("__main__"), "REAL_STDOUT_FD", None)   # the real stdout in REAL_STDOUT_FD
            os.write(fd if fd is not None else 1, (json.dumps(line) + "\n").encode())
        os._exit(0)

    threading.Thread(target=watch, daemon=True).start()
    return finished
