"""CPU / gloo, world_size 2: the data-parallel sharding + collective logic of mfcd.dist with the compute steps
served by the oracle (tests may use it; the product backend is HipCompute).  Checks that both exchange forms give
the single-process result for batch_size = B * world_size, and that the replicas stay bit-identical."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG, ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class OracleCompute:
    """Oracle-backed stand-in for mfcd.dist.HipCompute (numpy state, CPU tensors on the wire)."""

    def __init__(self, U0, V0, lr, wd):
        from oracle import oracle as O
        self.O, self.orc = O, O.COracle()
        self.st = O.new_state(U0, V0)
        self.lr, self.wd, self.step = lr, wd, 0

    @staticmethod
    def _split(rec):
        r = rec.numpy()
        return r[:, 0].astype(np.int64), r[:, 1].astype(np.int64), r[:, 2].astype(np.int64), r[:, 3].copy().view(np.float32)

    def _g(self, rec, divisor):
        u, i, j, z = self._split(rec)
        p, term = self.orc.forward(self.st["U"], self.st["V"], u, i, j, z)
        one = np.float32(1.0)
        den = np.maximum((one - p) * p, np.float32(1e-12))
        a = (np.float32(1.0 / divisor) * (p - z) / den).astype(np.float32)
        return (a * (one - p) * p).astype(np.float32), term

    def _scatter(self, rec, g):
        u, i, j, _ = self._split(rec)
        dU, dV = np.zeros_like(self.st["U"]), np.zeros_like(self.st["V"])
        U, V = self.st["U"], self.st["V"]
        for t in range(len(u)):
            dU[u[t]] += g[t] * (V[i[t]] - V[j[t]])
            dV[i[t]] += g[t] * U[u[t]]
            dV[j[t]] += -(g[t] * U[u[t]])
        return dU, dV

    def _adam(self, dU, dV):
        self.step += 1
        for nm, gr in (("U", dU), ("V", dV)):
            self.orc.adam(self.st[nm], self.st["m" + nm], self.st["v" + nm], gr, self.step, lr=self.lr, wd=self.wd)

    def coefficients(self, rec_local, divisor, out=None):
        g, term = self._g(rec_local, divisor)
        res = torch.from_numpy(np.stack([g, term]))
        if out is not None:
            out[:, : res.shape[1]] = res
        return res

    def apply(self, rec_global, g_global):
        self._adam(*self._scatter(rec_global, g_global.numpy()))

    def dense_grad(self, rec_local, divisor):
        g, term = self._g(rec_local, divisor)
        dU, dV = self._scatter(rec_local, g)
        return torch.from_numpy(np.concatenate([dU.reshape(-1), dV.reshape(-1)])), torch.from_numpy(term)

    def adam_dense(self, grad):
        n, d = self.st["U"].shape
        g = grad.numpy()
        self._adam(g[: n * d].reshape(n, d), g[n * d:].reshape(-1, d))


def _worker(rank, world, port, mode, out_dir):
    import sys
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mfcd import dist as mdist
        from mfcd.batching import pack_records
        rng = np.random.default_rng(5)
        n, m, d, B, N = 40, 30, 8, 16, 16 * 2 * 6 + 11       # short last global batch: rank 0 gets 11, rank 1 none
        U0 = (rng.standard_normal((n, d)) / np.sqrt(d)).astype(np.float32)
        V0 = (rng.standard_normal((m, d)) / np.sqrt(d)).astype(np.float32)
        u, i = rng.integers(0, n, N), rng.integers(0, m, N)
        j = (i + 1 + rng.integers(0, m - 1, N)) % m
        z = rng.integers(0, 2, N).astype(np.float64)
        stream = torch.from_numpy(pack_records(np.stack([u, i, j, z], 1), n, m))
        comp = OracleCompute(U0, V0, 1e-3, 1e-5)
        losses = mdist.train_steps_dp(comp, stream, B, mode=mode)
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), U=comp.st["U"], V=comp.st["V"], mU=comp.st["mU"],
                 vV=comp.st["vV"], losses=losses.numpy(), step=comp.step)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["allgather", "allreduce"])
def test_two_rank_data_parallel_equals_single_process_big_batch(tmp_path, mode, orc):
    from oracle import oracle as O
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), mode, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (dict(np.load(tmp_path / f"r{r}.npz")) for r in range(world))
    for k in ("U", "V", "mU", "vV", "losses"):
        np.testing.assert_array_equal(r0[k], r1[k], err_msg=f"replicas diverged in {k}")
    # single process, batch_size = B * world, same stream
    rng = np.random.default_rng(5)
    n, m, d, B, N = 40, 30, 8, 16, 16 * 2 * 6 + 11
    U0 = (rng.standard_normal((n, d)) / np.sqrt(d)).astype(np.float32)
    V0 = (rng.standard_normal((m, d)) / np.sqrt(d)).astype(np.float32)
    u, i = rng.integers(0, n, N), rng.integers(0, m, N)
    j = (i + 1 + rng.integers(0, m - 1, N)) % m
    z = rng.integers(0, 2, N).astype(np.float64)
    ref = O.new_state(U0, V0)
    ref_loss = orc.train_steps(ref, u, i, j, z, B * world, 0, lr=1e-3, wd=1e-5)
    assert int(r0["step"]) == len(ref_loss) == 7
    np.testing.assert_allclose(r0["losses"], ref_loss, rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(r0["U"], ref["U"], rtol=0, atol=5e-7)
    np.testing.assert_allclose(r0["V"], ref["V"], rtol=0, atol=5e-7)


def test_shard_bounds_cover_each_global_batch_once():
    from mfcd.dist import shard_bounds
    for lo, hi, B, world in ((0, 128, 64, 2), (128, 139, 64, 2), (0, 100, 16, 8), (50, 50, 16, 4)):
        seen = []
        for r in range(world):
            a, b = shard_bounds(lo, hi, B, r)
            assert lo <= a <= b <= hi and b - a <= B
            seen += list(range(a, b))
        assert seen == list(range(lo, min(hi, lo + B * world)))


# ---------------------------------------------------------------------------------------------------------------
# row-sharded state (mfcd.dist.train_steps_sharded): gloo, world 2, oracle-backed halves
# ---------------------------------------------------------------------------------------------------------------
class OracleShardCompute:
    """Oracle-backed stand-in for mfcd.dist.HipShardCompute: this rank's rows of U, V and of the moments as numpy."""

    def __init__(self, U0, V0, rank, world, lr, wd):
        from oracle import oracle as O
        from mfcd.dist import shard_rows
        self.orc = O.COracle()
        (n, self.d), m = U0.shape, V0.shape[0]
        self.u_lo, self.u_hi = shard_rows(n, rank, world)
        self.v_lo, self.v_hi = shard_rows(m, rank, world)
        self.U, self.V = U0[self.u_lo:self.u_hi].copy(), V0[self.v_lo:self.v_hi].copy()
        self.mU, self.vU = np.zeros_like(self.U), np.zeros_like(self.U)
        self.mV, self.vV = np.zeros_like(self.V), np.zeros_like(self.V)
        self.lr, self.wd = lr, wd

    def new_xbuf(self, B):
        return torch.zeros(3 * B * self.d, dtype=torch.float32)

    def pack(self, batch, B, xbuf):
        x = xbuf.numpy().reshape(3, B, self.d)
        x[:] = 0.0
        r = batch.numpy()
        for t in range(r.shape[0]):
            u, i, j = int(r[t, 0]), int(r[t, 1]), int(r[t, 2])
            if self.u_lo <= u < self.u_hi:
                x[0, t] = self.U[u - self.u_lo]
            if self.v_lo <= i < self.v_hi:
                x[1, t] = self.V[i - self.v_lo]
            if self.v_lo <= j < self.v_hi:
                x[2, t] = self.V[j - self.v_lo]

    def pack_ahead(self, next_batch, B, xbuf, step, hyper):
        """mfcd_shard_pack_ahead: the next batch's rows rolled forward over step `step` (zero sparse gradient)."""
        lr, b1, b2, eps, wd = hyper
        x = xbuf.numpy().reshape(3, B, self.d)
        x[:] = 0.0
        r = next_batch.numpy()

        def rolled(P, M1, M2, row):
            p, m1, m2 = P[row:row + 1].copy(), M1[row:row + 1].copy(), M2[row:row + 1].copy()
            self.orc.adam(p, m1, m2, np.zeros_like(p), step, lr=lr, betas=(b1, b2), eps=eps, wd=wd)
            return p[0]
        for t in range(r.shape[0]):
            u, i, j = int(r[t, 0]), int(r[t, 1]), int(r[t, 2])
            if self.u_lo <= u < self.u_hi:
                x[0, t] = rolled(self.U, self.mU, self.vU, u - self.u_lo)
            if self.v_lo <= i < self.v_hi:
                x[1, t] = rolled(self.V, self.mV, self.vV, i - self.v_lo)
            if self.v_lo <= j < self.v_hi:
                x[2, t] = rolled(self.V, self.mV, self.vV, j - self.v_lo)
        self.ahead_calls = getattr(self, "ahead_calls", 0) + 1

    def apply(self, batch, B, xbuf, step, hyper, terms):
        x = xbuf.numpy().reshape(3, B, self.d)
        r = batch.numpy()
        Bk = r.shape[0]
        z = r[:, 3].copy().view(np.float32)
        # forward on the gathered rows (a Bk x d "table" per role, sample t in row t)
        idx = np.arange(Bk)
        p, term = self.orc.forward(np.ascontiguousarray(x[0, :Bk]), np.ascontiguousarray(np.concatenate([x[1, :Bk], x[2, :Bk]])),
                                   idx, idx, idx + Bk, z)
        one = np.float32(1.0)
        den = np.maximum((one - p) * p, np.float32(1e-12))
        g = ((np.float32(1.0 / Bk) * (p - z) / den).astype(np.float32) * (one - p) * p).astype(np.float32)
        dU, dV = np.zeros_like(self.U), np.zeros_like(self.V)
        for t in range(Bk):
            u, i, j = int(r[t, 0]), int(r[t, 1]), int(r[t, 2])
            if self.u_lo <= u < self.u_hi:
                dU[u - self.u_lo] += g[t] * (x[1, t] - x[2, t])
            if self.v_lo <= i < self.v_hi:
                dV[i - self.v_lo] += g[t] * x[0, t]
            if self.v_lo <= j < self.v_hi:
                dV[j - self.v_lo] += -(g[t] * x[0, t])
        lr, b1, b2, eps, wd = hyper
        for prm, m1, m2, gr in ((self.U, self.mU, self.vU, dU), (self.V, self.mV, self.vV, dV)):
            if prm.size:
                self.orc.adam(prm, m1, m2, gr, step, lr=lr, betas=(b1, b2), eps=eps, wd=wd)
        terms[:Bk] = torch.from_numpy(term)


SHARD_CASES = {"dense": (41, 30, 8, 16, 16 * 9 + 5),       # every pair of consecutive batches shares a row
               "sparse": (900, 700, 8, 8, 8 * 40 + 3)}        # most pairs share none: the look-ahead exchange runs


def _shard_inputs(case):
    n, m, d, B, N = SHARD_CASES[case]
    rng = np.random.default_rng(8)
    U0 = (rng.standard_normal((n, d)) / np.sqrt(d)).astype(np.float32)
    V0 = (rng.standard_normal((m, d)) / np.sqrt(d)).astype(np.float32)
    u, i = rng.integers(0, n, N), rng.integers(0, m, N)
    j = (i + 1 + rng.integers(0, m - 1, N)) % m
    z = rng.integers(0, 2, N).astype(np.float64)
    return n, m, d, B, N, U0, V0, u, i, j, z


def _shard_worker(rank, world, port, out_dir, case):
    import sys
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mfcd import dist as mdist
        from mfcd.batching import pack_records
        n, m, d, B, N, U0, V0, u, i, j, z = _shard_inputs(case)
        stream = torch.from_numpy(pack_records(np.stack([u, i, j, z], 1), n, m))
        out = {}
        for tag, pipelined in (("", True), ("_strict", False)):
            comp = OracleShardCompute(U0, V0, rank, world, 1e-3, 1e-5)
            losses = mdist.train_steps_sharded(comp, stream, B, 0, (1e-3, 0.9, 0.999, 1e-8, 1e-5), pipelined=pipelined)
            out.update({"U" + tag: comp.U, "V" + tag: comp.V, "vV" + tag: comp.vV, "losses" + tag: losses.numpy(),
                        "ahead" + tag: np.array(getattr(comp, "ahead_calls", 0))})
        np.savez(os.path.join(out_dir, f"s{rank}.npz"), bounds=np.array([comp.u_lo, comp.u_hi, comp.v_lo, comp.v_hi]),
                 collide=mdist.batch_collisions(stream.numpy(), B), **out)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", ["dense", "sparse"])
def test_two_rank_row_sharded_run_equals_single_process_same_batch(tmp_path, orc, case):
    """World 2 over gloo: each rank holds half the rows of U, V, m, v; the batch stays B (NOT B*world); one int32
    all-reduce of the batch's rows per step.  The concatenated shards must equal the single-process oracle run with
    the SAME batch size, and both ranks must report the same step losses.  Both chains run: the pipelined one (the
    all-reduce of batch k+1 in flight under step k wherever the two batches share no row, its rows rolled forward over
    step k) must equal the strict pack -> all-reduce -> step chain BIT FOR BIT."""
    from oracle import oracle as O
    world = 2
    mp.spawn(_shard_worker, args=(world, _free_port(), str(tmp_path), case), nprocs=world, join=True)
    r0, r1 = (dict(np.load(tmp_path / f"s{r}.npz")) for r in range(world))
    np.testing.assert_array_equal(r0["losses"], r1["losses"])
    n, m, d, B, N, U0, V0, u, i, j, z = _shard_inputs(case)
    assert r0["bounds"].tolist() == [0, n // 2, 0, m // 2] and r1["bounds"].tolist() == [n // 2, n, m // 2, m]
    nsteps = (N + B - 1) // B
    free = int((~r0["collide"][:nsteps - 1]).sum())
    assert int(r0["ahead"]) == free and int(r0["ahead_strict"]) == 0
    assert free == 0 if case == "dense" else free > nsteps // 2
    for key in ("U", "V", "vV", "losses"):
        for r in (r0, r1):
            np.testing.assert_array_equal(r[key], r[key + "_strict"], err_msg=key)
    ref = O.new_state(U0, V0)
    ref_loss = orc.train_steps(ref, u, i, j, z, B, 0, lr=1e-3, wd=1e-5)
    np.testing.assert_allclose(r0["losses"], ref_loss, rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(np.concatenate([r0["U"], r1["U"]]), ref["U"], rtol=0, atol=5e-7)
    np.testing.assert_allclose(np.concatenate([r0["V"], r1["V"]]), ref["V"], rtol=0, atol=5e-7)
    np.testing.assert_allclose(np.concatenate([r0["vV"], r1["vV"]]), ref["vV"], rtol=1e-5, atol=1e-12)


def test_bench_starts_its_own_ranks_and_relays_one_json_line():
    """`python bench.py --gpus 2` outside torch.distributed.run must spawn the ranks itself, print exactly one JSON
    line (rank 0's) on stdout and exit 0; a failing child must give a non-zero exit.  Uses the
    launcher self-test mode (gloo rendezvous on CPU), so no GPU is touched."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    ok = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dp-mode", "selftest"],
                        env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert ok.returncode == 0, ok.stderr[-2000:]
    lines = [ln for ln in ok.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["sum_of_ranks_plus_one"] == 3.0
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dp-mode", "selftest",
                          "--workload", "no-such-workload"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         text=True, timeout=300)
    assert bad.returncode != 0


def _scan_worker(rank, world, port, out_dir):
    import sys
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import structure as S
        ran = []

        def fake_experiment(**kw):      # stands in for the GPU experiment: the scan logic is what is under test
            ran.append(kw["weight_decay"])
            return {"accuracy": [kw["weight_decay"] * 10.0], "device": kw["device"], "rank": rank}

        S.run_experiment, real = fake_experiment, S.run_experiment
        try:
            path = os.path.join(out_dir, "scan.pkl")
            got = S.parameter_scan(n=8, m=8, d=2, p=0.5, device="cuda:7", weight_decay=[1.0, 2.0, 3.0, 4.0, 5.0],
                                   save_path=path, save_every=2)
            in_memory = S.parameter_scan(n=8, m=8, d=2, p=0.5, device="cuda:7", weight_decay=[1.0, 2.0, 3.0])
        finally:
            S.run_experiment = real
        import pickle
        with open(os.path.join(out_dir, f"scan_r{rank}.pkl"), "wb") as f:
            pickle.dump({"ran": ran, "returned": got, "in_memory": in_memory}, f)
    finally:
        dist.destroy_process_group()


def test_parameter_scan_spreads_experiments_over_ranks_and_rank0_collects_in_order(tmp_path):
    """experiments of a scan are independent replicas — rank r runs experiments r, r+R, ...; rank 0 writes the
    .pkl in the serial order (reference layout, structure.py:183-184) and the other ranks return []."""
    import pickle
    world = 2
    mp.spawn(_scan_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [pickle.load(open(tmp_path / f"scan_r{k}.pkl", "rb")) for k in range(world)]
    assert r[0]["ran"] == [1.0, 3.0, 5.0, 1.0, 3.0] and r[1]["ran"] == [2.0, 4.0, 2.0]
    assert r[0]["returned"] == [] and r[1]["returned"] == [] and r[1]["in_memory"] == []
    saved = pickle.load(open(tmp_path / "scan.pkl", "rb"))
    assert [e["params"]["weight_decay"] for e in saved] == [1.0, 2.0, 3.0, 4.0, 5.0]
    assert [e["results"]["rank"] for e in saved] == [0, 1, 0, 1, 0]
    assert all(set(e) == {"params", "results"} for e in saved)
    mem = r[0]["in_memory"]
    assert [e["params"]["weight_decay"] for e in mem] == [1.0, 2.0, 3.0] and [e["results"]["rank"] for e in mem] == [0, 1, 0]


def _np_slab_pass(U, V, Xs, row0, s, what):
    """f64 restatement of one row slab of the UV^T pass (layout of include/mfcd.h: mfcd_uvt_stats_slab), for the
    sharding logic under test."""
    M = U.double() @ V.double().t()
    cm = M.mean(0)
    k = Xs.shape[0]
    Ms, Xd = M[row0:row0 + k], Xs.double()
    a, c = Ms - Ms.mean(1, keepdim=True), Xd - Xd.mean(1, keepdim=True)
    z = torch.zeros(k, dtype=torch.float64)
    rs = torch.stack([(a * c).sum(1), (a * a).sum(1), (c * c).sum(1), Ms.mean(1), Xd.mean(1), (Xd * Xd).sum(1), z, z], 1)
    share = torch.tensor([float((((Ms - cm) - s * Xd) ** 2).sum()), float(s * s * (Xd * Xd).sum()), 0.0, 0.0], dtype=torch.float64)
    return (rs if what & 1 else None), (share if what & 2 else None)


def _uvt_worker(rank, world, port, out_dir):
    import sys
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mfcd import dist as mdist
        g = torch.Generator().manual_seed(3)
        n, m, d = 50, 40, 8
        U, V, X = torch.randn(n, d, generator=g), torch.randn(m, d, generator=g), torch.randn(n, m, generator=g)
        cuts = [0, 31, 31, 50][: world + 1] if world == 3 else [0, 31, 50]     # world 3: the middle rank owns no row
        lo, hi = cuts[rank], cuts[rank + 1]
        rs, sc = mdist.uvt_stats_sharded(U, V, X[lo:hi], lo, 0.7, what=3, group=None, slab_pass=_np_slab_pass)
        err = mdist.reconstruction_error_sharded(U, V, X[lo:hi], lo, 0.7, slab_pass=_np_slab_pass)
        _, only = mdist.uvt_stats_sharded(U, V, X[lo:hi], lo, 0.7, what=2, slab_pass=_np_slab_pass)
        torch.save({"rs": rs, "sc": sc, "err": err, "only": only}, os.path.join(out_dir, f"uvt_r{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_row_block_sharded_uvt_pass_assembles_the_single_pass_result(tmp_path, world):
    """SURVEY 8e G1, eval pass: X in row blocks over the ranks (one rank may own none), shares summed in rank order, row
    blocks gathered in row order; every rank ends with the same full result = one pass over all rows."""
    mp.spawn(_uvt_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    g = torch.Generator().manual_seed(3)
    n, m, d = 50, 40, 8
    U, V, X = torch.randn(n, d, generator=g), torch.randn(m, d, generator=g), torch.randn(n, m, generator=g)
    rs_ref, sc_ref = _np_slab_pass(U, V, X, 0, 0.7, 3)
    for r in range(world):
        got = torch.load(tmp_path / f"uvt_r{r}.pt")
        assert got["rs"].shape == (n, 8)
        torch.testing.assert_close(got["rs"], rs_ref, rtol=1e-12, atol=1e-12)
        torch.testing.assert_close(got["sc"], sc_ref, rtol=1e-12, atol=0)
        torch.testing.assert_close(got["only"], got["sc"], rtol=0, atol=0)
        assert got["err"] == pytest.approx(float(torch.sqrt(sc_ref[0] / sc_ref[1])), rel=1e-12)
