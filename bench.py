#!/usr/bin/env python3
"""bench.py — triplet-updates/s of the fused MI355X training step at BASELINE.json's C2
(n=m=4096, d=64, p=0.01, random triplets, fp32, B=64, Adam lr=1e-3 wd=1e-5).

  python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

A "step" is one optimiser step = one pass of the hot path over one batch of 64 synthetic triplets.
Steps are consumed the way train_model consumes them: epoch by epoch (1049 steps per epoch at C2, the
last batch short), a fresh permutation per epoch, and the no-grad validation pass after every complete
epoch.  Inputs are resident in HBM when the timed region starts.
Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "matrix-factorization-with-comparison-data_amd")
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

os.environ.setdefault("OMP_NUM_THREADS", "4")  # the reference's own setting (structure.py:3)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)

C2 = dict(n=4096, m=4096, d=64, p=0.01, s=1.0, K=1, B=64, lr=1e-3, wd=1e-5)
# BASELINE.json configs[3], the configuration named for data parallelism (rehearsal only: --workload C4)
C4 = dict(n=65536, m=65536, d=64, p=0.0005, s=1.0, K=4, B=64, lr=1e-3, wd=1e-5)
WORKLOADS = {"C2": C2, "C4": C4}


def make_workload(cfg, seed):
    """Synthetic C2 inputs: rank-d "base"-law X kept factored (X = A B^T, entry std ~0.5),
    int(n*m*p/2) unique uniform triplets, hard BTL labels, 80/10/10 split, U,V ~ N(0, 1/d)."""
    n, m, d = cfg["n"], cfg["m"], cfg["d"]
    g = torch.Generator().manual_seed(seed)
    import generation_data as gd
    A, Bf = gd.generate_embedding_factors(n, m, d, "cpu", generator=g)
    rng = np.random.default_rng(seed)
    want = int(n * m * cfg["p"] / 2)
    keys = np.empty(0, dtype=np.int64)
    while keys.size < want:
        u = rng.integers(0, n, want)
        i = rng.integers(0, m, want)
        j = rng.integers(0, m, want)
        ok = i != j
        k = (u[ok].astype(np.int64) * m + i[ok]) * m + j[ok]
        keys = np.unique(np.concatenate([keys, k]))
    keys = rng.permutation(keys)[:want]
    u, i, j = keys // (m * m), (keys // m) % m, keys % m
    A, Bf = A.numpy().astype(np.float64), Bf.numpy().astype(np.float64)
    diff = np.einsum("td,td->t", A[u], Bf[i] - Bf[j])
    prob = 1.0 / (1.0 + np.exp(-cfg["s"] * diff))
    n_tr, n_va = int(0.8 * want), int(0.1 * want)
    K = int(cfg.get("K", 1))

    def labelled(lo, hi, reps):   # K independent hard labels per training triplet (structure.py:493-519, soft_label=False)
        idx = np.repeat(np.arange(lo, hi), reps)
        z = (rng.random(idx.size) < prob[idx]).astype(np.float64)
        return np.stack([u[idx], i[idx], j[idx], z], 1).astype(np.float64)

    U0 = (torch.randn(n, d, generator=g) / np.sqrt(d)).numpy()
    V0 = (torch.randn(m, d, generator=g) / np.sqrt(d)).numpy()
    return labelled(0, n_tr, K), labelled(n_tr, n_tr + n_va, 1), U0, V0


def algorithmic_bytes_per_step(cfg):
    """24*(n+m)*d [p,m,v read+write] + 12*B*d [three gathered rows] + 16*B [records]."""
    return 24 * (cfg["n"] + cfg["m"]) * cfg["d"] + 12 * cfg["B"] * cfg["d"] + 16 * cfg["B"]


class Runner:
    """Consumes optimiser steps exactly like mfcd.engine.fit, but in step-counted slices."""

    def __init__(self, cfg, dev, seed):
        import structure as S
        from mfcd import engine
        self.engine, self.cfg, self.dev = engine, cfg, dev
        tr, va, U0, V0 = make_workload(cfg, seed)
        model = S.MatrixFactorization(cfg["n"], cfg["m"], cfg["d"])
        with torch.no_grad():
            model.U.copy_(torch.from_numpy(U0))
            model.V.copy_(torch.from_numpy(V0))
        self.model = model.to(dev)
        self.opt = torch.optim.Adam(self.model.parameters(), lr=cfg["lr"], weight_decay=cfg["wd"])
        self.bind = engine.AdamBinding(self.model, self.opt)
        self.train = engine.SampleStore(tr, cfg["n"], cfg["m"], dev)
        self.val = engine.SampleStore(va, cfg["n"], cfg["m"], dev)
        self.gen = torch.Generator().manual_seed(seed + 1)
        self.steps_per_epoch = (self.train.N + cfg["B"] - 1) // cfg["B"]
        self.stream, self.pos = None, 0
        self.pre = engine.StreamPrefetch(dev)   # next epoch's record stream is built under the running step kernel
        self.train_events = []  # (start, stop, launches) around each fused-step call in the timed region

    def run(self, steps, record=False):
        """Enqueue `steps` optimiser steps (+ a validation pass after every completed epoch). No host sync.
        Returns the number of training samples consumed."""
        B, consumed = self.cfg["B"], 0
        while steps > 0:
            if self.stream is None:
                if self.pre.pending is None:
                    self.pre.start(self.train, torch.randperm(self.train.N, generator=self.gen))
                self.stream, self.pos = self.pre.take(), 0
            left = self.steps_per_epoch - self.pos
            take = min(left, steps)
            lo, hi = self.pos * B, min(self.train.N, (self.pos + take) * B)
            if record:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            self.engine.train_steps(self.bind, self.stream[lo:hi], B)
            if record:
                e1.record()
                self.train_events.append((e0, e1, take))
            consumed += hi - lo
            self.pos += take
            steps -= take
            if self.pos == self.steps_per_epoch and self.pre.pending is None:
                # the epoch's last steps are enqueued: start building the next epoch's stream underneath them
                self.pre.start(self.train, torch.randperm(self.train.N, generator=self.gen))
            if self.pos == self.steps_per_epoch:  # structure.py:858-868
                self.engine.eval_batches(self.model.U.data, self.model.V.data, self.val.dev, B)
                self.stream = None
        return consumed

    def kernel_sample(self, launches=512):
        """[avg, min, max] µs of single step-kernel launches, each bracketed by its own HIP event pair."""
        B = self.cfg["B"]
        order = torch.randperm(self.train.N, generator=self.gen)
        stream = self.train.ordered(order)[: launches * B]
        out = [0.0, 0.0, 0.0]
        self.engine.train_steps(self.bind, stream, B, kernel_us=out)
        return out


def cpu_baseline(cfg, seed, budget_s=12.0):
    """The CPU oracle ("port") timed on this box's host cores on a bounded sample of the same workload."""
    import shutil
    import tempfile
    from oracle import oracle as O
    from oracle import torch_port
    tr, _, U0, V0 = make_workload(cfg, seed)
    B = cfg["B"]
    tmp = tempfile.mkdtemp(prefix="mfcd_orc_")
    try:
        try:
            lib = O.build(force=True, archflags="-march=native", out=os.path.join(tmp, "liborc_native.so"))
        except Exception:
            lib = O.build()
        orc = O.COracle(lib)
        ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        results = {}
        for threads in sorted({1, min(ncpu, 16)}):
            st = O.new_state(U0, V0)
            probe = tr[: 32 * B]
            t0 = time.perf_counter()
            orc.train_steps(st, probe[:, 0], probe[:, 1], probe[:, 2], probe[:, 3], B, 0, lr=cfg["lr"], wd=cfg["wd"],
                            threads=threads)
            per = (time.perf_counter() - t0) / 32
            nsteps = int(max(64, min(200000, (budget_s / 2) / max(per, 1e-6))))
            reps = (nsteps * B + len(tr) - 1) // len(tr)
            sample = np.tile(tr, (reps, 1))[: nsteps * B]   # several epochs' worth of the same workload
            t0 = time.perf_counter()
            orc.train_steps(st, sample[:, 0], sample[:, 1], sample[:, 2], sample[:, 3], B, 32, lr=cfg["lr"],
                            wd=cfg["wd"], threads=threads)
            dt = time.perf_counter() - t0
            results[threads] = (len(sample) / dt, nsteps, dt)
        best = max(results, key=lambda k: results[k][0])
        # torch-op port at the reference's own OMP_NUM_THREADS=4 (structure.py:3), ~3 s
        old = torch.get_num_threads()
        torch.set_num_threads(min(4, ncpu))
        U, V = torch.from_numpy(U0.copy()), torch.from_numpy(V0.copy())
        state = {k: torch.zeros_like(U if k.endswith("U") else V) for k in ("mU", "vU", "mV", "vV")}
        nst = 600
        smp = tr[: nst * B]
        u, i, j = (torch.from_numpy(smp[:, c].astype(np.int64)) for c in range(3))
        z = torch.from_numpy(smp[:, 3].copy())
        torch_port.train_steps(U, V, state, u[:64 * 20], i[:64 * 20], j[:64 * 20], z[:64 * 20], B, 0)
        t0 = time.perf_counter()
        torch_port.train_steps(U, V, state, u, i, j, z, B, 20)
        tp = len(smp) / (time.perf_counter() - t0)
        torch.set_num_threads(old)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    v, nsteps, dt = results[best]
    return {"value": round(v, 1), "unit": "triplet-updates/s", "cores": int(best), "kind": "port",
            "sample": f"{nsteps} optimiser steps (B={B}) of the {cfg.get('name', 'C2')} workload in {dt:.1f}s, C oracle "
                      f"(oracle/mfcd_oracle.c, -O3 -march=native, OpenMP Adam sweep); host has {ncpu} usable cores",
            "by_threads": {str(k): round(r[0], 1) for k, r in results.items()},
            "torch_op_port_4thr": round(tp, 1)}


class _StdoutToStderr:
    """Route fd 1 to stderr while the benchmark runs (RCCL prints a version banner on stdout) so that the ONE JSON
    line is the only thing this script ever writes to stdout."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def main():
    with _StdoutToStderr():
        out, is_printer = _run()
    if is_printer:
        print(json.dumps(out), flush=True)


def _run():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10490)   # 10 epochs of C2
    ap.add_argument("--warmup", type=int, default=1049)   # 1 epoch
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dp-mode", choices=["native", "allgather", "allreduce"], default=None,
                    help="form of the data-parallel path (default native: the loop inside libmfcd_hip.so with one RCCL "
                         "all-gather per step; allgather / allreduce: the per-step torch.distributed loops); given "
                         "with --gpus 1 it rehearses that path on a one-rank group")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="C2",
                    help="C2 (default): the configuration the metric is quoted on; C4: BASELINE.json configs[3], for "
                         "rehearsing the data-parallel path at the size it is named for")
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if args.gpus > 1:
            raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cfg = dict(WORKLOADS[args.workload])
    cfg["name"] = args.workload

    if world > 1 or args.dp_mode:
        import torch.distributed as dist
        from mfcd import dist as mdist
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        else:
            dist.init_process_group("nccl", device_id=dev)
        out = mdist.bench_data_parallel(cfg, dev, args.steps, args.warmup, args.seed, mode=args.dp_mode or "native")
        dist.destroy_process_group()
        return out, rank == 0

    runner = Runner(cfg, dev, args.seed)
    runner.run(args.warmup)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    consumed = runner.run(args.steps, record=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    runner.engine.check_status()   # a resident launch that gave up on a bounded wait would invalidate the number

    from mfcd import _lib
    plan_resident = (cfg["d"] & (cfg["d"] - 1)) == 0 and 2 <= cfg["d"] <= 256 and (cfg["n"] + cfg["m"]) * cfg["d"] <= 2097152
    kernel_name = ("resident_train_kernel<D=64,Q=2,LOOK=4,fast> (persistent: one launch per epoch; figures are per optimiser "
                   "step = launch time / steps)") if plan_resident else "train_step_kernel (one launch per optimiser step)"
    launches = sum(k for _, _, k in runner.train_events)
    train_ms = sum(a.elapsed_time(b) for a, b, _ in runner.train_events)
    period_us = train_ms * 1e3 / max(launches, 1)          # launch-to-launch, gaps included
    kavg, kmin, kmax = runner.kernel_sample()
    abytes = algorithmic_bytes_per_step(cfg)
    achieved = abytes / (period_us * 1e-6) / 1e9
    traffic = None   # HBM bytes per optimiser step from rocprofv3 --pmc passes of this same command (profiles/)
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            traffic = json.load(f).get("hbm_bytes_per_step")
    except Exception:
        pass
    out = {
        "metric": "triplet-updates/sec", "value": round(consumed / dt, 1), "unit": "triplet-updates/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt * 1e3 / args.steps, 6),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{cfg['name']}: n={cfg['n']} m={cfg['m']} d={cfg['d']} p={cfg['p']} K={cfg['K']} random triplets, "
                               f"B={cfg['B']}, Adam lr=1e-3 wd=1e-5, {runner.steps_per_epoch} steps/epoch + validation "
                               "pass per epoch", "global_batch": cfg["B"],
                   "train_samples": runner.train.N, "parallelism": "single"},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                     "algorithmic_bytes_per_launch": abytes, "launch_period_us": round(period_us, 3),
                     "kernel_us_event_pairs": {"avg": round(kavg, 3), "min": round(kmin, 3), "max": round(kmax, 3)},
                     "kernel": kernel_name},
    }
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(cfg, args.seed)
    return out, True


if __name__ == "__main__":
    main()
