"""Experiment: is the resident B = 64 step bound by the sample stream's dependency chain?  Same shape (C2: n = m = 4096,
d = 64, B = 64, one epoch of 1049 steps per launch), same number of hits per step, three streams whose chains differ:
  uniform   triplets drawn uniformly (what bench.py times): ~170 dependent samples per epoch
  spread    every row is named again only 32 (items) / 64 (users) steps later: chains of ~1049/32 links
  ladder    the first sample of every step shares one row with the first sample of the step before, users and items
            alternating, the other 63 samples spread: one chain of 1049 links, a different pair of waves at every link
and the time of a second epoch call over each (HIP events around the call).  If the launch is bound by vector issue the
three run alike; if by the chain, time grows with the chain depth at one hand-off latency per link.
python tools/exp_chain_depth.py > profiles/rNN_chain_depth_experiment.txt"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "matrix-factorization-with-comparison-data_amd")]
os.environ.setdefault("OMP_NUM_THREADS", "4")
import numpy as np, torch
import structure as S
from mfcd import engine

n = m = 4096; d = 64; B = 64; K = 1049
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)


def chain_depth(u, i, j):
    depth = np.zeros(n + m, dtype=np.int64)
    longest = 0
    for k in range(K):
        a, b, c = u[k], i[k] + n, j[k] + n
        dd = np.maximum(np.maximum(depth[a], depth[b]), depth[c]) + 1
        new = depth.copy()
        np.maximum.at(new, np.concatenate([a, b, c]), np.concatenate([dd, dd, dd]))
        depth = new
        longest = max(longest, int(dd.max()))
    return longest


def uniform():
    u = rng.integers(0, n, (K, B)); i = rng.integers(0, m, (K, B))
    return u, i, (i + 1 + rng.integers(0, m - 1, (K, B))) % m


def spread():
    t = np.arange(K * B).reshape(K, B)
    return t % n, (2 * t) % m, (2 * t + 1) % m


def ladder():
    u, i, j = spread()
    u, i, j = u.copy(), i.copy(), j.copy()
    # rows far away from what the spread samples of the neighbouring steps name
    for k in range(K):
        u[k, 0] = (n // 2 + 64 * (k // 2) + 17) % n if k % 2 == 0 else u[k - 1, 0]     # odd steps share the USER with step k-1
        i[k, 0] = (m // 2 + 128 * ((k + 1) // 2) + 33) % m if k % 2 == 1 else (i[k - 1, 0] if k else 5)   # even steps share the ITEM
        j[k, 0] = (i[k, 0] + m // 3) % m
    return u, i, j


print(f"C2 shape, B = {B}, {K} steps per launch; us per step of the second epoch call (HIP events), 3 repetitions")
for name, make in (("uniform", uniform), ("spread", spread), ("ladder", ladder)):
    u, i, j = make()
    depth = chain_depth(u, i, j)
    rows = np.stack([u.reshape(-1), i.reshape(-1), j.reshape(-1), rng.integers(0, 2, K * B)], 1).astype(np.float64)
    model = S.MatrixFactorization(n, m, d).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-5)
    bind = engine.AdamBinding(model, opt)
    st = engine.SampleStore(rows, n, m, dev)
    engine.train_steps(bind, st.dev, B); torch.cuda.synchronize()
    times = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); engine.train_steps(bind, st.dev, B); e1.record(); torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) * 1e3 / K)
    engine.check_status()
    plan = engine.train_plan(K * B, B, n, m, d)["form_name"]
    print(f"{name:8s} chain depth {depth:5d}   {min(times):.3f} / {sorted(times)[1]:.3f} / {max(times):.3f} us per step   "
          f"({min(times) * K:.0f} us per launch, form {plan})", flush=True)
