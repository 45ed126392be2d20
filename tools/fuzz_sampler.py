"""Diagnostic: randomised sweep of the device triplet sampler (mfcd_sample_triplets through mfcd/sampling.py): random
shapes, strategies, request sizes (up to and beyond the support), exclusion sets and seeds; checks the structural
contract of the reference's loops — distinct, i != j, in range, inside the strategy's candidate sets, nothing from
`exclude`, a shorter request is a prefix of a longer one (attempt order), the whole support is reachable, an impossible
request raises.  python tools/fuzz_sampler.py [trials] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "matrix-factorization-with-comparison-data_amd")]
os.environ.setdefault("OMP_NUM_THREADS", "4")
import numpy as np, torch
from mfcd import sampling

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")
bad = 0
for t in range(trials):
    strategy = str(rng.choice(["random", "popularity", "variance", "proximity", "top_k", "margin"]))
    n, m = int(rng.integers(1, 400)), int(rng.integers(2, 300))
    X = torch.from_numpy(rng.standard_normal((n, m)).astype(np.float32))
    Xd = X.to(dev)
    kw, k = {}, None
    if strategy in ("proximity", "top_k") and rng.random() < 0.6:
        k = int(rng.integers(2, min(m, 40) + 1))
        kw["k"] = k
    if strategy == "popularity":
        kw.update(popularity_method=str(rng.choice(["zipf", "exponential", "uniform"])), alpha=float(rng.choice([0.1, 1.0, 1.5])))
    if strategy == "margin":
        kw["max_attempts"] = int(rng.choice([5000, 50000]))
    # support sizes (an upper bound for margin / proximity with overlapping lists)
    kk = min(100 if k is None else k, m) if strategy == "proximity" else (k or min(m, max(5, int(0.1 * m))))
    support = {"random": n * m * (m - 1), "popularity": n * m * (m - 1), "variance": n * m * (m - 1),
               "proximity": n * kk * kk, "top_k": n * kk * (kk - 1), "margin": n * m * (m - 1)}[strategy]
    want = int(min(rng.integers(1, 3000), max(1, support // int(rng.choice([1, 2, 8, 50])))))
    seed = int(rng.integers(0, 2 ** 62))
    msg = []
    try:
        a = sampling.sample_triplets(Xd, want, strategy, None, device=dev, seed=seed, **kw).cpu().numpy()
    except RuntimeError as e:
        a = None
        if not (strategy == "variance" and n == 1):                  # one user: NaN variances, torch.multinomial raises too
            msg.append(f"unexpected RuntimeError: {e}"[:120])
    except ValueError as e:
        a = None
        if strategy == "random":       # uniform law: every triplet of the support is reachable; the other laws have
            msg.append(f"unexpected ValueError: {e}"[:120])       # overlapping lists / vanishing tails (exp(-1.5 i)): the
                                                                  # reference would spin forever there, this build says so
    if a is not None:
        S = {tuple(r) for r in a.tolist()}
        if len(S) != a.shape[0] or a.shape[0] > want:
            msg.append("duplicates / too many")
        if a.size and (a.min() < 0 or a[:, 0].max() >= n or a[:, 1:].max() >= m or (a[:, 1] == a[:, 2]).any()):
            msg.append("range / i == j")
        if strategy in ("random", "popularity", "variance") and a.shape[0] != want:
            msg.append(f"short: {a.shape[0]} of {want}")
        if strategy in ("proximity", "top_k") and a.size:
            best = torch.topk(X, kk, dim=1)[1].numpy()
            worst = torch.topk(-X, kk, dim=1)[1].numpy()
            in_i = (best[a[:, 0]] == a[:, 1:2]).any(1)
            in_j = ((worst if strategy == "proximity" else best)[a[:, 0]] == a[:, 2:3]).any(1)
            if not (in_i.all() and in_j.all()):
                msg.append("outside the user's lists")
        if strategy == "margin" and a.size:
            head = X[:10].numpy()
            margin = np.mean(head.max(1) - head.min(1)) * want / (n * m)
            if not (np.abs(X.numpy()[a[:, 0], a[:, 1]] - X.numpy()[a[:, 0], a[:, 2]]) <= margin).all():
                msg.append("outside the margin")
        if a.shape[0] >= 2 and strategy != "margin":              # (the margin depends on the request size)
            half = a.shape[0] // 2
            b = sampling.sample_triplets(Xd, half, strategy, None, device=dev, seed=seed, **kw).cpu().numpy()
            if not np.array_equal(b, a[:half]):
                msg.append("a shorter request is not a prefix")
        if a.shape[0] >= 4:
            excl = {tuple(r) for r in a[: a.shape[0] // 2].tolist()}
            try:
                c = sampling.sample_triplets(Xd, min(want, max(1, (support - len(excl)) // 4)), strategy, excl, device=dev,
                                             seed=seed + 1, **kw).cpu().numpy()
                if {tuple(r) for r in c.tolist()} & excl:
                    msg.append("exclude violated")
            except ValueError:
                pass
    if msg:
        bad += 1
        print(f"trial {t}: {strategy} n={n} m={m} want={want} kw={kw}: " + "; ".join(msg[:3]), flush=True)
    if (t + 1) % 100 == 0:
        print(f"... {t + 1} trials, {bad} bad", flush=True)
print(f"done: {trials} trials, {bad} bad")
