#!/usr/bin/env python3
"""Golden vectors for the seeded triplet samplers (TEST INFRASTRUCTURE — runs in the dev container only).

Imports the *unmodified* reference module /root/reference/generation_data.py, runs its per-attempt sampler loops
(generation_data.py:16-26 random, 29-43 proximity, 87-99 variance, 103-128 popularity, 189-224 top_k, 229-247 cluster) under fixed torch / numpy seeds
and stores, per case, the inputs, the returned list IN ITS ORDER (the 80/10/10 split indexes into it,
structure.py:705-718) and one draw from each global generator taken right after the call (so a test can tell that the
vectorised forms leave both generators where the loops do).  Only data is committed (tests/golden/samplers.npz).

Usage:  OMP_NUM_THREADS=4 PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_samplers.py
"""
import os
import sys

os.environ.setdefault("OMP_NUM_THREADS", "4")
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402
import torch  # noqa: E402

import generation_data as RG  # noqa: E402  (the reference)

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "samplers.npz")

CASES = [
    # name, strategy, n, m, num_triplets, excluded (drawn first with the same strategy), kwargs
    ("random_a", "random", 24, 40, 300, 0, {}),
    ("random_excl", "random", 9, 7, 120, 60, {}),
    ("proximity_a", "proximity", 24, 40, 300, 0, {}),
    ("proximity_k8_excl", "proximity", 16, 20, 200, 80, {"k": 8}),
    ("proximity_wide", "proximity", 12, 260, 500, 0, {}),
    ("popularity_zipf", "popularity", 24, 40, 300, 50, {"method": "zipf", "alpha": 1.5}),
    ("popularity_exp", "popularity", 10, 30, 150, 0, {"method": "exponential", "alpha": 0.2}),
    ("top_k_a", "top_k", 24, 40, 150, 0, {}),
    ("top_k_short", "top_k", 6, 30, 200, 20, {}),
    ("top_k_k3", "top_k", 20, 12, 60, 10, {"k": 3}),
    ("variance_a", "variance", 16, 24, 120, 30, {}),
    ("cluster_a", "cluster", 30, 40, 150, 20, {"n_clusters": 6}),
]
FN = {"random": RG.choose_items_random, "proximity": RG.choose_items_by_proximity,
      "popularity": RG.choose_items_by_popularity, "top_k": RG.choose_items_top_k,
      "variance": RG.choose_items_by_variance, "cluster": RG.choose_items_cluster_based}


def as_rows(trips):
    return np.asarray([[int(u), int(i), int(j)] for u, i, j in trips], dtype=np.int64).reshape(-1, 3)


def main():
    out = {"names": np.asarray([c[0] for c in CASES])}
    for seed, (name, strategy, n, m, want, n_excl, kw) in enumerate(CASES):
        torch.manual_seed(1000 + seed)
        np.random.seed(1000 + seed)
        X = torch.randn(n, m)
        excl = FN[strategy](X, n_excl, set(), **kw) if n_excl else []
        excl_set = set((int(u), int(i), int(j)) for u, i, j in excl)
        torch.manual_seed(2000 + seed)
        np.random.seed(2000 + seed)
        got = FN[strategy](X, want, excl_set, **kw)
        out[f"{name}.X"] = X.numpy().copy()
        out[f"{name}.exclude"] = as_rows(excl)
        out[f"{name}.triplets"] = as_rows(got)
        out[f"{name}.after"] = np.asarray([float(torch.rand(1, dtype=torch.float64)), float(np.random.random_sample())])
        out[f"{name}.meta"] = np.asarray([n, m, want, 2000 + seed, kw.get("k", kw.get("n_clusters", -1))], dtype=np.int64)
        out[f"{name}.alpha"] = np.asarray([kw.get("alpha", 0.0)])
        out[f"{name}.method"] = np.asarray(kw.get("method", ""))
        out[f"{name}.strategy"] = np.asarray(strategy)
        print(name, len(got))
    np.savez_compressed(OUT, **out)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
