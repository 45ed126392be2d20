#!/usr/bin/env python3
"""Triplet samplers at BASELINE sizes: host forms (reference draws replayed in bulk) against the device sampler
(mfcd_sample_triplets).  Usage on the GPU box: python tools/bench_samplers.py > profiles/rNN_samplers.txt"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "matrix-factorization-with-comparison-data_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import generation_data as gd  # noqa: E402
from mfcd import sampling  # noqa: E402

dev = torch.device("cuda", 0)
CASES = [("C2 random", "random", 4096, 4096, 64, 83886), ("C4 random", "random", 65536, 65536, 64, 1073741),
         ("C3 margin", "margin", 16384, 16384, 128, 134217), ("C5 popularity", "popularity", 100000, 20000, 256, 500000),
         ("C2 proximity", "proximity", 4096, 4096, 64, 83886), ("C2 top_k", "top_k", 4096, 4096, 64, 83886)]
HOST = {"random": gd.choose_items_random, "margin": gd.choose_items_by_margin,
        "popularity": gd.choose_items_by_popularity, "proximity": gd.choose_items_by_proximity,
        "top_k": gd.choose_items_top_k}
print("strategy: triplets kept, seconds (host bulk form | device: law set-up + attempts, second call)")
for name, strategy, n, m, d, want in CASES:
    torch.manual_seed(0)
    np.random.seed(0)
    A, B = gd.generate_embedding_factors(n, m, d, "cpu", generator=torch.Generator().manual_seed(1))
    FX = gd.FactoredMatrix(A, B)
    X = FX if strategy in ("random", "margin", "popularity") else FX.dense()
    t0 = time.time()
    h = HOST[strategy](X, want, set())
    th = time.time() - t0
    Xd = X if strategy in ("random", "margin", "popularity") else X.to(dev)
    times = []
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.time()
        r = sampling.sample_triplets(Xd, want, strategy, None, device=dev, seed=rep)
        torch.cuda.synchronize()
        times.append(time.time() - t0)
    print(f"{name:14s} host {len(h):8d} in {th:7.3f} s | device {r.shape[0]:8d} in {times[0]:7.3f} s, {times[1]:7.3f} s",
          flush=True)
