"""Depth of the dependency chain a run of K optimiser steps carries at the C2 shape (n = m = 4096, B = 64, uniformly
drawn triplets): a sample's update reads three rows as the previous step left them, so it hangs on the latest earlier
sample that named any of them.  The resident kernel hands a re-touched row from wave to wave once per link of that chain
(publish -> visible -> poll: tools/diag_short_call_stamps.py measures ~2.7 us per link), which bounds short calls.
No GPU needed.  Usage: sim_chain_depth.py [n m B]"""
import sys
import numpy as np

n, m, B = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (4096, 4096, 64)
rng = np.random.default_rng(0)
print(f"n={n} m={m} B={B}: longest chain of dependent samples in K steps (5 draws)")
for K in (20, 100, 1049):
    out = []
    for _ in range(5):
        depth = np.zeros(n + m, dtype=np.int64)      # chain depth of the latest update of each row
        longest = 0
        for _k in range(K):
            u = rng.integers(0, n, B)
            i = rng.integers(0, m, B) + n
            j = rng.integers(0, m, B) + n
            d = np.maximum(np.maximum(depth[u], depth[i]), depth[j]) + 1     # reads the state before this step
            new = depth.copy()
            np.maximum.at(new, np.concatenate([u, i, j]), np.concatenate([d, d, d]))
            depth = new
            longest = max(longest, int(d.max()))
        out.append(longest)
    print(f"  K={K:5d}: {out}")
