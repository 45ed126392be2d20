#!/usr/bin/env python3
"""Golden vectors for the ground-truth generators (TEST INFRASTRUCTURE — runs in the dev container only).

Imports the *unmodified* reference (structure.generate_X, structure.py:590-663, over generation_data.py:346-715) and
stores, per `generation` keyword, the matrix X it returns under fixed seeds of the three generators the laws draw from
(torch, numpy, and Python's `random` — networkx's small-world graph uses the latter), plus one draw from each taken right
after the call.  "graph" is absent: the reference raises TypeError for it (a stray comma at generation_data.py:565 turns
`noise` into a tuple).  Only data is committed (tests/golden/generators.npz).

Usage:  OMP_NUM_THREADS=4 PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_generators.py
"""
import os
import random
import sys
import types

os.environ.setdefault("OMP_NUM_THREADS", "4")
sys.dont_write_bytecode = True
_stub = types.ModuleType("torch.utils.tensorboard")      # imported at structure.py:10, used only under `if False`
_stub.SummaryWriter = object
sys.modules["torch.utils.tensorboard"] = _stub
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402
import torch  # noqa: E402

import structure as R  # noqa: E402  (the reference)

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "generators.npz")
CASES = [("base", 24, 18, 4, {}), ("low_rank", 24, 18, 5, {"rank": 3}), ("low_rank", 10, 30, 4, {}), ("clustered", 24, 18, 4, {}),
         ("structured", 24, 18, 4, {}), ("svd", 24, 18, 4, {}), ("correlated", 24, 18, 4, {}), ("social", 24, 18, 4, {}),
         ("temporal", 24, 18, 4, {}), ("hierarchical", 24, 18, 4, {}), ("gmm", 40, 30, 3, {})]


def main():
    out = {"cases": np.asarray([f"{g}:{n}:{m}:{d}:{kw.get('rank', -1)}" for g, n, m, d, kw in CASES])}
    for k, (g, n, m, d, kw) in enumerate(CASES):
        torch.manual_seed(300 + k)
        np.random.seed(300 + k)
        random.seed(300 + k)
        X = R.generate_X(n, m, d, "cpu", generation=g, **kw)
        out[f"{k}.X"] = X.detach().cpu().numpy().copy()
        out[f"{k}.after"] = np.asarray([float(torch.rand(1, dtype=torch.float64)), float(np.random.random_sample()),
                                        random.random()])
        print(g, tuple(X.shape))
    try:
        R.generate_X(8, 6, 3, "cpu", generation="graph")
        out["graph_raises"] = np.asarray("")
    except Exception as e:      # recorded: what the reference does for this keyword
        out["graph_raises"] = np.asarray(type(e).__name__)
    np.savez_compressed(OUT, **out)
    print("graph:", out["graph_raises"], "| wrote", OUT)


if __name__ == "__main__":
    main()
