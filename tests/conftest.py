import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "matrix-factorization-with-comparison-data_amd")
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle as O
    return O.COracle()


KATS = ["kat_dups_d8.npz", "kat_short_soft_d64.npz", "kat_saturated_d8.npz", "kat_d2.npz",
        "kat_d128_wd.npz", "kat_d256_b1.npz", "kat_d5.npz"]
E2ES = ["e2e_c1.npz", "e2e_soft_k3.npz", "e2e_hard_k2_d16.npz"]
