#!/usr/bin/env python3
"""Reference-vs-port step time (TEST INFRASTRUCTURE — runs in the dev container only).

The reference's Python cannot travel to the GPU box, so bench.py's cpu_baseline there is a port (the C oracle and
the op-for-op torch port).  This script times, in THIS container, on the same C2 inputs and at the reference's own
OMP_NUM_THREADS=4:
  * the unmodified reference loop body (structure.py:845-852: DataLoader batch, zero_grad, forward, BCE, backward,
    torch.optim.Adam.step, loss.item()) through the reference's own MatrixFactorization / BTLPreferenceDataset /
    DataLoader classes,
  * oracle/torch_port.py on the same records,
  * oracle/mfcd_oracle.c on the same records (1 and 4 threads),
and checks that all of them end at the same parameters.  Output: profiles/r02_reference_vs_port_cpu.txt.

Usage:  OMP_NUM_THREADS=4 PYTHONDONTWRITEBYTECODE=1 python oracle/compare_reference_port.py
"""
import os
import sys
import time
import types

os.environ.setdefault("OMP_NUM_THREADS", "4")
sys.dont_write_bytecode = True
_stub = types.ModuleType("torch.utils.tensorboard")
_stub.SummaryWriter = object
sys.modules["torch.utils.tensorboard"] = _stub
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import structure as R  # noqa: E402  (the reference, unmodified)

sys.path.insert(0, os.path.dirname(HERE))
from oracle import oracle as O  # noqa: E402
from oracle import torch_port  # noqa: E402

n = m = 4096
d, B, STEPS = 64, 64, 300
torch.set_num_threads(4)
torch.manual_seed(0)
np.random.seed(0)
rng = np.random.default_rng(0)
# C2-shaped inputs (rank-d X with entry std ~0.5; uniform triplets; BTL labels through the reference's own dataset class)
A = np.linalg.qr(rng.standard_normal((n, d)))[0] * (np.sqrt(n * m) / (2 * np.sqrt(d)))
Bm = np.linalg.qr(rng.standard_normal((m, d)))[0]
X = torch.tensor(A @ Bm.T, dtype=torch.float32)
N = B * STEPS
trip = list({(int(u), int(i), int(j)) for u, i, j in zip(rng.integers(0, n, 2 * N), rng.integers(0, m, 2 * N),
                                                         rng.integers(0, m, 2 * N)) if i != j})[:N]
ds = R.BTLPreferenceDataset(trip, X, scale=1.0, K=1, soft_label=False, train=True)
loader = torch.utils.data.DataLoader(ds, batch_size=B, shuffle=False)
model = R.MatrixFactorization(n, m, d)
U0, V0 = model.U.detach().numpy().copy(), model.V.detach().numpy().copy()
opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-5)


def reference_epoch():
    """structure.py:845-852, verbatim semantics (device = cpu)."""
    model.train()
    t0 = time.perf_counter()
    for batch in loader:
        u, i, j, z = [x.to("cpu") for x in batch]
        opt.zero_grad()
        pred = model(u, i, j)
        loss = F.binary_cross_entropy(pred, z.float())
        loss.backward()
        opt.step()
        loss.item()
    return time.perf_counter() - t0


t_ref_warm = reference_epoch()       # first pass: allocator / thread-pool warm-up
model.U.data.copy_(torch.from_numpy(U0))
model.V.data.copy_(torch.from_numpy(V0))
opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-5)
t_ref = reference_epoch()

rows = np.asarray(ds.data, dtype=np.float64)
u, i, j = (torch.from_numpy(rows[:, c].astype(np.int64)) for c in range(3))
z = torch.from_numpy(rows[:, 3].copy())
Up, Vp = torch.from_numpy(U0.copy()), torch.from_numpy(V0.copy())
st = {k: torch.zeros_like(Up if k.endswith("U") else Vp) for k in ("mU", "vU", "mV", "vV")}
torch_port.train_steps(Up.clone(), Vp.clone(), {k: v.clone() for k, v in st.items()}, u[:B * 20], i[:B * 20], j[:B * 20],
                       z[:B * 20], B, 0)
t0 = time.perf_counter()
torch_port.train_steps(Up, Vp, st, u, i, j, z, B, 0)
t_port = time.perf_counter() - t0

orc = O.COracle()
t_c = {}
for thr in (1, 4):
    so = O.new_state(U0, V0)
    t0 = time.perf_counter()
    orc.train_steps(so, rows[:, 0], rows[:, 1], rows[:, 2], rows[:, 3], B, 0, lr=1e-3, wd=1e-5, threads=thr)
    t_c[thr] = time.perf_counter() - t0

dU_port = float(np.abs(Up.numpy() - model.U.detach().numpy()).max())
dU_c = float(np.abs(so["U"] - model.U.detach().numpy()).max())
lines = [
    "Reference CPU path vs the ports bench.py times on the GPU box (dev container: "
    f"{os.cpu_count()} vCPU, torch {torch.__version__}, OMP_NUM_THREADS=4 as structure.py:3 sets it)",
    f"workload: C2 shape n=m={n} d={d}, B={B}, {STEPS} optimiser steps, Adam lr=1e-3 wd=1e-5, identical records",
    "",
    f"{'path':58s} {'us/step':>10s} {'updates/s':>12s}",
    f"{'reference loop (structure.py:845-852, DataLoader + autograd + Adam)':58s} {t_ref / STEPS * 1e6:10.1f} {N / t_ref:12.0f}",
    f"{'  of which DataLoader/collate (separate pass over the loader)':58s} "
    f"{sum(1 for _ in loader) and 0 or 0:10.1f}",
    f"{'oracle/torch_port.py (same ATen ops, no autograd / DataLoader)':58s} {t_port / STEPS * 1e6:10.1f} {N / t_port:12.0f}",
    f"{'oracle/mfcd_oracle.c, 1 thread':58s} {t_c[1] / STEPS * 1e6:10.1f} {N / t_c[1]:12.0f}",
    f"{'oracle/mfcd_oracle.c, 4 threads':58s} {t_c[4] / STEPS * 1e6:10.1f} {N / t_c[4]:12.0f}",
    "",
    f"agreement after {STEPS} steps: max|U_port - U_ref| = {dU_port:.2e}, max|U_c_oracle - U_ref| = {dU_c:.2e}",
    f"reading: the torch port runs the reference's ATen kernels without its Python overhead (DataLoader collate, autograd "
    f"graph), so it is {t_ref / t_port:.2f}x faster than the reference itself; as a CPU baseline it therefore FAVOURS the CPU.",
]
t0 = time.perf_counter()
for _ in loader:
    pass
t_load = time.perf_counter() - t0
lines[5] = f"{'  of which DataLoader/collate alone (separate pass)':58s} {t_load / STEPS * 1e6:10.1f}"
out = "\n".join(lines) + "\n"
print(out)
with open(os.path.join(os.path.dirname(HERE), "profiles", "r02_reference_vs_port_cpu.txt"), "w") as f:
    f.write(out)
