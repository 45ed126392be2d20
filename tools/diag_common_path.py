"""Diagnostic: per-step time of the resident kernel when hits are rare (B = 1: three hits per step in total),
i.e. the cost of the common no-hit path executed by every wave, vs the normal B = 64 stream."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "matrix-factorization-with-comparison-data_amd")]
os.environ.setdefault("OMP_NUM_THREADS", "4")
import numpy as np, torch
import structure as S
from mfcd import engine
dev = torch.device("cuda:0")
size = [a for a in sys.argv[1:] if "=" not in a]
knobs = [a for a in sys.argv[1:] if "=" in a]
if knobs:     # e.g. resident_q=4 resident_lookahead=8
    engine.set_tuning(**{k: int(v) for k, v in (a.split("=") for a in knobs)})
n = m = 4096; d = 64
if size and size[0] == "C3":
    n = m = 16384; d = 128
elif size and "x" in size[0]:      # e.g. 4096x32: n = m = 4096, d = 32
    n = m = int(size[0].split("x")[0]); d = int(size[0].split("x")[1])
for B, N in ((1, 20000), (8, 160000), (64, 67108)):
    model = S.MatrixFactorization(n, m, d).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-5)
    bind = engine.AdamBinding(model, opt)
    rng = np.random.default_rng(0)
    rows = np.stack([rng.integers(0, n, N), rng.integers(0, m, N), rng.integers(0, m, N), rng.integers(0, 2, N)], 1).astype(np.float64)
    rows[:, 2] = (rows[:, 1] + 1 + rng.integers(0, m - 1, N)) % m
    st = engine.SampleStore(rows, n, m, dev)
    engine.train_steps(bind, st.dev, B); torch.cuda.synchronize()
    t0 = time.perf_counter(); engine.train_steps(bind, st.dev, B); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    steps = (N + B - 1) // B
    print(f"B={B:3d}: {steps} steps, {dt/steps*1e6:.3f} us/step ({3*B} row hits per step; n=m={n} d={d})", flush=True)
