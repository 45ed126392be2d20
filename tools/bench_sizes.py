"""Diagnostic: fused-step kernel period across problem sizes (streaming regime check; not the product)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "matrix-factorization-with-comparison-data_amd")]
os.environ.setdefault("OMP_NUM_THREADS", "4")
import numpy as np, torch
import structure as S
from mfcd import engine

dev = torch.device("cuda:0")
cases = [("C1", 256, 256, 8, torch.float32), ("nb", 1000, 1000, 2, torch.float32), ("C2", 4096, 4096, 64, torch.float32),
         ("C2bf16", 4096, 4096, 64, torch.bfloat16), ("mid", 8192, 8192, 64, torch.float32),
         ("C3f32", 16384, 16384, 128, torch.float32), ("C3bf16", 16384, 16384, 128, torch.bfloat16),
         ("C4", 65536, 65536, 64, torch.float32), ("C5", 100000, 20000, 256, torch.float32)]
for name, n, m, d, dt_ in cases:
    model = S.MatrixFactorization(n, m, d, dtype=dt_).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-5)
    bind = engine.AdamBinding(model, opt)
    B, steps = 64, 600
    rng = np.random.default_rng(0)
    N = B * steps
    rows = np.stack([rng.integers(0, n, N), rng.integers(0, m, N), rng.integers(0, m, N), rng.integers(0, 2, N)], 1).astype(np.float64)
    st = engine.SampleStore(rows, n, m, dev)
    engine.train_steps(bind, st.dev, B); torch.cuda.synchronize()
    t0 = time.perf_counter(); engine.train_steps(bind, st.dev, B); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    k = [0, 0, 0]
    if dt_ == torch.float32:
        engine.train_steps(bind, st.dev, B, kernel_us=k)
    # fp32 24 B/element + gathers; bf16 factors (2 + 2 + 16) B/element + 6 B*d gathers
    ab = (24 * (n + m) * d + 12 * B * d + 16 * B) if dt_ == torch.float32 else (20 * (n + m) * d + 6 * B * d + 16 * B)
    form = engine.train_plan(N, B, n, m, d, bf16=dt_ == torch.bfloat16)
    print(f"{name:6s} {form['form_name']:9s} Q={form['resident_q']:2d} n={n:6d} m={m:6d} d={d:3d} elems={(n+m)*d/1e6:7.2f}M  period={dt/steps*1e6:8.2f} us  "
          f"algGB/s={ab/(dt/steps)/1e9:8.1f}  ({ab/(dt/steps)/8e12*100:5.1f}% of 8TB/s)  evpair avg/min={k[0]:.2f}/{k[1]:.2f} us", flush=True)
    del model, opt, bind, st
    torch.cuda.empty_cache()
