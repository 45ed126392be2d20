"""Diagnostic: streaming-step period at C3 / C4 / C5 size by workgroup tile (16-byte chunks per thread and array).
VERDICT r1 item 8: why do d = 128 / 256 sit 12-20 points under C4's 78 % of 8 TB/s?  Not part of the product."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "matrix-factorization-with-comparison-data_amd")]
os.environ.setdefault("OMP_NUM_THREADS", "4")
import numpy as np, torch
import structure as S
from mfcd import engine

dev = torch.device("cuda:0")
engine.set_train_path("streaming")
cases = [("C3f32", 16384, 16384, 128, torch.float32), ("C3bf16", 16384, 16384, 128, torch.bfloat16),
         ("C4", 65536, 65536, 64, torch.float32), ("C5", 100000, 20000, 256, torch.float32),
         ("C5x2", 200000, 40000, 256, torch.float32)]
for name, n, m, d, dt_ in cases:
    model = S.MatrixFactorization(n, m, d, dtype=dt_).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-5)
    bind = engine.AdamBinding(model, opt)
    B, steps = 64, 200
    rng = np.random.default_rng(0)
    N = B * steps
    rows = np.stack([rng.integers(0, n, N), rng.integers(0, m, N), rng.integers(0, m, N), rng.integers(0, 2, N)], 1).astype(np.float64)
    st = engine.SampleStore(rows, n, m, dev)
    ab = (24 * (n + m) * d + 12 * B * d + 16 * B) if dt_ == torch.float32 else (20 * (n + m) * d + 6 * B * d + 16 * B)
    out = []
    for chunks in (0, 1, 2, 4, 8):
        engine.set_tuning(stream_chunks=chunks)
        engine.train_steps(bind, st.dev, B); torch.cuda.synchronize()
        t0 = time.perf_counter(); engine.train_steps(bind, st.dev, B); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        out.append(f"chunks={chunks}: {dt/steps*1e6:7.2f} us {ab/(dt/steps)/1e12:5.2f} TB/s")
    engine.set_tuning(stream_chunks=0)
    print(f"{name:7s} state {(n+m)*d*12/1e6:7.1f} MB  " + " | ".join(out), flush=True)
    del model, opt, bind, st
    torch.cuda.empty_cache()
engine.set_train_path("auto")
