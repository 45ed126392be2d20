"""Diagnostic: N fused steps of the STREAMING form at a BASELINE size, for profiling one step kernel in isolation
(rocprofv3 --pmc ... -- python3 tools/stream_step.py C5 40).  Not part of the product."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "matrix-factorization-with-comparison-data_amd")]
os.environ.setdefault("OMP_NUM_THREADS", "4")
import numpy as np, torch
import structure as S
from mfcd import engine

SIZES = {"C3": (16384, 16384, 128), "C4": (65536, 65536, 64), "C5": (100000, 20000, 256), "C5x2": (200000, 40000, 256)}
name = sys.argv[1] if len(sys.argv) > 1 else "C5"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dtype = torch.bfloat16 if len(sys.argv) > 3 and sys.argv[3] == "bf16" else torch.float32
n, m, d = SIZES[name]
dev = torch.device("cuda:0")
engine.set_train_path("streaming")
model = S.MatrixFactorization(n, m, d, dtype=dtype).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-5)
bind = engine.AdamBinding(model, opt)
rng = np.random.default_rng(0)
N = 64 * steps
rows = np.stack([rng.integers(0, n, N), rng.integers(0, m, N), rng.integers(0, m, N), rng.integers(0, 2, N)], 1).astype(np.float64)
st = engine.SampleStore(rows, n, m, dev)
engine.train_steps(bind, st.dev, 64)
torch.cuda.synchronize()
print(f"{name} {dtype}: {steps} streaming steps done; algorithmic bytes per step = "
      f"{(24 if dtype == torch.float32 else 20) * (n + m) * d + (12 if dtype == torch.float32 else 6) * 64 * d + 16 * 64}")
