"""Diagnostic: wall time of the drop-in metric functions at C2 (reference: 0.39 s + 16.0 s on 8 vCPUs)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "matrix-factorization-with-comparison-data_amd")]
os.environ.setdefault("OMP_NUM_THREADS", "4")
import torch
import structure as S
import generation_data as gd
dev = "cuda"
n = m = 4096; d = 64
A, B = gd.generate_embedding_factors(n, m, d, dev)
X = (A @ B.t()).contiguous()
model = S.MatrixFactorization(n, m, d).to(dev)
for name, fn in (("compute_reconstruction_error", lambda: S.compute_reconstruction_error(model, X, 1.0)),
                 ("compute_alpha_and_norm_ratios", lambda: S.compute_alpha_and_norm_ratios(model, X))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{name}: {dt*1e3:.1f} ms", flush=True)
from mfcd import metrics
import numpy as np
U, V = model.U.data, model.V.data
t0 = time.perf_counter(); rs, _ = metrics.uvt_stats(U, V, X, 1.0); rs = rs.cpu(); dt1 = time.perf_counter() - t0
ok = np.ones(n, bool)
xm = rs[:, 4].to(dev, torch.float32)
t0 = time.perf_counter(); metrics.spearman_and_svd(U, V, xm, X, 0.5, ok); torch.cuda.synchronize(); dt2 = time.perf_counter() - t0
print(f"  uvt_stats+D2H {dt1*1e3:.1f} ms, spearman+svd {dt2*1e3:.1f} ms")
