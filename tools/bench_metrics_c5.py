"""Diagnostic: the drop-in metric functions at BASELINE configs[4] size (100000 x 20000, d = 256; dense X, 8 GB)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "matrix-factorization-with-comparison-data_amd")]
import torch
from mfcd import metrics
dev = torch.device("cuda:0")
n, m, d = 100000, 20000, 256
g = torch.Generator(device=dev).manual_seed(1)
U = torch.randn(n, d, device=dev, generator=g) / d ** 0.5
V = torch.randn(m, d, device=dev, generator=g) / d ** 0.5
kind = sys.argv[1] if len(sys.argv) > 1 else "rank-d"
if kind == "rank-d":       # what the reference's generators produce: X = A B^T with d columns (entries of std 0.5)
    A = torch.randn(n, d, device=dev, generator=g)
    Bf = torch.randn(m, d, device=dev, generator=g)
    X = (A @ Bf.t()) * (0.5 / d ** 0.5)
    del A, Bf
else:                      # full rank with a flat spectrum: the block iteration hands over to the dense solver
    X = torch.empty(n, m, device=dev)
    for r0 in range(0, n, 8192):
        X[r0:r0 + 8192].normal_(0.0, 0.5, generator=g)
    X += 0.3 * (U @ V.t())
torch.cuda.synchronize()
print(f"X: {kind}")
for rep in range(2):
    t0 = time.perf_counter(); e = metrics.reconstruction_error(U, V, X, 1.0); torch.cuda.synchronize(); t1 = time.perf_counter()
    out = metrics.alpha_and_norm_ratios(U, V, X); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"C5 size rep {rep}: reconstruction_error {1e3*(t1-t0):.1f} ms ({e:.4f}); alpha_and_norm_ratios {t2-t1:.2f} s (spearman mean {out[6]:.4f}, svd err {out[8]:.4f})", flush=True)
t0 = time.perf_counter()
A = U[:2048] @ (V - V.mean(0, keepdim=True)).t()
rho = metrics.spearman_rows(A, X[:2048]); torch.cuda.synchronize()
t1 = time.perf_counter()
for _ in range(3):
    rho = metrics.spearman_rows(A, X[:2048])
torch.cuda.synchronize()
print(f"spearman kernel alone: 2048 rows of 20000 columns in {(time.perf_counter()-t1)/3*1e3:.1f} ms -> {100000/2048*(time.perf_counter()-t1)/3:.2f} s for all rows")
