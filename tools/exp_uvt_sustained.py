"""Diagnostic: UV^T pass time as a function of how long the chip has been running passes back to back
(short benchmarks of an MFMA-dense kernel can see a different power state than a sustained run)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "matrix-factorization-with-comparison-data_amd")]
import torch
from mfcd import metrics
dev = torch.device("cuda:0")
for name, n, m, d, per in [("C2", 4096, 4096, 64, 200), ("C3", 16384, 16384, 128, 40), ("C5", 100000, 20000, 256, 5)]:
    if len(sys.argv) > 1 and name not in sys.argv[1:]:
        continue
    U = torch.randn(n, d, device=dev) / d ** 0.5
    V = torch.randn(m, d, device=dev) / d ** 0.5
    X = torch.randn(n, m, device=dev) * 0.5
    metrics.uvt_stats(U, V, X, 1.0); torch.cuda.synchronize()
    time.sleep(0.5)
    t_start = time.perf_counter()
    rows = []
    while time.perf_counter() - t_start < 3.0:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(per):
            metrics.uvt_stats(U, V, X, 1.0)
        e1.record(); torch.cuda.synchronize()
        rows.append((time.perf_counter() - t_start, e0.elapsed_time(e1) * 1e3 / per))
    pick = [rows[0], rows[1], rows[len(rows) // 8], rows[len(rows) // 4], rows[len(rows) // 2], rows[-1]]
    print(name, " ".join(f"[t={t:.2f}s {us:.1f}us]" for t, us in pick), flush=True)
    del U, V, X
