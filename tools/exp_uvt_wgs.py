"""Diagnostic: UV^T pass time against the number of workgroups the column split aims for (MFCD_TUNE_UVT_TARGET_WGS)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "matrix-factorization-with-comparison-data_amd")]
import torch
from mfcd import metrics, engine
dev = torch.device("cuda:0")
for name, n, m, d, per in [("C2", 4096, 4096, 64, 200), ("mid", 8192, 8192, 64, 100), ("C3", 16384, 16384, 128, 40), ("C5", 100000, 20000, 256, 5)]:
    if len(sys.argv) > 1 and name not in sys.argv[1:]:
        continue
    U = torch.randn(n, d, device=dev) / d ** 0.5
    V = torch.randn(m, d, device=dev) / d ** 0.5
    X = torch.randn(n, m, device=dev) * 0.5
    for wgs, mst in ((512, 8), (512, 4), (512, 2), (1024, 8), (1024, 4), (1024, 2), (2048, 2), (4096, 8)):
        engine.set_tuning(uvt_target_wgs=wgs, uvt_min_stages=mst)
        res = []
        for what in (3, 1, 2):
            for phase in range(2):
                t0, k = time.perf_counter(), 0
                while time.perf_counter() - t0 < 0.25:
                    for _ in range(per):
                        metrics.uvt_stats(U, V, X, 1.0, what=what)
                    torch.cuda.synchronize(); k += per
                dt = (time.perf_counter() - t0) / k
            res.append(dt * 1e6)
        print(f"{name} target_wgs={wgs:5d} min_stages={mst}: full {res[0]:8.1f} us  rows {res[1]:8.1f}  err {res[2]:8.1f}  ({2.0*n*m*d/res[0]/1e6/157.3*100:.1f} % of peak)", flush=True)
    del U, V, X
