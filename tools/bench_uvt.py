"""Diagnostic: time the dense UV^T metric pass (mfcd_uvt_stats) and price it against its roofline."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "matrix-factorization-with-comparison-data_amd")]
os.environ.setdefault("OMP_NUM_THREADS", "4")
import torch
from mfcd import metrics
dev = torch.device("cuda:0")
only = set(sys.argv[1:])
for name, n, m, d in [("C1", 256, 256, 8), ("C2", 4096, 4096, 64), ("C3", 16384, 16384, 128), ("C5", 100000, 20000, 256)]:
    if only and name not in only:
        continue
    U = torch.randn(n, d, device=dev) / d ** 0.5
    V = torch.randn(m, d, device=dev) / d ** 0.5
    X = torch.randn(n, m, device=dev) * 0.5
    def timed(what, seconds=float(os.environ.get("UVT_BENCH_SECONDS", "0.3"))):
        """Average pass time over >= `seconds` of back-to-back passes after an untimed stretch of the same length (a chip
        that idled while X was generated runs its first passes 10-25 % slower: tools/exp_uvt_sustained.py)."""
        per = max(1, min(200, int(2e9 / (n * m * d) * 40)))
        for phase in range(2):
            t0, k = time.perf_counter(), 0
            while time.perf_counter() - t0 < seconds:
                for _ in range(per):
                    metrics.uvt_stats(U, V, X, 1.0, what=what)
                torch.cuda.synchronize()
                k += per
            dt_ = (time.perf_counter() - t0) / k
        return dt_
    reps = 5
    dt = timed(3)
    flops = 2.0 * n * m * d
    bytes_ = 4.0 * n * m + 4.0 * (n + m) * d          # X is read once (round 2: row statistics come out of the sweep)
    sel = {"rows": timed(1), "err": timed(2)}          # the passes the two metric functions issue
    # torch reference of the reference's own op sequence for the same quantity (GEMM + centring + norms)
    dt_t = float("nan")
    if not os.environ.get("MFCD_SKIP_TORCH"):
        t1 = time.perf_counter()
        for _ in range(reps):
            M = U @ V.t(); M -= M.mean(0, keepdim=True); e = torch.norm(M - X) / torch.norm(X)
        torch.cuda.synchronize()
        dt_t = (time.perf_counter() - t1) / reps
    split = d in (32, 64, 128, 256)      # bf16x3 split-product form (three bf16 MFMAs per fp32 product)
    pipe = (f"{3*flops/dt/1e12:7.1f} bf16 TFLOP/s issued ({3*flops/dt/2500e12*100:4.1f}% of the 2.5 PF bf16 MFMA peak)" if split
            else f"{flops/dt/157.3e12*100:5.1f}% of the 157.3 TF fp32 MFMA peak")
    print(f"{name}: n={n} m={m} d={d}  uvt_stats {dt*1e6:9.1f} us  = {flops/dt/1e12:6.2f} nominal TFLOP/s; {pipe}"
          f"; X read {bytes_/dt/1e9:7.1f} GB/s ({bytes_/dt/8e12*100:4.1f}% of 8 TB/s) | rows-only {sel['rows']*1e6:8.1f} us "
          f"({bytes_/sel['rows']/8e12*100:4.1f}% HBM), error-only {sel['err']*1e6:8.1f} us ({bytes_/sel['err']/8e12*100:4.1f}% HBM)"
          f" | torch-op sequence on the same GPU {dt_t*1e6:9.1f} us", flush=True)
    del U, V, X
    torch.cuda.empty_cache()
