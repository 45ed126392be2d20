"""Diagnostic: wall time of the driver-style timed region (ONE 20-step call between two device syncs) with and without
the HIP event pair bench.py records around it, and with the pieces of host work taken apart."""
import os, sys, time, statistics as st
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "matrix-factorization-with-comparison-data_amd")]
import torch
import bench

cfg = dict(bench.WORKLOADS["C2"])
dev = torch.device("cuda:0")
r = bench.Runner(cfg, dev, 0)
r.run(5)
bench.clock_ramp(r, 0.3)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20

def timed(record):
    out = []
    for _ in range(60):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r.run(K, record=record)
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) * 1e6)
        r.reset_events()
    return out

for rep in range(2):
    for rec in (True, False):
        x = timed(rec)
        print(f"record={rec!s:5}  wall us: min {min(x):.1f} med {st.median(x):.1f} p90 {sorted(x)[53]:.1f}  -> {st.median(x)/K:.2f} us/step", flush=True)
# first call after the ramp, as the bench does it (ramp, sync, ONE call)
for rec in (True, False):
    x = []
    for _ in range(8):
        bench.clock_ramp(r, 0.1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r.run(K, record=rec)
        torch.cuda.synchronize()
        x.append((time.perf_counter() - t0) * 1e6)
        r.reset_events()
    print(f"after ramp, record={rec!s:5}: " + " ".join(f"{v:.1f}" for v in x), flush=True)
