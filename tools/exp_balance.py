"""Experiment: does the imbalance of hits per wave bound the resident step?  Same state size and triplet count as C2,
different user/item split (item rows are touched twice as often as user rows when n == m)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "matrix-factorization-with-comparison-data_amd")]
os.environ.setdefault("OMP_NUM_THREADS", "4")
import torch, bench
dev = torch.device("cuda:0")
for n, m in [(4096, 4096), (2731, 5461), (5461, 2731), (1024, 7168)]:
    cfg = dict(bench.C2, n=n, m=m, p=bench.C2["p"] * 4096 * 4096 / (n * m))
    r = bench.Runner(cfg, dev, 0)
    r.run(r.steps_per_epoch); torch.cuda.synchronize()
    k = [0, 0, 0]
    t0 = time.perf_counter(); c = r.run(5 * r.steps_per_epoch); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"n={n} m={m} train={r.train.N} steps/epoch={r.steps_per_epoch}: {dt/(5*r.steps_per_epoch)*1e6:.3f} us/step wall, {c/dt/1e6:.1f} M/s", flush=True)
