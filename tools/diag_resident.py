"""Diagnostic: cycle accounting of the resident kernel (needs the -DMFCD_STAMPS build: make -C csrc diag;
run with MFCD_LIB=.../libmfcd_hip_diag.so).  Not part of the product."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "matrix-factorization-with-comparison-data_amd")]
os.environ.setdefault("OMP_NUM_THREADS", "4")
import numpy as np, torch, time
import bench
from mfcd import engine

dev = torch.device("cuda:0")
r = bench.Runner(bench.C2, dev, 0)
r.run(1049); torch.cuda.synchronize()
t0 = time.perf_counter(); r.run(1049); torch.cuda.synchronize(); dt = time.perf_counter() - t0
ws = engine.workspace_for(dev).buf
dbg = ws[256:256 + 4096 * 64].view(torch.int64).cpu().numpy().reshape(-1, 8)
dbg = dbg[dbg[:, 0] > 0]
K = 1049
clk = dbg[:, 0].mean() / (dt * 1e6) if dt > 0 else 0  # cycles per us (approx; includes val pass etc.)
print(f"waves {len(dbg)}  epoch wall {dt*1e3:.2f} ms = {dt/K*1e6:.2f} us/step; total cycles/wave mean {dbg[:,0].mean():.0f} (~{clk:.0f} cyc/us)")
names = ["total", "poll-wait", "hit-compute", "adam", "publish", "hits", "polls", "x"]
for c in range(1, 5):
    print(f"  {names[c]:12s} mean {dbg[:,c].mean():10.0f} cyc/wave = {100*dbg[:,c].mean()/dbg[:,0].mean():5.1f}% of total; per step {dbg[:,c].mean()/K:8.1f} cyc")
hits = dbg[:, 5].sum()
print(f"  hits total {hits} ({hits/K:.1f}/step); polls(spins) total {dbg[:,6].sum()} ({dbg[:,6].sum()/max(hits,1):.2f} re-polls per hit)")
print(f"  per hit: wait {dbg[:,1].sum()/hits:.0f} cyc, compute {dbg[:,2].sum()/hits:.0f} cyc")
other = dbg[:, 0] - dbg[:, 1:5].sum(1)
print(f"  other (scan, loop) mean {other.mean():.0f} cyc/wave = {100*other.mean()/dbg[:,0].mean():.1f}%; per step {other.mean()/K:.1f} cyc")

import numpy as np
w = dbg[:, 1] / dbg[:, 0]
order = np.argsort(w)
print("poll-wait share of wave time: min %.3f p10 %.3f median %.3f p90 %.3f max %.3f" % (w.min(), np.percentile(w,10), np.median(w), np.percentile(w,90), w.max()))
for name, idx in (("least-waiting 20 waves", order[:20]), ("most-waiting 20 waves", order[-20:])):
    sub = dbg[idx]
    print(f"  {name}: hits/wave {sub[:,5].mean():.1f}  wait {sub[:,1].mean()/K:.0f} ticks/step  hit-compute {sub[:,2].mean()/K:.0f}  adam {sub[:,3].mean()/K:.0f}  publish {sub[:,4].mean()/K:.0f}  other {(sub[:,0]-sub[:,1:5].sum(1)).mean()/K:.0f}  total {sub[:,0].mean()/K:.0f}")
print("hits per wave: min %d median %d max %d" % (dbg[:,5].min(), np.median(dbg[:,5]), dbg[:,5].max()))
