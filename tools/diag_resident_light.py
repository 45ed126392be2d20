"""Diagnostic: hit-step and publish-path time of the resident kernel with stamps ONLY on those paths
(-DMFCD_STAMPS=2 build; MFCD_LIB=.../libmfcd_hip_diag.so).  The common path carries no stamp, so the launch keeps
(almost) its normal pace and the tick counts can be read against it."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "matrix-factorization-with-comparison-data_amd")]
os.environ.setdefault("OMP_NUM_THREADS", "4")
import numpy as np, torch
import bench
from mfcd import engine
dev = torch.device("cuda:0")
r = bench.Runner(bench.C2 | {"name": "C2"}, dev, 0)
r.run(1049); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
order = torch.randperm(r.train.N, generator=r.gen)
stream = r.train.ordered(order)
torch.cuda.synchronize()
e0.record(); engine.train_steps(r.bind, stream, 64); e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
ws = engine.workspace_for(dev).buf
dbg = ws[256:256 + 4096 * 64].view(torch.int64).cpu().numpy().reshape(-1, 8)
dbg = dbg[dbg[:, 0] > 0]
K = 1049
tot = dbg[:, 0].mean()
tick_us = tot / (ms * 1e3)
print(f"waves {len(dbg)}  launch {ms*1e3:.0f} us = {ms*1e3/K:.3f} us/step; wave lifetime {tot:.0f} ticks (~{tick_us:.0f} ticks/us)")
waited = (dbg[:, 5] >> 32).sum()
dbg[:, 5] = dbg[:, 5] & 0xffffffff
hits, pubs = dbg[:, 5].sum(), dbg[:, 7].sum()
print(f"hit blocks: total hits {hits} ({hits/K:.1f}/step); time in hit blocks {dbg[:,3].mean()/tot*100:.1f}% of wave time; "
      f"per hit {dbg[:,3].sum()/hits/tick_us:.2f} us, of which granule wait {dbg[:,1].sum()/hits/tick_us:.2f} us "
      f"({dbg[:,6].sum()/hits:.2f} re-polls per hit)")
print(f"polls that had to wait: {waited} of {hits} hits; at success the newest needed publish was {dbg[:,2].sum()/max(waited,1)*0.01:.2f} us old "
      "(real-time clock: publish stamp taken by the owner just before its granule stores)")
print(f"publish slow paths: {pubs} ({pubs/K:.1f}/step); {dbg[:,4].mean()/tot*100:.1f}% of wave time; per slow path {dbg[:,4].sum()/max(pubs,1)/tick_us:.2f} us")
rest = tot - dbg[:, 3].mean() - dbg[:, 4].mean()
print(f"everything else (common path): {rest/tot*100:.1f}% of wave time = {rest/K/tick_us:.3f} us per step")
h = dbg[:, 5]
for lo, hi in ((0, 35), (35, 45), (45, 55), (55, 65), (65, 200)):
    sel = (h >= lo) & (h < hi)
    if sel.any():
        print(f"  waves with {lo:3d}-{hi:3d} hits: n={sel.sum():4d}  hit-block time {dbg[sel,3].mean()/tick_us:7.0f} us  publish {dbg[sel,4].mean()/tick_us:6.0f} us  "
              f"granule wait/hit {dbg[sel,1].sum()/dbg[sel,5].sum()/tick_us:.2f} us")
# who sets the pace?  per-wave total granule wait: the waves that (almost) never wait are the critical ones
wait_us = dbg[:, 1] / tick_us
order_ = np.argsort(wait_us)
pc = np.percentile(wait_us, [0, 1, 5, 25, 50, 75, 95, 100])
print("per-wave total granule wait [us] percentiles 0/1/5/25/50/75/95/100: " + " ".join(f"{v:.0f}" for v in pc))
print("the 12 waves that waited least (wave id, hits, publishes, wait us, hit-block non-wait us, publish us, common us):")
for w in order_[:12]:
    hb = dbg[w, 3] / tick_us
    print(f"  wave {w:5d} (CU-slot {w // 16:4d}, XCD {(w // 4) % 8})  hits {h[w]:3d}  pubs {dbg[w,7]:3d}  wait {wait_us[w]:6.0f}  "
          f"hit non-wait {hb - wait_us[w]:6.0f}  publish {dbg[w,4]/tick_us:5.0f}  common {(dbg[w,0]-dbg[w,3]-dbg[w,4])/tick_us:6.0f}")
byx = [wait_us[((np.arange(len(wait_us)) // 4) % 8) == x].mean() for x in range(8)]
print("mean wait by (workgroup id mod 8) = XCD: " + " ".join(f"{v:.0f}" for v in byx))
