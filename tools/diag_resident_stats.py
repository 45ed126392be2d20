"""Diagnostic: where a wave's time goes in the look-ahead resident kernel (C2 by default), from the per-wave cycle and
event counters of the -DMFCD_RES_STATS build (make -C csrc diag; run with MFCD_LIB=.../libmfcd_hip_diag.so).
Usage: diag_resident_stats.py [C2|C3|NxD] [knob=value ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "matrix-factorization-with-comparison-data_amd")]
os.environ.setdefault("OMP_NUM_THREADS", "4")
import numpy as np, torch
import bench
from mfcd import engine
dev = torch.device("cuda:0")
size = [a for a in sys.argv[1:] if "=" not in a]
knobs = [a for a in sys.argv[1:] if "=" in a]
if knobs:
    engine.set_tuning(**{k: int(v) for k, v in (a.split("=") for a in knobs)})
cfg = dict(bench.C2, name="C2")
if size and size[0] == "C3":
    cfg = dict(bench.C3, name="C3")
r = bench.Runner(cfg, dev, 0)
K = r.steps_per_epoch
r.run(K); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
order = torch.randperm(r.train.N, generator=r.gen)
stream = r.train.ordered(order)
torch.cuda.synchronize()
e0.record(); engine.train_steps(r.bind, stream, cfg["B"]); e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
ws = engine.workspace_for(dev).buf
raw = ws[256:256 + 64 + 4096 * 64].view(torch.int64).cpu().numpy()
print("first give-up record (wave, step, pos, what):", raw[:4])
dbg = raw[8:].reshape(-1, 8).copy()
dbg = dbg[dbg[:, 0] > 0]
hw_id = (dbg[:, 6] >> 32) & 0xFFFFFFFF
xcc = (dbg[:, 5] >> 32) & 0xF
dbg[:, 6] &= 0xFFFFFFFF
dbg[:, 5] &= 0xFFFFFFFF
tot = dbg[:, 0].mean()
tick_us = dbg[:, 0].max() / (ms * 1e3)          # the slowest wave's loop ~ the launch
print(f"waves {len(dbg)}  call {ms*1e3:.0f} us = {ms*1e3/K:.3f} us/step; step loop of a wave: mean {tot:.0f} max {dbg[:,0].max()} ticks "
      f"(~{tick_us:.0f} ticks/us if the slowest loop is the launch)")
hits = dbg[:, 4] & 0xFFFFF
first_ok = (dbg[:, 4] >> 40) & 0xFFFFF
pubs = dbg[:, 7] & 0xFFFFFFFF
evs = dbg[:, 7] >> 32
H = hits.sum()
print(f"hits {H} ({H/K:.1f}/step, {H/len(dbg):.1f}/wave); "
      f"first poll succeeded {first_ok.sum()} ({first_ok.sum()/H*100:.0f}%); failed polls {dbg[:,5].sum()} ({dbg[:,5].sum()/H:.2f}/hit)")
print(f"event steps {evs.sum()} ({evs.sum()/len(dbg):.1f}/wave = {evs.sum()/len(dbg)/K*100:.1f}% of steps); rows published {pubs.sum()}")
print(f"share of a wave's loop time: event steps {dbg[:,1].mean()/tot*100:.1f}% (of which waiting polls {dbg[:,2].mean()/tot*100:.1f}%, "
      f"publish passes {dbg[:,3].mean()/tot*100:.1f}%); per event step {dbg[:,1].sum()/evs.sum()/tick_us:.2f} us; "
      f"per waited poll {dbg[:,2].sum()/max((hits-first_ok).sum(),1)/tick_us:.2f} us; per publish pass {dbg[:,3].sum()/max(pubs.sum(),1)/tick_us:.2f} us")
rest = tot - dbg[:, 1].mean()
print(f"common path: {rest/tot*100:.1f}% of loop time = {rest/K/tick_us:.3f} us per step;  kernel start -> first step {dbg[:,6].mean()/tick_us:.2f} us")
wait_us = dbg[:, 2] / tick_us
pc = np.percentile(wait_us, [0, 1, 5, 25, 50, 75, 95, 100])
print("per-wave total poll wait [us] percentiles 0/1/5/25/50/75/95/100: " + " ".join(f"{v:.0f}" for v in pc))
busy = (dbg[:, 0] - dbg[:, 2]) / tick_us
pc = np.percentile(busy, [0, 5, 50, 95, 100])
print("per-wave loop time minus poll wait [us] percentiles 0/5/50/95/100: " + " ".join(f"{v:.0f}" for v in pc))
for lo, hi in ((0, 35), (35, 45), (45, 55), (55, 65), (65, 400)):
    sel = (hits >= lo) & (hits < hi)
    if sel.any():
        print(f"  waves with {lo:3d}-{hi:3d} hits: n={sel.sum():4d}  event-step time {dbg[sel,1].mean()/tick_us:6.0f} us  wait {dbg[sel,2].mean()/tick_us:6.0f} us  "
              f"publish {dbg[sel,3].mean()/tick_us:5.0f} us  busy {busy[sel].mean():6.0f} us")

# placement: which SIMD of which CU a wave ran on (HW_ID: simd [5:4], cu [11:8], sh [12], se [15:13]; XCC_ID)
simd = (hw_id >> 4) & 3
cu = (hw_id >> 8) & 15
sh = (hw_id >> 12) & 1
se = (hw_id >> 13) & 7
cu_key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
simd_key = cu_key * 4 + simd
common = (dbg[:, 0] - dbg[:, 1]) / tick_us / K            # us per step outside event steps
uc, cnt_cu = np.unique(cu_key, return_counts=True)
us_, cnt_simd = np.unique(simd_key, return_counts=True)
print(f"placement: {len(uc)} CUs hold waves; waves per CU min/median/max {cnt_cu.min()}/{int(np.median(cnt_cu))}/{cnt_cu.max()} "
      f"(histogram {dict(zip(*np.unique(cnt_cu, return_counts=True)))}); waves per SIMD histogram {dict(zip(*np.unique(cnt_simd, return_counts=True)))}")
per_simd = dict(zip(us_, cnt_simd))
crowd = np.array([per_simd[k_] for k_ in simd_key])
for cval in np.unique(crowd):
    sel = crowd == cval
    print(f"  waves on a SIMD with {cval} waves: n={sel.sum():4d}  common path {common[sel].mean():.3f} us/step  busy {busy[sel].mean():5.0f} us  wait {wait_us[sel].mean():5.0f} us")
pc = np.percentile(common, [0, 5, 50, 95, 100])
print("per-wave common-path us/step percentiles 0/5/50/95/100: " + " ".join(f"{v:.3f}" for v in pc))
byx = [busy[xcc == x].mean() for x in range(8) if (xcc == x).any()]
print("mean busy by XCC: " + " ".join(f"{v:.0f}" for v in byx) + "   waves by XCC: " + " ".join(str(int((xcc == x).sum())) for x in range(8)))
