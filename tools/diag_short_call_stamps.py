"""Diagnostic: where the FIXED time of a short resident call goes (C2, default 20 steps per call), from per-wave absolute
time stamps (100 MHz reference clock) of the -DMFCD_RES_STATS -DMFCD_RES_STAMPS build:
    make -C matrix-factorization-with-comparison-data_amd/csrc exp NAME=stamps EXPFLAGS="-DMFCD_RES_STATS -DMFCD_RES_STAMPS"
    MFCD_LIB=.../libmfcd_hip_stamps.so python tools/diag_short_call_stamps.py [steps] [knob=value ...]
Prints, relative to the first wave's start: when the last wave started (dispatch spread), when the waves entered and
left their step loops, when they ended; beside the HIP-event duration of the call."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "matrix-factorization-with-comparison-data_amd")]
os.environ.setdefault("OMP_NUM_THREADS", "4")
import numpy as np, torch
import bench
from mfcd import engine
dev = torch.device("cuda:0")
steps = [int(a) for a in sys.argv[1:] if a.isdigit()]
K = steps[0] if steps else 20
knobs = [a for a in sys.argv[1:] if "=" in a]
if knobs:
    engine.set_tuning(**{k: int(v) for k, v in (a.split("=") for a in knobs)})
cfg = dict(bench.C2, name="C2")
r = bench.Runner(cfg, dev, 0)
B = cfg["B"]
r.run(r.steps_per_epoch); torch.cuda.synchronize()
order = torch.randperm(r.train.N, generator=r.gen)
stream = r.train.ordered(order)
for _ in range(200):                                     # clocks up, prepared call made
    engine.train_steps(r.bind, stream[:K * B], B)
torch.cuda.synchronize()
ws = engine.workspace_for(dev).buf
rows = []
for rep in range(8):
    lo = (rep + 1) * K * B
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); engine.train_steps(r.bind, stream[lo:lo + K * B], B); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    raw = ws[256:256 + 64 + 2 * 4096 * 64].view(torch.int64).cpu().numpy()
    d = raw[8:8 + 4096 * 8].reshape(-1, 8)
    hb = raw[8 + 4096 * 8:8 + 2 * 4096 * 8].reshape(-1, 8)[d[:, 1] > 0]
    d = d[d[:, 1] > 0]
    t0, l0, l1, t1 = d[:, 1], d[:, 2], d[:, 3], d[:, 7]
    z = t0.min()
    f = lambda x: (x - z) / 100.0                        # noqa: E731  (us)
    mhz = np.median(d[:, 0] / np.maximum((l1 - l0) / 100.0, 1e-9))      # s_memtime ticks of the loop per us of the loop
    rows.append((us, f(t0).max(), np.median(f(l0)), f(l0).max(), np.median(f(l1)), f(l1).max(), np.median(f(t1)), f(t1).max(),
                 np.median((l1 - l0) / 100.0), len(d), mhz))
print(f"C2, {K}-step calls; times in us after the first wave's start (median / max over waves)")
print("call(events)  last wave start | loop entered med max | loop left med max | wave end med max | loop length med | waves")
for x in rows:
    print(f"{x[0]:8.1f}      {x[1]:6.2f}          | {x[2]:6.2f} {x[3]:6.2f}      | {x[4]:6.2f} {x[5]:6.2f}   | {x[6]:6.2f} {x[7]:6.2f}  | {x[8]:6.2f}         | {x[9]}  s_memtime/us {x[10]:.0f}")
# the slowest waves of the last call: how many hits they had, how many polls failed, when they entered / left the loop
hits = d[:, 4] & 0xFFFFF
first_ok = (d[:, 4] >> 40) & 0xFFFFF
fails = d[:, 5] & 0xFFFFFFFF
order_w = np.argsort(-(l1 - z))
print("slowest waves of the last call: loop entered, left [us]; hits; hits whose first poll succeeded; failed polls")
for w in order_w[:12]:
    print(f"  {f(l0)[w]:6.2f} {f(l1)[w]:6.2f}   hits {hits[w]:3d}  first-ok {first_ok[w]:3d}  failed polls {fails[w]:5d}")
print("by number of hits: waves, loop left median / max [us], failed polls per wave")
for h in range(int(hits.max()) + 1):
    sel = hits == h
    if sel.any():
        print(f"  {h:2d} hits: {sel.sum():5d} waves   {np.median(f(l1)[sel]):6.2f} {f(l1)[sel].max():6.2f}   {fails[sel].mean():7.1f}")
tk = rows[-1][10]                                            # s_memtime ticks per us
nhs = max(int(hb[:, 5].sum()), 1)
print(f"inside a hit step (mean over {nhs} hit steps of the last call, us): step start -> first granule load "
      f"{hb[:,0].sum()/nhs/tk:.2f}; load -> every tag right {hb[:,1].sum()/nhs/tk:.2f}; hit arithmetic {hb[:,2].sum()/nhs/tk:.2f}; "
      f"dense update {hb[:,3].sum()/nhs/tk:.2f}; publishing the touched rows {hb[:,4].sum()/nhs/tk:.2f}")
nf = max(int(hb[:, 7].sum()), 1)
print(f"hits whose first look succeeded: {int(hb[:,7].sum())} of {int(hits.sum())}; their load -> tags-right time {hb[:,6].sum()/nf/tk:.2f} us (one round trip)")
engine.check_status()
