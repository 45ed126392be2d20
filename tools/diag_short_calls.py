"""Diagnostic: where a SHORT fused-step call (the driver's --steps 20) spends its time: GPU-side event time of one call
from an idle chip, after a busy ramp, and back to back; host-side enqueue time per call.  Not part of the product."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "matrix-factorization-with-comparison-data_amd")]
os.environ.setdefault("OMP_NUM_THREADS", "4")
import torch  # noqa: E402

import bench  # noqa: E402
from mfcd import engine, metrics  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
r = bench.Runner(dict(bench.C2, name="C2"), dev, 0)
B = r.cfg["B"]
stream = r.train.ordered(torch.randperm(r.train.N, generator=r.gen))
X = torch.randn(4096, 4096, device=dev)


def one_call(k0):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    engine.train_steps(r.bind, stream[k0 * B:(k0 + K) * B], B)
    e1.record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return e0.elapsed_time(e1) * 1e3, (t1 - t0) * 1e6, (t2 - t0) * 1e6


engine.train_steps(r.bind, stream[:5 * B], B)
torch.cuda.synchronize()
time.sleep(0.5)
print(f"K={K}  [events us, host enqueue us, wall us]")
print("idle chip      :", [round(x, 1) for x in one_call(5)])
print("again (50 us later):", [round(x, 1) for x in one_call(5 + K)])
for _ in range(300):
    metrics.uvt_stats(r.model.U.data, r.model.V.data, X, 1.0)
print("after busy ramp:", [round(x, 1) for x in one_call(5 + 2 * K)])
ev, hs, wl = [], [], []
for c in range(40):
    a, b, w = one_call((5 + 3 * K + c * K) % (r.steps_per_epoch - K))
    ev.append(a); hs.append(b); wl.append(w)
ev.sort(); hs.sort(); wl.sort()
print(f"40 calls, sync between: events min {ev[0]:.1f} med {ev[20]:.1f} | host enqueue min {hs[0]:.1f} med {hs[20]:.1f} | "
      f"wall min {wl[0]:.1f} med {wl[20]:.1f}")
# back to back without sync: GPU-side period per call
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter()
e0.record()
for c in range(40):
    engine.train_steps(r.bind, stream[(c * K) * B:(c * K + K) * B], B)
e1.record()
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"40 calls back to back: {e0.elapsed_time(e1) * 1e3 / 40:.1f} us per call on the GPU, host {1e6 * (t1 - t0) / 40:.1f} us per call")
engine.check_status()

# host-side split of one call: Python before / the C-ABI call itself / Python after
from mfcd import _lib
L = _lib.load()
real = L.mfcd_train_call_run
spans = []


def timed_entry(*a):
    t0 = time.perf_counter()
    rc = real(*a)
    spans.append((t0, time.perf_counter()))
    return rc


L.mfcd_train_call_run = timed_entry
r.bind.drop_prepared()      # the fast path caches the bound entry: make it re-bind to the timed one
tot, pre, cabi, post = [], [], [], []
for c in range(60):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    engine.train_steps(r.bind, stream[(c * K) * B:(c * K + K) * B], B)
    t1 = time.perf_counter()
    a, b = spans[-1]
    tot.append(t1 - t0); pre.append(a - t0); cabi.append(b - a); post.append(t1 - b)
L.mfcd_train_call_run = real
med = lambda x: sorted(x)[len(x) // 2] * 1e6
print(f"host split (median of 60): total {med(tot):.1f} us = python before {med(pre):.1f} + C-ABI call {med(cabi):.1f} + python after {med(post):.1f}")
