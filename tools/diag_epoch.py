"""Diagnostic: GPU-side and host-side breakdown of one epoch of the bench loop, without intermediate syncs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "matrix-factorization-with-comparison-data_amd")]
os.environ.setdefault("OMP_NUM_THREADS", "4")
import torch, numpy as np
import bench
from mfcd import engine
dev = torch.device("cuda:0")
r = bench.Runner(bench.C2, dev, 0)
B = 64
r.run(1049 * 2); torch.cuda.synchronize()
E = 8
ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(E)]
host = []
t_all0 = time.perf_counter()
for e in range(E):
    h0 = time.perf_counter()
    order = torch.randperm(r.train.N, generator=r.gen)
    h1 = time.perf_counter()
    ev[e][0].record()
    stream = r.train.ordered(order)
    h2 = time.perf_counter()
    ev[e][1].record()
    engine.train_steps(r.bind, stream, B)
    h3 = time.perf_counter()
    ev[e][2].record()
    engine.eval_batches(r.model.U.data, r.model.V.data, r.val.dev, B)
    ev[e][3].record()
    h4 = time.perf_counter()
    host.append((h1 - h0, h2 - h1, h3 - h2, h4 - h3))
torch.cuda.synchronize()
t_all = time.perf_counter() - t_all0
print(f"{E} epochs wall {t_all*1e3:.2f} ms = {t_all/E*1e3:.3f} ms/epoch = {t_all/E/1049*1e6:.3f} us/step")
for e in range(E):
    g = [ev[e][i].elapsed_time(ev[e][i + 1]) for i in range(3)]
    gap = ev[e - 1][3].elapsed_time(ev[e][0]) if e else 0.0
    print(f"epoch {e}: GPU gather {g[0]*1e3:7.1f} us | train(memset+kernel+mean) {g[1]*1e3:8.1f} us | eval {g[2]*1e3:6.1f} us | idle before {gap*1e3:7.1f} us"
          f" || host randperm {host[e][0]*1e3:.2f} ms, ordered {host[e][1]*1e3:.2f}, train_steps {host[e][2]*1e3:.2f}, eval {host[e][3]*1e3:.2f}")
