"""Diagnostic: where does host time go in an epoch of fused steps? (not part of the product)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "matrix-factorization-with-comparison-data_amd")]
import torch, numpy as np
import bench
from mfcd import engine

dev = torch.device("cuda:0")
r = bench.Runner(bench.C2, dev, 0)
B = 64
r.run(1049); torch.cuda.synchronize()
for trial in range(3):
    t0 = time.perf_counter()
    order = torch.randperm(r.train.N, generator=r.gen)
    t1 = time.perf_counter()
    stream = r.train.ordered(order)
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    engine.train_steps(r.bind, stream, B)
    t4 = time.perf_counter()
    torch.cuda.synchronize()
    t5 = time.perf_counter()
    engine.eval_batches(r.model.U.data, r.model.V.data, r.val.dev, B)
    t6 = time.perf_counter()
    torch.cuda.synchronize()
    t7 = time.perf_counter()
    print(f"randperm {1e3*(t1-t0):.2f} ms | ordered(H2D+gather) {1e3*(t2-t1):.2f} | sync {1e3*(t3-t2):.2f} | "
          f"train_steps host {1e3*(t4-t3):.2f} ms ({1e6*(t4-t3)/1049:.2f} us/launch) | drain {1e3*(t5-t4):.2f} | "
          f"eval host {1e3*(t6-t5):.3f} | eval drain {1e3*(t7-t6):.3f}")
# many small calls: host cost of one call with 1 step
t0 = time.perf_counter()
for k in range(200):
    engine.train_steps(r.bind, stream[k*64:(k+1)*64], B)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"single-step calls: host {1e6*(t1-t0)/200:.1f} us/call, drain {1e3*(t2-t1):.2f} ms")
