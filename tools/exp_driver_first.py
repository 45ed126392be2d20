"""Diagnostic: the bench's timed region exactly (fresh process: Runner, 5 warm-up steps, ramp, sync, ONE 20-step call,
sync) in variants, to find what makes the call after the ramp ~9 us slower on the host than a repeated call."""
import os, sys, time, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "matrix-factorization-with-comparison-data_amd")]
import torch
import bench
from mfcd import engine

cfg = dict(bench.WORKLOADS["C2"]); cfg["name"] = "C2"
dev = torch.device("cuda:0")
r = bench.Runner(cfg, dev, 0)
r.run(5)
scratch = copy.deepcopy(r.model)
sopt = torch.optim.Adam(scratch.parameters(), lr=cfg["lr"], weight_decay=cfg["wd"])
sbind = engine.AdamBinding(scratch, sopt)

def ramp_keep(seconds):          # as bench.clock_ramp, but on a scratch binding that stays alive
    t0 = time.perf_counter()
    pos = r.pos
    while time.perf_counter() - t0 < seconds:
        for _ in range(16):
            r.pos = 0
            r.run(20, record=False, bind=sbind)
        torch.cuda.synchronize()
    r.pos = pos

for mode in sys.argv[1:] or ["plain"]:
    for it in range(5):
        if mode == "keep":
            ramp_keep(0.3)
        else:
            bench.clock_ramp(r, 0.3)
        if mode == "touch":      # touch the real model's state once, untimed, before the timed call
            r.run(20); torch.cuda.synchronize()
        if mode == "touch_other":
            pos = r.pos; r.run(20, bind=sbind); r.pos = pos; torch.cuda.synchronize()
        if mode == "sleep":
            time.sleep(0.0005)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r.run(20, record=False)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{mode} it={it}: enqueue {1e6*(t1-t0):.1f} us, wall {1e6*(t2-t0):.1f} us", flush=True)
