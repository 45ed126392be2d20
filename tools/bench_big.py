"""Step period of the big resident form (mfcd_train_steps_big: the whole Adam state of a d = 64 model of up to 8.39 M
elements in registers / LDS of one GPU) against the streaming form, on uniform triplets.
python tools/bench_big.py [steps] > profiles/rNN_big_resident.txt"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "matrix-factorization-with-comparison-data_amd")]
os.environ.setdefault("OMP_NUM_THREADS", "4")
import numpy as np, torch
import structure as S
from mfcd import engine

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
dev = torch.device("cuda:0")
B, d = 64, 64
print(f"B = {B}, d = {d}, {steps} steps per call, uniform triplets; us per optimiser step (HIP events around the call, best of 3)")
for name, n, m in (("C4: 65536 x 65536", 65536, 65536), ("32768 x 32768", 32768, 32768), ("16384 x 16384", 16384, 16384)):
    rng = np.random.default_rng(1)
    N = steps * B
    rows = np.stack([rng.integers(0, n, N), rng.integers(0, m, N), rng.integers(0, m, N), rng.integers(0, 2, N)], 1).astype(np.float64)
    rows[:, 2] = (rows[:, 1] + 1 + rng.integers(0, m - 1, N)) % m
    st = engine.SampleStore(rows, n, m, dev)
    out = {}
    for form in ("streaming", "auto", "big_ieee", "big"):
        model = S.MatrixFactorization(n, m, d).to(dev)
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-5)
        bind = engine.AdamBinding(model, opt)
        if form.startswith("big"):
            big = engine.BigResident(bind)
            engine.set_resident_math("ieee" if form == "big_ieee" else "fast")
            run = lambda: big.train_steps(st.dev, B)          # noqa: E731
        else:
            engine.set_train_path(form)
            run = lambda: engine.train_steps(bind, st.dev, B)   # noqa: E731
        try:
            run(); torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); run(); e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1e3 / steps)
            if form.startswith("big"):
                big.status()
            else:
                engine.check_status()
            out[form] = best
        finally:
            engine.set_train_path("auto")
            engine.set_resident_math("fast")
    plan = engine.train_plan(N, B, n, m, d)["form_name"]
    if plan == "streaming" and n + m <= 131072:
        plan = "big resident, chosen by engine.train_steps"
    print(f"{name:20s} streaming {out['streaming']:7.2f}   auto ({plan}) {out['auto']:7.2f}   big resident {out['big']:7.2f} (IEEE flavour {out['big_ieee']:7.2f})", flush=True)
