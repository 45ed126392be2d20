"""Diagnostic: randomised parity sweep of the fused step (every form that applies, fp32; split into several calls on one
planned workspace) against the C oracle and against each other.  python tools/fuzz_parity.py [trials] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "matrix-factorization-with-comparison-data_amd")]
os.environ.setdefault("OMP_NUM_THREADS", "4")
import numpy as np, torch
import structure as S
from mfcd import engine
from oracle import oracle as O

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
dev = torch.device("cuda:0")
orc = O.COracle()
bad = 0
t_start = time.time()
for t in range(trials):
    d = int(rng.choice([1, 2, 3, 4, 5, 8, 12, 16, 32, 33, 64, 100, 128, 256, 300]))
    n = int(rng.integers(1, 3000)) if rng.random() < 0.8 else int(rng.integers(1, 20))
    m = int(rng.integers(2, 3000)) if rng.random() < 0.8 else int(rng.integers(2, 20))
    B = int(rng.choice([1, 7, 64, 64, 64, 100, 256]))
    N = int(rng.integers(1, 30)) * B + int(rng.integers(0, B))
    soft = bool(rng.random() < 0.3)
    U0 = (rng.standard_normal((n, d)) / np.sqrt(d)).astype(np.float32)
    V0 = (rng.standard_normal((m, d)) / np.sqrt(d)).astype(np.float32)
    u, i = rng.integers(0, n, N), rng.integers(0, m, N)
    if rng.random() < 0.3:                                   # a hot head: duplicates inside every batch
        u[rng.random(N) < 0.5] = 0
        i[rng.random(N) < 0.5] = 0
    j = (i + 1 + rng.integers(0, max(m - 1, 1), N)) % m
    z = rng.integers(0, 5, N) / 4.0 if soft else rng.integers(0, 2, N).astype(np.float64)
    lr, wd = float(rng.choice([1e-3, 1e-2])), float(rng.choice([0.0, 1e-5, 1e-3]))
    cuts = sorted({0, N} | {int(c) * B for c in rng.integers(0, N // B + 1, 3)})
    rows = np.stack([u, i, j, z], 1).astype(np.float64)
    rec = engine.SampleStore(rows, n, m, dev).dev
    ref = O.new_state(U0, V0)
    ref_loss = orc.train_steps(ref, u, i, j, z, B, 0, lr=lr, wd=wd, threads=4)
    results = {}
    for form, math in (("streaming", "ieee"), ("resident", "ieee"), ("resident", "fast"), ("local", "ieee"), ("auto", "fast")):
        engine.set_train_path(form)
        engine.set_resident_math(math)
        try:
            plan = engine.train_plan(N, B, n, m, d)
        except Exception:
            plan = None
        if form in ("resident", "local") and (plan is None or plan["form_name"] != form):
            continue
        model = S.MatrixFactorization(n, m, d)
        with torch.no_grad():
            model.U.copy_(torch.from_numpy(U0)); model.V.copy_(torch.from_numpy(V0))
        model = model.to(dev)
        opt = torch.optim.Adam(model.parameters(), lr=lr, weight_decay=wd)
        bind = engine.AdamBinding(model, opt)
        try:
            loss = torch.cat([engine.train_steps(bind, rec[a:b], B).clone() for a, b in zip(cuts[:-1], cuts[1:]) if b > a])
            engine.check_status()
        except Exception as e:
            print(f"trial {t}: {form}/{math} n={n} m={m} d={d} B={B} N={N}: EXCEPTION {type(e).__name__}: {e}", flush=True)
            bad += 1
            continue
        results[(form, math)] = (loss.cpu().numpy(), model.U.data.cpu().numpy(), model.V.data.cpu().numpy())
    engine.set_train_path("auto"); engine.set_resident_math("fast")
    nsteps = len(ref_loss)
    tol = (2e-6 + 2e-8 * nsteps) * (lr / 1e-3)      # rounding differences scale with the update size
    base = results.get(("streaming", "ieee"))
    for key, (loss, U, V) in results.items():
        dl = np.abs(loss - ref_loss).max() / max(1.0, np.abs(ref_loss).max())
        du, dv = np.abs(U - ref["U"]), np.abs(V - ref["V"])
        frac = max((du > tol).mean(), (dv > tol).mean())
        worst = max(du.max(), dv.max())
        same = base is not None and np.array_equal(U, base[1]) and np.array_equal(V, base[2])
        ok = dl <= 3e-5 and frac <= 1e-3 and worst <= 20 * lr
        if not ok:
            bad += 1
        if not ok or (key[1] == "ieee" and not same):
            print(f"trial {t}: {key} n={n} m={m} d={d} B={B} N={N} soft={soft} lr={lr} wd={wd} cuts={cuts}: loss {dl:.1e} "
                  f"frac>{tol:.1e} {frac:.1e} worst {worst:.1e} bit-equal-to-streaming {same} {'OK' if ok else 'FAIL'}", flush=True)
    if t % 10 == 9:
        print(f"... {t + 1} trials, {bad} bad, {time.time() - t_start:.0f} s", flush=True)
print(f"done: {trials} trials, {bad} bad")
