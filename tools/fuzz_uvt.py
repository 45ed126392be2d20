"""Diagnostic: randomised sweep of the UV^T metric pass (all three epilogues, tiled and generic forms, ragged shapes)
and of the Spearman kernel against the C oracle / scipy.  python tools/fuzz_uvt.py [trials] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "matrix-factorization-with-comparison-data_amd")]
os.environ.setdefault("OMP_NUM_THREADS", "4")
import numpy as np, torch, scipy.stats
from mfcd import metrics
from oracle import oracle as O

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")
orc = O.COracle()
bad = 0
for t in range(trials):
    d = int(rng.choice([1, 2, 3, 8, 16, 32, 33, 64, 100, 128, 256, 300]))
    n = int(rng.integers(1, 700)) if rng.random() < 0.85 else int(rng.integers(1, 40))
    m = int(rng.integers(1, 900)) if rng.random() < 0.85 else int(rng.integers(1, 40))
    s = float(rng.choice([1.0, 0.3, 2.5]))
    U = (rng.standard_normal((n, d)) / np.sqrt(d)).astype(np.float32)
    V = (rng.standard_normal((m, d)) / np.sqrt(d)).astype(np.float32)
    X = (rng.standard_normal((n, m)) * 0.5 + rng.choice([0.0, 0.3, -2.0])).astype(np.float32)
    if rng.random() < 0.2:
        X[rng.integers(0, n)] = 0.25                     # a constant row (zero variance)
    Ud, Vd, Xd = (torch.from_numpy(a).to(dev) for a in (U, V, X))
    ref_rows, err2, ref2 = orc.uvt_stats(U, V, X, s)
    msg = []
    for what in (3, 1, 2):
        rs, sc = metrics.uvt_stats(Ud, Vd, Xd, s, what=what)
        if what & 1:
            r = rs.cpu().numpy()
            for col in range(3):
                if m == 1:
                    break                                             # one column: every centred sum is 0 up to rounding noise
                scale = max(np.abs(ref_rows[:, col]).max(), 1e-9)
                e = np.abs(r[:, col] - ref_rows[:, col]).max() / scale
                if not e <= 5e-5:
                    msg.append(f"what={what} col{col} rel {e:.1e}")
        if what & 2:
            c = sc.cpu().numpy()
            if not abs(c[0] - err2) <= 5e-5 * max(err2, 1e-20) or not abs(c[1] - ref2) <= 5e-5 * max(ref2, 1e-20):
                msg.append(f"what={what} scal {c[0]:.6g}/{err2:.6g} {c[1]:.6g}/{ref2:.6g}")
    if m >= 2 and t % 3 == 0:                                # Spearman kernel against scipy on a few rows
        A = Ud @ Vd.t()
        rho = metrics.spearman_rows(A, Xd).cpu().numpy()
        Ah = A.cpu().numpy()
        for r in rng.integers(0, n, min(n, 4)):
            want = scipy.stats.spearmanr(Ah[r], X[r]).statistic
            if np.isnan(want) != np.isnan(rho[r]) or (not np.isnan(want) and abs(want - rho[r]) > 1e-9):
                msg.append(f"spearman row {r}: {rho[r]} vs {want}")
    if msg:
        bad += 1
        print(f"trial {t}: n={n} m={m} d={d} s={s}: " + "; ".join(msg[:4]), flush=True)
print(f"done: {trials} trials, {bad} bad")
