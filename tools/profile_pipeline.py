"""Diagnostic: where one repetition of run_experiment spends its wall time (stage by stage), per configuration.

Mirrors the body of structure.run_experiment (ref:306-450) with a timer around every stage; not the product."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "matrix-factorization-with-comparison-data_amd")]
os.environ.setdefault("OMP_NUM_THREADS", "4")
import numpy as np, torch
import structure as S

dev = "cuda"
cases = {"C1": dict(n=256, m=256, d=8, p=0.05, epochs=30), "nb": dict(n=1000, m=1000, d=2, p=0.5, epochs=30),
         "nb20": dict(n=1000, m=1000, d=20, p=0.1, epochs=30), "C2": dict(n=4096, m=4096, d=64, p=0.01, epochs=30),
         "C3": dict(n=16384, m=16384, d=128, p=0.001, epochs=3, strategy="margin", reps=1),
         "C3r": dict(n=16384, m=16384, d=128, p=0.001, epochs=3, strategy="random", reps=1)}
want = sys.argv[1:] or list(cases)


class T:
    def __init__(self):
        self.rows = []

    def __call__(self, name, fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        self.rows.append((name, time.perf_counter() - t0))
        return out


for name in want:
    c = cases[name]
    n, m, d, p, epochs = c["n"], c["m"], c["d"], c["p"], c["epochs"]
    strategy = c.get("strategy", "random")
    for rep in range(c.get("reps", 2)):   # rep 0 warms caches / lazy init
        torch.manual_seed(rep); np.random.seed(rep)
        t = T()
        X = t("generate_X", lambda: S.generate_X(n, m, d, dev))
        loaders = t("split_dataset (sample+label)", lambda: S.split_dataset_from_triplets(X, int(n * m * p / 2), strategy=strategy))
        tr, va, te = loaders
        model = S.MatrixFactorization(n, m, d).to(dev)
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-5)
        t(f"train_model ({epochs} ep)", lambda: S.train_model(model, tr, va, opt, dev, num_epochs=epochs))
        t("evaluate_model", lambda: S.evaluate_model(model, te, dev))
        t("reconstruction_error", lambda: S.compute_reconstruction_error(model, X, 1.0))
        t("alpha_and_norm_ratios", lambda: S.compute_alpha_and_norm_ratios(model, X))
        t("ground_truth_metrics", lambda: S.compute_ground_truth_metrics(te, X, dev))
    tot = sum(v for _, v in t.rows)
    steps = epochs * ((len(tr.dataset) + 63) // 64)
    print(f"== {name}: n={n} m={m} d={d} p={p}  train samples={len(tr.dataset)}  steps={steps}  total={tot:.3f} s")
    for k, v in t.rows:
        print(f"   {k:32s} {v*1e3:10.1f} ms  {v/tot*100:5.1f}%")
    sys.stdout.flush()
