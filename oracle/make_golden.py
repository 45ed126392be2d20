#!/usr/bin/env python3
"""Golden-vector generator (TEST INFRASTRUCTURE — runs in the dev container only).

Imports the *unmodified* reference from /root/reference and dumps small input/output
vectors into tests/golden/*.npz (+ one JSON schema).  Only the generated data is
committed; the reference itself never travels.  Nothing under the product package
imports this file.

The only accommodation made for the reference is the one SURVEY.md §8c records:
`torch.utils.tensorboard` is not installed, and structure.py:10 imports
`SummaryWriter` from it although every use sits under `if False:` (structure.py:831,
871, 875).  A placeholder module object is registered under that name before the
import so that the import statement succeeds; no reference code path touches it.

Usage:  OMP_NUM_THREADS=4 PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py
"""
import json
import os
import sys
import types

os.environ.setdefault("OMP_NUM_THREADS", "4")
sys.dont_write_bytecode = True

_stub = types.ModuleType("torch.utils.tensorboard")
_stub.SummaryWriter = object
sys.modules["torch.utils.tensorboard"] = _stub
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import structure as R  # noqa: E402  (the reference)

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def _np(t):
    return t.detach().cpu().numpy().copy()


# --------------------------------------------------------------------------------------
# (i) single-step known-answer tests: reference forward (structure.py:773-795), BCE (849),
#     autograd backward (850) and torch.optim.Adam (364, 851) on hand-made batches.
# --------------------------------------------------------------------------------------
def kat(name, n, m, d, batches, lr, wd, scale_init=1.0, seed=0):
    torch.manual_seed(seed)
    model = R.MatrixFactorization(n, m, d)
    with torch.no_grad():
        model.U.mul_(scale_init)
        model.V.mul_(scale_init)
    opt = torch.optim.Adam(model.parameters(), lr=lr, weight_decay=wd)
    out = {"n": n, "m": m, "d": d, "lr": lr, "wd": wd, "n_steps": len(batches),
           "U0": _np(model.U), "V0": _np(model.V)}
    for k, (u, i, j, z) in enumerate(batches):
        # exactly the body of the reference loop, structure.py:846-852
        opt.zero_grad()
        pred = model(u, i, j)
        loss = F.binary_cross_entropy(pred, z.float())
        loss.backward()
        out[f"u{k}"], out[f"i{k}"], out[f"j{k}"], out[f"z{k}"] = _np(u), _np(i), _np(j), _np(z)
        out[f"p{k}"] = _np(pred)
        out[f"loss{k}"] = np.float64(loss.item())
        out[f"dU{k}"] = _np(model.U.grad)
        out[f"dV{k}"] = _np(model.V.grad)
        opt.step()
        out[f"U{k + 1}"], out[f"V{k + 1}"] = _np(model.U), _np(model.V)
        st_u, st_v = opt.state[model.U], opt.state[model.V]
        out[f"mU{k + 1}"], out[f"vU{k + 1}"] = _np(st_u["exp_avg"]), _np(st_u["exp_avg_sq"])
        out[f"mV{k + 1}"], out[f"vV{k + 1}"] = _np(st_v["exp_avg"]), _np(st_v["exp_avg_sq"])
    np.savez_compressed(os.path.join(OUT, f"kat_{name}.npz"), **out)
    print("wrote kat", name)


def rand_batch(g, n, m, B, soft_K=None):
    u = torch.randint(0, n, (B,), generator=g)
    i = torch.randint(0, m, (B,), generator=g)
    j = torch.randint(0, m, (B,), generator=g)
    j = torch.where(i == j, (j + 1) % m, j)
    if soft_K:
        z = torch.randint(0, soft_K + 1, (B,), generator=g).double() / soft_K
    else:
        z = torch.randint(0, 2, (B,), generator=g).double()
    return u, i, j, z  # label dtype float64, as the DataLoader collates Python floats


def make_kats():
    g = torch.Generator().manual_seed(1234)
    # A: heavy duplicates (n=16, m=12), full batches, hard labels
    kat("dups_d8", 16, 12, 8, [rand_batch(g, 16, 12, 64) for _ in range(3)], 1e-3, 1e-5)
    # B: short batch (B=30), soft labels K=4, d=64
    kat("short_soft_d64", 32, 32, 64, [rand_batch(g, 32, 32, 30, soft_K=4) for _ in range(3)], 1e-3, 1e-5)
    # C: saturated sigmoid (|x|>20 → p rounds to 0/1, loss term 100, zero gradient)
    kat("saturated_d8", 24, 24, 8, [rand_batch(g, 24, 24, 64) for _ in range(3)], 1e-3, 1e-5, scale_init=12.0)
    # D: notebook default d=2
    kat("d2", 50, 50, 2, [rand_batch(g, 50, 50, 64) for _ in range(3)], 1e-3, 1e-5)
    # E: d=128, stronger decay / lr (Runs.ipynb sweeps wd up to 5e-3)
    kat("d128_wd", 40, 40, 128, [rand_batch(g, 40, 40, 64) for _ in range(3)], 1e-2, 5e-3)
    # F: d=256 with a batch of one
    kat("d256_b1", 20, 20, 256, [rand_batch(g, 20, 20, 1), rand_batch(g, 20, 20, 64)], 1e-3, 1e-5)
    # G: odd d (not a multiple of 4)
    kat("d5", 30, 30, 5, [rand_batch(g, 30, 30, 64) for _ in range(2)], 1e-3, 1e-4)


# --------------------------------------------------------------------------------------
# (ii) end-to-end: generate_X → split → model/Adam → train_model → every metric.
# --------------------------------------------------------------------------------------
class RecordingLoader:
    """Pass-through around a reference DataLoader that remembers what each epoch yielded.
    train_model only iterates the loader and takes len() (structure.py:845, 854)."""

    def __init__(self, loader):
        self.loader, self.epochs = loader, []

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        rec = []
        self.epochs.append(rec)
        for batch in self.loader:
            rec.append([b.clone() for b in batch])
            yield batch


def data_array(ds):
    return np.asarray(ds.data, dtype=np.float64).reshape(-1, 4)


def e2e(name, n, m, d, p, s, K, soft_label, lr, wd, epochs, strategy="random", seed=0):
    torch.manual_seed(seed)
    np.random.seed(seed)
    X = R.generate_X(n, m, d, "cpu")
    num_triplets = int(n * m * p / 2)
    train_loader, val_loader, test_loader = R.split_dataset_from_triplets(
        X, num_triplets, scale=s, K=K, strategy=strategy, soft_label=soft_label)
    model = R.MatrixFactorization(n, m, d)
    opt = torch.optim.Adam(model.parameters(), lr=lr, weight_decay=wd)
    out = {"n": n, "m": m, "d": d, "p": p, "s": s, "K": K, "soft_label": soft_label,
           "lr": lr, "wd": wd, "epochs": epochs, "X": _np(X),
           "train_data": data_array(train_loader.dataset),
           "val_data": data_array(val_loader.dataset),
           "test_data": data_array(test_loader.dataset),
           "U0": _np(model.U), "V0": _np(model.V),
           "rng_state_before_train": _np(torch.get_rng_state())}
    rec = RecordingLoader(train_loader)
    tl, vl = R.train_model(model, rec, val_loader, opt, "cpu", num_epochs=epochs)
    out["train_losses"], out["val_losses"] = np.array(tl), np.array(vl)
    # per-epoch order as [E, N, 4] (u, i, j, z) in the order consumed
    out["epoch_stream"] = np.stack([
        np.concatenate([torch.stack([b[0].double(), b[1].double(), b[2].double(), b[3].double()], 1).numpy()
                        for b in ep]) for ep in rec.epochs])
    out["rng_state_after_train"] = _np(torch.get_rng_state())
    out["U_final"], out["V_final"] = _np(model.U), _np(model.V)
    for nm, prm in (("U", model.U), ("V", model.V)):
        out[f"m{nm}_final"] = _np(opt.state[prm]["exp_avg"])
        out[f"v{nm}_final"] = _np(opt.state[prm]["exp_avg_sq"])
    out["adam_step"] = np.float64(opt.state[model.U]["step"].item())
    te_loss, te_acc = R.evaluate_model(model, test_loader, "cpu")
    out["test_loss"], out["test_acc"] = np.float64(te_loss), np.float64(te_acc)
    out["rec_error"] = np.float64(R.compute_reconstruction_error(model, X, s))
    res = R.compute_alpha_and_norm_ratios(model, X)
    names = ["alpha", "norm_X", "norm_ratio", "rec_scaled", "pearson_mean", "pearson_std",
             "spearman_mean", "spearman_std", "svd_err", "slopes", "correlations",
             "spearman_scores", "rec_scaled_per_row", "alpha_per_row"]
    for nm, v in zip(names, res):
        out["m14_" + nm] = np.asarray(v, dtype=np.float64)
    with torch.no_grad():  # structure.py:388-392
        UVT_full = torch.matmul(model.U, model.V.t())
        rand_indices = torch.randperm(X.shape[0])[:2]
        out["sampled_idx"] = _np(rand_indices)
        out["sampled_X_rows"] = X[rand_indices].cpu().numpy()
        out["sampled_UVT_rows"] = UVT_full[rand_indices].cpu().numpy()
    gt_loss, gt_acc = R.compute_ground_truth_metrics(test_loader, X, "cpu")
    out["gt_loss"], out["gt_acc"] = np.float64(gt_loss), np.float64(gt_acc)
    np.savez_compressed(os.path.join(OUT, f"e2e_{name}.npz"), **out)
    print("wrote e2e", name, "train", len(train_loader.dataset), "losses", tl[:2], "…", tl[-1])


# --------------------------------------------------------------------------------------
# (iii) result-dict / .pkl layout (structure.py:420-444, 172-200) as a JSON schema.
# --------------------------------------------------------------------------------------
def describe(o):
    if isinstance(o, dict):
        return {"type": "dict", "items": {k: describe(v) for k, v in o.items()}}
    if isinstance(o, (list, tuple)):
        d = {"type": type(o).__name__, "len": len(o)}
        if len(o):
            d["elem0"] = describe(o[0])
        return d
    if isinstance(o, np.ndarray):
        return {"type": "ndarray", "dtype": str(o.dtype), "shape": list(o.shape)}
    return {"type": type(o).__module__ + "." + type(o).__name__
            if type(o).__module__ != "builtins" else type(o).__name__}


def make_schema():
    import pickle
    import tempfile
    torch.manual_seed(3)
    np.random.seed(3)
    kw = dict(n=40, m=30, d=4, p=0.5, s=1.0, device="cpu", lr=1e-3, weight_decay=1e-5,
              num_epochs=2, reps=2)
    res = R.run_experiment(kw["n"], kw["m"], kw["d"], kw["p"], kw["s"], "cpu", kw["lr"],
                           kw["weight_decay"], reps=2, num_epochs=2)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "sub", "scan.pkl")
        ret = R.parameter_scan(n=40, m=30, d=[2, 4], p=0.5, num_epochs=1, reps=1, save_path=path,
                               save_every=1)
        with open(path, "rb") as f:  # a file this script just wrote
            saved = pickle.load(f)
    ret2 = R.parameter_scan(n=40, m=30, d=2, p=0.5, num_epochs=1, reps=1)
    schema = {"run_experiment": describe(res),
              "parameter_scan_return_with_save_path": describe(ret),
              "parameter_scan_saved": describe(saved),
              "parameter_scan_saved_params0": saved[0]["params"],
              "parameter_scan_return_no_save": describe(ret2)}
    with open(os.path.join(OUT, "result_schema.json"), "w") as f:
        json.dump(schema, f, indent=1, sort_keys=True)
    print("wrote schema")


if __name__ == "__main__":
    make_kats()
    # C1 of BASELINE.json: n=m=256, d=8, p=0.05, s=1, K=1, random triplets
    e2e("c1", 256, 256, 8, 0.05, 1.0, 1, False, 1e-3, 1e-5, epochs=5)
    # soft labels, K=3, n != m, larger scale and decay
    e2e("soft_k3", 64, 48, 4, 0.3, 2.0, 3, True, 1e-3, 1e-3, epochs=3, seed=1)
    # hard labels with K=2 repeats and d=16
    e2e("hard_k2_d16", 96, 80, 16, 0.1, 1.0, 2, False, 5e-3, 1e-5, epochs=3, seed=2)
    make_schema()
