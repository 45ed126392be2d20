"""MI355X-native drop-in for the reference module of the same name.

Same function names, argument meaning, return types and `.pkl` layout as
MayeulCassier/Matrix-Factorization-With-Comparison-Data `structure.py` (cited below as ref:LINE),
so `Runs.ipynb` / `Plots.ipynb` work unchanged — but the training step, the evaluation pass and the
dense UV^T metrics run as hand-written gfx950 kernels behind the C-ABI in include/mfcd.h.
This file is host glue only; it holds no arithmetic of the hot path and has no CPU fallback:
`device` must name a GPU.
"""
import itertools
import os
import pickle

os.environ.setdefault("OMP_NUM_THREADS", "4")  # the reference pins this at import (ref:3); host work here is tiny

import numpy as np
import torch
import torch.nn as nn
from torch.utils.data import DataLoader, Dataset

from generation_data import *  # noqa: F401,F403  (ref:17 re-exports every sampler/generator name)
import generation_data as _gd
from mfcd import engine as _engine
from mfcd import metrics as _metrics
from mfcd import sampling as _sampling

if torch.get_num_threads() > 16:  # imported after torch: keep the host-side pool small (GPU boxes expose 100s of cores)
    torch.set_num_threads(4)

try:  # progress bars are cosmetic (ref:840)
    from tqdm import tqdm as _tqdm
except Exception:  # pragma: no cover
    _tqdm = None


# ------------------------------------------------------------------------------------------------
# model (ref:746-795)
# ------------------------------------------------------------------------------------------------
class MatrixFactorization(nn.Module):
    """BTL comparison model: P(u prefers i over j) = sigmoid(U[u] . (V[i] - V[j])).

    Parameters `.U [n_users, d]`, `.V [n_items, d]`, fp32, drawn N(0, 1/d) in this order (ref:770-771).
    Calling the module evaluates the forward kernel; `train_model` fuses forward, backward and Adam on the device
    for torch.optim.Adam and runs the reference's per-batch loop over the same kernels for any other optimiser."""

    def __init__(self, n_users, n_items, d, dtype=torch.float32):
        """`dtype=torch.bfloat16` (extension, BASELINE configs[2]) stores the factors in bf16: same fp32 draws,
        rounded to nearest even; training then keeps fp32 Adam moments and rounds once per step."""
        super().__init__()
        scale = torch.sqrt(torch.tensor(d, dtype=torch.float32))
        self.U = nn.Parameter((torch.randn(n_users, d) / scale).to(dtype))
        self.V = nn.Parameter((torch.randn(n_items, d) / scale).to(dtype))

    def forward(self, u, i, j):
        """ref:773-795 on the forward kernel.  fp32 factors that require grad get an autograd graph (custom Function:
        backward = sigmoid backward + the scatter-accumulate kernel), so `loss.backward()` on the result fills
        `.U.grad` / `.V.grad` as it does in the reference; bf16 factors and no-grad contexts return a plain tensor."""
        rec = _engine.records_from_indices(u, i, j, self.U.shape[0], self.V.shape[0], self.U.device)
        if torch.is_grad_enabled() and self.U.dtype == torch.float32 and (self.U.requires_grad or self.V.requires_grad):
            return _engine.TripletForward.apply(self.U, self.V, rec)
        _, _, p = _engine.eval_batches(self.U.data, self.V.data, rec, min(max(rec.shape[0], 1), 4096), want_p=True)
        return p


# ------------------------------------------------------------------------------------------------
# training / evaluation (ref:812-921)
# ------------------------------------------------------------------------------------------------
def _need_gpu(device):
    if torch.device(device).type != "cuda":
        raise RuntimeError(f"device={device!r}: this build runs the triplet hot path on an MI355X only; "
                           "pass device='cuda' (there is deliberately no CPU fallback)")


def train_model(model, train_loader, val_loader, optimizer, device, num_epochs=100, is_last=False,
                open_browser=False):
    """ref:812-878.  Returns (train_losses, val_losses), one Python float per epoch
    (mean over batches of the batch-mean BCE; the short last batch weighs like any other).
    `is_last` / `open_browser` only ever fed the reference's disabled TensorBoard block."""
    _need_gpu(device)
    model.train()
    progress = (lambda it: _tqdm(it, desc="Training Progress")) if _tqdm is not None else None
    if _engine.fused_step_applies(model, optimizer):
        out = _engine.fit(model, train_loader, val_loader, optimizer, num_epochs, progress)
    else:   # any other optimiser (or Adam flags the fused step does not implement): the generic loop, same kernels
        out = _engine.fit_generic(model, train_loader, val_loader, optimizer, num_epochs, progress)
    model.eval()
    return out


def evaluate_model(model, test_loader, device):
    """ref:881-921 → (mean batch BCE, accuracy of (p > 0.5) against the labels)."""
    _need_gpu(device)
    model.eval()
    return _engine.evaluate(model, test_loader)


def compute_reconstruction_error(model, X, s):
    """ref:925-955 → ||(UV^T - column mean) - sX||_F / ||sX||_F as a float."""
    return _metrics.reconstruction_error(model.U.data, model.V.data, X, s)


def compute_alpha_and_norm_ratios(model, X_init):
    """ref:958-1082 → the 14-tuple in the reference's order."""
    return _metrics.alpha_and_norm_ratios(model.U.data, model.V.data, X_init)


def compute_ground_truth_metrics(test_loader, X, device):
    """ref:1085-1127: MSE between sigmoid(X[u,i]-X[u,j]) (no scale) and the labels, per batch, and
    the accuracy of (diff > 0).  Two-element gather per sample, once per experiment: torch ops on
    `device`, not a kernel."""
    rows = torch.from_numpy(_engine.dataset_records(test_loader.dataset)).to(X.device)
    order, bs = _engine.epoch_order(test_loader)  # same RNG draw as iterating the loader
    rows = rows[order.to(X.device)]
    u, i, j = rows[:, 0].long(), rows[:, 1].long(), rows[:, 2].long()
    z = rows[:, 3].float()
    diff = X[u, i] - X[u, j]
    se = (torch.sigmoid(diff) - z) ** 2
    total, loss_sum, nb = rows.shape[0], 0.0, 0
    for off in range(0, total, bs):
        loss_sum += se[off:off + bs].mean().item()
        nb += 1
    correct = ((diff > 0).float() == z).sum().item()
    return loss_sum / max(nb, 1), (correct / total if total > 0 else 0.0)


# ------------------------------------------------------------------------------------------------
# data (ref:465-742)
# ------------------------------------------------------------------------------------------------
_LABEL_DEVICE = None


def set_label_device(device):
    """Extension (not in the reference): draw BTL labels on `device` (a GPU) from now on; None (default) restores the
    host path, which consumes torch's CPU generator exactly like the reference."""
    global _LABEL_DEVICE
    if device is not None:
        _need_gpu(device)
    _LABEL_DEVICE = None if device is None else torch.device(device)


class BTLPreferenceDataset(Dataset):
    """ref:465-531.  `.data` is a list of (u, i, j, label) tuples.  Labels are drawn with ONE
    vectorised torch.bernoulli call over all rows, which consumes the CPU generator exactly like the
    reference's one-call-per-row loop (same serial kernel, same order).

    The rows are kept as one float64 [N, 4] array (what the device path uploads); the Python list of
    tuples behind `.data` is only built when somebody reads `.data`, and from then on that list is
    the source of truth (callers may edit or replace it, as they can with the reference's)."""

    def __init__(self, triplets, X, scale=1.0, K=1, soft_label=False, train=False):
        """With `set_label_device(device)` in force (extension; off by default) the labels are drawn and the records
        built ON that GPU (mfcd_generate_labels) — same law per label, Philox stream keyed by one int64
        taken from torch's global generator instead of the CPU Mersenne-Twister stream, so runs are reproducible but
        not bit-comparable with the reference; nothing of the dataset touches host memory unless `.data` is read."""
        self.X, self.scale, self.soft_label = X, scale, soft_label
        self._data = None
        self._dev = None
        device_labels = _LABEL_DEVICE
        if device_labels is not None:
            seed = int(torch.empty((), dtype=torch.int64).random_().item())
            self._dev = _engine.generate_labels(triplets, X, scale=scale, K=K, soft=bool(soft_label and train),
                                                seed=seed, device=device_labels)
            self._rows = None
            return
        self._rows = self._label_rows(triplets, K, train)

    def _label_rows(self, triplets, K, train):
        idx = np.asarray(list(triplets) if not isinstance(triplets, (list, np.ndarray)) else triplets,
                         dtype=np.int64).reshape(-1, 3)
        if idx.shape[0] == 0:
            return np.empty((0, 4), dtype=np.float64)
        it = torch.from_numpy(idx)
        if isinstance(self.X, _gd.FactoredMatrix):     # X kept as factors (C4 / C5 sizes): entries on demand, fp32
            diff = torch.from_numpy(self.X.entries(idx[:, 0], idx[:, 1]) - self.X.entries(idx[:, 0], idx[:, 2]))
        else:
            Xc = self.X.detach()
            dev_idx = it.to(Xc.device)
            diff = (Xc[dev_idx[:, 0], dev_idx[:, 1]] - Xc[dev_idx[:, 0], dev_idx[:, 2]]).to("cpu")
        score = torch.sigmoid(self.scale * diff)                          # ref:509 (fp32, CPU op as there)
        draws = torch.bernoulli(score.repeat_interleave(K)).view(idx.shape[0], K)
        if self.soft_label and train:                                   # ref:510-513
            # torch.mean of K fp32 0/1 draws, then .item() -> Python float
            rows = np.empty((idx.shape[0], 4), dtype=np.float64)
            rows[:, :3] = idx
            rows[:, 3] = draws.mean(dim=1).double().numpy()
            return rows
        rows = np.empty((idx.shape[0] * K, 4), dtype=np.float64)         # ref:516-518: K rows per triplet
        rows[:, :3] = np.repeat(idx, K, axis=0)
        rows[:, 3] = draws.reshape(-1).double().numpy()
        return rows

    def _generate_labels(self, triplets, K, train=False):
        """ref:493-519 → list of (u, i, j, label)."""
        return self._tuples(self._label_rows(triplets, K, train))

    @staticmethod
    def _tuples(rows):
        ints = rows[:, :3].astype(np.int64)
        return list(zip(ints[:, 0].tolist(), ints[:, 1].tolist(), ints[:, 2].tolist(), rows[:, 3].tolist()))

    def _mfcd_records(self):
        """float64 [N, 4] rows for the device path, or None once `.data` has been handed out."""
        if self._rows is None and self._dev is not None and self._data is None:
            rec = self._dev.cpu().numpy()
            rows = np.empty((rec.shape[0], 4), dtype=np.float64)
            rows[:, :3] = rec[:, :3]
            rows[:, 3] = rec[:, 3].copy().view(np.float32)
            self._rows = rows
        return self._rows

    def _mfcd_device_records(self):
        """int32 [N, 4] device records when the labels were drawn on the device and `.data` was never handed out."""
        return self._dev if self._data is None else None

    @property
    def data(self):
        if self._data is None:
            self._data, self._rows = self._tuples(self._mfcd_records()), None
        return self._data

    @data.setter
    def data(self, value):
        self._data, self._rows = value, None

    def __len__(self):
        if self._data is not None:
            return len(self._data)
        return self._rows.shape[0] if self._rows is not None else self._dev.shape[0]

    def __getitem__(self, idx):
        if self._data is not None:
            return self._data[idx]
        r = self._mfcd_records()[idx]
        return (int(r[0]), int(r[1]), int(r[2]), float(r[3]))


_STRATEGIES = {
    "random": lambda X, k, ex, **kw: _gd.choose_items_random(X, num_triplets=k, exclude=ex),
    "proximity": lambda X, k, ex, **kw: _gd.choose_items_by_proximity(X, k, ex),
    "margin": lambda X, k, ex, **kw: _gd.choose_items_by_margin(X, k, ex),
    "variance": lambda X, k, ex, **kw: _gd.choose_items_by_variance(X, k, ex),
    "popularity": lambda X, k, ex, **kw: _gd.choose_items_by_popularity(
        X, k, ex, method=kw["popularity_method"], alpha=kw["alpha"]),
    "top_k": lambda X, k, ex, **kw: _gd.choose_items_top_k(X, k, ex),
    "cluster": lambda X, k, ex, **kw: _gd.choose_items_cluster_based(X, k, ex, n_clusters=kw["n_clusters"]),
    "user_similarity": lambda X, k, ex, **kw: _gd.choose_items_by_user_similarity(X, k, ex),
    "svd": lambda X, k, ex, **kw: _gd.choose_items_by_svd_projection(X, k, ex),
}


_SAMPLER_DEVICE = None


def set_sampler_device(device):
    """Extension (not in the reference): draw triplets ON `device` (a GPU; include/mfcd.h mfcd_sample_triplets) from now
    on for the strategies that have a device law (random, margin, popularity, variance, proximity, top_k, svd); the
    others (cluster, user_similarity — "Not used" in the reference's own comments) and None (default) use the host
    samplers, which consume torch's / numpy's generators exactly like the reference."""
    global _SAMPLER_DEVICE
    if device is not None:
        _need_gpu(device)
    _SAMPLER_DEVICE = None if device is None else torch.device(device)


def get_triplets_from_X(X, num_triplets, strategy="random", exclude=None, popularity_method="zipf", alpha=1.5,
                        n_clusters=10):
    """ref:533-588 → set of unique (u, i, j)."""
    if strategy not in _STRATEGIES:
        raise ValueError(f"Unknown triplet sampling strategy: {strategy}")
    if _SAMPLER_DEVICE is not None and strategy in _sampling.DEVICE_STRATEGIES:
        kw = dict(popularity_method=popularity_method, alpha=alpha) if strategy == "popularity" else {}
        rows = _sampling.sample_triplets(X, num_triplets, strategy, exclude, device=_SAMPLER_DEVICE, **kw).cpu().numpy()
        return set(zip(rows[:, 0].tolist(), rows[:, 1].tolist(), rows[:, 2].tolist()))
    found = _STRATEGIES[strategy](X, num_triplets, exclude or set(), popularity_method=popularity_method,
                                  alpha=alpha, n_clusters=n_clusters)
    return set(found)


_FACTOR_GENERATORS = ("structured", "svd", "correlated", "graph", "social", "temporal", "hierarchical", "gmm")


def generate_X(n, m, d, device, generation="base", **kwargs):
    """ref:590-663 → ground-truth preference matrix [n, m] fp32 on `device`."""
    if generation == "base":
        return _gd.generate_embeddings(n, m, d, device=device)
    if generation == "low_rank":
        A, B, S = _gd.generate_low_rank_matrix(n, m, d, rank=kwargs.get("rank", d), device=device)
        return (A * S) @ B.t()
    if generation == "clustered":
        return _gd.generate_clustered_matrix_from_embeddings(n, m, d, device=device)
    if generation in _FACTOR_GENERATORS:
        A, B = getattr(_gd, f"generate_{generation}_embeddings")(n, m, d, device=device)
        return A @ B.t()
    raise ValueError(f"Unknown generation method: {generation}")


def split_dataset_from_triplets(X, num_triplets, scale=1.0, K=1, train_ratio=0.8, val_ratio=0.1, batch_size=64,
                                strategy="random", popularity_method="zipf", alpha=1.5, soft_label=False):
    """ref:666-742 → (train_loader, val_loader, test_loader).

    With BOTH `set_sampler_device` and `set_label_device` in force (and a strategy that has a device law) the whole
    chain — triplets, the seed-42 80/10/10 split, the top-up of the test part to 500 rows, the labels and the 16-byte
    records — stays in HBM ("dataset materialisation on device"); `.data` of the datasets still yields the
    reference's list of tuples when somebody reads it."""
    if _SAMPLER_DEVICE is not None and _LABEL_DEVICE is not None and strategy in _sampling.DEVICE_STRATEGIES:
        kw = dict(popularity_method=popularity_method, alpha=alpha) if strategy == "popularity" else {}
        trip = _sampling.sample_triplets(X, num_triplets, strategy, None, device=_SAMPLER_DEVICE, **kw)
        total = trip.shape[0]
        if total < num_triplets:
            print(f"⚠️ Only {total} triplets generated for strategy: {strategy} (target={num_triplets})")
        n_train, n_val = int(train_ratio * total), int(val_ratio * total)
        order = torch.randperm(total, generator=torch.Generator().manual_seed(42)).to(trip.device)   # random_split's draw
        tr, va, te = trip[order[:n_train]], trip[order[n_train:n_train + n_val]], trip[order[n_train + n_val:]]
        if te.shape[0] * K < 500:                                                                # ref:721
            more = _sampling.sample_triplets(X, (500 + K - 1) // K - te.shape[0], strategy, trip,
                                             device=_SAMPLER_DEVICE, **kw)
            te = torch.cat((te, more))
        mk = lambda t, train: BTLPreferenceDataset(t, X, scale=scale, K=K, soft_label=soft_label, train=train)  # noqa: E731
        return (DataLoader(mk(tr, True), batch_size=batch_size, shuffle=True),
                DataLoader(mk(va, False), batch_size=batch_size, shuffle=False),
                DataLoader(mk(te, False), batch_size=batch_size, shuffle=False))
    triplets = list(get_triplets_from_X(X, num_triplets, strategy=strategy, popularity_method=popularity_method,
                                        alpha=alpha))
    if len(triplets) < num_triplets:
        print(f"⚠️ Only {len(triplets)} triplets generated for strategy: {strategy} (target={num_triplets})")
    total = len(triplets)
    n_train, n_val = int(train_ratio * total), int(val_ratio * total)
    parts = torch.utils.data.random_split(triplets, [n_train, n_val, total - n_train - n_val],
                                          generator=torch.Generator().manual_seed(42))      # ref:710-713
    tr, va, te = ([triplets[k] for k in part.indices] for part in parts)
    min_test = 500                                                                           # ref:721
    if len(te) * K < min_test:
        te = te + list(get_triplets_from_X(X, (min_test + K - 1) // K - len(te), strategy=strategy,
                                           popularity_method=popularity_method, alpha=alpha,
                                           exclude=set(tr + va + te)))
    mk = lambda t, train: BTLPreferenceDataset(t, X, scale=scale, K=K, soft_label=soft_label, train=train)  # noqa: E731
    return (DataLoader(mk(tr, True), batch_size=batch_size, shuffle=True),
            DataLoader(mk(va, False), batch_size=batch_size, shuffle=False),
            DataLoader(mk(te, False), batch_size=batch_size, shuffle=False))


# ------------------------------------------------------------------------------------------------
# experiment drivers (ref:81-450, 1154-1269): API / .pkl contract only
# ------------------------------------------------------------------------------------------------
_RESULT_KEYS = ("reconstruction_errors", "log_likelihoods", "accuracy", "gt_log_likelihoods", "gt_accuracy",
                "train_losses", "val_losses", "alpha", "norm_X", "norm_ratio", "reconstruction_error_scaled",
                "pearson_corr", "pearson_std", "spearman_corr", "spearman_std", "svd_error_scaled", "slopes",
                "pearson_corr_matrix", "spearman_corr_matrix", "reconstruction_error_scaled_per_row",
                "alpha_per_row", "sampled_UVT_rows", "sampled_X_rows")


def run_experiment(n, m, d, p, s, device, lr, weight_decay, reps=5, num_epochs=100, open_browser=False, K=1,
                   d1=None, strategy="random", popularity_method="zipf", alpha=1.5, soft_label=False,
                   generation="base"):
    """ref:306-450 → dict with the 23 keys of ref:420-444, one list entry per repetition."""
    res = {k: [] for k in _RESULT_KEYS}
    for rep in range(reps):
        X = generate_X(n, m, d, device, generation=generation)
        loaders = split_dataset_from_triplets(X, int(n * m * p / 2), scale=s, K=K, strategy=strategy,
                                              popularity_method=popularity_method, alpha=alpha,
                                              soft_label=soft_label)
        train_loader, val_loader, test_loader = loaders
        model = MatrixFactorization(n, m, d).to(device)
        optimizer = torch.optim.Adam(model.parameters(), lr=lr, weight_decay=weight_decay)
        t_losses, v_losses = train_model(model, train_loader, val_loader, optimizer, device, num_epochs=num_epochs,
                                         is_last=(rep == reps - 1), open_browser=open_browser)
        test_loss, test_acc = evaluate_model(model, test_loader, device)
        rec_error = compute_reconstruction_error(model, X, s)
        m14 = compute_alpha_and_norm_ratios(model, X)
        rows = torch.randperm(X.shape[0])[:2]                                     # ref:390 (global generator)
        gt_loss, gt_acc = compute_ground_truth_metrics(test_loader, X, device)
        for key, val in zip(("alpha", "norm_X", "norm_ratio", "reconstruction_error_scaled", "pearson_corr",
                             "pearson_std", "spearman_corr", "spearman_std", "svd_error_scaled", "slopes",
                             "pearson_corr_matrix", "spearman_corr_matrix", "reconstruction_error_scaled_per_row",
                             "alpha_per_row"), m14):
            res[key].append(val)
        res["train_losses"].append(t_losses)
        res["val_losses"].append(v_losses)
        res["accuracy"].append(test_acc)
        res["log_likelihoods"].append(-test_loss)
        res["reconstruction_errors"].append(rec_error)
        res["gt_log_likelihoods"].append(-gt_loss)
        res["gt_accuracy"].append(gt_acc)
        res["sampled_X_rows"].append(X[rows.to(X.device)].cpu().numpy())
        res["sampled_UVT_rows"].append(_metrics.uvt_rows(model.U.data, model.V.data, rows).cpu().numpy())
    return res


def _to_python(v):
    if isinstance(v, (np.float32, np.float64)):
        return float(v)
    if isinstance(v, np.integer):
        return int(v)
    return v


def _append_pickle(path, new_items):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    old = []
    if os.path.exists(path):
        with open(path, "rb") as f:
            old = pickle.load(f)
    old.extend(new_items)
    with open(path, "wb") as f:
        pickle.dump(old, f)
    print(f"✅ Saved {len(new_items)} new experiments to {path}")


_SCAN_KEYS = ("n", "m", "d", "p", "lr", "weight_decay", "num_epochs", "reps", "s", "K", "d1", "strategy",
              "popularity_method", "alpha", "soft_label", "generation")


def parameter_scan(n=1000, m=1000, d=2, p=0.5, s=1.0, device='cpu', lr=1e-3, weight_decay=1e-5, num_epochs=30,
                   reps=1, strategy="random", open_browser=False, linear=False, K=1, d1=None, save_path=None,
                   save_every=None, popularity_method="zipf", alpha=1.5, soft_label=False, generation="base"):
    """ref:81-255.  Scalar-or-list hyper-parameters → Cartesian product (default) or synchronised
    linear scan; each experiment yields {'params': ..., 'results': run_experiment(...)}.  With
    `save_path` the list is pickled (appending every `save_every` experiments) and, like the
    reference, the function then returns an empty list (ref:200-202)."""
    given = dict(n=n, m=m, d=d, p=p, lr=lr, weight_decay=weight_decay, num_epochs=num_epochs, reps=reps, s=s, K=K,
                 d1=d1, strategy=strategy, popularity_method=popularity_method, alpha=alpha, soft_label=soft_label,
                 generation=generation)
    grid, lists, synchronised = _normalise_grid({k: given[k] for k in _SCAN_KEYS})
    rank, world = _scan_ranks()
    if save_path and os.path.exists(save_path) and rank == 0:
        print(f"🧹 Removing existing file at {save_path}")
        os.remove(save_path)
    if not linear:
        configs = [dict(zip(grid.keys(), combo)) for combo in itertools.product(*grid.values())]
    elif synchronised:
        configs = [{k: (v[t] if len(v) > 1 else v[0]) for k, v in grid.items()} for t in range(len(lists[0]))]
    else:
        raise ValueError("The linear scan is not possible because the parameters are not synchronized.")
    def run_one(cfg, dev):
        print(f"\nRunning experiment with parameters: {cfg}")
        return run_experiment(n=cfg["n"], m=cfg["m"], d=cfg["d"], p=cfg["p"], s=cfg["s"], device=dev,
                              lr=cfg["lr"], weight_decay=cfg["weight_decay"], reps=cfg["reps"],
                              num_epochs=cfg["num_epochs"], open_browser=open_browser, K=cfg["K"], d1=cfg["d1"],
                              strategy=cfg["strategy"], popularity_method=cfg["popularity_method"],
                              alpha=cfg["alpha"], soft_label=cfg["soft_label"], generation=cfg["generation"])

    if world > 1:
        return scan_over_ranks(configs, run_one, rank, world, device, save_path, save_every)
    pending = []
    for cfg in configs:
        pending.append({"params": cfg, "results": run_one(cfg, device)})
        if save_path and save_every and len(pending) >= save_every:
            _append_pickle(save_path, pending)
            pending = []
    if save_path and pending:
        _append_pickle(save_path, pending)
        pending = []
    return pending


def _scan_ranks():
    """(rank, world) when the caller runs one process per GPU under torch.distributed, else (0, 1)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def scan_over_ranks(configs, run_one, rank, world, device, save_path=None, save_every=None, group=None):
    """the experiments of a scan are independent, so under `torchrun --nproc-per-node R` (one process
    per GPU, process group initialised by the caller) rank r runs experiments r, r+R, r+2R, ... on ITS GPU and rank 0
    collects them (one `gather_object` of plain Python results at the end; no data-path collective).  Rank 0 returns /
    pickles the list in the order of `configs`, exactly as the serial scan would; the other ranks return [].
    Not reproduced: the serial scan lets the global RNG state run on from one experiment into the next, so an experiment's
    random data here equals the serial run's only for rank 0's first experiment (seed per experiment for replay)."""
    import torch.distributed as dist
    dev = device
    if isinstance(device, str) and device == "cuda":
        dev = f"cuda:{int(os.environ.get('LOCAL_RANK', rank)) % max(torch.cuda.device_count(), 1)}"
    mine = [(k, {"params": configs[k], "results": run_one(configs[k], dev)}) for k in range(rank, len(configs), world)]
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(mine, gathered, dst=0, group=group)
    if rank != 0:
        return []
    done = [entry for _, entry in sorted((kv for part in gathered for kv in part), key=lambda kv: kv[0])]
    if not save_path:
        return done
    chunk = save_every if save_every else len(done)
    for a in range(0, len(done), max(chunk, 1)):
        _append_pickle(save_path, done[a:a + chunk])
    return []


def print_return_structure_types(obj, prefix="root"):
    """ref:258-302: debugging aid that prints the type tree of a nested result object."""
    if isinstance(obj, dict):
        for key, val in obj.items():
            print_return_structure_types(val, f"{prefix}.{key}")
    elif isinstance(obj, (list, tuple)):
        kinds = {type(e).__name__ for e in obj}
        inner = "empty" if not obj else (kinds.pop() if len(kinds) == 1 else "mixed")
        print(f"{prefix}: {type(obj).__name__}[{inner}]")
    elif isinstance(obj, torch.Tensor):
        print(f"{prefix}: torch.Tensor")
    else:
        print(f"{prefix}: {type(obj).__name__}")


def _normalise_grid(given):
    """Scalar-or-list kwargs → ({name: list}, lists_that_were_lists); NumPy scalars become Python ones (ref:128-148)."""
    grid = {}
    for key, v in given.items():
        if isinstance(v, np.ndarray):
            v = list(v)
        elif isinstance(v, list):
            v = [_to_python(x) for x in v]
        else:
            v = _to_python(v)
        grid[key] = v
    lists = [v for v in grid.values() if isinstance(v, list)]
    synchronised = len(lists) <= 1 or all(len(v) == len(lists[0]) for v in lists)
    grid = {k: (v if isinstance(v, (list, tuple)) else [v]) for k, v in grid.items()}
    return grid, lists, synchronised


def evaluate_ground_truth(n, m, p, d, s, device, K, reps=1, strategy="random", popularity_method="zipf", alpha=1.5,
                          soft_label=False, generation="base"):
    """ref:1154-1200 → (losses, accuracies) of the ground-truth matrix itself, one entry per repetition."""
    losses, accuracies = [], []
    for _ in range(reps):
        X = generate_X(n, m, d, device, generation=generation)
        _, _, test_loader = split_dataset_from_triplets(X, int(n * m * p / 2), scale=s, K=K, strategy=strategy,
                                                        popularity_method=popularity_method, alpha=alpha,
                                                        soft_label=soft_label)
        gt_loss, gt_acc = compute_ground_truth_metrics(test_loader, X, device)
        losses.append(gt_loss)
        accuracies.append(gt_acc)
    return losses, accuracies


def parameter_scan_ground_truth(n, m, p, d, s, device, K, linear=False, reps=1, strategy="random",
                                popularity_method="zipf", alpha=1.5, soft_label=False, generation="base"):
    """ref:1203-1269 → [{'params': ..., 'results': {'gt_loss': [...], 'gt_accuracy': [...]}}, ...].
    A linear scan over unsynchronised lists silently becomes a Cartesian scan, as in the reference."""
    grid, lists, synchronised = _normalise_grid(dict(n=n, m=m, p=p, d=d, s=s, K=K, strategy=strategy,
                                                     popularity_method=popularity_method, alpha=alpha,
                                                     soft_label=soft_label, generation=generation))
    if linear and synchronised:
        configs = [{k: (v[t] if len(v) > 1 else v[0]) for k, v in grid.items()} for t in range(len(lists[0]))]
    else:
        configs = [dict(zip(grid.keys(), combo)) for combo in itertools.product(*grid.values())]
    out = []
    for cfg in (_tqdm(configs, desc="Training Progress") if _tqdm is not None else configs):
        gt_loss, gt_accuracy = evaluate_ground_truth(**cfg, device=device, reps=reps)
        out.append({"params": cfg, "results": {"gt_loss": gt_loss, "gt_accuracy": gt_accuracy}})
    return out
