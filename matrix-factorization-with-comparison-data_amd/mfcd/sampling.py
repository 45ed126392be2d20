"""Triplet sampling on the device (include/mfcd.h: mfcd_sample_triplets).

Host side of csrc/sampler.hip: turns a strategy name of the reference (`get_triplets_from_X`, structure.py:533-588)
into a `mfcd_sampler` law, runs blocks of attempts until the request is met or the strategy's attempt budget is spent
(the budgets of the reference: margin 5 000 000 in blocks of 500, top_k 3x, svd 5x the request; the others unbounded),
and returns the triplets in attempt order as a device tensor.  Opt-in (`structure.set_sampler_device`): the default
host samplers consume torch's / numpy's generators draw for draw like the reference; this path has its own Philox
stream (one int64 seed taken from torch's global generator per request), so it is reproducible under
`torch.manual_seed` but distributionally — not bitwise — equal to a reference run.
"""
import ctypes

import numpy as np
import torch

from . import _lib

LAW_UNIFORM, LAW_ITEM_CDF, LAW_LISTS = 0, 1, 2
DEVICE_STRATEGIES = ("random", "margin", "popularity", "variance", "proximity", "top_k", "svd")


def _dense_on(X, device):
    return X.detach().to(device=device, dtype=torch.float32).contiguous()


class _Law:
    """A filled mfcd_sampler plus the tensors its pointers refer to (kept alive with it)."""

    def __init__(self, n, m, device):
        self.n, self.m, self.device = int(n), int(m), device
        self.c = _lib.Sampler()
        self.c.law, self.c.n, self.c.m = LAW_UNIFORM, int(n), int(m)
        self.keep = []
        self.budget = None           # attempt budget of the strategy (None: until the request is met)
        self.block_multiple = 1      # the margin strategy counts attempts in blocks of 500

    def hold(self, t):
        self.keep.append(t)
        return _lib.ptr(t)


def _is_factored(X):
    return hasattr(X, "pair_diff") and hasattr(X, "A")


def build_law(X, num_triplets, strategy, device, popularity_method="zipf", alpha=1.5, k=None, max_attempts=5_000_000):
    """The reference's per-strategy set-up (everything in front of its attempt loop) → a device law."""
    import generation_data as _gd
    n, m = X.shape
    law = _Law(n, m, device)
    c = law.c
    if strategy == "random":
        return law
    if strategy == "margin":                                             # generation_data.py:56-57
        head = X.rows(0, min(10, n)) if _is_factored(X) else X[:min(10, n)].detach().cpu().numpy()
        c.use_margin = 1
        c.margin = float(np.mean(head.max(axis=1) - head.min(axis=1)) * num_triplets / (n * m))
        if _is_factored(X):
            A, B = X.A.to(device).contiguous(), X.B.to(device).contiguous()
            c.A, c.B, c.dx = law.hold(A), law.hold(B), A.shape[1]
        else:
            c.X = law.hold(_dense_on(X, device))
        law.budget, law.block_multiple = int(max_attempts), 500
        law.margin = c.margin
        return law
    if strategy in ("popularity", "variance"):
        if strategy == "popularity":                                    # generation_data.py:110-119
            probs = _gd._popularity_probs(m, popularity_method, alpha)
            c.pair_rule = 0
        else:                                                            # generation_data.py:90-91
            var = torch.var(_dense_on(X, device), dim=0).double().cpu().numpy()
            probs = var / var.sum()
            c.pair_rule = 1
            if not np.isfinite(probs).all() or (probs < 0).any():      # e.g. one user: the unbiased variance is NaN
                raise RuntimeError("probability tensor contains either `inf`, `nan` or element < 0")   # as torch.multinomial (ref:95)
        cdf = np.cumsum(probs)
        cdf /= cdf[-1]
        c.law = LAW_ITEM_CDF
        c.cdf = law.hold(torch.from_numpy(cdf).to(device))
        return law
    if strategy in ("proximity", "top_k"):
        Xd = _dense_on(X, device)
        if strategy == "proximity":                                     # generation_data.py:36-37
            kk = min(100 if k is None else int(k), m)
            best = torch.topk(Xd, k=kk, dim=1)[1].to(torch.int32).contiguous()
            worst = torch.topk(-Xd, k=kk, dim=1)[1].to(torch.int32).contiguous()
            c.list_i, c.list_j, c.pair_rule = law.hold(best), law.hold(worst), 0
        else:                                                            # generation_data.py:198-213
            kk = min(m, max(5, int(0.1 * m))) if k is None else int(k)
            best = torch.topk(Xd, k=kk, dim=1)[1].to(torch.int32).contiguous()
            c.list_i = c.list_j = law.hold(best)
            c.pair_rule = 1
            law.budget = 3 * int(num_triplets)
            law.k = kk
        c.law, c.k, c.list_row_stride = LAW_LISTS, kk, kk
        return law
    if strategy == "svd":                                                # generation_data.py:144-162
        import scipy.sparse.linalg as spla
        rank = int(num_triplets / (n * m) * max(n, m))
        Us, S, Vt = spla.svds(X.detach().cpu().numpy(), k=rank)
        top_users = np.argsort(np.linalg.norm(Us * S, axis=1))[-max(1, int(0.3 * n)):]
        top_items = np.argsort(np.linalg.norm(Vt.T * S, axis=1))[-max(2, int(0.3 * m)):]
        items = torch.from_numpy(top_items.astype(np.int32)).to(device)
        c.law, c.k, c.list_row_stride, c.pair_rule = LAW_LISTS, int(items.numel()), 0, 1
        c.list_i = c.list_j = law.hold(items)
        c.users, c.n_users = law.hold(torch.from_numpy(top_users.astype(np.int32)).to(device)), int(top_users.size)
        law.budget = 5 * int(num_triplets)
        return law
    raise ValueError(f"no device law for triplet sampling strategy: {strategy}")


def triplet_keys(rows, m):
    rows = np.asarray(rows, dtype=np.int64).reshape(-1, 3)
    return (rows[:, 0] * m + rows[:, 1]) * m + rows[:, 2]


def run_law(law, num_triplets, exclude=None, seed=0):
    """Blocks of attempts until `num_triplets` are kept or the budget is spent → (int32 [T, 3] device tensor in attempt
    order, attempts consumed)."""
    L = _lib.load()
    device, m = law.device, law.m
    want = int(num_triplets)
    out = torch.empty((max(want, 0), 3), dtype=torch.int32, device=device)
    if want <= 0:
        return out, 0
    if isinstance(exclude, torch.Tensor):          # int [E, 3] device triplets (what an earlier request returned)
        e = exclude.to(device=device, dtype=torch.int64).reshape(-1, 3)
        barred = ((e[:, 0] * m + e[:, 1]) * m + e[:, 2]).contiguous()
    else:
        barred = torch.from_numpy(triplet_keys(sorted(exclude), m)).to(device) if exclude else \
            torch.empty(0, dtype=torch.int64, device=device)
    have = attempts = idle = 0
    stream = _lib.stream_ptr(device)
    while have < want and (law.budget is None or attempts < law.budget):
        need = want - have
        A = max(65536, need + need // 4)
        if law.budget is not None:
            A = min(A, law.budget - attempts)
        A = -(-A // law.block_multiple) * law.block_multiple
        ws_bytes = L.mfcd_sample_workspace_bytes(A, barred.numel())
        if ws_bytes == 0:
            raise _lib.MfcdError("triplet request too large for one sampling block")
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
        keys = torch.empty(need, dtype=torch.int64, device=device)
        counts = torch.zeros(2, dtype=torch.int64, device=device)
        part = out[have:]
        _lib.check(L.mfcd_sample_triplets(ctypes.byref(law.c), _lib.ptr(barred) if barred.numel() else None,
                                          barred.numel(), attempts, A, int(seed) & 0xFFFFFFFFFFFFFFFF, need,
                                          _lib.ptr(part), _lib.ptr(keys), _lib.ptr(counts), _lib.ptr(ws), ws_bytes,
                                          stream))
        got, used = (int(v) for v in counts.tolist())
        used = -(-used // law.block_multiple) * law.block_multiple
        attempts += used
        have += got
        if got:
            barred = torch.cat((barred, keys[:got]))
        idle = 0 if got else idle + 1
        if idle >= 16 and law.budget is None:      # the reference's loop would spin forever: nothing left to draw
            raise ValueError(f"cannot draw {want} distinct triplets with this strategy: {have} found, none in the "
                             f"last {idle} blocks of attempts")
    return out[:have], attempts


def sample_triplets(X, num_triplets, strategy="random", exclude=None, device=None, seed=None, **kw):
    """Device form of `get_triplets_from_X` → int32 [T, 3] device tensor (T <= num_triplets), attempt order."""
    if device is None:
        device = X.device if torch.is_tensor(X) and X.is_cuda else torch.device("cuda")
    device = torch.device(device)
    if device.type != "cuda":
        raise _lib.MfcdError("device triplet sampling needs a GPU device")
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    if seed is None:
        seed = int(torch.empty((), dtype=torch.int64).random_().item())
    law = build_law(X, int(num_triplets), strategy, device, **kw)
    trip, attempts = run_law(law, num_triplets, exclude, seed)
    if trip.shape[0] < num_triplets:                                     # the reference's own messages
        if strategy == "margin":
            top = float(X.A.max()) if _is_factored(X) else float(X.max())
            print(f"⚠️ Only {trip.shape[0]} triplets generated (target={num_triplets}, margin={law.margin:.4f}) "
                  f"after {attempts} attempts.maximum : {top}")
        elif strategy == "top_k":
            print(f"⚠️ Only {trip.shape[0]} triplets generated (target={num_triplets}, k={law.k})")
        elif strategy == "svd":
            print(f"⚠️ Only {trip.shape[0]} triplets generated (target={num_triplets})")
    return trip
