"""CPU oracle for the triplet-MF hot path — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package never does; it raises when its HIP library is missing.

Two parts:
  * `COracle` — ctypes binding of oracle/mfcd_oracle.c (training step, eval pass, UV^T sums);
  * numpy restatements of the reference's metric functions, vectorised (the reference loops
    over rows in Python), each citing the reference lines it follows.

Parity status: PINNED against tests/golden/ (vectors produced by the unmodified reference via
oracle/make_golden.py); see tests/test_oracle_golden.py.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libmfcd_oracle.so")

_f32p = ctypes.POINTER(ctypes.c_float)
_f64p = ctypes.POINTER(ctypes.c_double)
_i64p = ctypes.POINTER(ctypes.c_int64)
_i32p = ctypes.POINTER(ctypes.c_int32)


def build(force=False, archflags="", out=None):
    """Compile oracle/mfcd_oracle.c with gcc (building the checker is not using it)."""
    out = out or _LIB
    src = os.path.join(_HERE, "mfcd_oracle.c")
    if force or not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        cmd = ["make", "-C", _HERE, "-B", f"OUT={out}"]
        if archflags:
            cmd.append(f"ARCHFLAGS={archflags}")
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL)
    return out


def _p(a, typ):
    return a.ctypes.data_as(typ) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


class COracle:
    def __init__(self, path=None):
        path = path or build()
        L = ctypes.CDLL(path)
        L.mfcd_orc_forward.argtypes = [_f32p, _f32p, _i64p, _i64p, _i64p, _f32p, ctypes.c_int, ctypes.c_int, _f32p, _f32p]
        L.mfcd_orc_grad.argtypes = [_f32p, _f32p, _i64p, _i64p, _i64p, _f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                    ctypes.c_int, _f32p, _f32p, _f32p, _f32p, _f32p]
        L.mfcd_orc_adam.argtypes = [_f32p, _f32p, _f32p, _f32p, ctypes.c_int64] + [ctypes.c_double] * 6 + [ctypes.c_int]
        L.mfcd_orc_train_steps.argtypes = [_f32p] * 6 + [_i64p, _i64p, _i64p, _f32p, ctypes.c_int64, ctypes.c_int,
                                                        ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int] + \
            [ctypes.c_double] * 5 + [_f32p, ctypes.c_int]
        L.mfcd_orc_train_steps.restype = ctypes.c_int
        L.mfcd_orc_train_steps_bf16.argtypes = L.mfcd_orc_train_steps.argtypes
        L.mfcd_orc_train_steps_bf16.restype = ctypes.c_int
        L.mfcd_orc_round_bf16.argtypes = [_f32p, ctypes.c_int64]
        L.mfcd_orc_eval_batches.argtypes = [_f32p, _f32p, _i64p, _i64p, _i64p, _f32p, ctypes.c_int64, ctypes.c_int,
                                            ctypes.c_int, _f32p, _i32p, _f32p]
        L.mfcd_orc_eval_batches.restype = ctypes.c_int
        L.mfcd_orc_uvt_stats.argtypes = [_f32p, _f32p, _f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_double, _f64p, _f64p]
        L.mfcd_orc_max_threads.restype = ctypes.c_int
        self.L = L

    def max_threads(self):
        return self.L.mfcd_orc_max_threads()

    def forward(self, U, V, u, i, j, z):
        U, V, u, i, j, z = _f32(U), _f32(V), _i64(u), _i64(i), _i64(j), _f32(z)
        B, d = len(u), U.shape[1]
        p, term = np.empty(B, np.float32), np.empty(B, np.float32)
        self.L.mfcd_orc_forward(_p(U, _f32p), _p(V, _f32p), _p(u, _i64p), _p(i, _i64p), _p(j, _i64p), _p(z, _f32p),
                                B, d, _p(p, _f32p), _p(term, _f32p))
        return p, term

    def grad(self, U, V, u, i, j, z):
        """→ (p, loss, dU, dV) of one batch, dense gradients as autograd builds them."""
        U, V, u, i, j, z = _f32(U), _f32(V), _i64(u), _i64(i), _i64(j), _f32(z)
        (n, d), m, B = U.shape, V.shape[0], len(u)
        dU, dV, sc = np.empty_like(U), np.empty_like(V), np.empty_like(V)
        p, loss = np.empty(B, np.float32), np.zeros(1, np.float32)
        self.L.mfcd_orc_grad(_p(U, _f32p), _p(V, _f32p), _p(u, _i64p), _p(i, _i64p), _p(j, _i64p), _p(z, _f32p),
                             B, n, m, d, _p(dU, _f32p), _p(dV, _f32p), _p(sc, _f32p), _p(p, _f32p), _p(loss, _f32p))
        return p, float(loss[0]), dU, dV

    def adam(self, p, m, v, g, step, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, wd=0.0, threads=1):
        """In place on p, m, v (contiguous fp32)."""
        for a in (p, m, v):
            assert a.dtype == np.float32 and a.flags.c_contiguous
        g = _f32(g) if g is not None else None
        self.L.mfcd_orc_adam(_p(p, _f32p), _p(m, _f32p), _p(v, _f32p), _p(g, _f32p), p.size, lr, betas[0], betas[1],
                             eps, wd, float(step), threads)

    def round_bf16(self, a):
        """In place: nearest-even rounding of a contiguous fp32 array to bf16-representable values."""
        assert a.dtype == np.float32 and a.flags.c_contiguous
        self.L.mfcd_orc_round_bf16(_p(a, _f32p), a.size)
        return a

    def train_steps(self, state, u, i, j, z, B, step0, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, wd=0.0, threads=1,
                    bf16_factors=False):
        """state = dict(U,V,mU,vU,mV,vV) of contiguous fp32 arrays, updated in place.
        → fp32 array of per-step batch-mean losses.  bf16_factors: U, V are rounded to bf16 after every step."""
        u, i, j, z = _i64(u), _i64(i), _i64(j), _f32(z)
        N = len(u)
        n, d = state["U"].shape
        m = state["V"].shape[0]
        nsteps = (N + B - 1) // B
        losses = np.empty(nsteps, np.float32)
        for k in ("U", "V", "mU", "vU", "mV", "vV"):
            assert state[k].dtype == np.float32 and state[k].flags.c_contiguous
        fn = self.L.mfcd_orc_train_steps_bf16 if bf16_factors else self.L.mfcd_orc_train_steps
        got = fn(
            _p(state["U"], _f32p), _p(state["V"], _f32p), _p(state["mU"], _f32p), _p(state["vU"], _f32p),
            _p(state["mV"], _f32p), _p(state["vV"], _f32p), _p(u, _i64p), _p(i, _i64p), _p(j, _i64p), _p(z, _f32p),
            N, B, step0, n, m, d, lr, betas[0], betas[1], eps, wd, _p(losses, _f32p), threads)
        assert got == nsteps
        return losses

    def eval_batches(self, U, V, u, i, j, z, B):
        """→ (per-batch mean BCE fp32[nb], per-batch correct int32[nb], p fp32[N])."""
        U, V, u, i, j, z = _f32(U), _f32(V), _i64(u), _i64(i), _i64(j), _f32(z)
        N, d = len(u), U.shape[1]
        nb = (N + B - 1) // B
        loss, corr, p = np.empty(nb, np.float32), np.empty(nb, np.int32), np.empty(N, np.float32)
        self.L.mfcd_orc_eval_batches(_p(U, _f32p), _p(V, _f32p), _p(u, _i64p), _p(i, _i64p), _p(j, _i64p),
                                     _p(z, _f32p), N, B, d, _p(loss, _f32p), _p(corr, _i32p), _p(p, _f32p))
        return loss, corr, p

    def uvt_stats(self, U, V, X, s=1.0):
        """→ (row_stats f64[n,3] = (xu,uu,xx) of row-centred UV^T vs row-centred X, err2, ref2)."""
        U, V, X = _f32(U), _f32(V), _f32(X)
        n, d = U.shape
        m = V.shape[0]
        rs, o2 = np.empty((n, 3), np.float64), np.empty(2, np.float64)
        self.L.mfcd_orc_uvt_stats(_p(U, _f32p), _p(V, _f32p), _p(X, _f32p), n, m, d, s, _p(rs, _f64p), _p(o2, _f64p))
        return rs, float(o2[0]), float(o2[1])


def new_state(U0, V0):
    """Fresh Adam state around copies of U0, V0 (exp_avg = exp_avg_sq = 0, as torch initialises them)."""
    U, V = _f32(U0).copy(), _f32(V0).copy()
    return {"U": U, "V": V, "mU": np.zeros_like(U), "vU": np.zeros_like(U),
            "mV": np.zeros_like(V), "vV": np.zeros_like(V)}


def epoch_losses(batch_losses):
    """structure.py:852-855 / 865-868: Python-float (f64) sum of the per-batch fp32 means / #batches."""
    return float(np.sum(np.asarray(batch_losses, dtype=np.float64))) / len(batch_losses)


# ----------------------------------------------------------------------------------------------
# numpy restatements of the dense metric functions
# ----------------------------------------------------------------------------------------------
def reconstruction_error(U, V, X, s):
    """structure.py:925-955: ||(UV^T - colmean) - sX||_F / ||sX||_F, fp32 tensors."""
    M = _f32(U) @ _f32(V).T                                   # 940
    M = M - M.mean(axis=0, keepdims=True, dtype=np.float32)   # 943
    sX = np.float32(s) * _f32(X)
    num = np.sqrt(np.sum((M - sX).astype(np.float64) ** 2))   # 949
    den = np.sqrt(np.sum(sX.astype(np.float64) ** 2))         # 946
    return float(num / den)                                   # 952


def _rankdata_rows(a):
    """Average ranks per row (what scipy.stats.spearmanr uses), vectorised over rows."""
    from scipy.stats import rankdata
    return rankdata(a, axis=1)


def spearman_rows(A, X):
    """structure.py:1023-1031 per row: scipy.stats.spearmanr(a, x).correlation = Pearson correlation of the average
    ranks, float64.  NaN for a constant row (scipy warns and returns nan)."""
    ra, rx = _rankdata_rows(np.asarray(A)), _rankdata_rows(np.asarray(X))
    ra = ra - ra.mean(1, keepdims=True)
    rx = rx - rx.mean(1, keepdims=True)
    with np.errstate(invalid="ignore", divide="ignore"):
        return np.sum(ra * rx, 1) / np.sqrt(np.sum(ra * ra, 1) * np.sum(rx * rx, 1))


def alpha_and_norm_ratios(U, V, X_init):
    """structure.py:958-1082 → the same 14-tuple (lists hold Python/NumPy floats)."""
    UVT = _f32(U) @ _f32(V).T                                            # 982
    UVT = UVT - UVT.mean(axis=1, keepdims=True, dtype=np.float32)        # 985
    X = _f32(X_init).copy()
    X = X - X.mean(axis=1, keepdims=True, dtype=np.float32)              # 987
    A, C = UVT.astype(np.float64), X.astype(np.float64)
    dot = float(np.sum(A * C))                                           # 990
    norm_UVT = float(np.sqrt(np.sum(A * A)))                             # 991
    norm_X = float(np.sqrt(np.sum(C * C)))                               # 992
    alpha = dot / (norm_UVT ** 2 + 1e-8)                                 # 994
    norm_ratio = norm_UVT / (norm_X + 1e-8)                              # 995
    rec_scaled = float(np.sqrt(np.sum((alpha * A - C) ** 2))) / (norm_X + 1e-8)  # 996
    # rows with both std > 1e-8 (1006, 1027)
    ok = (X.std(axis=1) > 1e-8) & (UVT.std(axis=1) > 1e-8)
    xu, uu, xx = np.sum(A * C, 1), np.sum(A * A, 1), np.sum(C * C, 1)
    # np.corrcoef subtracts the (f64) row means again; rows are already centred up to fp32 rounding
    Ac = A - A.mean(1, keepdims=True)
    Cc = C - C.mean(1, keepdims=True)
    corr_all = np.sum(Ac * Cc, 1) / np.sqrt(np.sum(Ac * Ac, 1) * np.sum(Cc * Cc, 1))
    correlations = [float(c) for c in corr_all[ok]]                      # 1003-1008
    pearson_mean = float(np.mean(correlations)) if correlations else 0.0  # 1009
    s1 = np.linalg.svd(C, compute_uv=False)                              # 1013
    s2 = np.linalg.svd(A, compute_uv=False)                              # 1014
    k = min(len(s1), len(s2))
    svd_err = float(np.linalg.norm(alpha * s2[:k] - s1[:k]) / (np.linalg.norm(s1[:k]) + 1e-8))  # 1016-1017
    rx, ru = _rankdata_rows(X), _rankdata_rows(UVT)                      # 1028 (Spearman = Pearson of ranks)
    rx = rx - rx.mean(1, keepdims=True)
    ru = ru - ru.mean(1, keepdims=True)
    with np.errstate(invalid="ignore", divide="ignore"):
        rho = np.sum(rx * ru, 1) / np.sqrt(np.sum(rx * rx, 1) * np.sum(ru * ru, 1))
    spearman_scores = [float(r) for r, o in zip(rho, ok) if o and not np.isnan(r)]  # 1027-1030
    spearman_mean = float(np.mean(spearman_scores)) if spearman_scores else 0.0
    pearson_std = float(np.std(correlations)) if correlations else 0.0   # 1034
    spearman_std = float(np.std(spearman_scores)) if spearman_scores else 0.0
    # fp32 dot products of fp32 rows (np.dot on float32 rows), 1042-1045
    xx32 = np.einsum("ij,ij->i", X, X)
    xu32 = np.einsum("ij,ij->i", X, UVT)
    uu32 = np.einsum("ij,ij->i", UVT, UVT)
    sl_ok = (xx32 > 1e-8) & (UVT.std(axis=1) > 1e-8)
    slopes = [float(v) for v in (xu32[sl_ok] / xx32[sl_ok])]
    alpha_i = np.where(uu32 > 1e-8, xu32 / np.where(uu32 > 1e-8, uu32, 1), 0.0).astype(np.float32)  # 1057-1058
    adj = (alpha_i[:, None] * UVT).astype(np.float32)                    # 1060, 1063
    rec_rows = float(np.sqrt(np.sum((adj - X).astype(np.float64) ** 2))) / (norm_X + 1e-8)  # 1064
    del xu, uu, xx
    return (alpha, norm_X, norm_ratio, rec_scaled, pearson_mean, pearson_std, spearman_mean, spearman_std,
            svd_err, slopes, correlations, spearman_scores, rec_rows, [float(a) for a in alpha_i])


def ground_truth_metrics(test_data, X, B=64):
    """structure.py:1085-1127 on the (u,i,j,z) rows of the test set, batches of B in order."""
    td = np.asarray(test_data, dtype=np.float64).reshape(-1, 4)
    X = _f32(X)
    tot, correct, nb = 0.0, 0, 0
    for off in range(0, len(td), B):
        blk = td[off:off + B]
        u, i, j = blk[:, 0].astype(np.int64), blk[:, 1].astype(np.int64), blk[:, 2].astype(np.int64)
        z = blk[:, 3].astype(np.float32)
        diff = X[u, i] - X[u, j]                                     # 1108 (no scale factor)
        prob = (1.0 / (1.0 + np.exp(-diff, dtype=np.float32))).astype(np.float32)  # 1111
        tot += float(np.mean((prob - z) ** 2, dtype=np.float32))     # 1114-1115
        correct += int(np.sum((diff > 0).astype(np.float32) == z))   # 1118-1121
        nb += 1
    return tot / nb, (correct / len(td) if len(td) else 0.0)         # 1125-1127
