"""Dense reconstruction / alignment metrics on the device path.

The n x m product UV^T is never materialised for the Frobenius / Pearson / slope / alpha metrics:
libmfcd_hip.so's MFMA pass returns per-row sums and two global sums (include/mfcd.h,
mfcd_uvt_stats), and everything the reference derives from them is formed here in f64.

Spearman correlations (structure.py:1023-1031) run on the HIP rank kernel
(mfcd_spearman_rows: bitonic sort in LDS, exact integer rank sums) over rows of U V^T formed by a plain
library GEMM.  The singular-value error (structure.py:1011-1017) stays on torch's dense eigen-solver: U V^T
has rank <= d, so its spectrum comes from a d x d problem, and X's spectrum is cached per X.
"""
import weakref

import numpy as np
import torch

from . import _lib

_ws = {}


def _workspace(nbytes, device):
    buf = _ws.get(device)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        _ws[device] = buf
    return buf


def uvt_stats(U, V, X, s=1.0, what=3):
    """→ (row_stats f64 [n,8] on device, scal f64 [4] on device); layout in include/mfcd.h.
    what = 1: only the per-row sums are computed (scal is None), 2: only the global sums (row_stats is None), 3: both."""
    L = _lib.load()
    if U.dtype == torch.bfloat16:   # bf16 factors: exact widening for the one-off metric pass
        U, V = U.float(), V.float()
    for t, nm in ((U, "U"), (V, "V"), (X, "X")):
        if not t.is_cuda or t.dtype != torch.float32:
            raise _lib.MfcdError(f"{nm} must be a float32 GPU tensor (no CPU fallback)")
    U, V, X = U.contiguous(), V.contiguous(), X.contiguous()
    (n, d), m = U.shape, V.shape[0]
    if X.shape != (n, m):
        raise ValueError(f"X must be [{n},{m}], got {tuple(X.shape)}")
    row_stats = torch.empty((n, 8), dtype=torch.float64, device=U.device) if what & 1 else None
    scal = torch.empty(4, dtype=torch.float64, device=U.device) if what & 2 else None
    ws = _workspace(L.mfcd_uvt_workspace_bytes(n, m, d), U.device)
    _lib.check(L.mfcd_uvt_stats_select(_lib.ptr(U), _lib.ptr(V), _lib.ptr(X), n, m, d, float(s), int(what),
                                       _lib.ptr(row_stats), _lib.ptr(scal), _lib.ptr(ws), ws.numel(),
                                       _lib.stream_ptr(U.device)))
    return row_stats, scal


def uvt_stats_factored(U, V, FX, s=1.0, what=3, slab_rows=4096):
    """The UV^T pass against a ground truth kept as factors (generation_data.FactoredMatrix, X = A B^T): X is formed
    `slab_rows` rows at a time by a plain library GEMM and each slab goes through mfcd_uvt_stats_slab, so nothing of size
    n x m exists (C4: 16 GiB dense, 1 GiB per slab).  Returns what uvt_stats returns."""
    L = _lib.load()
    if U.dtype == torch.bfloat16:
        U, V = U.float(), V.float()
    U, V = U.contiguous(), V.contiguous()
    dev = U.device
    (n, d), m = U.shape, V.shape[0]
    if tuple(FX.shape) != (n, m):
        raise ValueError(f"X must be [{n},{m}], got {tuple(FX.shape)}")
    A, B = FX.A.to(dev), FX.B.to(dev)
    slab_rows = max(1, min(int(slab_rows), n))
    row_stats = torch.empty((n, 8), dtype=torch.float64, device=dev) if what & 1 else None
    scal = torch.zeros(4, dtype=torch.float64, device=dev) if what & 2 else None
    share = torch.empty(4, dtype=torch.float64, device=dev) if what & 2 else None
    ws = _workspace(L.mfcd_uvt_slab_workspace_bytes(n, m, d, slab_rows), dev)
    for r0 in range(0, n, slab_rows):
        r1 = min(n, r0 + slab_rows)
        Xs = (A[r0:r1] @ B.t()).contiguous()
        _lib.check(L.mfcd_uvt_stats_slab(_lib.ptr(U), _lib.ptr(V), _lib.ptr(Xs), n, m, d, float(s), int(what), r0,
                                         r1 - r0, _lib.ptr(row_stats[r0:r1]) if what & 1 else None, _lib.ptr(share),
                                         _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev)))
        if what & 2:
            scal += share                      # f64, slab order: deterministic
    return row_stats, scal


def _is_factored(X):
    return hasattr(X, "A") and hasattr(X, "B") and not isinstance(X, torch.Tensor)


def uvt_rows(U, V, row_ids):
    """Rows `row_ids` of UV^T as a [k, m] fp32 device tensor (structure.py:389-392 without the full GEMM)."""
    L = _lib.load()
    U, V = U.float().contiguous(), V.float().contiguous()
    host_ids = torch.as_tensor(row_ids).reshape(-1).cpu()
    if host_ids.numel() and (int(host_ids.min()) < -U.shape[0] or int(host_ids.max()) >= U.shape[0]):
        raise IndexError(f"row index out of range for U with {U.shape[0]} rows")     # as U[row] would
    host_ids = torch.where(host_ids < 0, host_ids + U.shape[0], host_ids)
    ids = host_ids.to(dtype=torch.int32, device=U.device).contiguous()
    k, m = ids.numel(), V.shape[0]
    out = torch.empty((k, m), dtype=torch.float32, device=U.device)
    _lib.check(L.mfcd_uvt_rows(_lib.ptr(U), _lib.ptr(V), _lib.ptr(ids), k, U.shape[0], m, U.shape[1], _lib.ptr(out),
                               _lib.stream_ptr(U.device)))
    return out


def reconstruction_error(U, V, X, s):
    """compute_reconstruction_error (structure.py:925-955) → float."""
    _, scal = uvt_stats_factored(U, V, X, s, what=2) if _is_factored(X) else uvt_stats(U, V, X, s, what=2)
    e2, r2 = scal[:2].cpu().tolist()
    return float(np.sqrt(e2) / np.sqrt(r2)) if r2 > 0 else float("nan") if e2 == 0 else float("inf")


def _rank_rows(M):
    """Average ranks along dim 1 (what scipy.stats.rankdata gives), on the GPU."""
    n, m = M.shape
    srt, idx = torch.sort(M, dim=1, stable=True)
    base = torch.arange(1, m + 1, device=M.device, dtype=torch.float64).expand(n, m)
    # ties: average the positions of equal runs
    new_run = torch.ones_like(srt, dtype=torch.bool)
    new_run[:, 1:] = srt[:, 1:] != srt[:, :-1]
    run_id = torch.cumsum(new_run, dim=1) - 1
    flat = run_id + (torch.arange(n, device=M.device) * m)[:, None]
    sums = torch.zeros(n * m, dtype=torch.float64, device=M.device).scatter_add_(0, flat.reshape(-1), base.reshape(-1))
    cnts = torch.zeros(n * m, dtype=torch.float64, device=M.device).scatter_add_(
        0, flat.reshape(-1), torch.ones(n * m, dtype=torch.float64, device=M.device))
    avg = (sums / cnts.clamp_min(1))[flat.reshape(-1)].reshape(n, m)
    ranks = torch.empty_like(avg)
    ranks.scatter_(1, idx, avg)
    return ranks


def spearman_rows_sorted(A, X):
    """Per-row Spearman rho by plain torch ops (device sort, run-averaged ranks, f64 sums; NaN in either row -> NaN).
    NOT on the product path since round 3 (rows of any length go through the HIP rank kernels): kept as an independent
    formulation the GPU tests hold the kernels against beside scipy."""
    ra, rx = _rank_rows(A), _rank_rows(X)
    ra = ra - ra.mean(1, keepdim=True)
    rx = rx - rx.mean(1, keepdim=True)
    rho = (ra * rx).sum(1) / torch.sqrt((ra * ra).sum(1) * (rx * rx).sum(1))
    bad = torch.isnan(A).any(1) | torch.isnan(X).any(1)
    return torch.where(bad, torch.full_like(rho, float("nan")), rho)


def spearman_rows_any(A, X):
    """Per-row Spearman rho for any row length: the LDS rank kernel up to its column limit, the global-memory form of the
    same kernel (mfcd_spearman_rows_long: segmented device sort of row blocks) above it."""
    if A.shape[1] <= _lib.load().mfcd_spearman_max_columns():
        return spearman_rows(A, X)
    return spearman_rows_long(A, X)


def spearman_rows_long(A, X):
    """include/mfcd.h mfcd_spearman_rows_long: rows of any length (BASELINE configs[3]: 65536 items), f64 [rows]."""
    L = _lib.load()
    rows, m = A.shape
    if X.shape != A.shape or A.dtype != torch.float32 or X.dtype != torch.float32 or not A.is_cuda or not X.is_cuda:
        raise _lib.MfcdError("spearman_rows_long needs two float32 GPU matrices of the same shape")
    if A.stride(1) != 1 or X.stride(1) != 1:
        A, X = A.contiguous(), X.contiguous()
    rho = torch.empty(rows, dtype=torch.float64, device=A.device)
    if rows == 0:
        return rho
    need = L.mfcd_spearman_long_workspace_bytes(rows, m)
    if need == 0:
        raise _lib.MfcdError(f"rows of {m} columns are beyond the rank kernels")
    ws = _workspace(need, A.device)
    _lib.check(L.mfcd_spearman_rows_long(A.data_ptr(), A.stride(0) if rows > 1 else m, X.data_ptr(),
                                         X.stride(0) if rows > 1 else m, rows, m, _lib.ptr(rho), _lib.ptr(ws), ws.numel(),
                                         _lib.stream_ptr(A.device)))
    return rho


def spearman_rows(A, X):
    """Per-row Spearman rho of two [rows, m] fp32 GPU matrices (rows may be strided views) → f64 [rows] on device.
    HIP kernel (include/mfcd.h: mfcd_spearman_rows), m <= mfcd_spearman_max_columns() = 20448."""
    L = _lib.load()
    rows, m = A.shape
    if X.shape != A.shape or A.dtype != torch.float32 or X.dtype != torch.float32 or not A.is_cuda or not X.is_cuda:
        raise _lib.MfcdError("spearman_rows needs two float32 GPU matrices of the same shape")
    if A.stride(1) != 1 or X.stride(1) != 1:
        A, X = A.contiguous(), X.contiguous()
    rho = torch.empty(rows, dtype=torch.float64, device=A.device)
    _lib.check(L.mfcd_spearman_rows(A.data_ptr(), A.stride(0) if rows > 1 else m, X.data_ptr(),
                                    X.stride(0) if rows > 1 else m, rows, m, _lib.ptr(rho),
                                    _lib.stream_ptr(A.device)))
    return rho


class _PerMatrixCache:
    """One entry, tied to the tensor OBJECT it was computed for: a weak reference that must still resolve to the very
    same tensor, plus its in-place version counter and the caller's extra key.  (Keys made of data_ptr/shape alone go
    stale: the caching allocator hands a freed X's block to the next X of the same shape — ADVICE r2.)"""

    def __init__(self):
        self._ref, self._key, self._val = None, None, None

    def clear(self):
        self._ref, self._key, self._val = None, None, None

    def get(self, X, extra=()):
        if self._ref is not None and self._ref() is X and self._key == (X._version, X.data_ptr(), extra):
            return True, self._val
        return False, None

    def put(self, X, val, extra=()):
        self._ref, self._key, self._val = weakref.ref(X), (X._version, X.data_ptr(), extra), val


_x_spectrum_cache = _PerMatrixCache()


def _x_singular_values(X, xmean):
    """Singular values of the row-centred X from the Gram matrix on its smaller side (f64): same spectrum as
    torch.linalg.svd(X) to ~4e-8 relative at C2 and 40x cheaper than svdvals (118 ms vs 4.6 s at 4096^2).
    X is constant while a model trains and is evaluated against it, so the spectrum is cached per X (keyed by
    tensor object and in-place version counter)."""
    found, hit = _x_spectrum_cache.get(X)
    if found:
        return hit
    n, m = X.shape
    Xc = (X - xmean[:, None]).double()
    G = Xc @ Xc.t() if n <= m else Xc.t() @ Xc
    s1 = torch.sqrt(torch.clamp(torch.linalg.eigvalsh(G), min=0.0)).flip(0)
    _x_spectrum_cache.put(X, s1)         # one X at a time
    return s1


_x_top_cache = _PerMatrixCache()


def _x_top_spectrum(X, xmean, r, tol=1e-11, max_iter=40, oversample=16):
    """(the r largest singular values of the row-centred X, descending, f64; ||X_c||_F^2) — or None when a few block
    power iterations do not pin them down (slowly decaying spectrum: the caller takes the dense eigen-solver).

    The singular-value error only needs these: U V^T has at most r = min(d, n, m) non-zero singular values, so
        ||alpha s2 - s1||^2 = sum_{i<r} (alpha s2_i - s1_i)^2 + (||X_c||_F^2 - sum_{i<r} s1_i^2),   ||s1||^2 = ||X_c||_F^2.
    Subspace iteration with Rayleigh-Ritz on the Gram operator of the smaller side, f64, block r + 16; every accepted
    Ritz value carries a residual below tol * lambda_max (|lambda - theta| <= ||residual||).  For the generators'
    rank-d matrices this converges in two or three iterations: 118 ms -> a few ms at 4096^2, 3.5 s -> ~0.1 s at 16384^2."""
    n, m = X.shape
    dim = min(n, m)
    b = r + oversample
    if r <= 0 or 2 * b > dim:
        return None
    found, hit = _x_top_cache.get(X, r)
    if found:
        return hit                              # (None is cached too: a spectrum that did not converge is not retried)
    Xc = (X - xmean[:, None]).double()
    fro2 = float((Xc * Xc).sum())
    if n <= m:
        apply = lambda Q: Xc @ (Xc.t() @ Q)
    else:
        apply = lambda Q: Xc.t() @ (Xc @ Q)
    g = torch.Generator(device=X.device).manual_seed(20240)
    Q = torch.linalg.qr(torch.randn(dim, b, dtype=torch.float64, device=X.device, generator=g)).Q
    out = None
    prev_res = None
    for it in range(max_iter):
        Z = apply(Q)
        T = Q.t() @ Z
        w, S = torch.linalg.eigh(0.5 * (T + T.t()))            # ascending
        top = S[:, -r:]
        res = torch.linalg.norm(Z @ top - (Q @ top) * w[-r:], dim=0).max()
        lam_max = w[-1].clamp_min(1e-300)
        rel = float(res / lam_max)
        if rel <= tol:
            out = (torch.sqrt(torch.clamp(w[-r:], min=0.0)).flip(0), fro2)
            break
        if it >= 3 and prev_res is not None and rel > 0.5 * prev_res and rel > 1e-6:
            break                                               # not contracting: leave it to the dense solver
        prev_res = rel
        Q = torch.linalg.qr(Z).Q
    _x_top_cache.put(X, out, r)                 # one X at a time
    return out


def spearman_and_svd(U, V, X_centred_rows_mean, X, alpha, ok_rows, row_block=2048):
    """per-row Spearman rho (HIP rank kernel on rows of U V^T formed by a plain library GEMM; torch
    sort-based ranks only for rows longer than the kernel's 20448 columns) and the singular-value error."""
    n, m = X.shape
    rho = torch.empty(n, dtype=torch.float64, device=X.device)
    vbar = V.mean(dim=0, keepdim=True)
    Vc = V - vbar
    in_kernel = m <= _lib.load().mfcd_spearman_max_columns()
    for r0 in range(0, n, row_block):
        r1 = min(n, r0 + row_block)
        A = U[r0:r1] @ Vc.t()                              # row-centred UV^T block (ranks ignore the shift)
        rho[r0:r1] = spearman_rows(A, X[r0:r1]) if in_kernel else spearman_rows_any(A, X[r0:r1])
    rho = rho.cpu().numpy()
    scores = [np.float64(r) for r, o in zip(rho, ok_rows) if o and not np.isnan(r)]  # spearmanr yields np.float64
    # singular values: X centred (n x m); UV^T centred has rank <= d -> spectrum from a d x d problem
    try:
        Vc = Vc.double()
        # sigma(U Vc^T) = sqrt(eig( (U^T U)^{1/2} (Vc^T Vc) (U^T U)^{1/2} )) ; use QR-free form via svdvals of R factors
        Ru = torch.linalg.qr(U.double(), mode="r").R
        Rv = torch.linalg.qr(Vc, mode="r").R
        s2 = torch.linalg.svdvals(Ru @ Rv.t())
        r = min(len(s2), n, m)
        top = _x_top_spectrum(X, X_centred_rows_mean, r)
        if top is not None:
            # only the r largest singular values of X meet a non-zero partner; the rest enter through ||X_c||_F^2
            s1r, fro2 = top
            head = float(((alpha * s2[:r] - s1r) ** 2).sum())
            tail = max(0.0, fro2 - float((s1r ** 2).sum()))
            svd_err = float(np.sqrt(head + tail) / (np.sqrt(fro2) + 1e-8))
        else:
            s1 = _x_singular_values(X, X_centred_rows_mean)
            k = min(len(s1), n, m)
            s2p = torch.zeros(k, dtype=torch.float64, device=X.device)
            s2p[: min(k, len(s2))] = s2[: min(k, len(s2))]
            svd_err = float((torch.linalg.norm(alpha * s2p - s1[:k]) / (torch.linalg.norm(s1[:k]) + 1e-8)).item())
        failed = False
    except Exception:  # the reference swallows SVD failures the same way (structure.py:1018-1020)
        svd_err, failed = 1.0, True
    return scores, svd_err, failed


def alpha_and_norm_ratios(U, V, X):
    """compute_alpha_and_norm_ratios (structure.py:958-1082) → the same 14-tuple."""
    if U.dtype == torch.bfloat16:
        U, V = U.float(), V.float()
    row_stats, _ = uvt_stats(U, V, X, 1.0, what=1)
    rs = row_stats.cpu().numpy()
    n, m = X.shape
    sac, saa, scc = rs[:, 0], rs[:, 1], rs[:, 2]
    dot, nu2, nx2 = float(sac.sum()), float(saa.sum()), float(scc.sum())
    norm_UVT, norm_X = float(np.sqrt(nu2)), float(np.sqrt(nx2))
    alpha = dot / (norm_UVT ** 2 + 1e-8)                                        # 994
    norm_ratio = norm_UVT / (norm_X + 1e-8)                                     # 995
    rec_scaled = float(np.sqrt(max(alpha * alpha * nu2 - 2 * alpha * dot + nx2, 0.0))) / (norm_X + 1e-8)  # 996
    std_x, std_u = np.sqrt(scc / m), np.sqrt(saa / m)                            # np.std of centred rows
    ok = (std_x > 1e-8) & (std_u > 1e-8)                                         # 1006, 1027
    with np.errstate(invalid="ignore", divide="ignore"):
        corr_all = sac / np.sqrt(saa * scc)
    correlations = [c for c in corr_all[ok]]                                     # np.float64 like np.corrcoef
    pearson_mean = float(np.mean(correlations)) if correlations else 0.0
    xm = torch.from_numpy(rs[:, 4]).to(device=X.device, dtype=torch.float32)
    spearman_scores, svd_err, failed = spearman_and_svd(U, V, xm, X, alpha, ok)
    if failed:
        pearson_mean = 0.0                                                       # structure.py:1019
    spearman_mean = float(np.mean(spearman_scores)) if spearman_scores else 0.0
    pearson_std = float(np.std(correlations)) if correlations else 0.0
    spearman_std = float(np.std(spearman_scores)) if spearman_scores else 0.0
    sl_ok = (scc > 1e-8) & (std_u > 1e-8)                                        # 1042-1043
    slopes = [np.float32(v) for v in (sac[sl_ok] / scc[sl_ok])]                  # np.dot of fp32 rows -> np.float32
    with np.errstate(invalid="ignore", divide="ignore"):
        a_i = np.where(saa > 1e-8, sac / np.where(saa > 1e-8, saa, 1.0), 0.0)    # 1057-1058
    alpha_per_row = [np.float32(v) if saa[k] > 1e-8 else 0.0 for k, v in enumerate(a_i)]
    rec_rows = float(np.sqrt(max(float(np.sum(a_i * a_i * saa - 2 * a_i * sac + scc)), 0.0))) / (norm_X + 1e-8)  # 1064
    return (alpha, norm_X, norm_ratio, rec_scaled, pearson_mean, pearson_std, spearman_mean, spearman_std,
            svd_err, slopes, correlations, spearman_scores, rec_rows, alpha_per_row)
