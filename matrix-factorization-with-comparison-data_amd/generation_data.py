"""Host-side data preparation with the reference module's names (ref = reference generation_data.py).

Triplet samplers and ground-truth generators are one-off host work outside the device hot path
(SURVEY §8f rows N2/N3).  They keep the reference's names, arguments, return types, printed warnings
and — where it is cheap — the same order of RNG draws, so seeded runs line up with the reference.
`generate_embeddings` additionally has a factored large-scale form (`generate_embedding_factors`):
the reference's O(n^3) full orthogonal matrices are only ever used through their first d columns.
"""
import math

import numpy as np
import torch

# ------------------------------------------------------------------------------------------------
# triplet samplers (ref:16-338): each returns a list of unique (u, i, j) tuples
# ------------------------------------------------------------------------------------------------


def _accept(t, i, j, exclude, found):
    return i != j and t not in exclude and t not in found


def _choose_items_random_serial(n, m, num_triplets, exclude):
    found = set()
    while len(found) < num_triplets:
        u = int(torch.randint(0, n, (1,)))
        i, j = torch.randint(0, m, (2,)).tolist()
        if _accept((u, i, j), i, j, exclude, found):
            found.add((u, i, j))
    return list(found)


def choose_items_random(X, num_triplets, exclude):
    """Uniform u, i, j (ref:16-26): per attempt one randint(n) and one randint(m, size 2), rejected when i == j,
    excluded or already drawn.

    Vectorised without changing a single draw: ATen's CPU randint takes one 32-bit Mersenne-Twister word w per
    element and returns w % range, serially, so `randint(0, L, (3A,))` with L = lcm(n, m) yields words whose
    residues mod n / mod m are exactly the A attempts' (u, i, j).  Attempts are drawn in blocks, filtered with
    numpy, and the generator is then rewound and advanced by exactly the number of attempts the one-at-a-time
    loop would have used, so everything drawn afterwards (labels, initial factors, shuffles) is unchanged.
    The accepted triplets go into a Python set in attempt order: `list(set)` order is part of the contract
    (the 80/10/10 split indexes into it, ref:705-718)."""
    n, m = X.shape
    exclude = exclude or set()
    L = math.lcm(int(n), int(m))
    if L >= 2 ** 32 or num_triplets <= 0:
        return _choose_items_random_serial(n, m, num_triplets, exclude)
    if num_triplets + len(exclude) > n * m * (m - 1):
        raise ValueError(f"cannot draw {num_triplets} distinct triplets from a {n} x {m} matrix")
    enc = lambda a: (a[:, 0] * m + a[:, 1]) * m + a[:, 2]                       # noqa: E731
    barred = np.sort(enc(np.asarray(list(exclude), dtype=np.int64).reshape(-1, 3))) if exclude else None
    state0 = torch.get_rng_state()
    taken, taken_keys, attempts, need = [], np.empty(0, dtype=np.int64), 0, int(num_triplets)
    while need > 0:
        block = max(4096, need + need // 8 + 64)
        w = torch.randint(0, L, (3 * block,)).numpy().reshape(block, 3)
        cand = np.stack((w[:, 0] % n, w[:, 1] % m, w[:, 2] % m), axis=1)
        key = enc(cand)
        ok = cand[:, 1] != cand[:, 2]
        if barred is not None:
            ok &= ~np.isin(key, barred)
        if taken_keys.size:
            ok &= ~np.isin(key, taken_keys)
        idx = np.flatnonzero(ok)
        _, first = np.unique(key[idx], return_index=True)                      # first attempt of every new triplet
        idx = idx[np.sort(first)]
        if idx.size >= need:
            idx = idx[:need]
            attempts += int(idx[-1]) + 1
        else:
            attempts += block
        taken.append(cand[idx])
        taken_keys = np.concatenate((taken_keys, key[idx]))
        need -= idx.size
    torch.set_rng_state(state0)
    torch.randint(0, L, (3 * attempts,))                                        # leave the generator where the loop would
    rows = np.concatenate(taken)
    found = set()
    for t in zip(rows[:, 0].tolist(), rows[:, 1].tolist(), rows[:, 2].tolist()):
        found.add(t)
    return list(found)


def _choose_items_by_proximity_serial(X, num_triplets, exclude, k=100):
    n, m = X.shape
    kk = min(k, m)
    found, cache = set(), {}
    while len(found) < num_triplets:
        u = int(torch.randint(0, n, (1,)))
        if u not in cache:
            row = X[u]
            cache[u] = (torch.topk(row, k=kk)[1].tolist(), torch.topk(-row, k=kk)[1].tolist())
        best, worst = cache[u]
        i, j = int(np.random.choice(best)), int(np.random.choice(worst))
        if _accept((u, i, j), i, j, exclude, found):
            found.add((u, i, j))
    return list(found)


def _legacy_choice_block(k, nwords):
    """What the next calls of the legacy `np.random.choice(a_list_of_length_k)` return, in bulk.

    numpy's RandomState.choice without p draws `randint(0, k)`, which for k - 1 < 2^32 takes one 32-bit
    Mersenne-Twister word at a time, masks it with the smallest all-ones mask >= k - 1 and rejects values above k - 1
    (distributions.c, buffered_bounded_masked_uint32): the sequence of returned indices is the masked word stream with
    the rejected words dropped, whoever asks.  Draws `nwords` raw words from the global numpy generator and returns
    (indices, pos): the indices the successive choice() calls would return and, for each, the position of its word in
    the block (pos[t] + 1 words are consumed once t + 1 calls have returned).  Needs k >= 2 (k = 1 consumes nothing)."""
    raw = np.random.randint(0, 2 ** 32, size=nwords, dtype=np.uint32)
    top = k - 1
    mask = top
    for sh in (1, 2, 4, 8, 16):
        mask |= mask >> sh
    val = raw & np.uint32(mask)
    pos = np.flatnonzero(val <= top)
    return val[pos].astype(np.int64), pos


def _advance_generators(t_state, n_state, torch_words, n, numpy_words):
    """Put torch's CPU generator `torch_words` draws of randint(0, n) and numpy's global generator `numpy_words` 32-bit
    words past the given states: where a one-attempt-at-a-time loop that used that many would have left them."""
    torch.set_rng_state(t_state)
    np.random.set_state(n_state)
    if torch_words:
        torch.randint(0, n, (torch_words,))
    if numpy_words:
        np.random.randint(0, 2 ** 32, size=numpy_words, dtype=np.uint32)


def _user_item_tables(X, users, kk, worst):
    """Rows of torch.topk indices (the reference's per-user lists) for the distinct users of a block."""
    uu, inv = np.unique(users, return_inverse=True)
    rows = X[torch.from_numpy(uu).to(X.device)]
    best = torch.topk(rows, k=kk, dim=1)[1].cpu().numpy()
    low = torch.topk(-rows, k=kk, dim=1)[1].cpu().numpy() if worst else None
    return inv, best, low


def _first_new(key, ok, barred, found_keys):
    """Indices (ascending) of the attempts that add a triplet: allowed, not barred, first occurrence of their key."""
    if barred is not None:
        ok = ok & ~np.isin(key, barred)
    if found_keys.size:
        ok = ok & ~np.isin(key, found_keys)
    idx = np.flatnonzero(ok)
    _, first = np.unique(key[idx], return_index=True)
    first.sort()
    return idx[first]


def _barred_keys(exclude, m):
    if not exclude:
        return None
    return np.sort(np.fromiter(((u * m + i) * m + j for u, i, j in exclude), dtype=np.int64, count=len(exclude)))


def _as_tuple_list(rows):
    """list(set) of the rows inserted in attempt order: the order the reference's `list(triplets)` has."""
    got = np.concatenate(rows) if rows else np.empty((0, 3), dtype=np.int64)
    found = set()
    for t in zip(got[:, 0].tolist(), got[:, 1].tolist(), got[:, 2].tolist()):
        found.add(t)
    return list(found)


def choose_items_by_proximity(X, num_triplets, exclude, k=100):
    """"Min-Max": i among the user's k best items, j among the k worst (ref:29-43): per attempt one torch.randint(n),
    two torch.topk over the user's row and two legacy `np.random.choice(list)` calls.

    Consumed in bulk without changing a draw (as `choose_items_random` does for its strategy): the users of a block of
    attempts are one randint call, their top / bottom lists one batched topk over the distinct users, and the two
    choices of attempt a are entries 2a and 2a + 1 of the masked-rejection stream of `_legacy_choice_block` (both
    lists have k entries, so the two calls share range and mask).  Both generators are left where the
    one-at-a-time loop would leave them.  The reference recomputes both topk for every attempt (O(m) each)."""
    n, m = X.shape
    kk = min(k, m)
    exclude = exclude or set()
    if kk < 2 or n >= 2 ** 32 or num_triplets <= 0 or not torch.is_tensor(X):
        return _choose_items_by_proximity_serial(X, num_triplets, exclude, k)
    barred = _barred_keys(exclude, m)
    t_state, n_state = torch.get_rng_state(), np.random.get_state()
    rows, found_keys = [], np.empty(0, dtype=np.int64)
    attempts = words = 0
    need = int(num_triplets)
    span = 1 << (kk - 1).bit_length()                                            # words per accepted draw ~ span / kk
    while need > 0:
        A = max(4096, need + need // 4 + 64)
        idx_stream, pos = _legacy_choice_block(kk, int(2 * A * span / kk * 1.05) + 256)
        A = min(A, idx_stream.size // 2)
        us = torch.randint(0, n, (A,)).numpy()
        inv, best, low = _user_item_tables(X, us, kk, worst=True)
        ii = best[inv, idx_stream[0:2 * A:2]].astype(np.int64)
        jj = low[inv, idx_stream[1:2 * A:2]].astype(np.int64)
        key = (us.astype(np.int64) * m + ii) * m + jj
        idx = _first_new(key, ii != jj, barred, found_keys)
        if idx.size >= need:
            idx = idx[:need]
            used = int(idx[-1]) + 1
        else:
            used = A
        attempts += used
        words += int(pos[2 * used - 1]) + 1
        rows.append(np.stack((us[idx], ii[idx], jj[idx]), axis=1))
        found_keys = np.concatenate((found_keys, key[idx]))
        need -= idx.size
        _advance_generators(t_state, n_state, attempts, n, words)
    return _as_tuple_list(rows)


def _x_pair_diff(X, us, ii, jj):
    """X[u, i] - X[u, j] for index arrays, as float64 numpy.  `X` is a dense [n, m] tensor / array or a
    FactoredMatrix (below), for which nothing of size n x m is ever formed."""
    if isinstance(X, FactoredMatrix):
        return X.pair_diff(us, ii, jj)
    Xn = X if isinstance(X, np.ndarray) else X.detach().cpu().numpy()
    return Xn[us, ii].astype(np.float64) - Xn[us, jj].astype(np.float64)


class FactoredMatrix:
    """Ground-truth matrix kept as its factors, X = A @ B.T (A [n, d], B [m, d], fp32 CPU tensors): the form the
    "base" law has at BASELINE sizes C4 / C5, where the dense matrix would be 16 GiB / 7.5 GiB.  It offers exactly what
    the samplers and the label generator read from X: its shape, the first rows, single entries."""

    def __init__(self, A, B):
        self.A, self.B = A.detach().float().cpu().contiguous(), B.detach().float().cpu().contiguous()
        self.shape = (self.A.shape[0], self.B.shape[0])

    def rows(self, r0, r1):
        return (self.A[r0:r1] @ self.B.t()).numpy()

    def entries(self, us, ii):
        A, B = self.A.numpy(), self.B.numpy()
        return np.einsum("td,td->t", A[us], B[ii]).astype(np.float32)

    def pair_diff(self, us, ii, jj):
        return self.entries(us, ii).astype(np.float64) - self.entries(us, jj).astype(np.float64)

    def dense(self, device="cpu"):
        return (self.A.to(device) @ self.B.to(device).t())


def choose_items_by_margin(X, num_triplets, exclude, max_attempts=5000_000):
    """"Close-Call": |X[u,i]-X[u,j]| below an adaptive margin (ref:46-84).  As there: the margin is the mean range
    of the first ten rows times the sampling density, attempts come in blocks of 500 from an UNSEEDED numpy Generator
    (so there is no draw order to keep), at most `max_attempts` of them, and a short result is reported, not raised.
    Many blocks are drawn and filtered per numpy pass (the reference walks every attempt in Python); the attempt
    counter still advances in blocks of 500 and stops with the block that completes the request."""
    n, m = X.shape
    exclude = exclude or set()
    head = X.rows(0, min(10, n)) if isinstance(X, FactoredMatrix) else X[:min(10, n)].cpu().numpy()
    margin = np.mean(head.max(axis=1) - head.min(axis=1)) * num_triplets / (n * m)
    Xn = X if isinstance(X, FactoredMatrix) else X.cpu().numpy()
    rng = np.random.default_rng()
    enc = lambda u, i, j: (u.astype(np.int64) * m + i) * m + j                   # noqa: E731
    barred = np.sort(np.fromiter(((u * m + i) * m + j for u, i, j in exclude), dtype=np.int64, count=len(exclude))) \
        if exclude else None
    found_keys = np.empty(0, dtype=np.int64)
    rows, attempts, block = [], 0, 500
    need = int(num_triplets)
    while need > 0 and attempts < max_attempts:
        nblk = max(1, min(400, (max_attempts - attempts + block - 1) // block))   # blocks in this pass
        us = rng.integers(0, n, size=nblk * block)
        ij = rng.integers(0, m, size=(nblk * block, 2))
        ii, jj = ij[:, 0], ij[:, 1]
        close = ii != jj
        cand = np.flatnonzero(close)
        close[cand] = np.abs(_x_pair_diff(Xn, us[cand], ii[cand], jj[cand])) <= margin
        idx = np.flatnonzero(close)
        key = enc(us[idx], ii[idx], jj[idx])
        ok = np.ones(idx.size, dtype=bool)
        if barred is not None:
            ok &= ~np.isin(key, barred)
        if found_keys.size:
            ok &= ~np.isin(key, found_keys)
        idx, key = idx[ok], key[ok]
        _, first = np.unique(key, return_index=True)                              # first attempt of every new triplet
        first.sort()
        idx, key = idx[first], key[first]
        if idx.size >= need:                                                      # the request completes inside this pass
            idx, key = idx[:need], key[:need]
            attempts += (int(idx[-1]) // block + 1) * block
        else:
            attempts += nblk * block
        rows.append(np.stack((us[idx], ii[idx], jj[idx]), axis=1))
        found_keys = np.concatenate((found_keys, key))
        need -= idx.size
    got = np.concatenate(rows) if rows else np.empty((0, 3), dtype=np.int64)
    found = set()
    for t in zip(got[:, 0].tolist(), got[:, 1].tolist(), got[:, 2].tolist()):
        found.add(t)
    if len(found) < num_triplets:
        top = float(np.max(X.A.numpy()) if isinstance(X, FactoredMatrix) else np.max(Xn))
        print(f"⚠️ Only {len(found)} triplets generated (target={num_triplets}, margin={margin:.4f}) "
              f"after {attempts} attempts.maximum : {top}")
    return list(found)


def choose_items_by_variance(X, num_triplets, exclude):
    """Items drawn proportionally to their variance across users (ref:87-99)."""
    n, m = X.shape
    var = torch.var(X, dim=0)
    probs = var / var.sum()
    found = set()
    while len(found) < num_triplets:
        u = int(torch.randint(0, n, (1,)))
        i, j = torch.multinomial(probs, 2, replacement=False).tolist()
        t = (u, i, j)
        if t not in found and t not in exclude:
            found.add(t)
    return list(found)


def _popularity_probs(m, method, alpha):
    if method == "zipf":
        w = 1.0 / (np.arange(1, m + 1) ** alpha)
    elif method == "exponential":
        w = np.exp(-alpha * np.arange(m))
    elif method == "uniform":
        w = np.ones(m)
    else:
        raise ValueError(f"Unknown popularity method: {method}")
    return w / w.sum()


def _choose_items_by_popularity_serial(n, m, probs, num_triplets, exclude):
    items = np.arange(m)
    found = set()
    while len(found) < num_triplets:
        u = int(torch.randint(0, n, (1,)))
        i, j = np.random.choice(items, size=2, replace=False, p=probs).tolist()
        if _accept((u, i, j), i, j, exclude, found):
            found.add((u, i, j))
    return list(found)


def choose_items_by_popularity(X, num_triplets, exclude, method="zipf", alpha=1.5):
    """i, j drawn without replacement from an item-popularity law over the item index (ref:103-128): per attempt one
    torch.randint(n) and one numpy `choice(m, size=2, replace=False, p=probs)`.

    The reference pays O(m) per attempt inside `choice` (a cumsum over the catalogue): ~100 s for C5's 500 000
    triplets.  Here the SAME draws are consumed in bulk.  numpy's legacy choice (mtrand.pyx, replace=False with p)
    draws two uniforms x0, x1, maps them through cdf = cumsum(p) / cdf[-1] with searchsorted(side='right'); if the two
    items differ they are (i, j); if they coincide it keeps i, draws ONE more uniform and maps it through the cdf of
    p with p[i] = 0.  So an attempt consumes 2 or 3 uniforms (two 32-bit Mersenne-Twister words each) and one
    torch word (w % n).  Blocks of attempts are evaluated with numpy, the walk "position += 2 or 3" is a cheap integer
    pass, and both generators are then rewound and advanced by exactly what the one-at-a-time loop would have used, so
    everything drawn afterwards is unchanged.  `list(set)` order is kept (attempt order), as the split depends on it."""
    n, m = X.shape
    exclude = exclude or set()
    probs = _popularity_probs(m, method, alpha)
    if n >= 2 ** 32 or num_triplets <= 0 or m < 2:
        return _choose_items_by_popularity_serial(n, m, probs, num_triplets, exclude)
    cdf = np.cumsum(probs)
    cdf /= cdf[-1]
    enc = lambda u, i, j: (u.astype(np.int64) * m + i) * m + j                   # noqa: E731
    barred = np.sort(np.fromiter(((u * m + i) * m + j for u, i, j in exclude), dtype=np.int64, count=len(exclude))) \
        if exclude else None
    t_state, n_state = torch.get_rng_state(), np.random.get_state()
    rows, found_keys = [], np.empty(0, dtype=np.int64)
    attempts = uniforms = 0
    need = int(num_triplets)
    cdf_without = {}
    while need > 0:
        A = max(4096, need + need // 4 + 64)
        us = torch.randint(0, n, (A,)).numpy()
        x = np.random.random_sample(3 * A)
        first = cdf.searchsorted(x, side="right")
        same = (first[:-1] == first[1:]).tolist()
        starts = np.empty(A, dtype=np.int64)                                      # position of attempt a's first uniform
        pos = 0
        for a in range(A):
            starts[a] = pos
            pos += 3 if same[pos] else 2
        ii = first[starts]
        jj = first[starts + 1]
        coll = np.flatnonzero(ii == jj)
        if coll.size:                                                             # second round of choice(): p[i] = 0
            extra = x[starts[coll] + 2]
            for item in np.unique(ii[coll]).tolist():
                c2 = cdf_without.get(item)
                if c2 is None:
                    p2 = probs.copy()
                    p2[item] = 0
                    c2 = np.cumsum(p2)
                    c2 /= c2[-1]
                    if len(cdf_without) < 512:
                        cdf_without[item] = c2
                sel = ii[coll] == item
                jj[coll[sel]] = c2.searchsorted(extra[sel], side="right")
        key = enc(us, ii, jj)
        ok = ii != jj
        if barred is not None:
            ok &= ~np.isin(key, barred)
        if found_keys.size:
            ok &= ~np.isin(key, found_keys)
        idx = np.flatnonzero(ok)
        _, fst = np.unique(key[idx], return_index=True)
        fst.sort()
        idx = idx[fst]
        if idx.size >= need:
            idx = idx[:need]
            last = int(idx[-1])
            attempts += last + 1
            uniforms += int(starts[last]) + (3 if same[int(starts[last])] else 2)
        else:
            attempts += A
            uniforms += pos
            # the next block must continue where this one ended in BOTH streams
            torch.set_rng_state(t_state)
            np.random.set_state(n_state)
            torch.randint(0, n, (attempts,))
            np.random.random_sample(uniforms)
        rows.append(np.stack((us[idx], ii[idx], jj[idx]), axis=1))
        found_keys = np.concatenate((found_keys, key[idx]))
        need -= idx.size
    torch.set_rng_state(t_state)
    np.random.set_state(n_state)
    torch.randint(0, n, (attempts,))                                              # leave both generators where the
    np.random.random_sample(uniforms)                                             # one-at-a-time loop would
    got = np.concatenate(rows)
    found = set()
    for t in zip(got[:, 0].tolist(), got[:, 1].tolist(), got[:, 2].tolist()):
        found.add(t)
    return list(found)


def choose_items_by_svd_projection(X, num_triplets, exclude, rank=10, top_fraction=0.3):
    """Users/items with the largest truncated-SVD projection norms (ref:131-179).  As there, `rank` is overridden
    from the sampling density and at most 5*num_triplets attempts are made: u uniform over the top users, (i, j) an
    ordered pair of distinct top items.  The attempts come from an UNSEEDED numpy Generator in the reference (no draw
    order to keep), so they are drawn and filtered in one numpy pass each block instead of one Python iteration each."""
    import scipy.sparse.linalg as spla
    n, m = X.shape
    exclude = exclude or set()
    rank = int(num_triplets / (n * m) * max(n, m))
    Us, S, Vt = spla.svds(X.cpu().numpy(), k=rank)
    u_norm = np.linalg.norm(Us * S, axis=1)
    i_norm = np.linalg.norm((Vt.T * S), axis=1)
    top_users = np.argsort(u_norm)[-max(1, int(top_fraction * n)):]
    top_items = np.argsort(i_norm)[-max(2, int(top_fraction * m)):]
    rng = np.random.default_rng()
    barred = _barred_keys(exclude, m)
    rows, found_keys = [], np.empty(0, dtype=np.int64)
    need, budget = int(num_triplets), 5 * int(num_triplets)
    while need > 0 and budget > 0:
        A = min(budget, max(4096, need + need // 4 + 64))
        us = top_users[rng.integers(0, top_users.size, size=A)].astype(np.int64)
        a = rng.integers(0, top_items.size, size=A)
        b = rng.integers(0, top_items.size - 1, size=A)
        b += b >= a                                                              # uniform over the ordered distinct pairs
        ii, jj = top_items[a].astype(np.int64), top_items[b].astype(np.int64)
        key = (us * m + ii) * m + jj
        idx = _first_new(key, ii != jj, barred, found_keys)[:need]
        budget -= A
        rows.append(np.stack((us[idx], ii[idx], jj[idx]), axis=1))
        found_keys = np.concatenate((found_keys, key[idx]))
        need -= idx.size
    found = _as_tuple_list(rows)
    if len(found) < num_triplets:
        print(f"⚠️ Only {len(found)} triplets generated (target={num_triplets})")
    return found


def estimate_k(num_triplets):
    return math.ceil((1 + math.sqrt(1 + 8 * num_triplets)) / 2)


def _choose_items_top_k_serial(X, num_triplets, exclude, k=None):
    n, m = X.shape
    if k is None:
        k = min(m, max(5, int(0.1 * m)))
    found, cache = set(), {}
    for _ in range(num_triplets * 3):
        u = int(torch.randint(0, n, (1,)))
        if u not in cache:
            cache[u] = torch.topk(X[u], k=k).indices.tolist()
        best = cache[u]
        i = int(np.random.choice(best))
        j = int(np.random.choice(best))
        while i == j:
            j = int(np.random.choice(best))
        t = (u, i, j)
        if t not in found and t not in exclude:
            found.add(t)
        if len(found) >= num_triplets:
            break
    if len(found) < num_triplets:
        print(f"⚠️ Only {len(found)} triplets generated (target={num_triplets}, k={k})")
    return list(found)


def choose_items_top_k(X, num_triplets, exclude, k=None):
    """"top_10%": both items among the user's k best, k = 10 % of the catalogue, >= 5 (ref:189-224); at most
    3 * num_triplets attempts, j redrawn while it equals i.

    Same draws in bulk (see `choose_items_by_proximity`): every choice has range k, so the attempts walk one
    masked-rejection stream — attempt a takes the next index for i, then indices until one differs from it for j
    (a cheap integer pass); the users are one randint call, their lists one batched topk."""
    n, m = X.shape
    exclude = exclude or set()
    if k is None:
        k = min(m, max(5, int(0.1 * m)))
    if k < 2 or n >= 2 ** 32 or num_triplets <= 0 or not torch.is_tensor(X):
        return _choose_items_top_k_serial(X, num_triplets, exclude, k)
    barred = _barred_keys(exclude, m)
    t_state, n_state = torch.get_rng_state(), np.random.get_state()
    rows, found_keys = [], np.empty(0, dtype=np.int64)
    attempts = words = 0
    need, budget = int(num_triplets), 3 * int(num_triplets)
    span = 1 << (k - 1).bit_length()
    while need > 0 and attempts < budget:
        A = min(budget - attempts, max(4096, need + need // 4 + 64))
        idx_stream, pos = _legacy_choice_block(k, int(2 * A * (1 + 2.0 / k) * span / k * 1.05) + 256)
        stream = idx_stream.tolist()
        i_at, j_at = [], []
        p, last = 0, len(stream) - 1
        while len(i_at) < A and p < last:                                         # the reference's while i == j loop
            q = p + 1
            while q <= last and stream[q] == stream[p]:
                q += 1
            if q > last:
                break
            i_at.append(p)
            j_at.append(q)
            p = q + 1
        A = len(i_at)
        i_at, j_at = np.asarray(i_at, dtype=np.int64), np.asarray(j_at, dtype=np.int64)
        us = torch.randint(0, n, (A,)).numpy()
        inv, best, _ = _user_item_tables(X, us, k, worst=False)
        ii = best[inv, idx_stream[i_at]].astype(np.int64)
        jj = best[inv, idx_stream[j_at]].astype(np.int64)
        key = (us.astype(np.int64) * m + ii) * m + jj
        idx = _first_new(key, np.ones(A, dtype=bool), barred, found_keys)
        if idx.size >= need:
            idx = idx[:need]
            used = int(idx[-1]) + 1
        else:
            used = A
        attempts += used
        words += int(pos[j_at[used - 1]]) + 1
        rows.append(np.stack((us[idx], ii[idx], jj[idx]), axis=1))
        found_keys = np.concatenate((found_keys, key[idx]))
        need -= idx.size
        _advance_generators(t_state, n_state, attempts, n, words)
    found = _as_tuple_list(rows)
    if len(found) < num_triplets:
        print(f"⚠️ Only {len(found)} triplets generated (target={num_triplets}, k={k})")
    return found


def choose_items_cluster_based(X, num_triplets, exclude, n_clusters=20):
    """i and j from two different k-means clusters of the item columns (ref:229-247)."""
    from sklearn.cluster import KMeans
    n, m = X.shape
    labels = KMeans(n_clusters=n_clusters, n_init="auto").fit_predict(X.T.cpu().numpy())
    members = {c: np.where(labels == c)[0] for c in range(n_clusters)}
    ids = list(members)
    found = set()
    while len(found) < num_triplets:
        u = int(torch.randint(0, n, (1,)))
        c1, c2 = np.random.choice(ids, 2, replace=False)
        i, j = np.random.choice(members[c1]), np.random.choice(members[c2])
        if _accept((u, i, j), i, j, exclude, found):
            found.add((u, i, j))
    return list(found)


def choose_items_by_user_similarity(X, num_triplets, exclude=None, max_attempts=10000, verbose=False,
                                    fallback_random=False, error_if_incomplete=False):
    """Contrast a user's favourite items with those of cosine-similar users (ref:251-338)."""
    from sklearn.metrics.pairwise import cosine_similarity
    n, m = X.shape
    sim = cosine_similarity(X.cpu().numpy())
    np.fill_diagonal(sim, -1.0)
    exclude = exclude or set()
    rng = np.random.default_rng()
    n_neigh = min(20, max(3, num_triplets // n))
    top_k = max(3, min(m // 10, 10 + num_triplets // (5 * n)))
    if verbose:
        print(f"→ Adaptive config: top_k={top_k}, neighbors/user={n_neigh}, target={num_triplets}")
    favourites = torch.topk(X, k=min(top_k, m), dim=1).indices.tolist()
    found, attempts = set(), 0
    while len(found) < num_triplets and attempts < max_attempts:
        u = rng.integers(0, n)
        mine = set(favourites[u])
        for v in np.argsort(-sim[u])[:n_neigh]:
            theirs = set(favourites[v])
            only_u, only_v = list(mine - theirs), list(theirs - mine)
            if only_u and only_v:
                i, j = rng.choice(only_u), rng.choice(only_v)
            elif len(mine) >= 2:
                i, j = rng.choice(list(mine), size=2, replace=False)
            else:
                continue
            t = (u, i, j)
            if i != j and t not in found and t not in exclude:
                found.add(t)
                break
        attempts += 1
        if verbose and attempts % 1000 == 0:
            print(f"{len(found)} triplets generated after {attempts} attempts.")
    if len(found) < num_triplets:
        msg = f"⚠️ Only {len(found)} triplets generated (target={num_triplets}) after {attempts} attempts."
        if error_if_incomplete:
            raise RuntimeError(msg)
        if fallback_random:
            found.update(choose_items_random(X, num_triplets - len(found), exclude=found | exclude))
        else:
            print(msg)
    if verbose:
        print(f"✅ Returned {len(found)} triplets.")
    return list(found)


# ------------------------------------------------------------------------------------------------
# ground-truth generators (ref:346-715)
# ------------------------------------------------------------------------------------------------
def _haar_columns(dim, k):
    """First k columns of scipy.stats.ortho_group.rvs(dim) drawn from numpy's global RNG, without the O(dim^3) work.

    ortho_group (scipy 1.15 _multivariate.py: rvs) draws z = normal(size=(dim, dim)), takes q, r = qr(z) and flips
    the sign of column c of q by sign(r[c, c]).  Columns 0..k-1 of q and of r's diagonal only depend on columns
    0..k-1 of z (Householder QR proceeds column by column), so the same dim*dim normals are drawn here in row
    blocks — which leaves numpy's generator in exactly the state the full call would — and only the dim x k panel
    is factorised.  Agrees with the full call to ~1e-16 (tests/test_host_logic.py)."""
    k = min(k, dim)
    panel = np.empty((dim, k))
    block = max(1, min(dim, (1 << 22) // max(dim, 1)))      # ~32 MiB of normals at a time
    for r0 in range(0, dim, block):
        r1 = min(dim, r0 + block)
        panel[r0:r1] = np.random.normal(size=(r1 - r0, dim))[:, :k]
    q, r = np.linalg.qr(panel)
    diag = np.diagonal(r)
    return q * (diag / abs(diag))


def generate_embeddings(n, m, d, device="cpu"):
    """"base" X (ref:346-370): Q_n diag(1/sqrt(d) on the first d) Q_m^T * sqrt(nm)/2 with Haar orthogonal
    Q_n, Q_m drawn (in this order) by scipy's ortho_group from numpy's global RNG.  Only the first d
    columns of each matter, so only those are formed (same RNG consumption, see _haar_columns)."""
    k = min(d, n, m)                      # the reference's spectrum has min(d, n, m) non-zero entries, each 1/sqrt(d)
    A = _haar_columns(n, k)
    B = _haar_columns(m, k)
    X = (A / np.sqrt(d)) @ B.T * (np.sqrt(n * m) / 2)
    return torch.tensor(X, dtype=torch.float32, device=device)


def generate_embedding_factors(n, m, d, device="cpu", generator=None):
    """Large-scale form of the "base" law: Haar n x d and m x d frames from the QR of Gaussian matrices
    (sign-fixed), O((n+m) d^2) instead of O(n^3).  X = A @ B.T with A = Q_n sqrt(nm)/(2 sqrt(d))."""
    def frame(rows):
        G = torch.randn(rows, d, dtype=torch.float64, generator=generator)
        Q, R = torch.linalg.qr(G)
        return Q * torch.sign(torch.diagonal(R))
    A = frame(n) * (math.sqrt(n * m) / (2 * math.sqrt(d)))
    return A.float().to(device), frame(m).float().to(device)


def generate_low_rank_matrix(n, m, d, rank, device="cpu"):
    """Orthonormal n x d, m x d frames and a 0/1 spectrum with `rank` ones (ref:373-391)."""
    A = torch.tensor(_haar_columns(n, d), dtype=torch.float32, device=device)
    B = torch.tensor(_haar_columns(m, d), dtype=torch.float32, device=device)
    S = torch.zeros(d)
    S[:rank] = 1.0
    return A, B, S.to(device) if torch.device(device).type != "cpu" else S


def generate_clustered_matrix_from_embeddings(n, m, d, n_clusters=5, device="cpu", scale=1.0, shift_strength=0.5):
    """"base" X whose item columns are pulled towards their k-means cluster mean (ref:394-434)."""
    from sklearn.cluster import KMeans
    X = generate_embeddings(n, m, d, device="cpu").numpy()
    labels = KMeans(n_clusters=n_clusters, n_init="auto", random_state=42).fit_predict(X.T)
    out = X.copy()
    for c in range(n_clusters):
        cols = np.where(labels == c)[0]
        if len(cols):
            out[:, cols] = (1 - shift_strength) * X[:, cols] + shift_strength * X[:, cols].mean(axis=1, keepdims=True)
    return torch.tensor(out, dtype=torch.float32, device=device) * scale


def generate_structured_embeddings(n, m, d, num_clusters=5, cluster_std=0.1, device="cpu"):
    """Items scattered around cluster centres, users as mixtures of the centres (ref:437-467)."""
    centres = torch.randn(num_clusters, d, device=device)
    assign = torch.randint(0, num_clusters, (m,))
    V = centres[assign.to(centres.device)] + cluster_std * torch.randn(m, d, device=device)
    U = torch.randn(n, num_clusters, device=device) @ centres
    return U, V


def generate_svd_embeddings(n, m, d, noise_level=0.1, device="cpu"):
    """Top-d SVD factors of a Gaussian matrix, sqrt-spectrum on both sides, plus noise (ref:470-502).  As the reference
    computes it: `torch.svd` returns V, which ref:492-496 names `Vt` and slices as `Vt[:d, :].T` — the item factor is
    therefore the transposed first d ROWS of V, [min(n, m), d]; kept, so that seeded runs give the reference's matrix."""
    Uf, S, Vf = torch.svd(torch.randn(n, m, device=device))
    root = torch.sqrt(S[:d])
    U, V = Uf[:, :d] * root, Vf[:d, :].T * root
    U = U + noise_level * torch.randn_like(U)
    V = V + noise_level * torch.randn_like(V)
    return U.to(device), V.to(device)


def generate_correlated_embeddings(n, m, d, correlation_factor=0.8, device="cpu"):
    """Gaussian factors mixed by (1-c) I + c 11^T, divided by d (ref:505-534)."""
    U, V = torch.randn(n, d, device=device), torch.randn(m, d, device=device)
    C = torch.eye(d, device=device) * (1 - correlation_factor) + correlation_factor * torch.ones((d, d), device=device)
    return (U @ C) / d, (V @ C) / d


def _smooth_over_graph(U, influence):
    """In-place sequential neighbour averaging on a Watts-Strogatz(k=5, p=0.1) graph (ref:567-574, 610-617)."""
    import networkx as nx
    G = nx.watts_strogatz_graph(U.shape[0], k=5, p=0.1)
    for u in range(U.shape[0]):
        friends = list(G.neighbors(u))
        if friends:
            U[u] = (1 - influence) * U[u] + influence * U[friends].mean(dim=0)
    return U


def generate_graph_embeddings(n, m, d, device="cpu"):
    """Two socially-smoothed signal dimensions + 0.1-scale noise dimensions; V / sqrt(d) (ref:539-585).  The reference
    itself raises TypeError here (a stray comma at ref:565 makes `noise` a tuple); this is the evident intent, with the
    reference's draw order.  The graph comes from networkx, i.e. from Python's `random` module, as there."""
    d_eff = min(d, 2)
    U_low, V_low = torch.randn(n, d_eff, device=device), torch.randn(m, d_eff, device=device)
    U_low = _smooth_over_graph(U_low, 0.3)
    U = torch.cat([U_low, 0.1 * torch.randn(n, d - d_eff, device=device)], dim=1)
    V = torch.cat([V_low, 0.1 * torch.randn(m, d - d_eff, device=device)], dim=1)
    return U, V / np.sqrt(d)


def generate_social_embeddings(n, m, d, social_influence=0.5, device="cpu"):
    """Gaussian users smoothed over a small-world graph, U / log(d+1) (ref:588-619)."""
    U, V = torch.randn(n, d, device=device), torch.randn(m, d, device=device)
    return _smooth_over_graph(U, social_influence) / np.log(d + 1), V


def generate_temporal_embeddings(n, m, d, timesteps=5, device="cpu"):
    """Base factors plus `timesteps` x 0.02-scale drift; V / sqrt(d) (ref:622-651)."""
    U0, V0 = torch.randn(n, d, device=device), torch.randn(m, d, device=device)
    U = U0 + timesteps * (torch.randn(n, d, device=device) * 0.02)
    V = V0 + timesteps * (torch.randn(m, d, device=device) * 0.02)
    return U, V / np.sqrt(d)


def generate_hierarchical_embeddings(n, m, d, num_groups=5, device="cpu"):
    """Users = group centre + 10 x Gaussian; V / log(d+1) (ref:653-683)."""
    groups = torch.randn(num_groups, d, device=device)
    assign = torch.randint(0, num_groups, (n,))
    U = groups[assign.to(groups.device)] + 10 * torch.randn(n, d, device=device)
    return U, torch.randn(m, d, device=device) / np.log(d + 1)


def generate_gmm_embeddings(n, m, d, num_clusters=5, device="cpu"):
    """Every user/item snapped to the mean of its Gaussian-mixture component; note both use the means
    of the SECOND fit, as in the reference (ref:686-715)."""
    from sklearn.mixture import GaussianMixture
    gmm = GaussianMixture(n_components=num_clusters, random_state=42)
    uc = gmm.fit_predict(torch.randn(n, d).numpy())
    ic = gmm.fit_predict(torch.randn(m, d).numpy())
    return (torch.tensor(gmm.means_[uc], dtype=torch.float32, device=device),
            torch.tensor(gmm.means_[ic], dtype=torch.float32, device=device))


# ------------------------------------------------------------------------------------------------
# unused preference helpers kept for name compatibility (ref:723-742)
# ------------------------------------------------------------------------------------------------
def sigmoid_preference(U, V, u, i, j, scale=1.0):
    return int(torch.sigmoid(scale * torch.dot(U[u], V[i] - V[j])).item() > 0.5)


def softmax_preference(U, V, u, i, j, temp=1.0):
    probs = torch.softmax((V @ U[u]) / temp, dim=0)
    return int(probs[i].item() > probs[j].item())


def max_preference(U, V, u, i, j):
    return int(torch.dot(U[u], V[i] - V[j]).item() > 0)
