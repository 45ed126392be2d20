"""Diagnostic: randomised single-process rehearsals of the two multi-GPU loops (every rank's part played in this process,
no communicator): the data-parallel loop for world R against the streaming step with batch 64 R, the row-sharded loop
against the streaming step with batch 64 — both must be bit-identical.  python tools/fuzz_multi.py [trials] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "matrix-factorization-with-comparison-data_amd")]
os.environ.setdefault("OMP_NUM_THREADS", "4")
import numpy as np, torch
import structure as S
from mfcd import dist as mdist, engine

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")
bad = 0


def fresh(U0, V0):
    n, d = U0.shape
    model = S.MatrixFactorization(n, V0.shape[0], d)
    with torch.no_grad():
        model.U.copy_(torch.from_numpy(U0)); model.V.copy_(torch.from_numpy(V0))
    model = model.to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-5)
    return model, opt, engine.AdamBinding(model, opt)


for t in range(trials):
    d = int(rng.choice([1, 3, 8, 16, 32, 33, 64, 128]))
    n = int(rng.integers(1, 600)) if rng.random() < 0.8 else int(rng.integers(1, 6))
    m = int(rng.integers(2, 600)) if rng.random() < 0.8 else int(rng.integers(2, 6))
    world = int(rng.choice([1, 2, 3, 4, 5, 8]))
    B = 64
    N = int(rng.integers(1, 12)) * B * world + int(rng.integers(0, B * world))
    U0 = (rng.standard_normal((n, d)) / np.sqrt(d)).astype(np.float32)
    V0 = (rng.standard_normal((m, d)) / np.sqrt(d)).astype(np.float32)
    u, i = rng.integers(0, n, N), rng.integers(0, m, N)
    j = (i + 1 + rng.integers(0, max(m - 1, 1), N)) % m
    z = rng.integers(0, 5, N) / 4.0
    rec = engine.SampleStore(np.stack([u, i, j, z], 1).astype(np.float64), n, m, dev).dev
    msg = []
    engine.set_train_path("streaming")
    try:
        mo, op, bi = fresh(U0, V0)
        ref_dp_loss = engine.train_steps(bi, rec, B * world).clone()
        ref_dp = (mo.U.data.clone(), mo.V.data.clone())
        mo, op, bi = fresh(U0, V0)
        ref_sh_loss = engine.train_steps(bi, rec, B).clone()
        ref_sh = (mo.U.data.clone(), mo.V.data.clone(), op.state[mo.U]["exp_avg"].clone(), op.state[mo.V]["exp_avg_sq"].clone())
    finally:
        engine.set_train_path("auto")
    try:
        mo, op, bi = fresh(U0, V0)
        loss = mdist.NativeDP(bi, simulate_world=world).train_steps(rec, B)
        if not (torch.equal(mo.U.data, ref_dp[0]) and torch.equal(mo.V.data, ref_dp[1])):
            msg.append("data-parallel rehearsal: parameters differ")
        if not torch.allclose(loss, ref_dp_loss, rtol=1e-6, atol=1e-7):
            msg.append("data-parallel rehearsal: losses differ")
        mo, op, bi = fresh(U0, V0)
        loss = mdist.NativeShard(bi, simulate_world=world).train_steps(rec, B)
        if not (torch.equal(mo.U.data, ref_sh[0]) and torch.equal(mo.V.data, ref_sh[1]) and torch.equal(loss, ref_sh_loss)
                and torch.equal(op.state[mo.U]["exp_avg"], ref_sh[2]) and torch.equal(op.state[mo.V]["exp_avg_sq"], ref_sh[3])):
            msg.append("row-sharded rehearsal: not bit-identical")
    except Exception as e:
        msg.append(f"EXCEPTION {type(e).__name__}: {e}")
    if msg:
        bad += 1
        print(f"trial {t}: n={n} m={m} d={d} world={world} N={N}: " + "; ".join(msg), flush=True)
print(f"done: {trials} trials, {bad} bad")
