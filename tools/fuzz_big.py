"""Diagnostic: randomised sweep of the big resident form (mfcd_train_steps_big) against the streaming form: random table
shapes with 16 000 <= n + m <= 131 072 (d = 64), batch sizes <= 64, ragged last batches, soft labels, several calls in a
row; IEEE flavour: parameters, moments and losses bit-equal; fast flavour: within the resident tolerance.
python tools/fuzz_big.py [trials] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "matrix-factorization-with-comparison-data_amd")]
os.environ.setdefault("OMP_NUM_THREADS", "4")
import numpy as np, torch
import structure as S
from mfcd import engine

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 50
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")
d, bad, refused = 64, 0, 0
for t in range(trials):
    total = int(rng.integers(16000, 131073))
    n = int(rng.integers(1, total - 1))
    m = total - n
    if m < 2:
        continue
    B = int(rng.choice([1, 7, 16, 33, 64, 64]))
    N = int(rng.integers(1, 40)) * B + int(rng.integers(0, B))
    soft = bool(rng.random() < 0.3)
    U0 = (rng.standard_normal((n, d)) / np.sqrt(d)).astype(np.float32)
    V0 = (rng.standard_normal((m, d)) / np.sqrt(d)).astype(np.float32)
    u, i = rng.integers(0, n, N), rng.integers(0, m, N)
    j = (i + 1 + rng.integers(0, m - 1, N)) % m
    z = rng.integers(0, 5, N) / 4.0 if soft else rng.integers(0, 2, N).astype(np.float64)
    lr, wd = float(rng.choice([1e-3, 1e-2])), float(rng.choice([0.0, 1e-5, 1e-3]))
    rec = engine.SampleStore(np.stack([u, i, j, z], 1).astype(np.float64), n, m, dev).dev
    calls = int(rng.integers(1, 4))

    def fresh():
        model = S.MatrixFactorization(n, m, d)
        with torch.no_grad():
            model.U.copy_(torch.from_numpy(U0)); model.V.copy_(torch.from_numpy(V0))
        model = model.to(dev)
        opt = torch.optim.Adam(model.parameters(), lr=lr, weight_decay=wd)
        return model, opt, engine.AdamBinding(model, opt)
    engine.set_train_path("streaming")
    try:
        ms, os_, bs = fresh()
        ref = torch.cat([engine.train_steps(bs, rec, B).clone() for _ in range(calls)])
    finally:
        engine.set_train_path("auto")
    msg = []
    for flavour in ("ieee", "fast"):
        mb, ob, bb = fresh()
        big = engine.BigResident(bb)
        if not big.takes(rec, B):
            # few item (or user) rows: a batch's 2 B item references land in a handful of wave slices — the form's
            # pre-check says no and engine.train_steps streams such calls
            if min(n, m) >= 128 * 2 * B:
                msg.append("pre-check refused a stream spread over many slices")
            refused += 1
            break
        engine.set_resident_math(flavour)
        try:
            got = torch.cat([big.train_steps(rec, B).clone() for _ in range(calls)])
            big.status()
        except Exception as e:
            msg.append(f"{flavour}: {type(e).__name__}: {e}"[:100])
            continue
        finally:
            engine.set_resident_math("fast")
        if flavour == "ieee":
            same = torch.equal(got, ref) and torch.equal(mb.U.data, ms.U.data) and torch.equal(mb.V.data, ms.V.data) and \
                all(torch.equal(os_.state[a][k], ob.state[b][k]) for a, b in ((ms.U, mb.U), (ms.V, mb.V))
                    for k in ("exp_avg", "exp_avg_sq"))
            if not same:
                msg.append("ieee flavour not bit-equal to the streaming form")
        else:
            dU = (mb.U.data - ms.U.data).abs().max().item()
            dV = (mb.V.data - ms.V.data).abs().max().item()
            dl = (got - ref).abs().max().item()
            if max(dU, dV) > 0.05 * lr + 3e-6 or dl > 2e-5:
                msg.append(f"fast flavour off: dU {dU:.1e} dV {dV:.1e} dloss {dl:.1e}")
    if msg:
        bad += 1
        print(f"trial {t}: n={n} m={m} B={B} N={N} soft={soft} lr={lr} wd={wd} calls={calls}: " + "; ".join(msg), flush=True)
    if (t + 1) % 25 == 0:
        print(f"... {t + 1} trials, {bad} bad", flush=True)
print(f"done: {trials} trials, {bad} bad ({refused} narrow tables refused by the pre-check, as intended)")
