"""Diagnostic: event trace of one resident epoch at C2 (build: -DMFCD_STAMPS=2 -DMFCD_TRACE, MFCD_LIB=.../libmfcd_hip_diag.so).
Every hit leaves {sample position, arrival time, poll-success time, spins}; every publish leaves its time per (sample, role)
slot.  From those the script rebuilds who waited for whom and walks the critical path of the launch backwards from the
wave that finished last: how much of it is execution, how much hand-off latency, how often it changes wave."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "matrix-factorization-with-comparison-data_amd")]
os.environ.setdefault("OMP_NUM_THREADS", "4")
import numpy as np, torch
import bench
from mfcd import engine, _lib

dev = torch.device("cuda:0")
cfg = bench.C2 | {"name": "C2"}
r = bench.Runner(cfg, dev, 0)
r.run(1049); torch.cuda.synchronize()
order = torch.randperm(r.train.N, generator=r.gen)
stream = r.train.ordered(order)
torch.cuda.synchronize()
engine.train_steps(r.bind, stream, 64); torch.cuda.synchronize()
N, B, n, m, d = r.train.N, 64, cfg["n"], cfg["m"], cfg["d"]
L = _lib.load()
L.mfcd_diag_mailbox_offset.restype = ctypes.c_size_t
L.mfcd_diag_mailbox_offset.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
ws = engine.workspace_for(dev, (n, m, d))
Ncap = ws.plan[0]
off = L.mfcd_diag_mailbox_offset(Ncap, B, n, m, d)
u64 = ws.buf[off:].view(torch.int64)
pub = u64[Ncap * 3 * d: Ncap * 3 * d + N * 3].cpu().numpy().reshape(N, 3)            # publish time per (sample, role), 10 ns
log = u64[Ncap * 3 * d + Ncap * 3: Ncap * 3 * d + Ncap * 3 + 4096 * 128 * 4].cpu().numpy().reshape(4096, 128, 4)
rec = stream.cpu().numpy()                                                             # (u, i, j, z) in table ids
# owner wave of a row (C2: interleaved virtual order, two rows per wave): user u -> wave u, item i -> wave i
owner = np.stack([rec[:, 0], rec[:, 1], rec[:, 2]], 1)

hits = []   # (wave, k, pos, own-mask, t_arrive, t_success, spins)
for w in range(4096):
    for e in log[w]:
        if e[2] == 0:
            break
        pos = int(e[0] & ((1 << 56) - 1)); own = int((e[0] >> 56) & 7)
        hits.append((w, pos // B, pos, own, int(e[1]), int(e[2]), int(e[3])))
hits = np.array(hits, dtype=np.int64)
t0 = hits[:, 4].min()
hits[:, 4] -= t0; hits[:, 5] -= t0
pubt = pub - t0
print(f"hits logged {len(hits)}; launch spans {hits[:,5].max()/100:.0f} us; waited {np.mean(hits[:,6] > 0)*100:.0f} % of hits, "
      f"mean wait of those {np.mean((hits[:,5]-hits[:,4])[hits[:,6] > 0])/100:.2f} us")
# for every waited hit: which publisher was last, how late relative to my arrival, and how far behind in steps was it
by_wave = {}
for row in hits:
    by_wave.setdefault(int(row[0]), []).append(row)
for w in by_wave:
    by_wave[w].sort(key=lambda x: x[4])
lat, late = [], []
for row in hits:
    if row[6] == 0:
        continue
    w, k, pos, own = int(row[0]), int(row[1]), int(row[2]), int(row[3])
    others = [rr for rr in range(3) if not (own >> rr) & 1]
    tp = max(pubt[pos, rr] for rr in others)
    lat.append(row[5] - tp)
    late.append(tp - row[4])
lat, late = np.array(lat) / 100.0, np.array(late) / 100.0
print(f"waited hits: success comes {np.median(lat):.2f} us (median) after the last needed publish [p10 {np.percentile(lat,10):.2f}, p90 {np.percentile(lat,90):.2f}]; "
      f"that publish came {np.median(late):.2f} us after my arrival [p10 {np.percentile(late,10):.2f}, p90 {np.percentile(late,90):.2f}]")

# critical path, backwards: at a waited hit jump to the wave that published last (at its publish time); otherwise keep
# walking back on the same wave.  The publish of (pos, role) by owner wave P happened at pubt; P's own activity before
# that is its hit list.
def last_hit_before(w, t):
    best = None
    for row in by_wave.get(w, []):
        if row[5] <= t:
            best = row
        else:
            break
    return best
end_w = int(hits[np.argmax(hits[:, 5]), 0])
t = int(hits[:, 5].max())
w = end_w
hops, lat_sum, wait_free, segs = 0, 0.0, 0, []
guard = 0
while guard < 100000:
    guard += 1
    h = last_hit_before(w, t)
    if h is None:
        segs.append((w, 0, t)); break
    if h[6] == 0:                       # did not wait: the path stays on this wave, go further back
        t = int(h[4]) - 1
        wait_free += 1
        continue
    pos, own = int(h[2]), int(h[3])
    others = [rr for rr in range(3) if not (own >> rr) & 1]
    rr = max(others, key=lambda q: pubt[pos, q])
    tp = int(pubt[pos, rr])
    segs.append((w, int(h[5]), t))
    lat_sum += (int(h[5]) - tp) / 100.0
    hops += 1
    w = int(owner[pos, rr]); t = tp
total = hits[:, 5].max() / 100.0
print(f"critical path from wave {end_w}: {hops} changes of wave over {total:.0f} us; hand-off latency on the path {lat_sum:.0f} us "
      f"({lat_sum/total*100:.0f} %), the rest ({total - lat_sum:.0f} us) is execution of whichever wave the path is on; it passes "
      f"{hops + wait_free} hits in {int(hits[:,1].max()) + 1} steps ({(hops + wait_free)/(hits[:,1].max() + 1):.3f} per step; an average wave has "
      f"{len(hits)/4096/(hits[:,1].max() + 1):.3f})")
