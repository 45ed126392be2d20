"""Diagnostic: per-phase in-kernel cycle accounting of the tiled UV^T kernel (build with -DMFCD_UVT_STAMPS=1,
point MFCD_LIB at it).  Prints average cycles per wave-tile for: chain (X-load issue + fragment reads + MFMAs),
epilogue, stage-end wait + barrier (per stage), LDS-DMA issue (per stage)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "matrix-factorization-with-comparison-data_amd")]
import torch
from mfcd import metrics, _lib
L = _lib.load()
rd = L.mfcd_uvt_debug_read
dev = torch.device("cuda:0")
SECONDS = float(os.environ.get("UVT_DIAG_SECONDS", "0"))   # of back-to-back passes before the measured one
for name, n, m, d in [("C2", 4096, 4096, 64), ("C3", 16384, 16384, 128), ("C5", 100000, 20000, 256)]:
    if len(sys.argv) > 1 and name not in sys.argv[1:]:
        continue
    U = torch.randn(n, d, device=dev) / d ** 0.5
    V = torch.randn(m, d, device=dev) / d ** 0.5
    X = torch.randn(n, m, device=dev) * 0.5
    out = (ctypes.c_ulonglong * 8)()
    metrics.uvt_stats(U, V, X, 1.0); rd(out)
    import time
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < SECONDS:      # back-to-back passes: lets the clock settle under this load
        for _ in range(20):
            metrics.uvt_stats(U, V, X, 1.0)
        torch.cuda.synchronize()
    rd(out)
    metrics.uvt_stats(U, V, X, 1.0); rd(out)
    chain, epi, sync, dma, tiles, waves = [int(x) for x in out[:6]]
    cyc, rt = int(out[6]), int(out[7])
    if rt:
        print(f"{name}: stage loop per wave: {cyc/waves:.0f} shader cycles in {rt/waves/100:.1f} us -> in-kernel clock {cyc/rt*0.1:.3f} GHz; "
              f"{cyc/tiles:.0f} cycles per tile (MFMA issue floor {d//2*64})", flush=True)
    if not chain:
        continue
    print(f"{name}: waves {waves} tiles {tiles} ({tiles/waves:.1f}/wave) | per tile: chain {chain/tiles:8.0f}  epilogue {epi/tiles:7.0f}  "
          f"sync {sync/tiles:7.0f}  dma-issue {dma/tiles:6.0f}  total {(chain+epi+sync+dma)/tiles:8.0f} cycles "
          f"(MFMA issue floor {d//2*64})", flush=True)
    del U, V, X
