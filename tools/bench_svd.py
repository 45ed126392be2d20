"""Diagnostic: singular values of X by Gram matrix + eigvalsh against torch.linalg.svdvals (time and agreement)."""
import os, sys, time, torch
sys.path[:0] = ["/root/repo", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "matrix-factorization-with-comparison-data_amd")]
import generation_data as gd
dev = "cuda"
n = m = 4096; d = 64
A, B = gd.generate_embedding_factors(n, m, d, dev)
X = (A @ B.t()); Xc = X - X.mean(1, keepdim=True)
def t(name, fn):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter()-t0)*1e3:.1f} ms", flush=True); return r
s64 = t("svdvals f64", lambda: torch.linalg.svdvals(Xc.double()))
s32 = t("svdvals f32", lambda: torch.linalg.svdvals(Xc))
def gram():
    G = Xc.double() @ Xc.double().t()
    return torch.sqrt(torch.clamp(torch.linalg.eigvalsh(G), min=0)).flip(0)
sg = t("eigvalsh(Gram f64)", gram)
ref = torch.linalg.norm(s64)
print("rel diff f32 svdvals:", float(torch.linalg.norm(s32.double() - s64) / ref), " gram:", float(torch.linalg.norm(sg - s64) / ref))
