import sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from merv_amd import ops
dev = torch.device("cuda:0")
M, N, K = 16448, 3072, 1024
b = torch.zeros(N, device=dev)
def bench(a, w, n=50):
    out = ops.gemm(a, w, b)
    for _ in range(5): ops.gemm(a, w, b, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): ops.gemm(a, w, b, out=out)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, mk in [("random", lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)), ("zeros", lambda *s: torch.zeros(*s, device=dev, dtype=torch.bfloat16)),
                 ("ones", lambda *s: torch.ones(*s, device=dev, dtype=torch.bfloat16))]:
    for (m, n_, k) in [(16448, 3072, 1024), (16448, 1024, 4096)]:
        a, w = mk(m, k), mk(n_, k)
        b = torch.zeros(n_, device=dev)
        t = bench(a, w, 200)
        print(f"{name:7s} M={m} N={n_} K={k}: {t:.1f} us  {2*m*n_*k/t/1e6:.0f} TF/s")
