import sys
sys.path.insert(0, "/root/repo")
import torch
from merv_amd import _lib, ops
dev = torch.device("cuda:0"); lib = _lib.load()
g = torch.Generator(device=dev).manual_seed(0)
for name, M, N, K in [("lb.fc1", 32896, 4096, 1024), ("vv.fc1", 25096, 3072, 768)]:
    a = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16); w = (torch.randn(N, K, generator=g, device=dev) * K**-0.5).to(torch.bfloat16)
    bias = torch.randn(N, generator=g, device=dev); out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    line = name
    for act in ("none", "gelu_erf", "gelu_tanh", "quick_gelu"):
        for bz in (None, bias):
            best = 1e9
            for _ in range(3):
                ops.gemm(a, w, bias=bz, act=act, out=out)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): ops.gemm(a, w, bias=bz, act=act, out=out)
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 10)
            line += f" | {act}{'+b' if bz is not None else ''}: {2.0*M*N*K/best/1e9:6.1f}"
    print(line, flush=True)
