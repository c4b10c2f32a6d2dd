import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import os
os.environ.setdefault("MERV_TUNING_HOOKS", "1")  # forced tile configurations / kernel forms: the hooks build (merv_amd/_lib.py)
import torch
from merv_amd import _lib, ops
dev = torch.device("cuda:0"); lib = _lib.load()
g = torch.Generator(device=dev).manual_seed(0)
for name, M, N, K in [("qkv", 32768, 3072, 1024), ("fc1", 32768, 4096, 1024), ("fc2", 32768, 1024, 4096), ("proj", 32768, 1024, 1024),
                      ("vqkv", 25088, 2304, 768), ("vfc1", 25088, 3072, 768), ("vfc2", 25088, 768, 3072), ("vproj", 25088, 768, 768),
                      ("projector", 8192, 4096, 1024)]:
    a = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16); w = (torch.randn(N, K, generator=g, device=dev) * K**-0.5).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev); line = name
    for gm in (1, 2, 4, 6, 8, 12, 16):
        lib.merv_debug_set_gemm_variant(7 | (gm << 8))
        best = 1e9
        for _ in range(3):
            ops.gemm(a, w, out=out)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): ops.gemm(a, w, out=out)
            e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1) / 10 * 1e3)
        line += f" | gm{gm}: {best:6.1f}"
    lib.merv_debug_set_gemm_variant(0); print(line)
