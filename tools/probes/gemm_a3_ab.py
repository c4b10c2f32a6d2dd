#!/usr/bin/env python3
"""Three-deep A ring of the eight-phase GEMM (gemm.hip, A3) against the two-deep product form: the encoder stacks' shapes at B videos,
random data, both forms interleaved in ONE process (merv_debug_set_gemm_variant, third byte), outputs compared bit for bit.
usage: gemm_a3_ab.py [videos=16] [rounds=5] -> one line per shape + gpurun_out/gemm_a3_ab.json"""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch

from merv_amd import _lib, ops

dev = torch.device("cuda:0")
lib = _lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 5
M_lb, M_dn, M_vv, M_sg = 16 * 257 * B, 16 * 261 * B, 3137 * B, 16 * 196 * B
shapes = [
    ("lb.qkv", M_lb, 3072, 1024, "none", False), ("lb.proj", M_lb, 1024, 1024, "none", True),
    ("lb.fc1", M_lb, 4096, 1024, "quick_gelu", False), ("lb.fc2", M_lb, 1024, 4096, "none", True),
    ("dino.proj", M_dn, 1024, 1024, "none", True), ("dino.fc2", M_dn, 1024, 4096, "none", True),
    ("vv.qkv", M_vv, 2304, 768, "none", False), ("vv.proj", M_vv, 768, 768, "none", True),
    ("vv.fc1", M_vv, 3072, 768, "gelu_tanh", False), ("vv.fc2", M_vv, 768, 3072, "none", True),
    ("sig.proj", M_sg, 768, 768, "none", True), ("sig.fc2", M_sg, 768, 3072, "none", True),
    ("projector", 1024 * B, 4096, 1024, "none", False),
]
MODES = {"two-deep": 1 << 16, "A3": 2 << 16}  # third byte k + 1 -> mode k (0 never, 1 always)
g = torch.Generator(device=dev).manual_seed(0)
rows = []
for name, M, N, K, act, res in shapes:
    a = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g, device=dev) * K**-0.5).to(torch.bfloat16)
    bias = torch.randn(N, generator=g, device=dev)
    r = torch.randn(M, N, generator=g, device=dev).to(torch.bfloat16) if res else None
    outs, times = {}, {m: [] for m in MODES}
    for rnd in range(ROUNDS):
        for m, code in MODES.items():
            lib.merv_debug_set_gemm_variant(code)
            out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            ops.gemm(a, w, bias=bias, act=act, res=r, out=out)
            if rnd == 0:
                outs[m] = out
            n = 20
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                ops.gemm(a, w, bias=bias, act=act, res=r, out=out)
            e1.record()
            torch.cuda.synchronize()
            times[m].append(e0.elapsed_time(e1) / n * 1e3)
    equal = bool(torch.equal(outs["two-deep"], outs["A3"]))
    ref = a.float() @ w.float().t() + bias
    if act == "quick_gelu":
        ref = ref * torch.sigmoid(1.702 * ref)
    elif act == "gelu_tanh":
        ref = 0.5 * ref * (1 + torch.tanh(ref * 0.7978845608 * (1 + 0.044715 * ref * ref)))
    if res:
        ref = ref + r.float()
    err = float((outs["A3"].float() - ref).norm() / ref.norm())
    med = {m: sorted(t)[len(t) // 2] for m, t in times.items()}
    row = {"shape": name, "M": M, "N": N, "K": K, "bit_equal": equal, "rel_err_vs_fp32": err,
           "us_two_deep": [round(x, 1) for x in times["two-deep"]], "us_a3": [round(x, 1) for x in times["A3"]],
           "median_gain_pct": round((med["two-deep"] / med["A3"] - 1) * 100, 2)}
    rows.append(row)
    print(f"{name:10s} M={M:6d} N={N:5d} K={K:5d} two-deep {med['two-deep']:7.1f} us  A3 {med['A3']:7.1f} us  gain {row['median_gain_pct']:+5.2f} %  "
          f"bit-equal {equal}  err {err:.2e}", flush=True)
    assert equal and err < 6e-3, row
lib.merv_debug_set_gemm_variant(1 << 16)
Path("gpurun_out").mkdir(exist_ok=True)
json.dump({"videos": B, "rounds": ROUNDS, "rows": rows}, open("gpurun_out/gemm_a3_ab.json", "w"), indent=1)
