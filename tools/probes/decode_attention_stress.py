import sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from merv_amd import _lib
from merv_amd._lib import check, ptr
lib = _lib.load()
dev = torch.device("cuda:0")
st = lambda: torch.cuda.current_stream(dev).cuda_stream
H, Hkv, hd, max_len, ns = 32, 8, 128, 2048, 8
g = torch.Generator().manual_seed(1)
bf = lambda t: t.to(torch.bfloat16).to(dev)
Kc, Vc = bf(torch.randn(Hkv, max_len, hd, generator=g)), bf(torch.randn(Hkv, max_len, hd, generator=g))
inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
emb = torch.outer(torch.arange(max_len, dtype=torch.float32), inv); emb = torch.cat([emb, emb], -1)
cos, sin = bf(emb.cos()), bf(emb.sin())
ws = torch.zeros(lib.merv_decode_attention_fused_workspace_floats(H, ns), dtype=torch.float32, device=dev)
ws_a = torch.empty(lib.merv_decode_attention_workspace_floats(H, ns), dtype=torch.float32, device=dev)
bad = 0
N = 3000
side = torch.cuda.Stream(dev)
big = torch.randn(64, 1024, 1024, device=dev)  # background traffic on another stream to perturb timing
for it in range(N):
    pos = int(torch.randint(0, max_len - 1, (1,), generator=g))
    q, k, v = bf(torch.randn(H * hd, generator=g)), bf(torch.randn(Hkv * hd, generator=g)), bf(torch.randn(Hkv * hd, generator=g))
    p = torch.tensor([pos], dtype=torch.int64, device=dev)
    Ka, Va, q2 = Kc.clone(), Vc.clone(), torch.empty_like(q)
    out_a = torch.empty(H * hd, dtype=torch.bfloat16, device=dev)
    check(lib.merv_decode_rope_cache(ptr(q), ptr(k), ptr(v), ptr(q2), ptr(Ka), ptr(Va), ptr(cos), ptr(sin), ptr(p), H, Hkv, hd, max_len, st()), "rope")
    check(lib.merv_decode_attention(ptr(q2), ptr(Ka), ptr(Va), ptr(out_a), ptr(ws_a), ptr(p), H, Hkv, hd, max_len, ns, hd**-0.5, st()), "attn")
    Kb, Vb = Kc.clone(), Vc.clone()
    out_b = torch.full((H * hd,), float("nan"), dtype=torch.bfloat16, device=dev)
    if it % 3 == 0:
        with torch.cuda.stream(side):
            big.mul_(1.0001)
    for rep in range(2):  # back to back on the same workspace
        check(lib.merv_decode_attention_fused(ptr(q), ptr(k), ptr(v), ptr(cos), ptr(sin), ptr(p), ptr(Kb), ptr(Vb), ptr(out_b), ptr(ws), H, Hkv, hd, max_len, ns, hd**-0.5, st()), "fused")
    if not (torch.equal(out_a, out_b) and torch.equal(Ka, Kb) and torch.equal(Va, Vb)):
        bad += 1
torch.cuda.synchronize()
print("iterations", N, "mismatches", bad, "counters", int(ws[H * ns * 130:].view(torch.int32).abs().sum()))
