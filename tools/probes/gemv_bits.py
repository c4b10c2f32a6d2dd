"""Output bits of the decode GEMV shapes under the current MERV_GEMV_* hooks: prints one sha1 per case (two processes with different
hooks must print the same lines)."""
import hashlib, sys, torch
sys.path.insert(0, ".")
from merv_amd import _lib
from merv_amd._lib import check, ptr

lib = _lib.load()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
g = torch.Generator().manual_seed(5)


def sha(*ts):
    torch.cuda.synchronize()
    h = hashlib.sha1()
    for t in ts:
        h.update(t.cpu().contiguous().view(torch.int16 if t.dtype == torch.bfloat16 else torch.int32).numpy().tobytes())
    return h.hexdigest()


def mat(n, k):
    return (torch.randn(n, k, generator=g) * k**-0.5).to(torch.bfloat16).to(dev)


for N, K, res, norm, gated in [(4096, 4096, True, False, False), (4096, 11008, True, False, False), (4098, 11008, False, False, False),
                               (32000, 4096, False, True, False), (32000, 4096, False, False, False), (1000, 5120, True, False, False),
                               (77, 1024, False, False, False), (11008, 4096, False, True, True), (11006, 4096, False, True, True),
                               (11008, 4096, False, False, True), (4096, 4096, False, True, False), (4095, 4000, True, True, False)]:
    W, W2 = mat(N, K), mat(N, K) if gated else None
    x = (torch.randn(K, generator=g) * 2).to(torch.bfloat16).to(dev)
    wn = (1 + 0.2 * torch.randn(K, generator=g)).to(torch.bfloat16).to(dev)
    r = torch.randn(N, generator=g).to(torch.bfloat16).to(dev)
    y = torch.empty(N, dtype=torch.bfloat16, device=dev)
    check(lib.merv_decode_gemv(ptr(W), ptr(W2) if gated else 0, ptr(x), ptr(r) if res else 0, ptr(y), 0, N, K, ptr(wn) if norm else 0, 1e-5, st), "gemv")
    print(N, K, res, norm, gated, sha(y))
K = 4096
for Ns, norm in [((4096, 4096, 4096), True), ((4096, 1024, 1024), True), ((4096, 1024, 1026), False)]:
    Ws = [mat(n, K) for n in Ns]
    x = torch.randn(K, generator=g).to(torch.bfloat16).to(dev)
    wn = (1 + 0.2 * torch.randn(K, generator=g)).to(torch.bfloat16).to(dev)
    ys = [torch.empty(n, dtype=torch.bfloat16, device=dev) for n in Ns]
    check(lib.merv_decode_gemv3(ptr(Ws[0]), ptr(Ws[1]), ptr(Ws[2]), ptr(x), ptr(ys[0]), ptr(ys[1]), ptr(ys[2]), *Ns, K, ptr(wn) if norm else 0, 1e-5, st), "gemv3")
    print("gemv3", Ns, norm, sha(*ys))
