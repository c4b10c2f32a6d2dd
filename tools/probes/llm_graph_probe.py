# Does the HF Llama decode step capture into a hipGraph with a StaticCache? (transformers 5.x, PyTorch-ROCm)
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from transformers import LlamaConfig, LlamaForCausalLM, StaticCache
from merv_amd.llm import llama2_7b_config

dev = torch.device("cuda:0")
layers = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cfg = LlamaConfig(**dict(llama2_7b_config(), num_hidden_layers=layers)); cfg._attn_implementation = "sdpa"
torch.manual_seed(0)
with torch.device(dev):
    llm = LlamaForCausalLM(cfg)
llm = llm.to(torch.bfloat16).eval().requires_grad_(False)
S, NEW = 1049, 64
emb = torch.randn(1, S, cfg.hidden_size, device=dev, dtype=torch.bfloat16) * 0.02
with torch.inference_mode():
    # eager dynamic-cache reference
    out = llm(inputs_embeds=emb, use_cache=True); past = out.past_key_values
    tok = out.logits[:, -1].argmax(-1); ref = [int(tok)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(NEW - 1):
        out = llm(input_ids=tok[:, None], past_key_values=past, use_cache=True); past = out.past_key_values
        tok = out.logits[:, -1].argmax(-1); ref.append(int(tok))
    torch.cuda.synchronize(); t_eager = time.perf_counter() - t0
    # static cache + graph
    cache = StaticCache(config=cfg, max_cache_len=S + NEW)
    out = llm(inputs_embeds=emb, past_key_values=cache, cache_position=torch.arange(S, device=dev), use_cache=True)
    tok_buf = out.logits[:, -1].argmax(-1)[:, None].clone()
    pos_buf = torch.tensor([S], device=dev)
    got = [int(tok_buf)]
    def step():
        o = llm(input_ids=tok_buf, past_key_values=cache, cache_position=pos_buf, use_cache=True)
        return o.logits[:, -1].argmax(-1)
    nxt = step()  # warm-up at position S (eager): writes the cache slot S, which the replay below rewrites identically
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        nxt_static = step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(NEW - 1):
        g.replay()
        tok_buf.copy_(nxt_static[:, None]); pos_buf.add_(1)
        got.append(int(tok_buf))
    torch.cuda.synchronize(); t_graph = time.perf_counter() - t0
    # static cache, eager (no graph): must reproduce the graph's tokens exactly; and its logits vs the dynamic cache's
    cache2 = StaticCache(config=cfg, max_cache_len=S + NEW)
    o2 = llm(inputs_embeds=emb, past_key_values=cache2, cache_position=torch.arange(S, device=dev), use_cache=True)
    t2 = o2.logits[:, -1].argmax(-1)[:, None]; eager_static = [int(t2)]
    o_dyn = llm(inputs_embeds=emb, use_cache=True)
    l_dyn = llm(input_ids=t2, past_key_values=o_dyn.past_key_values, use_cache=True).logits[:, -1].float()
    for i in range(NEW - 1):
        o2 = llm(input_ids=t2, past_key_values=cache2, cache_position=torch.tensor([S + i], device=dev), use_cache=True)
        if i == 0:
            l_st = o2.logits[:, -1].float()
            print("step-2 logits static vs dynamic: rel l2 %.3e, top-2 gap of dynamic %.3e" % (
                float((l_st - l_dyn).norm() / l_dyn.norm()), float(l_dyn.topk(2).values.diff().abs())))
        t2 = o2.logits[:, -1].argmax(-1)[:, None]; eager_static.append(int(t2))
    print("graph == static-eager tokens:", eager_static == got)
print("layers", layers, "eager %.1f ms/token, graph %.1f ms/token" % (t_eager / (NEW - 1) * 1e3, t_graph / (NEW - 1) * 1e3))
print("tokens equal:", ref == got, ref[:8], got[:8])
