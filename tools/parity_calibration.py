#!/usr/bin/env python3
"""Calibrates the stated bf16 tolerance against the reference stack's own numerics ("next round" item 2).

One video, full depth (23 / 23 / 12 / 11 consumed blocks), full width, ONE set of weights and pixels, three computations:
  oracle   -- oracle/merv_oracle.py, fp32 on the CPU (the checker);
  hip      -- the product: libmerv_hip.so through MervVisualPath;
  refstack -- the reference's own stack on this GPU: plain PyTorch-ROCm bf16 ops (library GEMM, SDPA, layer_norm, gelu), what
              timm / transformers launch under vidlm.to(bf16) + autocast (tools/torch_rocm_baseline.py).
(here built with separate LayerNorm kernels) and, with --ln-fold, the HIP path with LayerNorm folded into the qkv / fc1
GEMMs -- the product default.
Reports err(refstack vs oracle), err(hip vs oracle), err(hip vs refstack) per encoder, per projector and on the fused
[1,1024,4096] tokens. Prints one JSON object (commit it as profiles/rNN_parity_calibration.json)."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
import torch

import bench
from oracle.parity import compare, reference_video
from torch_rocm_baseline import path_bf16, to_ref_stack


def main():
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    specs, _, path, extras = bench.build_models(dev, concurrent=False, want_ref=True, ln_fold=False)
    ref = extras["ref"]
    pix = bench.synth_pixels(specs, 1, dev, seed=4242)

    def hip_run():
        fused, w = path.forward(pix)
        torch.cuda.synchronize()
        return {"tokens": [path.buffers(i, 1)["tokens"].float().cpu() for i in range(len(specs))],
                "projected": [path.buffers(i, 1)["proj"].float().cpu() for i in range(len(specs))],
                "fused": fused.float().cpu(), "weights": w.float().cpu()}

    hip = hip_run()
    torch.set_num_threads(min(16, torch.get_num_threads()))
    orc, secs = reference_video([p.float().cpu() for p in pix], specs, ref["enc_W"], ref["proj_W"], ref["Fw"])

    # the reference stack on the same values: every parameter bf16 (vidlm.to(bf16)), same folded fusion query
    Ws = [to_ref_stack(W, dev) for W in ref["enc_W"]]
    u = extras["fusion"].fold().to(dev)
    fused_r, w_r, proj_r = path_bf16(pix, specs, Ws, [(w.to(dev), b.to(dev)) for w, b in ref["proj_W"]], u)
    from torch_rocm_baseline import encoder_bf16
    tok_r = [encoder_bf16(p, s, W).float().cpu() for p, s, W in zip(pix, specs, Ws)]
    rs = {"tokens": tok_r, "projected": [p.float().cpu() for p in proj_r], "fused": fused_r.float().cpu(), "weights": w_r.float().cpu()}
    del Ws

    def table(a, b):
        out = {s.name: {"tokens": compare(a["tokens"][i], b["tokens"][i]), "projected": compare(a["projected"][i], b["projected"][i])}
               for i, s in enumerate(specs)}
        out["fused"] = compare(a["fused"], b["fused"])
        out["fusion_weights_max_abs_diff"] = round(float((a["weights"] - b["weights"]).abs().max()), 6)
        return out

    res = {"what": "one video, full depth 23/23/12/11, shared weights and pixels; rel_l2 / min per-token cosine",
           "refstack_vs_oracle": table(rs, orc), "hip_vs_oracle": table(hip, orc), "hip_vs_refstack": table(hip, rs),
           "oracle_seconds": round(secs, 1), "torch": torch.__version__}
    if "--ln-fold" in sys.argv:
        for enc in path.encoders:
            enc.enable_ln_fold()
        path._bufs.clear()
        folded = hip_run()
        res["hip_lnfold_vs_oracle"] = table(folded, orc)
        res["hip_lnfold_vs_hip"] = table(folded, hip)
    worst = lambda t: max([t["fused"]["rel_l2"]] + [t[s.name]["tokens"]["rel_l2"] for s in specs])
    res["summary"] = {"worst_rel_l2_refstack_vs_oracle": worst(res["refstack_vs_oracle"]),
                      "worst_rel_l2_hip_vs_oracle": worst(res["hip_vs_oracle"]),
                      "hip_error_not_above_reference_stack_error": worst(res["hip_vs_oracle"]) <= worst(res["refstack_vs_oracle"]) * 1.05}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
