#!/usr/bin/env python3
"""Is the eight-phase GEMM power-bound? Board power, the power cap and the sysfs clock sampled (rocm-smi, a background thread) while one
workload at a time runs back to back for a few seconds: the LanguageBind qkv GEMM on random operands, the same launch on zero operands
(same instruction stream: the data-dependent part of the power), the K = 4096 fc2 GEMM, the resident attention launch, the HBM-bound
temporal attention, and the whole 16-video step. MI355X_MICROARCH.md ("DVFS give-back") warns that board power and pp_dpm_sclk are not
the test of the in-kernel clock (profiles/r05_gemm_energy_bound.json has that); this file adds what the board itself reports beside it."""
import json
import subprocess
import sys
import threading
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch

from merv_amd import ops

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower", "--json"], capture_output=True, text=True, timeout=10).stdout
        d = json.loads(out)
        card = d.get("card0", next(iter(d.values())))
        return {k: v for k, v in card.items() if any(s in k.lower() for s in ("power", "sclk", "mclk", "fclk"))}
    except Exception as e:  # noqa: BLE001
        return {"error": str(e)}


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.samples, self.stop = [], False

    def run(self):
        while not self.stop:
            self.samples.append(smi())
            time.sleep(0.05)


def run_for(fn, seconds):
    fn(); torch.cuda.synchronize()
    s = Sampler(); s.start()
    t0 = time.perf_counter(); n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            fn()
        n += 20
        torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    s.stop = True; s.join()
    return e0.elapsed_time(e1) / n * 1e3, s.samples


def num(v):
    try:
        return float(str(v).replace("Mhz", "").replace("(", "").replace(")", "").split()[0])
    except Exception:  # noqa: BLE001
        return None


def summarize(samples):
    keys = sorted({k for s in samples for k in s})
    out = {}
    for k in keys:
        vals = [num(s[k]) for s in samples if k in s and num(s[k]) is not None]
        if vals:
            vals.sort()
            out[k] = {"median": vals[len(vals) // 2], "max": vals[-1], "n": len(vals)}
        else:
            out[k] = str(samples[0].get(k))
    return out


M = 65792
rand = lambda *s: torch.randn(*s, generator=g, device=dev).to(torch.bfloat16)
work = {}
a, w, b = rand(M, 1024), (rand(3072, 1024) * 0.03), torch.zeros(3072, device=dev)
o = torch.empty(M, 3072, dtype=torch.bfloat16, device=dev)
work["gemm qkv N=3072 K=1024, random operands"] = lambda: ops.gemm(a, w, b, out=o)
az, wz = torch.zeros_like(a), torch.zeros_like(w)
work["gemm qkv N=3072 K=1024, zero operands"] = lambda: ops.gemm(az, wz, b, out=o)
a2, w2, b2 = rand(M, 4096), (rand(1024, 4096) * 0.015), torch.zeros(1024, device=dev)
r2, o2 = rand(M, 1024), torch.empty(M, 1024, dtype=torch.bfloat16, device=dev)
work["gemm fc2 N=1024 K=4096 + residual, random operands"] = lambda: ops.gemm(a2, w2, b2, res=r2, out=o2)
qkv = (torch.randn(256 * 257, 3072, generator=g, device=dev) * 1.5).to(torch.bfloat16)
work["attention, 257 tokens x 16 heads x 256 sequences (resident)"] = lambda: ops.attention(qkv, 256, 257, 16)
work["temporal attention (HBM-bound)"] = lambda: ops.temporal_attention(qkv, 32, 8, 257, 16)
res = {"idle": summarize([smi() for _ in range(5)])}
for name, fn in work.items():
    us, samples = run_for(fn, 5.0)
    res[name] = {"us_per_launch": round(us, 1), "smi": summarize(samples)}
    print(name, round(us, 1), {k: (v["median"] if isinstance(v, dict) else v) for k, v in res[name]["smi"].items()}, file=sys.stderr, flush=True)
import bench
specs, path = bench.build_path(dev)
pix = bench.synth_pixels(specs, 16, dev, 0)
us, samples = run_for(lambda: path.forward(pix), 8.0)
res["whole step, 16 videos, concurrent encoder streams"] = {"ms_per_step": round(us / 1e3, 2), "smi": summarize(samples)}
print(json.dumps({"what": "rocm-smi --showpower --showclocks --showmaxpower sampled every ~0.3 s while one workload runs back to back (5-8 s each); medians / maxima of the samples",
                  "workloads": res}))
