#!/usr/bin/env python3
"""One video per step: when does each encoder's chain END inside the concurrent step? (A timing event behind every chain, against one at the start.)
Tells which chain is the critical one under the current launch policy."""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch

import bench

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
specs, _, path, _ = bench.build_models(dev)
path.threaded_enqueue = False
names = [s.name for s in specs]
res = {}
for B in [int(a) for a in sys.argv[1:]] or [1, 2]:
    pix = bench.synth_pixels(specs, B, dev, seed=0)
    for _ in range(5):
        path.forward(pix)
    torch.cuda.synchronize()
    ends = {n: [] for n in names}
    total = []
    for _ in range(20):
        main = torch.cuda.current_stream(dev)
        smap = path.stream_map(B)
        e0 = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record(main)
        evs = {}
        for i in path.enqueue_order(B):
            st = path.streams[smap[i]]
            st.wait_event(e0)
            path.encode_project(i, pix[i], st)
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(st)
            evs[i] = ev
            main.wait_event(ev)
        path.fuse([path.buffers(i, B)["proj"] for i in range(len(specs))])
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record(main)
        torch.cuda.synchronize()
        for i, ev in evs.items():
            ends[names[i]].append(e0.elapsed_time(ev))
        total.append(e0.elapsed_time(e1))
    med = lambda v: round(sorted(v)[len(v) // 2], 3)
    res[f"{B} videos"] = {"chain_end_ms_after_start (median of 20 single calls, one host thread)": {n: med(v) for n, v in ends.items()}, "step_ms": med(total),
                          "stream_map": path.stream_map(B)}
    print(B, json.dumps(res[f"{B} videos"]), flush=True)
print(json.dumps(res))
