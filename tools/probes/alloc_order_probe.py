#!/usr/bin/env python3
"""alloc_order_probe.py: does the position of the visual path's buffers in the allocation order change its speed? (the e2e leg's
`visual_path_ms` was 2 ms above the same call measured alone)"""
import sys, time, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import bench
from merv_amd.vidlm import MERVVisual
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
specs, bbs, path, extras = bench.build_models(dev)
syn = [torch.randn(s.pixel_shape(1), device=dev).to(torch.bfloat16) for s in specs]
def lat(label):
    mm = MERVVisual(bbs, llm_dim=4096)
    for _ in range(3): mm.encode(syn)
    ts = []
    for _ in range(10):
        torch.cuda.synchronize(); t0 = time.perf_counter(); mm.encode(syn); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(label, "min %.2f ms median %.2f ms" % (min(ts) * 1e3, sorted(ts)[5] * 1e3), "reserved GB %.1f" % (torch.cuda.memory_reserved() / 2**30), flush=True)
    return mm
keep = [lat("fresh process:")]
keep.append(lat("second path, nothing in between:"))
big = torch.empty(14 * 2**30, dtype=torch.uint8, device=dev)
keep.append(lat("after one 14 GB allocation:"))
del big; torch.cuda.empty_cache()
keep.append(lat("after freeing it (empty_cache):"))
many = [torch.empty(4096, 4096, dtype=torch.bfloat16, device=dev) for _ in range(400)]
keep.append(lat("after 400 x 33.5 MB allocations:"))
mode = sys.argv[1] if len(sys.argv) > 1 else ""
def time_path(mm, label):
    for _ in range(3): mm.encode(syn)
    ts = []
    for _ in range(10):
        torch.cuda.synchronize(); t0 = time.perf_counter(); mm.encode(syn); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(label, "min %.2f ms median %.2f ms" % (min(ts) * 1e3, sorted(ts)[5] * 1e3), [hex(s.cuda_stream) for s in mm._path.streams], flush=True)
a, b = keep[0], keep[1]
time_path(a, "path A again:")
time_path(b, "path B again:")
sa, sb = a._path.streams, b._path.streams
b._path.streams = sa
time_path(b, "path B on A's streams:")
a._path.streams = sb
time_path(a, "path A on B's streams:")
b._path.streams = [torch.cuda.Stream(dev, priority=-1) for _ in range(4)]
time_path(b, "path B on high-priority streams:")
