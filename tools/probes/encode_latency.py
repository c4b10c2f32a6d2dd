import sys, time, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import bench
from merv_amd.vidlm import MERVVisual
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
specs, bbs, path, extras = bench.build_models(dev)
m = MERVVisual(bbs, llm_dim=4096)
for dt in (torch.float32, torch.bfloat16):
    vv = [torch.randn(s.pixel_shape(1), device=dev).to(dt) for s in specs]
    for mode in (False, True):
        m.graph_replay = mode
        for _ in range(3): m.encode(vv)
        torch.cuda.synchronize()
        ts = []
        for _ in range(10):
            torch.cuda.synchronize(); t0 = time.perf_counter(); m.encode(vv); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        print(dt, "graph" if mode else "eager", "min %.2f ms median %.2f ms" % (min(ts) * 1e3, sorted(ts)[5] * 1e3))
