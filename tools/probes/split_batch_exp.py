# experiment: two half-batches per encoder on 8 streams vs one batch per encoder on 4 streams
import sys, time
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[2]))
import torch
import bench
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
specs, path = bench.build_path(dev, concurrent=True)
pix = bench.synth_pixels(specs, 8, dev, seed=0)
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("4 streams, B=8:", timeit(lambda: path.forward(pix)))
# 8 streams: second path object sharing nothing (weights duplicated: fine for an experiment)
specs2, path2 = bench.build_path(dev, concurrent=True)
pa = [p[:4].contiguous() for p in pix]; pb = [p[4:].contiguous() for p in pix]
s2 = torch.cuda.Stream(dev)
def two():
    ev = torch.cuda.Event(); ev.record(torch.cuda.current_stream(dev))
    a = path.forward(pa)
    s2.wait_event(ev)
    with torch.cuda.stream(s2):
        b = path2.forward(pb)
    ev2 = torch.cuda.Event(); ev2.record(s2); torch.cuda.current_stream(dev).wait_event(ev2)
    return a, b
print("8 streams, 2 x B=4:", timeit(two))
