"""
cpu_worker.py -- one encoder branch (a4-a9: encoder -> 3davg+linear projector) of the CPU oracle in its own process.
*** TEST INFRASTRUCTURE ONLY *** (bench.py's cpu_baseline leg, through oracle.parity.reference_video_workers; merv_amd/ never runs it).

    python -m oracle.cpu_worker <in.pt> <out.pt> <threads>

Loads {"pix", "cfg", "W", "pw", "pb", "out_size"}, builds its intra-op pool, prints "ready", waits for a line on stdin, runs the branch
(oracle.merv_oracle.encoder_forward + projector_forward: the restatement of merv.py:563-589), saves the projected tokens and prints
"done <seconds>".
"""
import sys
import time

import torch

from . import merv_oracle as O


def main() -> None:
    src, dst, threads = sys.argv[1], sys.argv[2], int(sys.argv[3])
    torch.set_num_threads(threads)
    d = torch.load(src, weights_only=False)
    with torch.no_grad():
        torch.zeros(64, 64) @ torch.zeros(64, 64)  # the pool exists before the clock starts
        print("ready", flush=True)
        sys.stdin.readline()
        t0 = time.perf_counter()
        tok = O.encoder_forward(d["pix"], d["cfg"], d["W"])
        proj = O.projector_forward(tok, d["cfg"].t_out, d["cfg"].hp, d["out_size"], d["pw"], d["pb"])
        secs = time.perf_counter() - t0
        torch.save(proj, dst)
    print(f"done {secs:.4f}", flush=True)


if __name__ == "__main__":
    main()
