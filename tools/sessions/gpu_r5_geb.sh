#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 600 ./build/gemm_energy_bound 5 0.3 > gpurun_out/geb2.json 2> gpurun_out/geb2.err; echo "rc $?"
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/geb2.json"))
for sh in d["shapes"]:
    print(sh["shape"])
    for m in sh["modes"]:
        print("   %-60s %8.2f us clk %.3f x%.4f"%(m["mode"][:60],m["us_per_launch_median"],m["clock_ghz"],m["speedup_vs_product"]))
PY
