#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/power
rocm-smi --showpower --showclocks --showmaxpower --json > gpurun_out/power/smi_idle.json 2>&1; head -c 800 gpurun_out/power/smi_idle.json; echo
timeout 600 python3 tools/probes/power_clock_probe.py > gpurun_out/power/power_clock.json 2> gpurun_out/power/power_clock.err; echo "power rc $?"; tail -8 gpurun_out/power/power_clock.err
timeout 600 ./build/gemm_energy_bound 5 0.3 > gpurun_out/power/geb3.json 2> gpurun_out/power/geb3.err; echo "geb rc $?"
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/power/geb3.json"))
for sh in d["shapes"]:
    m=sh["modes"][0]
    print("%-60s %8.2f us clk %.3f kloop %.3f epi %.3f kshare %.3f"%(sh["shape"][:60],m["us_per_launch_median"],m["clock_ghz"],m["clock_ghz_kloop"],m["clock_ghz_epilogue"],m["kloop_share_of_block_life"]))
PY
