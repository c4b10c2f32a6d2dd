#!/bin/bash
# Round-end GPU run: the whole -m gpu suite, the forced-distributed legs on one rank, smoke, the profiling round, attention PMC, batch 1.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/final
python3 -m pytest tests -m gpu -x -q > gpurun_out/final/pytest.log 2>&1; tail -3 gpurun_out/final/pytest.log
MERV_BENCH_FORCE_DISTRIBUTED=1 python3 bench.py --steps 2 --warmup 1 --batch 4 > gpurun_out/final/forcedist_dp.json 2> gpurun_out/final/forcedist_dp.err; tail -2 gpurun_out/final/forcedist_dp.err | cut -c1-400
MERV_BENCH_FORCE_DISTRIBUTED=1 python3 bench.py --steps 2 --warmup 1 --batch 4 --parallelism units > gpurun_out/final/forcedist_units.json 2> gpurun_out/final/forcedist_units.err; echo "units rc=$?"; tail -1 gpurun_out/final/forcedist_units.err | cut -c1-400
python3 __graft_entry__.py smoke 2>&1 | tail -2
bash tools/gpu_profile_round.sh > gpurun_out/round_stdout.log 2>&1; grep -E "^gemm|^attention|^temporal|^Layer|^pool|^data" gpurun_out/round_stdout.log
bash tools/pmc_attn.sh > gpurun_out/pmc_attn_stdout.log 2>&1; grep "dispatches" gpurun_out/pmc_attn_stdout.log | cut -c1-500
python3 bench.py --batch 1 --no-cpu-baseline --no-e2e > gpurun_out/final/bench_b1.json 2>/dev/null; cut -c1-300 gpurun_out/final/bench_b1.json
