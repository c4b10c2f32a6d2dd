#!/bin/bash
# Round-6 refresh of the single-GPU legs of BASELINE.json configs[3] / [4], the reference stack's own kernels on the same box, and batch 1
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/configs
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 600 python3 bench.py --mxfp8 --steps 20 --warmup 5 > $OUT/bench_mxfp8.json 2> $OUT/bench_mxfp8.err; echo "mxfp8 rc $?"; head -c 400 $OUT/bench_mxfp8.json; echo
timeout 600 python3 bench.py --steps 20 --warmup 5 --no-e2e > $OUT/bench_bf16_same_box.json 2> $OUT/bench_bf16.err; echo "bf16 rc $?"; head -c 300 $OUT/bench_bf16_same_box.json; echo
timeout 600 python3 tools/torch_rocm_baseline.py 16 5 > $OUT/torch_rocm_baseline.json 2> $OUT/torch_rocm_baseline.err; echo "torch rc $?"; cat $OUT/torch_rocm_baseline.json
timeout 900 python3 tools/train_bench.py > $OUT/train_step.json 2> $OUT/train_step.err; echo "train rc $?"; tail -1 $OUT/train_step.json
timeout 300 python3 bench.py --batch 1 --steps 50 --warmup 10 --no-cpu-baseline --no-e2e > $OUT/bench_batch1.json 2> $OUT/bench_batch1.err; echo "b1 rc $?"; head -c 300 $OUT/bench_batch1.json; echo
