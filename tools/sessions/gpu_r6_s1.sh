#!/bin/bash
# Round 6, session 1: the hooks / no-hooks builds on the GPU (tests that force kernel forms, smoke), same-box pair against the round-5 library,
# small-batch diagnostics (chains alone / together, tile configurations at one video, per-queue kernel timeline at one video).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s1
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 1500 python3 -m pytest tests/test_gemm_variants_gpu.py tests/test_kernels_gpu.py tests/test_decode_gpu.py tests/test_sampler_gpu.py -m gpu -q -x > $OUT/tests.log 2>&1; echo "pytest rc $?" >> $OUT/tests.log; tail -4 $OUT/tests.log
timeout 300 python3 __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc $?" >> $OUT/smoke.log; tail -2 $OUT/smoke.log
export MERV_HIP_LIB_AB=1
for rep in 1 2; do for lib in ab/libmerv_hip_r5.so merv_amd/lib/libmerv_hip.so; do
  MERV_HIP_LIB=$R/$lib timeout 400 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e 2>$OUT/ab_err.log | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('rep $rep $lib: ms_per_step', d['ms_per_step'], 'gemm frac', r['frac'], 'gemm_ms', r['gemm_ms_per_step'])
" | tee -a $OUT/ab16.txt
done; done
unset MERV_HIP_LIB_AB
timeout 600 python3 tools/probes/batch_chains.py 1 2 4 > $OUT/batch_chains.txt 2>$OUT/batch_chains.err; tail -1 $OUT/batch_chains.txt > $OUT/batch_chains.json; head -3 $OUT/batch_chains.txt
timeout 600 python3 tools/gemm_bench.py 1 0,1,4,6,7,9 > $OUT/gemm_bench_b1.txt 2>&1; cat $OUT/gemm_bench_b1.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_b1 -- python3 $R/bench.py --batch 1 --steps 30 --warmup 10 --no-cpu-baseline --no-e2e --no-prof > $OUT/trace_b1.json 2> $OUT/trace_b1.err
cd $R
f=$(find $OUT/trace_b1 -name "*kernel_trace.csv" | head -1)
python3 tools/probes/timeline_chains.py $f 30 > $OUT/timeline_b1.txt 2>&1; head -80 $OUT/timeline_b1.txt
python3 tools/probes/timeline.py $f > $OUT/timeline_b1_overlap.txt 2>&1; head -20 $OUT/timeline_b1_overlap.txt
rm -rf $OUT/trace_b1
