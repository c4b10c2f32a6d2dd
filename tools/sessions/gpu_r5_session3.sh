#!/bin/bash
# Round-5 GPU session 3: attention score chains seeded by one MFMA step + scalar-base K / V pieces: parity tests, same-box A/B against the round-4 library.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s3
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 900 python3 -m pytest tests/test_kernels_gpu.py tests/test_encoder_gpu.py tests/test_fulldepth_parity_gpu.py tests/test_goldens_gpu.py tests/test_fullsize_gpu.py tests/test_outlier_statistics_gpu.py -q -k "atten or parity or golden or fullsize or outlier or encoder" > $OUT/tests.log 2>&1; echo "pytest rc $?" >> $OUT/tests.log; tail -5 $OUT/tests.log
bash tools/probes/ab_attn.sh ab/libmerv_hip_r4.so merv_amd/lib/libmerv_hip.so > $OUT/ab_attn.txt 2>&1; cat $OUT/ab_attn.txt
for rep in 1 2; do for lib in ab/libmerv_hip_r4.so merv_amd/lib/libmerv_hip.so; do
  MERV_HIP_LIB=$R/$lib timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e > $OUT/bench_$(basename $lib .so)_$rep.json 2> $OUT/bench_$(basename $lib .so)_$rep.err
  python3 -c "
import json,sys
d=json.loads(open('$OUT/bench_$(basename $lib .so)_$rep.json').read().strip().splitlines()[-1])
print('$lib rep $rep', d['ms_per_step'], d['value'], d['roofline']['frac'], [(k['name'][:22], k['ms_per_step']) for k in d['roofline']['by_kernel'] if 'attention' in k['name']])
"
done; done
