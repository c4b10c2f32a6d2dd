#!/bin/bash
# Round-5 GPU session 2: full GPU suite (no -x), parity under injected trained-tower statistics, MLP chunk-locality probe.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s2
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 1500 python3 -m pytest tests -m gpu -q > $OUT/tests.log 2>&1; echo "pytest rc $?" >> $OUT/tests.log; tail -8 $OUT/tests.log
timeout 900 python3 tools/parity_outliers.py > $OUT/parity_outliers.json 2> $OUT/parity_outliers.err; echo "outliers rc $?"; head -c 3000 $OUT/parity_outliers.json; tail -3 $OUT/parity_outliers.err
timeout 600 python3 tools/probes/mlp_chunk_locality.py > $OUT/mlp_chunk_locality.json 2> $OUT/mlp_chunk_locality.err; echo "mlp rc $?"; cat $OUT/mlp_chunk_locality.json
