#!/bin/bash
export MERV_HIP_LIB_AB=1  # tolerant binding for a previous build (merv_amd/_lib.py)
# ab_attn.sh LIB...: tools/attn_bench.py 16 per library, interleaved twice
for rep in 1 2; do for lib in "$@"; do echo "== rep $rep $lib"; MERV_HIP_LIB=$PWD/$lib python3 tools/attn_bench.py 16 2>&1 | grep "^attn"; done; done
