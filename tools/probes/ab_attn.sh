#!/bin/bash
# ab_attn.sh LIB...: tools/attn_bench.py 16 per library, interleaved twice
for rep in 1 2; do for lib in "$@"; do echo "== rep $rep $lib"; MERV_HIP_LIB=$PWD/$lib python3 tools/attn_bench.py 16 2>&1 | grep "^attn"; done; done
