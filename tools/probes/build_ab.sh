#!/bin/bash
# build_ab.sh NAME [hipcc flags...]: a second copy of the library with gemm.hip compiled under the given flags -> ab/libmerv_hip_NAME.so
# (same-box A/B pairs and ablations: MERV_HIP_LIB selects the library inside one GPU session)
set -e
cd "$(dirname "$0")/../.."
name=$1; shift
mkdir -p ab
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Iinclude "$@" -Imerv_amd/csrc -c ${SRC:-merv_amd/csrc/gemm.hip} -o ab/gemm_$name.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ab/libmerv_hip_$name.so ab/gemm_$name.o $(ls merv_amd/csrc/*.o | grep -v '/gemm.o')
echo ab/libmerv_hip_$name.so
