#!/bin/bash
# build_ab_src.sh NAME FILE [hipcc flags...]: a second copy of the library with merv_amd/csrc/FILE.hip compiled under the given flags
# -> ab/libmerv_hip_NAME.so (same-box A/B pairs: MERV_HIP_LIB selects the library inside one GPU session). FILE = gemm | attention | decode | ...
set -e
cd "$(dirname "$0")/../.."
name=$1; file=$2; shift 2
mkdir -p ab
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Iinclude "$@" -Imerv_amd/csrc -c merv_amd/csrc/$file.hip -o ab/${file}_$name.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ab/libmerv_hip_$name.so ab/${file}_$name.o $(ls merv_amd/csrc/*.o | grep -v "/$file.o")
echo ab/libmerv_hip_$name.so
