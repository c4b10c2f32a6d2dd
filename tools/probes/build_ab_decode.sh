#!/bin/bash
# build_ab_decode.sh NAME [flags...]: library copy with decode.hip compiled under the given flags -> ab/libmerv_hip_NAME.so
set -e
cd "$(dirname "$0")/../.."
name=$1; shift
mkdir -p ab
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Iinclude -Imerv_amd/csrc "$@" -c merv_amd/csrc/decode.hip -o ab/decode_$name.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ab/libmerv_hip_$name.so ab/decode_$name.o $(ls merv_amd/csrc/*.o | grep -v '/decode.o')
echo ab/libmerv_hip_$name.so
