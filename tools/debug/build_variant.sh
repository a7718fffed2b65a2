#!/bin/bash
# Development build of the library with extra compiler flags: bash tools/debug/build_variant.sh NAME -DFLAG ...
# -> abtmp/libparticles_NAME.so (used through PCL_DEBUG_LIB by the tools in this directory)
set -e
name=$1; shift
mkdir -p abtmp build/$name
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fPIC -std=c++17 -Iinclude "$@" -c -o build/$name/particles_hip.o particles_amd/csrc/particles_hip.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o abtmp/libparticles_$name.so build/$name/particles_hip.o build/pair_sort.o build/particles_comm.o build/authoring.o -ldl
