#!/bin/bash
# Development build with phase clocks in the scatter kernels: abtmp/libparticles_phase.so (see phase_clocks.py)
set -e
mkdir -p abtmp build/phase
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fPIC -std=c++17 -Iinclude -DPCL_PHASE_CLOCKS -c -o build/phase/particles_hip.o particles_amd/csrc/particles_hip.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o abtmp/libparticles_phase.so build/phase/particles_hip.o build/pair_sort.o build/particles_comm.o build/authoring.o -ldl
