#!/bin/bash
# dev tool: the instrumented library of the WORKING TREE (-DPHC_SIM_PROFILE: phase profile, ablation switches, single-wavefront timeline) -> phc_amd/_obj/libphc_amd_prof.so
set -e
cd "$(dirname "$0")/../../phc_amd/csrc"
O=../_obj; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment -fno-slp-vectorize -ffp-contract=off -DPHC_SIM_PROFILE -c phc_kernels.hip -o $O/kp.o
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment -ffast-math -fno-slp-vectorize -DPHC_SIM_PROFILE -c phc_sim.hip -o $O/sp.o
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment -ffp-contract=off -c phc_learn.hip -o $O/lp.o
hipcc --offload-arch=gfx950 -shared -fPIC $O/kp.o $O/sp.o $O/lp.o -o $O/libphc_amd_prof.so
ls -la $O/libphc_amd_prof.so
