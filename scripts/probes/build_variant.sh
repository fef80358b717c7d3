#!/bin/bash
# dev tool: build the library of another git revision next to the working tree's (same-box A/B runs: PHC_AMD_LIB=phc_amd/_obj/libphc_amd_<tag>.so)
#   bash scripts/probes/build_variant.sh <rev> <tag>
set -e
rev=$1; tag=$2; tmp=$(mktemp -d)
git archive $rev phc_amd/csrc include | tar -x -C $tmp
cd $tmp/phc_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment -fno-slp-vectorize -ffp-contract=off -c phc_kernels.hip -o k.o
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment -ffast-math -fno-slp-vectorize -c phc_sim.hip -o s.o
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment -ffp-contract=off -c phc_learn.hip -o l.o
hipcc --offload-arch=gfx950 -shared -fPIC k.o s.o l.o -o $OLDPWD/phc_amd/_obj/libphc_amd_$tag.so
cd $OLDPWD; rm -rf $tmp; ls -la phc_amd/_obj/libphc_amd_$tag.so
