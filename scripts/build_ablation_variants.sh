#!/bin/bash
# Builds ab/libexp_<mask>.so: the library with parts of the level-1 hot loop compiled out (CM2_EXP bit mask in
# knn_phase1_f16.h: 1 no DMA pieces, 2 no barrier, 4 no fragment reads, 8 no min trees) for scripts/ablate_skeleton.sh.
# Run on the build host (hipcc cross-compiles); needs an up-to-date cellmapper_amd/csrc/build/ (make -C cellmapper_amd/csrc).
set -e
cd "$(dirname "$0")/../cellmapper_amd/csrc"
mkdir -p ../../ab /tmp/cm_exp
for x in 0 1 2 4 8 15; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I. -I../../include -DCM2_EXP=$x -c knn_phase1_f16_inst.hip -o /tmp/cm_exp/inst_$x.o
  hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--no-undefined -o ../../ab/libexp_$x.so build/cm_api.o build/knn_kernels.o build/graph_kernels.o \
        build/jaccard_kernels.o build/grid_knn.o build/rank_kernels.o build/embed_kernels.o /tmp/cm_exp/inst_$x.o build/knn_phase1_direct_inst.o build/knn_phase1_metric_inst.o build/knn_phase1_direct2_inst.o build/p1_g0.o build/p1_g1.o build/p1_g2.o build/p1_g3.o
done
