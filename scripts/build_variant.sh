#!/bin/bash
# build_variant.sh NAME "-DFLAG=.. ..."  ->  build_ab/lib_NAME.so: the library with the phase-1 kernel (and its host
# launcher, which sizes the LDS ring) compiled under extra defines; everything else is linked from csrc/build.
set -e
name=$1; flags=$2
cd "$(dirname "$0")/../cellmapper_amd/csrc"
mkdir -p ../../build_ab /tmp/cm_var_$name
CXX="/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -I. -I../../include $flags"
$CXX -c knn_phase1_f16_inst.hip -o /tmp/cm_var_$name/f16.o &
$CXX -c knn_kernels.hip -o /tmp/cm_var_$name/knn.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--no-undefined -o ../../build_ab/lib_$name.so build/cm_api.o /tmp/cm_var_$name/knn.o \
  build/graph_kernels.o build/jaccard_kernels.o build/grid_knn.o build/rank_kernels.o build/embed_kernels.o /tmp/cm_var_$name/f16.o build/knn_phase1_direct_inst.o build/knn_phase1_metric_inst.o build/knn_phase1_direct2_inst.o \
  build/p1_g0.o build/p1_g1.o build/p1_g2.o build/p1_g3.o
echo built build_ab/lib_$name.so
