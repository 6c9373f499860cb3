#!/bin/bash
# build_f16_variant.sh NAME "-D..."  ->  build_ab/lib_NAME.so: only knn_phase1_f16_inst.hip recompiled under extra defines.
set -e
name=$1; flags=$2
cd "$(dirname "$0")/../cellmapper_amd/csrc"
mkdir -p ../../build_ab /tmp/cm_f16var
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -I. -I../../include $flags -c knn_phase1_f16_inst.hip -o /tmp/cm_f16var/f16_$name.o
objs=$(ls build/*.o | grep -v knn_phase1_f16_inst.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--no-undefined -o ../../build_ab/lib_$name.so $objs /tmp/cm_f16var/f16_$name.o
echo built build_ab/lib_$name.so
