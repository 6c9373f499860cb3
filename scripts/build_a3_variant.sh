#!/bin/bash
# build_a3_variant.sh NAME "-DCM_A3_EXP=.. ..."  ->  build_ab/lib_NAME.so: the library with graph_kernels.hip (A3 and the
# other graph kernels) compiled under extra defines; every other object is linked from csrc/build (make first).
set -e
name=$1; flags=$2
cd "$(dirname "$0")/../cellmapper_amd/csrc"
mkdir -p ../../build_ab /tmp/cm_a3var
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -I. -I../../include $flags -c graph_kernels.hip -o /tmp/cm_a3var/gk_$name.o
objs=$(ls build/*.o | grep -v graph_kernels.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--no-undefined -o ../../build_ab/lib_$name.so $objs /tmp/cm_a3var/gk_$name.o
echo built build_ab/lib_$name.so
