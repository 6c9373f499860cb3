#!/bin/bash
# build_q4_variant.sh NAME "-DCM2Q_EXP=.. ..."  ->  build_ab/lib_NAME.so: the library with only the four-accumulator filter
# kernel (knn_phase1_q4_inst.hip) compiled under extra defines; every other object is linked from csrc/build (make first).
set -e
name=$1; flags=$2
cd "$(dirname "$0")/../cellmapper_amd/csrc"
mkdir -p ../../build_ab /tmp/cm_q4var
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -I. -I../../include -DCM_ABLATE $flags -c knn_phase1_q4_inst.hip -o /tmp/cm_q4var/q4_$name.o
objs=$(ls build/*.o | grep -v knn_phase1_q4_inst.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--no-undefined -o ../../build_ab/lib_$name.so $objs /tmp/cm_q4var/q4_$name.o
echo built build_ab/lib_$name.so
