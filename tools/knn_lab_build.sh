#!/bin/bash
# Variant builds of knn.hip alone for tools/knn_lab.py: tools/knn_lab_build.sh NAME [-DFLAG ...]  ->  _ab/libknn_NAME.so
set -e
cd "$(dirname "$0")/../svnet_amd/csrc"
name=$1; shift
mkdir -p ../../_ab/_knn_$name
F="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $F -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 "$@" -c knn.hip -o ../../_ab/_knn_$name/knn.o
/opt/rocm/bin/hipcc $F -c error.cpp -o ../../_ab/_knn_$name/error.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../_ab/libknn_$name.so ../../_ab/_knn_$name/knn.o ../../_ab/_knn_$name/error.o
echo built _ab/libknn_$name.so
