#!/bin/bash
# The attainable floor of the exact k-NN (VERDICT r4 #6b), measured: variant builds of knn.hip alone - whole call, no selection
# (distance phase + hand-over), no distance phase (hand-over + selection), table preparation alone - timed per call by tools/knn_lab.py
# at the bench's shape (B = 32, N = 1024, k = 20; C = 3 / 62 / 127 = the four graphs of a step, C = 62 twice).
# Run ON the GPU box: bash tools/knn_floor.sh > gpurun_out/r05_knn_floor.txt
set -e
cd "$(dirname "$0")/.."
bash tools/knn_lab_build.sh nosel -DSVNET_KNN_ABL=1 > /dev/null
bash tools/knn_lab_build.sh nodist -DSVNET_KNN_ABL=4 > /dev/null
bash tools/knn_lab_build.sh preponly -DSVNET_KNN_ABL=5 > /dev/null
python tools/knn_lab.py nosel=_ab/libknn_nosel.so nodist=_ab/libknn_nodist.so preponly=_ab/libknn_preponly.so
