#!/bin/bash
# usage (here, repo root): tools/build_variant_lib.sh <suffix> <file.hip> "<extra flags>"  -> 4dgaussians_amd/libgs4d_<suffix>.so
# (kernel experiments only: one translation unit recompiled with extra -D flags, everything else the regular objects;
# load with GS4D_LIB_NAME=libgs4d_<suffix>.so)
set -e
suffix=$1; file=$2; extra=$3
cd 4dgaussians_amd/csrc
make -s
base=${file%.hip}
strict=""
case "$base" in preprocess|geom_bwd|knn|binning|train_step) strict="-ffp-contract=off";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wall -Wno-unused-function -I../../include $strict $extra -c $file -o /tmp/${base}_$suffix.o
objs=$(ls *.o | grep -v "^$base.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libgs4d_$suffix.so $objs /tmp/${base}_$suffix.o
ls -la ../libgs4d_$suffix.so
