#!/bin/bash
# usage (here, repo root): tools/build_timeline_lib.sh  -> 4dgaussians_amd/libgs4d_tl.so (kernel experiments only:
# blend_bwd.hip, blend_fwd.hip and binning.hip compiled with -DGS4D_TIMELINE, everything else the regular objects; load with GS4D_LIB_NAME=libgs4d_tl.so)
set -e
cd 4dgaussians_amd/csrc
make -s
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wall -Wno-unused-function -I../../include -DGS4D_TIMELINE $TL_EXTRA"
/opt/rocm/bin/hipcc $F -c blend_bwd.hip -o /tmp/blend_bwd_tl.o
/opt/rocm/bin/hipcc $F -c blend_fwd.hip -o /tmp/blend_fwd_tl.o
/opt/rocm/bin/hipcc $F -ffp-contract=off -c binning.hip -o /tmp/binning_tl.o
objs=$(ls *.o | grep -v '^blend_bwd.o$' | grep -v '^blend_fwd.o$' | grep -v '^binning.o$')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libgs4d_tl.so $objs /tmp/blend_bwd_tl.o /tmp/blend_fwd_tl.o /tmp/binning_tl.o
ls -la ../libgs4d_tl.so
