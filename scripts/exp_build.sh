#!/bin/bash
# Build experimental variants of the library next to the product one: fre-nctools_amd/libfregrid_hip_exp<N>.so with
# xgrid_kernels.hip compiled -DFG_EXP=<N> (timing experiments only; selected with FREGRID_HIP_LIB).
# usage: scripts/exp_build.sh N [file.hip] [extra compiler flags, e.g. -DCLIP_COMPACT=0]     (default file: xgrid_kernels.hip)
set -e
N=$1; F=${2:-xgrid_kernels.hip}; B=${F%.hip}; shift; shift || true; X="$@"
cd "$(dirname "$0")/../fre-nctools_amd/csrc"
make -j6 >/dev/null
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -std=c++17 -Wall -Wno-unused-function -I../../include -I. -DFG_EXP=$N $X -c $F -o /tmp/${B}_exp$N.o
OBJS=$(ls *.o | grep -v "^$B.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic-functions -o ../libfregrid_hip_exp$N.so /tmp/${B}_exp$N.o $OBJS -lm -lpthread
echo built ../libfregrid_hip_exp$N.so
