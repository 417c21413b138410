#!/bin/bash
# variants/libslim_NAME.so = the current build with gramr_k13.hip (cd_gramr_kernel<10,3>) recompiled
# under extra flags, e.g. "-DSLIM_K13_AH=2", "-DSLIM_K13_DMA=0 -DSLIM_GRAMR_SYNCROW=1"; the other
# objects are the ones `make` left in slim_amd/csrc/build.  A/B runs: SLIM_AMD_LIB=variants/libslim_NAME.so
# usage: scripts/build_k13_variant.sh NAME "FLAGS" [save-temps]
set -e
NAME=$1; FLAGS=$2
R=$(cd "$(dirname "$0")/.." && pwd)
S=$R/slim_amd/csrc
W=/tmp/k13_$NAME
rm -rf $W && mkdir -p $W $R/variants
EXTRA=""
[ -n "$3" ] && EXTRA="--save-temps"
(cd $W && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-result --offload-arch=gfx950 \
  -munsafe-fp-atomics $FLAGS $EXTRA -I$S -c -o $W/gramr_k13.o $S/gramr_k13.hip)
OBJS=$(ls $S/build/*.o | grep -v gramr_k13.o)
/opt/rocm/bin/hipcc -O3 -fPIC --offload-arch=gfx950 -shared -o $R/variants/libslim_$NAME.so $OBJS $W/gramr_k13.o
ls -la $R/variants/libslim_$NAME.so
