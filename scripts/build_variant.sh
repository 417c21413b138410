#!/bin/bash
# Build variants/libslim_NAME.so from the current sources with extra compiler flags
# (A/B runs of kernel experiments on one GPU box: SLIM_AMD_LIB=variants/libslim_NAME.so).
# usage: scripts/build_variant.sh NAME "-DSOME_EXPERIMENT=1 ..."   (any extra hipcc flags)
set -e
NAME=$1; FLAGS=$2
R=$(cd "$(dirname "$0")/.." && pwd)
W=/tmp/variant_$NAME
rm -rf $W && mkdir -p $W $R/variants
cp -r $R/slim_amd/csrc/. $W/
rm -rf $W/build
sed -i "s#^HIPFLAGS = #HIPFLAGS = $FLAGS #" $W/Makefile
sed -i "s#^OUT      = ../libslim.so#OUT      = $R/variants/libslim_$NAME.so#" $W/Makefile
sed -i "s#../../include#$R/include#g" $W/Makefile $W/engine.hpp $W/host_csr.hpp
make -j8 -C $W $R/variants/libslim_$NAME.so > $W/build.log 2>&1 || { tail -20 $W/build.log; exit 1; }
ls -la $R/variants/libslim_$NAME.so
