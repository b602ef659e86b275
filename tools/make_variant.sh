#!/bin/bash
# A/B build: tools/ab/libcoflux_<tag>.so from the production sources with extra -D flags in every translation unit
# (LIBCOFLUX=tools/ab/libcoflux_<tag>.so COFLUX_ALLOW_STALE_LIBRARY=1 python bench.py …).  usage: make_variant.sh <tag> <-Dflags…>
set -e
TAG=$1; shift
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
B=/tmp/coflux_variant_$TAG
rm -rf $B; mkdir -p $B/climaocean.jl_amd $B/include
cp -r $ROOT/climaocean.jl_amd/csrc $B/climaocean.jl_amd/csrc; cp $ROOT/include/coflux.h $B/include/
cd $B/climaocean.jl_amd/csrc; make -s clean
make -s -j8 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -Wno-unused-function -Wno-unused-value $*" libcoflux.so > /dev/null 2>&1
mkdir -p $ROOT/tools/ab; cp libcoflux.so $ROOT/tools/ab/libcoflux_$TAG.so
echo built tools/ab/libcoflux_$TAG.so
