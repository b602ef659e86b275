#!/bin/bash
# Builds scratch/libcoflux_<tag>.so from ALL sources with extra -D flags (table shape, chunk capacity, waves per SIMD …).
# usage: make_full_variant.sh <tag> <-Dflags...>
set -e
TAG=$1; shift
cd "$(dirname "$0")/../climaocean.jl_amd/csrc"
O=/tmp/_variant_$TAG; mkdir -p $O
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -Wno-unused-value"
pids=""
for f in coflux_interp coflux_solver coflux_solver_lean coflux_solver_libm coflux_net coflux_halo; do
  hipcc $FL "$@" -c $f.hip -o $O/$f.o & pids="$pids $!"
done
for f in coflux_abi coflux_window coflux_steps; do
  hipcc $FL "$@" -x hip -c $f.cpp -o $O/$f.o & pids="$pids $!"
done
g++ -O2 -std=c++17 -fPIC "$@" -c coflux_tables.cpp -o $O/coflux_tables.o
for p in $pids; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scratch/libcoflux_$TAG.so $O/*.o -ldl
echo built scratch/libcoflux_$TAG.so
