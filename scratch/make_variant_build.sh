#!/bin/bash
# Builds scratch/libcoflux_<tag>.so from the production sources with extra -D flags for the solver translation unit
# (A/B experiments: LIBCOFLUX=scratch/libcoflux_<tag>.so python scratch/ab.py).  usage: make_variant_build.sh <tag> <-Dflags...>
set -e
TAG=$1; shift
cd "$(dirname "$0")/../climaocean.jl_amd/csrc"
make -s libcoflux.so > /dev/null
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on "$@" -c coflux_solver.hip -o /tmp/_solver_$TAG.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scratch/libcoflux_$TAG.so coflux_interp.o /tmp/_solver_$TAG.o coflux_solver_lean.o coflux_solver_libm.o coflux_net.o coflux_halo.o coflux_abi.o coflux_window.o coflux_steps.o coflux_tables.o -ldl
echo built scratch/libcoflux_$TAG.so
