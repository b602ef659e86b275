#!/bin/bash
# Builds scratch/libcoflux_<tag>.so with extra -D flags for the lean ocean kernel's translation unit (A/B experiments:
# python scratch/ab_libs.py <tag> ...).  usage: make_lean_variant.sh <tag> <-Dflags...>
set -e
TAG=$1; shift
cd "$(dirname "$0")/../climaocean.jl_amd/csrc"
make -s libcoflux.so > /dev/null
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on "$@" -c coflux_solver_lean.hip -o /tmp/_lean_$TAG.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scratch/libcoflux_$TAG.so coflux_interp.o coflux_solver.o /tmp/_lean_$TAG.o coflux_solver_libm.o coflux_net.o coflux_halo.o coflux_abi.o coflux_window.o coflux_steps.o coflux_tables.o -ldl
echo built scratch/libcoflux_$TAG.so
