#!/bin/bash
# Builds scratch/libcoflux_leanstamp.so: the production library with the lean ocean kernel's per-wave time stamps
# compiled in (scratch/phases_lean.py reads them through cf_debug_phase_read).
set -e
cd "$(dirname "$0")/../climaocean.jl_amd/csrc"
make -s libcoflux.so > /dev/null
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -DCF_LEAN_STAMPS "$@" -c coflux_solver_lean.hip -o /tmp/_solver_lean_stamp.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scratch/libcoflux_leanstamp.so coflux_interp.o coflux_solver.o /tmp/_solver_lean_stamp.o coflux_solver_slab.o coflux_solver_libm.o coflux_net.o coflux_halo.o coflux_abi.o coflux_window.o coflux_steps.o coflux_tables.o -ldl
echo built scratch/libcoflux_leanstamp.so
