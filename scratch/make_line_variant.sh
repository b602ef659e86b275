#!/bin/bash
# Builds scratch/libcoflux_<tag>.so whose slab translation unit (coflux_solver_slab.hip) went through tools/gcn_sched.py with
# other options than the Makefile's (A/B of the scheduler's model): make_line_variant.sh <tag> [--plain | gcn_sched options…]
set -e
TAG=$1; shift
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd $ROOT/climaocean.jl_amd/csrc
make -s libcoflux.so > /dev/null
LL=/opt/rocm/lib/llvm/bin
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -Wno-unused-value"
T=/tmp/_line_$TAG; mkdir -p $T
DEV=coflux_solver_slab.dev.s
if [ "${1#-D}" != "$1" ]; then   # a -D flag first: the device assembly is recompiled with it
  hipcc $FL $1 -S --cuda-device-only -o $T/dev.s coflux_solver_slab.hip 2>/dev/null; DEV=$T/dev.s; shift
fi
if [ "$1" = "--plain" ]; then
  hipcc $FL -c coflux_solver_slab.hip -o $T/coflux_solver_slab.o 2>/dev/null
else
  python3 tools/gcn_sched.py $DEV $T/sched.s --function ao_lean_line_kernel --all-matching --vgprs 256 --report "$@" 2> $T/report.txt
  $LL/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $T/sched.s -o $T/dev.o
  $LL/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $T/slab.hsaco $T/dev.o
  $LL/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$T/slab.hsaco -output=$T/slab.hipfb
  hipcc $FL --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $T/slab.hipfb -c coflux_solver_slab.hip -o $T/coflux_solver_slab.o 2>/dev/null
fi
hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/scratch/libcoflux_$TAG.so coflux_interp.o coflux_solver.o coflux_solver_lean.o $T/coflux_solver_slab.o coflux_solver_libm.o coflux_net.o coflux_halo.o coflux_abi.o coflux_window.o coflux_steps.o coflux_tables.o -ldl
echo built scratch/libcoflux_$TAG.so
