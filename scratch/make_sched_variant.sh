#!/bin/bash
# Builds scratch/libcoflux_<tag>.so whose lean-solver translation unit went through tools/gcn_sched.py:
# hipcc -S (device) → gcn_sched.py → assemble → lld → bundle → host compile with that device image.
# usage: make_sched_variant.sh <tag> [gcn_sched options…]
set -e
TAG=$1; shift
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd $ROOT/climaocean.jl_amd/csrc
make -s libcoflux.so > /dev/null
LL=/opt/rocm/lib/llvm/bin
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -Wno-unused-value"
T=/tmp/_sched_$TAG; mkdir -p $T
hipcc $FL -S --cuda-device-only -o $T/lean.s coflux_solver_lean.hip
FN=$(grep -o "^_ZN6coflux14ao_lean_kernelILb[01]ELi256E[A-Za-z0-9]*EEvNS_8LeanArgsE" $T/lean.s | sort -u | sed 's/^/--function /')
python3 $ROOT/climaocean.jl_amd/csrc/tools/gcn_sched.py $T/lean.s $T/lean_sched.s $FN --report "$@" 2> $T/report.txt
$LL/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $T/lean_sched.s -o $T/lean_dev.o
$LL/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $T/lean.hsaco $T/lean_dev.o
$LL/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$T/lean.hsaco -output=$T/lean.hipfb
hipcc $FL --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $T/lean.hipfb -c coflux_solver_lean.hip -o $T/coflux_solver_lean.o 2>/dev/null
hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/scratch/libcoflux_$TAG.so coflux_interp.o coflux_solver.o $T/coflux_solver_lean.o coflux_solver_libm.o coflux_net.o coflux_halo.o coflux_abi.o coflux_window.o coflux_steps.o coflux_tables.o -ldl
echo built scratch/libcoflux_$TAG.so; grep -c "lines" $T/report.txt
