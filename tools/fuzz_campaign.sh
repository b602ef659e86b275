#!/bin/bash
# Round-end fuzz campaign: the GPU fuzzers of tools/fuzz/ on fresh seeds ($1 = first seed, $2 = output log).
export COFLUX_EXPERIMENTS=1   # (the fuzzers draw chunk plans: CF_OPT_AO_CHUNK is an experiment option)
seed=${1:-61}
log=${2:-gpurun_out/r06_fuzz.log}
mkdir -p $(dirname $log)
: > $log
run() { echo "== $*" >> $log; timeout 900 python "$@" >> $log 2>&1; echo "rc=$?" >> $log; }
run tools/fuzz/fuzz_geometry.py $seed ${CASES:-60}
run tools/fuzz/fuzz_interp_ice.py $((seed+1)) 40
run tools/fuzz/fuzz_stale_mask.py $((seed+2)) 30
run tools/fuzz/fuzz_ice_geometry.py $((seed+3)) 20
run tools/fuzz/fuzz_steps.py $((seed+4)) 30
run tools/fuzz/fuzz_ice_steps.py $((seed+5)) 20
run tools/fuzz/fuzz_certified.py $((seed+6)) 30
run tools/fuzz/fuzz_line.py $((seed+7)) 60
run tools/fuzz/fuzz_halo_in_launch.py $((seed+8)) 60
grep -E "^==|^rc=|bad|FAIL|Error|error|cases" $log | tail -60
