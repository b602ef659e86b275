#!/bin/bash
# rocprofv3 kernel stats of one bench.py command line (scratch): profile_cmd.sh <tag> <name> <bench args...>
# → gpurun_out/<tag>/kernel_stats_<name>.csv and bench_profiled_<name>.json
TAG=$1; NAME=$2; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt_$NAME -o kt --output-format csv -- python $GRAFT_REPO_ROOT/bench.py "$@" > $OUT/bench_profiled_$NAME.json 2> $OUT/kt_$NAME.log)
F=$(find /tmp/kt_$NAME -name "*kernel_stats.csv" | head -1)
if [ -n "$F" ]; then cp "$F" $OUT/kernel_stats_$NAME.csv; head -14 "$F" | cut -c1-170; else echo "no kernel_stats.csv"; tail -5 $OUT/kt_$NAME.log; fi
