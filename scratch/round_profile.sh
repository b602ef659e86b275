#!/bin/bash
# End-of-step measurement set: bench lines, rocprofv3 kernel stats of the same command, PMC traffic passes.
# usage: round_profile.sh <tag>
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TAG=${1:-r01d}; OUT=gpurun_out/$TAG; mkdir -p $OUT
python bench.py > $OUT/bench.json 2> $OUT/bench.err
python bench.py --flux-configuration corrected --no-cpu-baseline > $OUT/bench_corrected.json 2>> $OUT/bench.err
python bench.py --flux-configuration ncar --no-cpu-baseline > $OUT/bench_ncar.json 2>> $OUT/bench.err
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt --output-format csv -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline > $OUT/bench_profiled.json 2> $OUT/kt.log
cp $(find $OUT/kt -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
bash scratch/pmc_traffic.sh > $OUT/pmc_traffic.log 2>&1
cp gpurun_out/pmc_traffic/summary.json $OUT/pmc_traffic_raw.json
cat $OUT/bench.json $OUT/bench_corrected.json $OUT/bench_ncar.json $OUT/bench_profiled.json
head -8 $OUT/kernel_stats.csv
cat $OUT/pmc_traffic_raw.json
