#!/bin/bash
# End-of-step measurement set: bench lines, rocprofv3 kernel stats of the same command, PMC traffic passes.
# usage: round_profile.sh <tag>     (run on the GPU box; results under gpurun_out/<tag>/, copy what is cited into profiles/)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TAG=${1:-r02a}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
python bench.py > $OUT/bench.json 2> $OUT/bench.err
python bench.py --steps 20 --no-cpu-baseline > $OUT/bench_steps20.json 2>> $OUT/bench.err
python bench.py --flux-configuration corrected --no-cpu-baseline > $OUT/bench_corrected.json 2>> $OUT/bench.err
python bench.py --flux-configuration ncar --no-cpu-baseline > $OUT/bench_ncar.json 2>> $OUT/bench.err
python bench.py --ny 70 --no-cpu-baseline > $OUT/bench_slab70.json 2>> $OUT/bench.err
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_profiled.json 2> $OUT/kt.log)
cp $(find $OUT/kt -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && rocprofv3 --pmc $c -d $OUT/pmc_$c -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --repetitions 1 --no-cpu-baseline > $OUT/pmc_$c.log 2>&1)
done
python - <<PY
import csv,glob,collections,json
res=collections.defaultdict(dict)
for name in ("FETCH_SIZE","WRITE_SIZE"):
    for f in glob.glob("$OUT/pmc_%s/**/*counter_collection.csv"%name, recursive=True):
        agg=collections.defaultdict(float); cnt=collections.Counter()
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"]==name:
                k=r["Kernel_Name"].split("(")[0].replace("void coflux::","").replace("coflux::","")
                agg[k]+=float(r["Counter_Value"]); cnt[k]+=1
        for k in agg: res[k][name]=agg[k]/cnt[k]; res[k]["launches"]=cnt[k]
open("$OUT/pmc_traffic_raw.json","w").write(json.dumps(res,indent=1))
print(json.dumps(res,indent=1))
PY
rm -rf $OUT/kt $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
for f in bench bench_steps20 bench_corrected bench_ncar bench_slab70 bench_profiled; do python -c "
import json,sys
d=json.load(open('$OUT/$f.json')); r=d['roofline']
print('$f', 'ms/step %.4f'%d['ms_per_step'], 'value %.3e'%d['value'], 'ao %.4f frac %.4f nohint %s'%(r['avg_launch_ms'], r['frac'], r.get('avg_launch_ms_without_hints')), 'cpu', (d.get('cpu_baseline') or {}).get('value'))"; done
head -8 $OUT/kernel_stats.csv | cut -c1-160
