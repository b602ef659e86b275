#!/bin/bash
# End-of-step measurement set: bench lines, rocprofv3 kernel stats of the same command, PMC traffic passes.
# usage: round_profile.sh <tag>     (run on the GPU box; results under gpurun_out/<tag>/, copy what is cited into profiles/)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TAG=${1:-r02a}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
python bench.py > $OUT/bench.json 2> $OUT/bench.err
python bench.py --steps 20 --warmup 5 > $OUT/bench_steps20.json 2>> $OUT/bench.err
python bench.py --flux-configuration corrected --no-cpu-baseline > $OUT/bench_corrected.json 2>> $OUT/bench.err
python bench.py --flux-configuration ncar --no-cpu-baseline > $OUT/bench_ncar.json 2>> $OUT/bench.err
python bench.py --ny 70 --no-cpu-baseline > $OUT/bench_slab70.json 2>> $OUT/bench.err
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-sorted-pass > $OUT/bench_profiled.json 2> $OUT/kt.log)
cp $(find $OUT/kt -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && rocprofv3 --pmc $c -d $OUT/pmc_$c -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --repetitions 1 --no-cpu-baseline --no-sorted-pass > $OUT/pmc_$c.log 2>&1)
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
# FP64-issue side: VALU instructions and busy cycles per launch (one SQ pass, counters only)
(cd /tmp && rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY -d $OUT/pmc_SQ -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --repetitions 1 --no-cpu-baseline --no-sorted-pass > $OUT/pmc_SQ.log 2>&1)
(cd /tmp && rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $OUT/pmc_SQ2 -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --repetitions 1 --no-cpu-baseline --no-sorted-pass > $OUT/pmc_SQ2.log 2>&1)
python - <<PY
import csv,glob,collections,json
res=collections.defaultdict(dict)
for d in ("pmc_SQ","pmc_SQ2"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv"%d, recursive=True):
        agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(collections.Counter)
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0].replace("void coflux::","").replace("coflux::","")
            agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[k][r["Counter_Name"]]+=1
        for k in agg:
            for c in agg[k]: res[k][c]=agg[k][c]/cnt[k][c]
            res[k]["launches"]=max(cnt[k].values())
open("$OUT/pmc_sq_raw.json","w").write(json.dumps(res,indent=1))
print(json.dumps({k:v for k,v in res.items() if "ao_" in k or "interp" in k or "net_" in k},indent=1))
PY
rm -rf $OUT/kt $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_SQ $OUT/pmc_SQ2
python tools/slab_curve.py $TAG > $OUT/slab_curve.log 2>&1
python bench.py --config sea_ice --no-cpu-baseline > $OUT/bench_sea_ice.json 2>> $OUT/bench.err
python bench.py --config sea_ice --ice-free-cells zero --no-cpu-baseline > $OUT/bench_sea_ice_zero.json 2>> $OUT/bench.err
python bench.py --solver-path certified --no-cpu-baseline > $OUT/bench_certified.json 2>> $OUT/bench.err
python bench.py --solver-path certified --flux-configuration corrected --no-cpu-baseline > $OUT/bench_certified_corrected.json 2>> $OUT/bench.err
python bench.py --grid tripolar --nx 2160 --ny 1080 --flux-configuration corrected --no-cpu-baseline --no-sorted-pass > $OUT/bench_tripolar_2160x1080.json 2>> $OUT/bench.err
python bench.py --grid tripolar --nx 360 --ny 180 --flux-configuration corrected --no-cpu-baseline --no-sorted-pass > $OUT/bench_tripolar_360x180.json 2>> $OUT/bench.err
python bench.py --grid tripolar --nx 2160 --ny 1080 --flux-configuration corrected --days 30 --dt 300 > $OUT/bench_30day_tripolar_2160x1080.json 2>> $OUT/bench.err
python bench.py --grid tripolar --nx 360 --ny 180 --flux-configuration corrected --days 30 --dt 1200 --check-every 180 > $OUT/bench_30day_tripolar_360x180.json 2>> $OUT/bench.err
for f in bench bench_steps20 bench_corrected bench_ncar bench_slab70 bench_profiled; do python -c "
import json,sys
d=json.load(open('$OUT/$f.json')); r=d['roofline']
print('$f', 'ms/step %.4f'%d['ms_per_step'], 'value %.3e'%d['value'], 'ao %.4f frac %.4f sorted %s'%(r['avg_launch_ms'], r['frac'], r.get('avg_launch_ms_batches_sorted_by_trip_hints')), 'cpu', (d.get('cpu_baseline') or {}).get('value'), 'parity', (d.get('parity_measured') or {}).get('max'))"; done
head -8 $OUT/kernel_stats.csv | cut -c1-160
# the 1/8 slab, both ocean presets, with and without the latency layout / the split second layer (A/B on this box)
for cfg in default corrected; do
  python bench.py --ny 70 --flux-configuration $cfg --no-cpu-baseline > $OUT/bench_slab70_$cfg.json 2>> $OUT/bench.err
  python bench.py --ny 70 --flux-configuration $cfg --latency-layout never --no-cpu-baseline > $OUT/bench_slab70_${cfg}_layout_never.json 2>> $OUT/bench.err
  COFLUX_EXPERIMENTS=1 COFLUX_SLAB_SPLIT=0 python bench.py --ny 70 --flux-configuration $cfg --latency-layout never --no-cpu-baseline > $OUT/bench_slab70_${cfg}_round4_plan.json 2>> $OUT/bench.err
done
python bench.py --ny 140 --flux-configuration corrected --no-cpu-baseline > $OUT/bench_slab140_corrected.json 2>> $OUT/bench.err
python bench.py --ny 280 --flux-configuration corrected --no-cpu-baseline > $OUT/bench_slab280_corrected.json 2>> $OUT/bench.err
