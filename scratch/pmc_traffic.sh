#!/bin/bash
# HBM traffic of the three kernels (separate --pmc passes, counters only), per launch.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/pmc_traffic; mkdir -p $OUT
rocprofv3 --pmc FETCH_SIZE -d $OUT/f -o f --output-format csv -- python bench.py --steps 20 --warmup 2 --no-cpu-baseline > $OUT/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/w -o w --output-format csv -- python bench.py --steps 20 --warmup 2 --no-cpu-baseline > $OUT/w.log 2>&1
python - <<PY
import csv,glob,collections,json
res=collections.defaultdict(dict)
for p,name in (("f","FETCH_SIZE"),("w","WRITE_SIZE")):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv"%p, recursive=True):
        agg=collections.defaultdict(float); cnt=collections.Counter()
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"]==name:
                k=r["Kernel_Name"].split("(")[0][-40:]; agg[k]+=float(r["Counter_Value"]); cnt[k]+=1
        for k in agg: res[k][name]=agg[k]/cnt[k]
print(json.dumps(res,indent=1))
open("$OUT/summary.json","w").write(json.dumps(res,indent=1))
PY
