#!/bin/bash
# usage: pmc.sh <tag> <script args...>   — PMC passes for the solver kernel (counters only, no tracing)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TAG=$1; shift
OUT=gpurun_out/pmc_$TAG; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY -d $OUT/p1 -o p1 --output-format csv -- python "$@" > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD -d $OUT/p2 -o p2 --output-format csv -- python "$@" > $OUT/p2.log 2>&1
python - <<PY
import csv,glob,collections
for p in ("p1","p2"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv"%p, recursive=True):
        agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"][:60]; agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
        for k,v in agg.items():
            if "ao_flux" in k or "fused" in k or "interp" in k:
                print(k, {a:round(b/5) for a,b in v.items()})
PY
