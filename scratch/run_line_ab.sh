# A/B of slab-TU variants: iteration slope on the 1/8 slab (FixedIterations scan), layout forced on
export COFLUX_ALLOW_STALE_LIBRARY=1
R=$PWD
: > gpurun_out/line_ab.log
for tag in prod0 prod "$@"; do
  L=$R/scratch/libcoflux_$tag.so; LAY=1
  [ $tag = prod ] && L=$R/climaocean.jl_amd/csrc/libcoflux.so
  [ $tag = prod0 ] && L=$R/climaocean.jl_amd/csrc/libcoflux.so && LAY=0
  echo "== $tag" >> gpurun_out/line_ab.log
  NY=70 LAYOUT=$LAY LIBCOFLUX=$L python scratch/iters.py 2>/dev/null | grep -E "^(default|corrected)" >> gpurun_out/line_ab.log
done
python - <<'P'
import json,re
tag=None
for l in open('gpurun_out/line_ab.log'):
    if l.startswith('=='): tag=l.split()[1]; continue
    name,js=l.split(' ',1); d=json.loads(js)
    print(f"{tag:10s} {name:10s} intercept {d['0']:5.1f}  slope16-32 {(d['32']-d['16'])/16:.3f}  slope1-4 {(d['4']-d['1'])/3:.3f} us/trip")
P
