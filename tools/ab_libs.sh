#!/bin/bash
# Same-box A/B of library builds on the driver's workload: $1 = output prefix, the rest = tags of tools/ab/libcoflux_<tag>.so ("prod" = the production library)
out=$1; shift
mkdir -p $(dirname $out); rm -f ${out}.jsonl
for r in 1 2 3; do
  for tag in "$@"; do
    if [ $tag = prod ]; then unset LIBCOFLUX COFLUX_ALLOW_STALE_LIBRARY; else export LIBCOFLUX=$PWD/tools/ab/libcoflux_$tag.so COFLUX_ALLOW_STALE_LIBRARY=1; fi
    python bench.py --steps 300 --warmup 50 --no-cpu-baseline --no-sorted-pass --solver-path exact ${EXTRA} 2>/dev/null | \
      python -c "import json,sys; d=json.loads(sys.stdin.readline()); s=d['stages_ms']; print(json.dumps(dict(lib='$tag', ms_per_step=d['ms_per_step'], launch_ms=s['ao_fluxes'], stress_ms=s['net_fluxes'], solver_alone_ms=s['ao_fluxes_standalone'], interp_alone_ms=s['interpolate_tiled_standalone'], stress_alone_ms=s['net_fluxes_standalone'])))" >> ${out}.jsonl
  done
done
cat ${out}.jsonl
