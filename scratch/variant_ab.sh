# A/B of library variants (scratch/libcoflux_<tag>.so; "base" = the production library): exact solver path, full surface and 1/8 slab
P="import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['avg_launch_ms'], d['stages_ms']['ao_fluxes_standalone'])"
for tag in "$@"; do
  if [ "$tag" = base ]; then unset LIBCOFLUX; else export LIBCOFLUX=scratch/libcoflux_$tag.so; fi
  for ny in 560 70; do echo "== $tag ny=$ny"; COFLUX_ALLOW_STALE_LIBRARY=1 python bench.py --ny $ny --no-cpu-baseline --no-sorted-pass --solver-path ${SOLVER_PATH:-exact} 2>/dev/null | python -c "$P"; done
done
