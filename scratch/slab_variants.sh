# exact path on thin latitude slabs under library variants (one wave's dependent chain sets the time there)
P="import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],5), round(d['roofline']['avg_launch_ms'],5), round(d['stages_ms']['ao_fluxes_standalone'],5))"
for tag in "$@"; do
  if [ "$tag" = base ]; then unset LIBCOFLUX; else export LIBCOFLUX=scratch/libcoflux_$tag.so; fi
  for ny in 70 35; do echo "== $tag ny=$ny"; COFLUX_ALLOW_STALE_LIBRARY=1 python bench.py --ny $ny --no-cpu-baseline --no-sorted-pass --solver-path exact 2>/dev/null | python -c "$P"; done
done
