# where the next step's interpolation riders sit in a slab's solver launch (COFLUX_TAIL_POS) and how many (COFLUX_TAIL_BLOCKS)
for cfg in default corrected; do
for knob in "" "COFLUX_TAIL_POS=0" "COFLUX_TAIL_POS=256" "COFLUX_TAIL_BLOCKS=32" "COFLUX_TAIL_BLOCKS=128" "COFLUX_TAIL_BLOCKS=512"; do
  env COFLUX_EXPERIMENTS=1 $knob python bench.py --ny 70 --flux-configuration $cfg --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$cfg', '$knob'.ljust(24), round(d['ms_per_step']*1e3,2))"
done; done
