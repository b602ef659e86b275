# certified solver path in the pipelined step (bench.py) under different chunk plans (COFLUX_LAYERS, experiments only)
P="import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline'].get('avg_launch_ms'))"
for budget in ${BUDGETS_SCAN:-800}; do
for lay in "" 1024,768,512 896,768,640 832,768,704 1152,768,384 1024,896,384; do
  echo "== budget $budget layers ${lay:-plan}"
  if [ -z "$lay" ]; then python bench.py --no-cpu-baseline --solver-path certified --certified-budget $budget ${EXTRA} 2>/dev/null | python -c "$P"
  else COFLUX_EXPERIMENTS=1 COFLUX_LAYERS=$lay python bench.py --no-cpu-baseline --solver-path certified --certified-budget $budget ${EXTRA} 2>/dev/null | python -c "$P"; fi
done; done
echo "== exact"; python bench.py --no-cpu-baseline 2>/dev/null | python -c "$P"
