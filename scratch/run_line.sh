# slab-layout kernels: tests, iteration slope and slab bench, layout never vs auto (same box)
( timeout 1200 python -m pytest tests/test_slab_line.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -8 ) > gpurun_out/line_tests.log 2>&1
for lay in 0 1 0 1; do
  echo "== layout $lay slab70 iters" >> gpurun_out/line_iters.log
  NY=70 LAYOUT=$lay python scratch/iters.py 2>/dev/null | grep -E "^(default|corrected)" >> gpurun_out/line_iters.log
done
for lay in never auto; do
  python bench.py --ny 70 --latency-layout $lay 2>/dev/null | tail -1 > gpurun_out/line_slab70_$lay.json
done
python bench.py --ny 140 --latency-layout always 2>/dev/null | tail -1 > gpurun_out/line_slab140_always.json
python bench.py --ny 140 --latency-layout never 2>/dev/null | tail -1 > gpurun_out/line_slab140_never.json
cat gpurun_out/line_tests.log gpurun_out/line_iters.log
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/line_slab*.json')):
    d=json.loads(open(f).read()); print(f, d['ms_per_step'], d.get('solver_paths_ms_per_step'), d['stages_ms'].get('ao_fluxes'), d['stages_ms'].get('ao_fluxes_standalone'))
P
