( timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5 ) > gpurun_out/line2_tests.log 2>&1
for cfg in default corrected; do for lay in never auto; do
  python bench.py --ny 70 --flux-configuration $cfg --latency-layout $lay 2>/dev/null | tail -1 > gpurun_out/line2_slab70_${cfg}_$lay.json
done; done
for lay in never auto; do
  python bench.py --grid tripolar --nx 360 --ny 180 --latency-layout $lay 2>/dev/null | tail -1 > gpurun_out/line2_tripolar360_$lay.json
  python bench.py --ny 35 --latency-layout $lay 2>/dev/null | tail -1 > gpurun_out/line2_slab35_$lay.json
done
cat gpurun_out/line2_tests.log
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/line2_*.json')):
    try:
        d=json.loads(open(f).read()); print(f, round(d['ms_per_step']*1e3,2), d['config'].get('solver_path'), {k:round(v*1e3,2) for k,v in (d.get('solver_paths_ms_per_step') or {}).items()})
    except Exception as e: print(f, 'ERR', e)
P
