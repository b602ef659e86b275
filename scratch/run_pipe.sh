for cfg in default corrected; do for p in off tail auto; do
  python bench.py --ny 70 --flux-configuration $cfg --pipeline $p --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/pipe_${cfg}_$p.json
  python -c "
import json;d=json.loads(open('gpurun_out/pipe_${cfg}_$p.json').read());print('$cfg pipeline=$p', round(d['ms_per_step']*1e3,2), {k:round(v*1e3,2) for k,v in d['stages_ms'].items() if v})"
done; done
