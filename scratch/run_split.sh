( timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5 ) > gpurun_out/split_tests.log 2>&1
for cfg in default corrected; do for sp in 0 1 0 1; do
  COFLUX_EXPERIMENTS=1 COFLUX_SLAB_SPLIT=$sp python bench.py --ny 70 --flux-configuration $cfg 2>/dev/null | tail -1 > gpurun_out/split_slab70_${cfg}_$sp.json
  python -c "
import json;d=json.loads(open('gpurun_out/split_slab70_${cfg}_$sp.json').read());print('$cfg split=$sp', round(d['ms_per_step']*1e3,2), round(d['stages_ms'].get('ao_fluxes_standalone',0)*1e3,2))"
done; done
for ny in 80 100 120; do for sp in 0 1; do
  COFLUX_EXPERIMENTS=1 COFLUX_SLAB_SPLIT=$sp python bench.py --ny $ny 2>/dev/null | tail -1 > gpurun_out/split_slab${ny}_$sp.json
  python -c "
import json;d=json.loads(open('gpurun_out/split_slab${ny}_$sp.json').read());print('ny=$ny split=$sp', round(d['ms_per_step']*1e3,2), round(d['stages_ms'].get('ao_fluxes_standalone',0)*1e3,2))"
done; done
cat gpurun_out/split_tests.log
