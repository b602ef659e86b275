# same-box A/B of two library builds on the 1/4-degree surface: scratch/libcoflux_prev.so against the tree's
export COFLUX_ALLOW_STALE_LIBRARY=1
( timeout 1500 python -m pytest tests/test_steps.py tests/test_gpu_parity.py tests/test_full_size.py tests/test_model_api.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3 )
for i in 1 2 3; do for lib in prev new; do
  L=$PWD/climaocean.jl_amd/csrc/libcoflux.so; [ $lib = prev ] && L=$PWD/scratch/libcoflux_prev.so
  LIBCOFLUX=$L python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$lib default', {k:round(v*1e3,2) for k,v in d['solver_paths_ms_per_step'].items()}, round(d['stages_ms']['net_fluxes']*1e3,2), round(d['stages_ms']['net_fluxes_standalone']*1e3,2))"
done; done
for lib in prev new; do
  L=$PWD/climaocean.jl_amd/csrc/libcoflux.so; [ $lib = prev ] && L=$PWD/scratch/libcoflux_prev.so
  LIBCOFLUX=$L python bench.py --ny 70 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$lib slab70', round(d['ms_per_step']*1e3,2), round(d['stages_ms']['net_fluxes_standalone']*1e3,2))"
  LIBCOFLUX=$L python bench.py --config sea_ice --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$lib sea_ice', round(d['ms_per_step']*1e3,2))"
done
