# same-box A/B of two library builds on the 1/4-degree surface: scratch/libcoflux_prev.so against the tree's
export COFLUX_ALLOW_STALE_LIBRARY=1
for i in 1 2 3; do for lib in prev new; do
  L=$PWD/climaocean.jl_amd/csrc/libcoflux.so; [ $lib = prev ] && L=$PWD/scratch/libcoflux_prev.so
  LIBCOFLUX=$L python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$lib default', {k:round(v*1e3,2) for k,v in d['solver_paths_ms_per_step'].items()})"
done; done
for lib in prev new prev new; do
  L=$PWD/climaocean.jl_amd/csrc/libcoflux.so; [ $lib = prev ] && L=$PWD/scratch/libcoflux_prev.so
  LIBCOFLUX=$L python bench.py --flux-configuration corrected --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$lib corrected', {k:round(v*1e3,2) for k,v in d['solver_paths_ms_per_step'].items()})"
done
