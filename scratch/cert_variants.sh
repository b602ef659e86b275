# certified path under library variants: solver stage (cert_ab.py) and the pipelined step (bench.py)
P="import sys,json
for l in sys.stdin:
    n,_,j=l.partition(' '); d=json.loads(j)
    print(n, {k:(v['ao_us'], v['update_state_us']) for k,v in d.items()})"
for tag in "$@"; do
  if [ "$tag" = base ]; then unset LIBCOFLUX; else export LIBCOFLUX=scratch/libcoflux_$tag.so; fi
  echo "== $tag"; COFLUX_ALLOW_STALE_LIBRARY=1 BUDGETS=800 ORACLE=0 python scratch/cert_ab.py 2>&1 | tail -2 | python -c "$P"
  COFLUX_ALLOW_STALE_LIBRARY=1 python bench.py --no-cpu-baseline --no-sorted-pass 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['solver_paths_ms_per_step'])"
done
