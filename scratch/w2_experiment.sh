P="import sys,json
for l in sys.stdin:
    n,_,j=l.partition(' '); d=json.loads(j)
    for k,v in d.items(): print(n,k,v['ao_us'],v['update_state_us'],v.get('exact_share'))
"
for lib in w2 w2p; do for lay in 1280,1024,64 1280,1088,64; do echo == $lib $lay; COFLUX_EXPERIMENTS=1 COFLUX_LAYERS=$lay LIBCOFLUX=scratch/libcoflux_$lib.so BUDGETS=800,1000000 ORACLE=0 python scratch/cert_ab.py 2>&1 | tail -2 | python -c "$P"; done; done
