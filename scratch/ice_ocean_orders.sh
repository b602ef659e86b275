# scan of dispatch orders for ice_ocean_kernel (experiment knob COFLUX_ICE_OCEAN_ORDER; profiles/r04_experiments.md §17)
export COFLUX_EXPERIMENTS=1
run() { echo -n "$1 => "; COFLUX_ICE_OCEAN_ORDER="$1" python bench.py --config sea_ice --no-cpu-baseline --no-sorted-pass 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d[\"ms_per_step\"]*1e3,2))"; }
for i in 1 2; do
run ""
run "O0:256,I0:768,O256:512,T"
run "O0:256,I0:768,O512:256,O256:256,T"
run "O0:128,I0:768,O128:640,T"
run "O0:192,I0:768,O192:576,T"
run "O0:320,I0:768,O320:448,T"
run "I0:256,O0:256,I256:512,O256:512,T"
done
