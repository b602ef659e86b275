#!/bin/bash
# sclk / power of the device while the bench loop runs (scratch): is the FP64-bound solver clock- or power-limited?
cd $GRAFT_REPO_ROOT
python bench.py --steps 400000 --warmup 10 --repetitions 1 --no-cpu-baseline --no-sorted-pass > /tmp/b.json 2>/tmp/b.err &
PID=$!
sleep 6
for n in 1 2 3 4 5 6 7 8 9 10 11 12; do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor (edge|junction|hotspot)" | tr '\n' ' '; echo; sleep 3; done
wait $PID
python -c "import json;d=json.load(open('/tmp/b.json'));print('ms/step',d['ms_per_step'])"
echo idle; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo
rocm-smi --showmaxpower 2>/dev/null | grep -i power
