#!/bin/bash
# same-box A/B of two library builds: bash scripts/probes/ab_two.sh libA.so libB.so ["bench args"]
A=$1; B=$2; ARGS=${3:---batch 64 --steps 20 --no-cpu-baseline --no-variants}
for rep in 1 2; do for lib in $A $B; do
    GPMP2MI_LIB=$PWD/$lib timeout -k 10 200 python3 bench.py $ARGS > /tmp/ab.json 2>/tmp/ab.err || { tail -3 /tmp/ab.err; exit 1; }
    python3 -c "
import json; d=json.load(open('/tmp/ab.json')); print('$lib', round(d['value']), 'traj/s', {k: round(v['avg_ms']*1e3,1) for k,v in d['roofline']['kernels'].items()})"
done; done
