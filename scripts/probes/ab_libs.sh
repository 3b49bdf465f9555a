#!/bin/bash
# A/B of library builds: bench at B = 64 and B = 1024 for every lib given (GPMP2MI_LIB)
set -o pipefail
for lib in "$@"; do
  for B in 64 1024; do
    GPMP2MI_LIB=$PWD/$lib timeout -k 10 200 python3 bench.py --batch $B --steps 10 --no-cpu-baseline --no-variants > /tmp/ab.json 2>/tmp/ab.err || { tail -3 /tmp/ab.err; exit 1; }
    python3 -c "
import json; d=json.load(open('/tmp/ab.json')); print('$lib B=$B', round(d['value']), 'traj/s', {k: round(v['avg_ms']*1e3,1) for k,v in d['roofline']['kernels'].items()})"
  done
done
