#!/bin/bash
# k_linearize: shared-walk 4-wavefront form (with the fused finish) against the replicated 2-wavefront form (+ k_finish_step)
# over the batch size
for B in ${BATCHES:-32 64 128 256 512}; do for S in 2 4; do GPMP2MI_LIN_SPLIT=$S timeout -k 10 200 python3 bench.py --batch $B --steps 10 --no-cpu-baseline --no-variants > gpurun_out/ss.json 2>gpurun_out/ss.err && python3 -c "
import json; d=json.load(open('gpurun_out/ss.json')); print('B=$B split=$S', round(d['value']), 'traj/s', {k: round(v['avg_ms']*1e3,1) for k,v in d['roofline']['kernels'].items()})"; done; done
