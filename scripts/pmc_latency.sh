#!/bin/bash
# average in-flight latency per instruction class and cache hit rates (two PMC passes), default bench workload
set -o pipefail
O=gpurun_out/pmc_lat
mkdir -p $O
export TMPDIR=/tmp
source scripts/lib_run.sh
BENCH="python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-variants"
PASSES=${PASSES:-}; run a 300 rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_INSTS_SMEM SQ_INSTS_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VMEM_RD --output-format csv -d $O/a -o a -- $BENCH && PASSES="$PASSES $O/a"
PASSES=${PASSES:-}; run b 300 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/b -o b -- $BENCH && PASSES="$PASSES $O/b"
PASSES=${PASSES:-}; run c 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_LEVEL_WAVES SQ_WAVES --output-format csv -d $O/c -o c -- $BENCH && PASSES="$PASSES $O/c"
python3 scripts/pmc_sq_to_json.py $O/lat.json $PASSES
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/pmc_lat/lat.json'))['kernels']
for k,v in d.items():
    if not k.startswith('k_') or 'pack' in k or 'reset' in k: continue
    g=v.get
    out={}
    if g('SQ_INSTS_VMEM_RD'): out['vmem_lat_cyc']=round(4*g('SQ_INST_LEVEL_VMEM',0)/(g('SQ_INSTS_VMEM_RD')+g('SQ_INSTS_VMEM_WR',0)),0)
    if g('SQ_INSTS_SMEM'): out['smem_lat_cyc']=round(4*g('SQ_INST_LEVEL_SMEM',0)/g('SQ_INSTS_SMEM'),0)
    if g('SQ_INSTS_LDS'): out['lds_lat_cyc']=round(4*g('SQ_INST_LEVEL_LDS',0)/g('SQ_INSTS_LDS'),0)
    if g('SQ_IFETCH'): out['ifetch_lat_cyc']=round(4*g('SQ_IFETCH_LEVEL',0)/g('SQ_IFETCH'),0); out['ifetch_per_wave']=round(g('SQ_IFETCH')/g('SQ_WAVES',1),0)
    if g('TCP_TCC_READ_REQ_sum'): out['l1miss_lat']=round(g('TCP_TCC_READ_REQ_LATENCY_sum',0)/g('TCP_TCC_READ_REQ_sum'),0)
    if g('TCC_HIT_sum') is not None: out['l2_hit']=round(g('TCC_HIT_sum')/(g('TCC_HIT_sum')+g('TCC_MISS_sum',0)+1e-9),3)
    for c in ('SQ_INSTS_SMEM','SQ_INSTS_LDS','SQ_INSTS_VMEM_RD','SQ_INSTS_SALU','SQ_INSTS_BRANCH','SQ_WAVES','SQ_WAVE_CYCLES','SQ_INST_CYCLES_SMEM','SQ_INST_CYCLES_SALU'):
        if g(c) is not None: out[c]=round(g(c))
    print(k, out)
PY
