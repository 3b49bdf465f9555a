#!/bin/bash
# Round evidence on the GPU box: bench lines, kernel stats, PMC traffic, SQ counters.
# usage: bash scripts/evidence.sh rNN [quick]      (.evidence_commit, written by the caller, names the commit measured)
set -o pipefail
R=${1:-r03}
O=gpurun_out/evidence_$R
mkdir -p $O
export TMPDIR=/tmp
source scripts/lib_run.sh
COMMIT=$(cat .evidence_commit 2>/dev/null || echo unknown)
BENCH="python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-variants"
timeout -k 10 500 python3 bench.py > $O/${R}_bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
tail -c 400 $O/${R}_bench.json; echo
if [ "$2" != "quick" ]; then
timeout -k 10 200 python3 bench.py --batch 1024 --steps 10 --no-cpu-baseline --no-variants > $O/${R}_bench_b1024.json 2>> $O/bench.err || exit 1
# config 4 with its CPU leg: every sampled window is compared with the oracle before the timing counts
timeout -k 10 300 python3 bench.py --workload windows --batch 128 --cpu-sample 128 > $O/${R}_bench_windows128.json 2>> $O/bench.err || exit 1
timeout -k 10 200 python3 bench.py --opt LM --steps 10 --no-cpu-baseline > $O/${R}_bench_lm.json 2>> $O/bench.err || exit 1
timeout -k 10 200 python3 bench.py --opt DOGLEG --steps 10 --no-cpu-baseline > $O/${R}_bench_dogleg.json 2>> $O/bench.err || exit 1
echo "bench lines done"
fi
run ks 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o ks -- $BENCH || exit 1
run pmc_f 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -o f -- $BENCH || exit 1
run pmc_w 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -o w -- $BENCH || exit 1
python3 scripts/pmc_to_json.py $O/pmc_f $O/pmc_w $O/${R}_pmc_traffic.json $COMMIT
# SQ counter groups, one pass each (<= 8 SQ counters per pass); a pass whose step failed is left out of the summary
SQ=""
run sq_a 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $O/sq_a -o a -- $BENCH && SQ="$SQ $O/sq_a"
run sq_b 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD --output-format csv -d $O/sq_b -o b -- $BENCH && SQ="$SQ $O/sq_b"
run sq_c 300 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_LDS_IDX_ACTIVE --output-format csv -d $O/sq_c -o c -- $BENCH && SQ="$SQ $O/sq_c"
[ -n "$SQ" ] && python3 scripts/pmc_sq_to_json.py $O/${R}_pmc_sq.json $SQ
if [ "$2" != "quick" ]; then
B1K="python3 bench.py --batch 1024 --steps 3 --warmup 1 --no-cpu-baseline --no-variants"
SQ=""
run sq_a_b1024 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $O/sq_a_b1024 -o a -- $B1K && SQ="$O/sq_a_b1024"
run ks_b1024 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_b1024 -o ks -- $B1K || exit 1
run pmc_f_b1024 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f_b1024 -o f -- $B1K || exit 1
run pmc_w_b1024 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w_b1024 -o w -- $B1K || exit 1
[ -n "$SQ" ] && python3 scripts/pmc_sq_to_json.py $O/${R}_pmc_sq_b1024.json $SQ
python3 scripts/pmc_to_json.py $O/pmc_f_b1024 $O/pmc_w_b1024 $O/${R}_pmc_traffic_b1024.json $COMMIT
fi
find $O -name "*stats*.csv" | head -20
