#!/bin/bash
# Round evidence on the GPU box: bench lines, kernel stats, PMC traffic.  usage: bash scripts/evidence.sh rNN
set -o pipefail
R=${1:-r01}
O=gpurun_out/evidence_$R
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 400 python bench.py > $O/${R}_bench.json 2> $O/bench.err || exit 1
tail -c 600 $O/${R}_bench.json; echo
timeout -k 10 200 python bench.py --batch 1024 --steps 10 --no-cpu-baseline > $O/${R}_bench_b1024.json 2>> $O/bench.err || exit 1
timeout -k 10 200 python bench.py --workload windows --batch 128 --no-cpu-baseline > $O/${R}_bench_windows128.json 2>> $O/bench.err || exit 1
timeout -k 10 200 python bench.py --opt LM --no-cpu-baseline > $O/${R}_bench_lm.json 2>> $O/bench.err || exit 1
echo "bench lines done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o ks -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/ks.log 2>&1 || exit 1
echo "kernel stats done"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -o f -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_f.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -o w -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_w.log 2>&1 || exit 1
echo "pmc done"
python scripts/pmc_to_json.py $O/pmc_f $O/pmc_w $O/${R}_pmc_traffic.json
find $O -name "*.csv" | head -20
