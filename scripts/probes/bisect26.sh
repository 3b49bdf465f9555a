set -o pipefail
mkdir -p gpurun_out/par
for c in f0194fa 03878a5 34ee47c 4b08f7d a45227e 38cf430; do
  GPMP2MI_LIB=$PWD/build/ab/lib_$c.so timeout -k 10 200 python3 scripts/parity_sensitivity.py 27 26 >> gpurun_out/par/bisect26.log 2>&1 || echo "rc $? for $c" >> gpurun_out/par/bisect26.log
done
timeout -k 10 200 python3 scripts/parity_sensitivity.py 27 26 >> gpurun_out/par/bisect26.log 2>&1
GPMP2MI_WIDE_DENSE=1 timeout -k 10 200 python3 scripts/parity_sensitivity.py 27 26 >> gpurun_out/par/bisect26.log 2>&1
GPMP2MI_WIDE_H0=2 timeout -k 10 200 python3 scripts/parity_sensitivity.py 27 26 >> gpurun_out/par/bisect26.log 2>&1
grep "case" gpurun_out/par/bisect26.log
timeout -k 10 600 python3 scripts/parity_sensitivity.py 50 > gpurun_out/par/sens50.log 2>&1; echo rc=$?
grep "over\|above" gpurun_out/par/sens50.log
for w in 1 4 16; do timeout -k 10 120 ./build/probes/elim_probe $w 20 >> gpurun_out/par/elim_probe.log 2>&1 || echo "probe rc $?" >> gpurun_out/par/elim_probe.log; done
cat gpurun_out/par/elim_probe.log
