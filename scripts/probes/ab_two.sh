for rep in 1 2; do for lib in gpmp2_amd/csrc/build/ab/libsimple.so gpmp2_amd/csrc/build/ab/libpair.so; do
    GPMP2MI_LIB=$PWD/$lib timeout -k 10 200 python3 bench.py --batch 64 --steps 20 --no-cpu-baseline --no-variants > /tmp/ab.json 2>/tmp/ab.err || { tail -3 /tmp/ab.err; exit 1; }
    python3 -c "
import json; d=json.load(open('/tmp/ab.json')); print('$lib', round(d['value']), 'traj/s', {k: round(v['avg_ms']*1e3,1) for k,v in d['roofline']['kernels'].items()})"
done; done
