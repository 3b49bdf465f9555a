import sys, ctypes as C, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from gpmp2_amd import engine, problems
path = sys.argv[1]
e = engine.Engine(path)
p = problems.wam_restarts(B=64)
r, s = e.robot(p.model), e.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)
st = p.setting; st.fixed_iterations = 2
pl = e.plan(r, s, st, p.B)
pl.set_problem(p.start_conf, p.start_vel, p.end_conf, p.end_vel, p.init)
for _ in range(3): pl.optimize()
for b in (0, 33):
    out = (C.c_ulonglong * 64)()
    e._ck(e.lib.gpmp2mi_plan_debug_stamps(pl.h.ptr, b, out))
    raw = np.array(list(out), dtype=np.float64)
    tl = raw[48:64]
    print('   linearize wave (b, chunk 1): [loads+interp | sphere 0..10 (FK link walk + lookup + J) | rest of spheres | stores | gp] cycles', [int(x) for x in np.diff(tl[tl > 0])])
    ta = raw[32:40]
    print('   assemble wave (b, i=1): stage/build/misc/elim/store cycles', [int(x) for x in np.diff(ta[:6])])
    tb = raw[24:29]
    print('   build_tiles (b, i=1): owner rows / constants + unary / sub-step loop / replanner priors + shuffles', [int(x) for x in np.diff(tb)])
    t2 = raw[40:48]
    print('   assemble wave (b, i=2, level 2): stage/build/-/-/-/wait for odd blocks/level-2 products+elim+store', [int(x) for x in np.diff(t2[t2 > 0])])
    print('   assemble i=1 start -> i=2 end:', int(raw[47] - raw[32]))
    print('   step kernel: control', int(raw[1] - raw[0]), 'forward', int(raw[2] - raw[1]), 'backward', int(raw[3] - raw[2]), 'retract', int(raw[4] - raw[3]))
    fw = [raw[1]] + [raw[5 + k] for k in range(1, 9) if raw[5 + k] > 0]
    print('   forward levels h=2,4,..:', [int(x) for x in np.diff(fw)])
    bw = [raw[2]] + [raw[16 + k] for k in range(8, -1, -1) if raw[16 + k] > 0]
    print('   backward levels h=final..1:', [int(x) for x in np.diff(bw)])
