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
    t = np.array(list(out), dtype=np.float64)
    k = int((t > 0).sum())
    d = np.diff(t[:k])
    tl = np.array(list(out)[48:56], dtype=np.float64)
    print('   linearize wave (b, chunk 1): stage_robot/state+interp/sweep1/lookups/sweep2/stores/gp cycles', [int(x) for x in np.diff(tl)])
    ta = np.array(list(out)[32:40], dtype=np.float64)
    print('   assemble wave (b, i=1): stage/build/misc/elim/store cycles', [int(x) for x in np.diff(ta[:6])])
    print(path.split('/')[-1], 'traj', b, 'phases(cycles/100MHz ticks?):', [int(x) for x in d], 'total', int(t[k-1]-t[0]))
