import ctypes as C, sys
import numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from gpmp2_amd import engine, problems
e = engine.Engine(sys.argv[1])
p = problems.wam_restarts(B=64, opt="LM")
r, s = e.robot(p.model), e.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)
pl = e.plan(r, s, p.setting, p.B)
pl.set_problem(p.start_conf, p.start_vel, p.end_conf, p.end_vel, p.init)
for _ in range(2): pl.optimize()
for b in (0, 33, 50):
    out = (C.c_ulonglong * 64)()
    e._ck(e.lib.gpmp2mi_plan_debug_stamps(pl.h.ptr, p.B + b, out))
    raw = np.array(list(out), dtype=np.float64)[32:38]
    print("k_decide trajectory", b, "(last call): start -> error sum issued+reduced / barrier / decision by thread 0 / barrier / moves:", [int(x) for x in np.diff(raw)])
