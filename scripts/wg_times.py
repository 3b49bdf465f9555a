"""start / end wall-clock (100 MHz) of every k_linearize workgroup of the last launch (diagnostic -DG2_WGTIMES build)"""
import sys, ctypes as C, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from gpmp2_amd import engine, problems
e = engine.Engine(sys.argv[1])
p = problems.wam_restarts(B=64)
r, s = e.robot(p.model), e.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)
st = p.setting; st.fixed_iterations = 2
pl = e.plan(r, s, st, p.B)
pl.set_problem(p.start_conf, p.start_vel, p.end_conf, p.end_vel, p.init)
for _ in range(3): pl.optimize()
raw = []
for b in range(64):
    out = (C.c_ulonglong * 64)()
    e._ck(e.lib.gpmp2mi_plan_debug_stamps(pl.h.ptr, b, out))
    raw += list(out)
t = np.array(raw[:1280], dtype=np.float64).reshape(640, 2) * 10.0   # ns
t0 = t[:, 0].min()
print("workgroups", len(t), " first start 0, last start %.1f us, first end %.1f us, last end %.1f us" % ((t[:, 0].max() - t0) / 1e3, (t[:, 1].min() - t0) / 1e3, (t[:, 1].max() - t0) / 1e3))
d = (t[:, 1] - t[:, 0]) / 1e3
print("workgroup duration us: min %.1f median %.1f max %.1f" % (d.min(), np.median(d), d.max()))
print("start time percentiles us:", [round(float(x), 1) for x in np.percentile((t[:, 0] - t0) / 1e3, [10, 50, 90, 99])])
