import sys, time
import numpy as np
sys.path.insert(0, '.')
import gpmp2_amd as g
from gpmp2_amd import problems
p = problems.wam_restarts(B=1, total_step=10, obs_check_inter=9, opt="DOGLEG", sdf="synth200")
sdf = g.SignedDistanceField(p.sdf_origin, p.sdf_cell, p.sdf_data.shape[1], p.sdf_data.shape[2], p.sdf_data.shape[0])
sdf.data_ = p.sdf_data
t0 = time.perf_counter(); sdf.handle(); print("sdf upload+pack %.1f ms" % (1e3 * (time.perf_counter() - t0)))
for k in range(6):
    t0 = time.perf_counter()
    res = g.BatchTrajOptimize3DArm(p.model, sdf, p.start_conf[0], p.start_vel[0], p.end_conf[0], p.end_vel[0], p.init[0], p.setting)
    print("one-shot call %d: %.2f ms" % (k, 1e3 * (time.perf_counter() - t0)))
from gpmp2_amd.planner import _eng, _robot_handle
eng = _eng()
pl = eng.plan(_robot_handle(p.model), sdf.handle(), p.setting, 1)
for k in range(4):
    t0 = time.perf_counter()
    pl.set_problem(p.start_conf, p.start_vel, p.end_conf, p.end_vel, p.init); pl.optimize(); r = pl.result()
    print("resident plan call %d: %.2f ms  iters %s" % (k, 1e3 * (time.perf_counter() - t0), r["iters"]))
# the three parts of a resident call
import numpy as np
ts = np.zeros((20, 3))
for k in range(20):
    t0 = time.perf_counter(); pl.set_problem(p.start_conf, p.start_vel, p.end_conf, p.end_vel, p.init)
    t1 = time.perf_counter(); pl.optimize()
    t2 = time.perf_counter(); r = pl.result()
    t3 = time.perf_counter()
    ts[k] = [t1 - t0, t2 - t1, t3 - t2]
print("set_problem / optimize / result (median, ms):", [round(float(x), 3) for x in np.median(ts[5:], 0) * 1e3])
