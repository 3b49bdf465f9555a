import sys
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
import gpmp2_amd as g
from gpmp2_amd import engine as E
from oracle import Oracle
import test_gpu_robots as T
eng=E.Engine(); orc=Oracle()
model = g.generateMobileArm("PR2")
r, ro = eng.robot(model), orc.robot(model)
rng = np.random.default_rng(41)
q = rng.uniform(-1.0, 1.0, size=(32, 18))
p = T._tree_problem(model, N=8, inter=1, opt="GN")
p.end_conf[0, 3] = 0.2
p.end_conf[0, 4:] = np.tile(np.linspace(0.2, 0.8, 7), 2) * np.r_[np.ones(7), -np.ones(7)]
for i in range(9):
    p.init[0, i, :18] = p.start_conf[0] * (8 - i) / 8 + p.end_conf[0] * i / 8
p.init[0, :, 18:] = (p.end_conf[0] - p.start_conf[0])[None, :] / 3.0
s, so = eng.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data), orc.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)
args = (p.start_conf, p.start_vel, p.end_conf, p.end_vel)
traj = p.init + 0.05 * rng.normal(size=p.init.shape)
(ea, ha), (eb, hb) = eng.obstacle_factor(r, s, 0.6, traj[0, :, :18]), orc.obstacle_factor(ro, so, 0.6, traj[0, :, :18])
print("obstacle_factor on traj: max err diff", np.abs(ea-eb).max(), "H diff", np.abs(ha-hb).max())
a = eng.linearize(r, s, p.setting, *args, traj)
b = orc.linearize(ro, so, p.setting, *args, traj)
for k,(x,y) in enumerate(zip(a[:3], b[:3])):
    d=np.abs(x-y); print(k, x.shape, "max diff", d.max(), "at", np.unravel_index(d.argmax(), d.shape), "n>tol", (d>1e-9*np.abs(y).max()).sum())
    idx=np.argwhere(d>1e-9*np.abs(y).max())
    print(idx[:20].tolist())
