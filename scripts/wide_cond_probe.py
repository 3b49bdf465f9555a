"""Where does the 2x2-tile path differ from the oracle on long mobile-manipulator trajectories?  Runs the N = 35 case of
tests/test_gpu_robots.py::test_wide_long_trajectories through (a) the tile path, (b) the dense path
(GPMP2MI_WIDE_DENSE=1, set by the caller) and prints the differences to the oracle per optimizer."""
import os
import sys

import numpy as np

sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import test_gpu_robots as T
from gpmp2_amd import engine as E
from oracle import Oracle

eng, orc = E.Engine(), Oracle()
name, N = sys.argv[1] if len(sys.argv) > 1 else "mobile WAM (dof 10)", int(sys.argv[2]) if len(sys.argv) > 2 else 35
model = T._wide_models()[name]
D = model.dof()
p = T._tree_problem(model, N=N, inter=2, opt="GN")
B = 3
rng = np.random.default_rng(41)
start, end = np.repeat(p.start_conf, B, 0), np.repeat(p.end_conf, B, 0)
start[1:, 3:] += 0.2 * rng.normal(size=(B - 1, D - 3))
end[1:, :2] += 0.3 * rng.normal(size=(B - 1, 2))
init = np.zeros((B, N + 1, 2 * D))
for b in range(B):
    for i in range(N + 1):
        init[b, i, :D] = start[b] * (N - i) / N + end[b] * i / N
    init[b, :, D:] = (end[b] - start[b])[None, :] / 3.0
z = np.zeros((B, D))
args = (start, z, end, z)
r, ro = eng.robot(p.model), orc.robot(p.model)
s, so = eng.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data), orc.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)
print("dense" if os.environ.get("GPMP2MI_WIDE_DENSE") == "1" else "tiles", name, "N", N)
for opt in ("GN", "LM", "DOGLEG"):
    {"GN": p.setting.setGaussNewton, "LM": p.setting.setLM, "DOGLEG": p.setting.setDogleg}[opt]()
    res = eng.batch_optimize(r, s, p.setting, *args, init)
    ref = orc.batch_optimize(ro, so, p.setting, *args, init)
    print(opt, "iters", list(res["iters"]), list(ref["iters"]), "status", list(res["status"]), list(ref["status"]),
          "traj diff per trajectory", [float(f"{np.abs(res['traj'][b] - ref['traj'][b]).max():.2e}") for b in range(B)],
          "err rel", [float(f"{abs(res['final_error'][b] / ref['final_error'][b] - 1):.1e}") for b in range(B)])
