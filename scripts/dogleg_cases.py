"""The two Dogleg cases of the round-1 robot sweep that missed the 1e-6 trajectory gate (cases 6 and 42): per-iteration
divergence GPU vs oracle and the trust-region scalars of both sides.  usage: python scripts/dogleg_cases.py"""
import copy
import sys

import numpy as np

sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from gpmp2_amd import engine
from oracle import Oracle
from sweep_cases import robot_sweep_cases

eng, orc = engine.Engine(), Oracle()
np.set_printoptions(linewidth=200)
for case, name, opt, p in robot_sweep_cases(43):
    if case not in (6, 42):
        continue
    r, s = eng.robot(p.model), eng.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)
    ro, so = orc.robot(p.model), orc.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)
    args = (p.start_conf, p.start_vel, p.end_conf, p.end_vel)
    res = eng.batch_optimize(r, s, p.setting, *args, p.init)
    ref = orc.batch_optimize(ro, so, p.setting, *args, p.init)
    d = np.abs(res["traj"] - ref["traj"]).reshape(p.B, -1).max(axis=1)
    b = int(np.argmax(d))
    print(f"case {case} {name} {opt} B={p.B}: worst trajectory {b}, max|dtraj| {d[b]:.2e}, iters {res['iters'][b]} / {ref['iters'][b]}")
    one = [a[b:b + 1] for a in args]
    for k in range(1, int(ref["iters"][b]) + 1):
        st = copy.copy(p.setting)
        st.fixed_iterations = k
        pl = eng.plan(r, s, st, 1)
        pl.set_problem(*one, p.init[b:b + 1])
        pl.optimize()
        gt = pl.result()
        sc = pl.debug_scalars(0)
        with orc.dogleg_probe() as pr:
            ot = orc.batch_optimize(ro, so, st, *one, p.init[b:b + 1])
        row = pr.rows[-1]
        uu_g = (sc["gg"] / sc["ghg"]) ** 2 * sc["gg"]
        print(f"  k={k:2d} |dtraj| {np.abs(gt['traj'] - ot['traj']).max():.2e}  err rel {abs(gt['final_error'][0] - ot['final_error'][0]) / ot['final_error'][0]:.1e}"
              f"  trials(oracle) {len(pr.rows)}  tau {row[6]: .6f}  Delta {row[7]:.3e}  rho {row[8]: .4f}")
        print(f"        rel diff gg {abs(sc['gg'] - row[0]) / row[0]:.1e}  gHg {abs(sc['ghg'] - row[1]) / row[1]:.1e}  g.dxn {abs(sc['gn'] - row[2]) / abs(row[2]):.1e}"
              f"  nn {abs(sc['nn'] - row[3]) / row[3]:.1e}  uu {abs(uu_g - row[4]) / row[4]:.1e}   a = |dx_n - dx_u|^2 = {row[4] - 2 * row[5] + row[3]:.3e}  nn {row[3]:.3e} uu {row[4]:.3e}")
