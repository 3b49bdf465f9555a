"""Latency of the replanner step on a resident plan (ISAM2TrajOptimizer::update semantics, gpmp2mi_plan_update): WAM,
N = 100, I = 5, Synth200 field; fix the current state, move the goal, `iterations` Gauss-Newton iterations warm-started
from the previous solution.  Host wall time per call, median of 50."""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
from gpmp2_amd import engine, problems

eng = engine.Engine()
for B in (1, 16):
    p = problems.wam_restarts(B=B, total_step=100, obs_check_inter=5, opt="GN")
    r, s = eng.robot(p.model), eng.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)
    pl = eng.plan(r, s, p.setting, B)
    pl.set_problem(p.start_conf, p.start_vel, p.end_conf, p.end_vel, p.init)
    pl.optimize()
    first = pl.result()["traj"]
    D = p.setting.dof
    for iters in (1, 2, 3):
        ts = []
        for k in range(60):
            for b in range(B):
                pl.clear_state_priors(b)
                pl.fix_state(b, 5, first[b, 5, :D], first[b, 5, D:])
                pl.change_goal(b, p.end_conf[b] + 0.01 * ((k % 5) - 2), np.zeros(D))
            t0 = time.perf_counter()
            pl.update(iterations=iters)
            ts.append(time.perf_counter() - t0)
        ts = np.array(ts[10:]) * 1e3
        print(f"B={B:2d} update(iterations={iters}): median {np.median(ts):.3f} ms, min {ts.min():.3f} ms", flush=True)
