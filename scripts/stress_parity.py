"""Randomised parity sweep of whole plans against the CPU oracle (not part of the test suite: minutes of oracle
time).  Random start / goal configurations, trajectory lengths, interpolation counts, optimizers and batch
sizes on the WAM arm in the down-scaled desk scene; prints one line per case and a summary.
usage: python scripts/stress_parity.py [cases]"""
import sys

import numpy as np

sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from gpmp2_amd import engine, problems
from gpmp2_amd.trajutils import initArmTrajStraightLine
from oracle import Oracle

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
eng, orc = engine.Engine(), Oracle()
rng = np.random.default_rng(2026)
bad = 0
worst = 0.0
for case in range(cases):
    N = int(rng.choice([10, 25, 50, 100, 128]))
    inter = int(rng.choice([0, 1, 3, 5, 9]))
    opt = str(rng.choice(["GN", "LM", "DOGLEG"]))
    B = int(rng.choice([1, 3, 16, 64]))
    p = problems.wam_restarts(B=B, total_step=N, obs_check_inter=inter, opt=opt, sdf="40", max_iter=30)
    start = problems.WAM_START + rng.normal(0, 0.3, size=7)
    goal = problems.WAM_END + rng.normal(0, 0.3, size=7)
    base = initArmTrajStraightLine(start, goal, N)
    for b in range(B):
        p.start_conf[b], p.end_conf[b] = start, goal
        p.init[b] = base
        if b:
            p.init[b, :, :7] += np.sin(np.pi * np.arange(N + 1) / N)[:, None] * rng.normal(0, 0.4, size=7)[None, :]
    p.setting.set_epsilon(float(rng.choice([0.1, 0.2, 0.3])))
    p.setting.set_cost_sigma(float(rng.choice([0.02, 0.05, 0.1])))
    r, s = eng.robot(p.model), eng.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)
    ro, so = orc.robot(p.model), orc.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)
    args = (p.start_conf, p.start_vel, p.end_conf, p.end_vel)
    res = eng.batch_optimize(r, s, p.setting, *args, p.init)
    ref = orc.batch_optimize(ro, so, p.setting, *args, p.init)
    same_flow = list(res["iters"]) == list(ref["iters"]) and list(res["status"]) == list(ref["status"])
    dtraj = float(np.abs(res["traj"] - ref["traj"]).max())
    derr = float(np.max(np.abs(res["final_error"] - ref["final_error"]) / np.abs(ref["final_error"])))
    ok = same_flow and dtraj < 1e-6 and derr < 1e-8
    bad += not ok
    worst = max(worst, dtraj)
    print(f"case {case:3d} N={N:3d} I={inter} {opt:6s} B={B:2d} iters {int(ref['iters'].min())}..{int(ref['iters'].max())} "
          f"flow {'same' if same_flow else 'DIFF'} max|dtraj| {dtraj:.1e} rel final err {derr:.1e} {'ok' if ok else 'FAIL'}",
          flush=True)
print(f"{cases - bad}/{cases} cases agree (identical iteration counts and status, trajectories < 1e-6); worst {worst:.1e}")
