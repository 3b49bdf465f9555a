"""Throughput of the dense (8 <= dof <= 11) path: Pose2 mobile base + 7-joint WAM arm (dof 10), N = 100, I = 5,
Synth200 field scaled to the robot, LM and GN, against the same problem on the CPU oracle."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import gpmp2_amd as g
from gpmp2_amd import engine, problems
from gpmp2_amd.settings import TrajOptimizerSetting
from oracle import Oracle

eng, orc = engine.Engine(), Oracle()
wam = g.generateArm("WAMArm")
a7 = wam.fk_model()
mob = g.Pose2MobileArm(g.Arm(7, a7.a, a7.alpha, a7.d), g.pose3(t=(0.0, 0.0, 0.3)))
model = g.ArmModel(mob, [g.BodySphere(0, 0.3, (0, 0, 0.15))] + [g.BodySphere(s.link_id + 1, s.radius, s.center) for s in wam.spheres])
origin, cell, data = problems.small3d_sdf(40)
origin, cell, data = list(np.array(origin) * 3), cell * 3, data * 3
D, N, B = 10, 100, 64
for opt in ("GN", "LM"):
    st = TrajOptimizerSetting(D)
    st.set_total_step(N); st.set_total_time(10.0); st.set_obs_check_inter(5); st.set_cost_sigma(0.05); st.set_epsilon(0.3)
    st.set_conf_prior_model(1e-3); st.set_vel_prior_model(1e-3); st.set_Qc_model(np.eye(D)); st.set_max_iter(int(os.environ.get('WIDE_MAX_ITER', '50')))
    {"GN": st.setGaussNewton, "LM": st.setLM}[opt]()
    start = np.concatenate([[-2.0, -1.5, 0.0], problems.WAM_START])
    end = np.concatenate([[2.0, 1.5, 0.5], problems.WAM_END])
    rng = np.random.default_rng(5)
    init = np.zeros((B, N + 1, 2 * D))
    for b in range(B):
        amp = rng.normal(0, 0.3, size=D) * (b > 0)
        for i in range(N + 1):
            init[b, i, :D] = start * (N - i) / N + end * i / N + np.sin(np.pi * i / N) * amp
        init[b, :, D:] = (end - start)[None, :] / 10.0
    z = np.zeros((B, D))
    args = (np.repeat(start[None], B, 0), z, np.repeat(end[None], B, 0), z)
    r, s = eng.robot(model), eng.sdf(origin, cell, data)
    pl = eng.plan(r, s, st, B)
    pl.set_problem(*args, init)
    pl.optimize()
    pl.enable_timing(True)
    t0 = time.perf_counter()
    pl.optimize()
    dt = time.perf_counter() - t0
    res = pl.result()
    print(opt, f"GPU {B / dt:.0f} traj/s ({dt * 1e3:.1f} ms per batch of {B}); iters {res['iters'].min()}..{res['iters'].max()}",
          {k: round(v['ms'] / v['launches'] * 1e3, 1) for k, v in pl.timing().items()}, flush=True)
    if os.environ.get('WIDE_NO_ORACLE'):
        continue
    ro, so = orc.robot(model), orc.sdf(origin, cell, data)
    t0 = time.perf_counter()
    ref = orc.batch_optimize(ro, so, st, *args, init, nthreads=64)
    dc = time.perf_counter() - t0
    print(opt, f"oracle {B / dc:.0f} traj/s on 64 threads; iteration counts equal: {np.array_equal(ref['iters'], res['iters'])}; "
          f"max traj diff {np.abs(ref['traj'] - res['traj']).max():.1e}", flush=True)
