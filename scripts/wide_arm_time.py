"""Per-kernel times of the 2x2-tile path for a vector-space robot (10-joint planar chain, dof 10): the same kernels as
the mobile-base + WAM case of wide_time.py without the Pose2 (Lie) blocks in the assembler."""
import sys
import time

import numpy as np

sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import gpmp2_amd as g
from gpmp2_amd import engine, problems
from gpmp2_amd.settings import TrajOptimizerSetting

eng = engine.Engine()
D, N, B = 10, 100, 64
arm = g.Arm(D, [0.3] * D, [0.0] * D, [0.0] * D)
model = g.ArmModel(arm, [g.BodySphere(j, 0.1, (-0.15 * (s + 0.5), 0, 0)) for j in range(D) for s in range(2)][:17])
origin, cell, data = problems.small3d_sdf(40)
origin, cell, data = list(np.array(origin) * 3), cell * 3, data * 3
for opt in ("GN", "LM"):
    st = TrajOptimizerSetting(D)
    st.set_total_step(N); st.set_total_time(10.0); st.set_obs_check_inter(5); st.set_cost_sigma(0.05); st.set_epsilon(0.3)
    st.set_conf_prior_model(1e-3); st.set_vel_prior_model(1e-3); st.set_Qc_model(np.eye(D)); st.set_max_iter(50)
    {"GN": st.setGaussNewton, "LM": st.setLM}[opt]()
    start, end = np.full(D, -0.2), np.full(D, 0.25)
    rng = np.random.default_rng(5)
    init = np.zeros((B, N + 1, 2 * D))
    for b in range(B):
        amp = rng.normal(0, 0.2, size=D) * (b > 0)
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
