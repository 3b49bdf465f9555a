"""Throughput of the dense path for the PR2 model (dof 18, 65 spheres): N = 50, I = 2, 16 trajectories, LM."""
import sys, time
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import gpmp2_amd as g
from gpmp2_amd import engine, problems
from gpmp2_amd.settings import TrajOptimizerSetting

eng = engine.Engine()
model = g.generateMobileArm("PR2")
origin, cell, data = problems.small3d_sdf(40)
origin, cell, data = list(np.array(origin) * 3), cell * 3, data * 3
D, N, B = 18, 50, int(sys.argv[1]) if len(sys.argv) > 1 else 16
for opt in ("GN", "LM"):
    st = TrajOptimizerSetting(D)
    st.set_total_step(N); st.set_total_time(10.0); st.set_obs_check_inter(2); st.set_cost_sigma(0.1); st.set_epsilon(0.4)
    st.set_conf_prior_model(1e-3); st.set_vel_prior_model(1e-3); st.set_Qc_model(np.eye(D)); st.set_max_iter(30)
    {"GN": st.setGaussNewton, "LM": st.setLM}[opt]()
    start, end = np.zeros(D), np.zeros(D)
    start[:3] = [-1.5, -1.0, 0.3]; end[:3] = [1.5, 1.2, -0.4]; end[3] = 0.2
    end[4:] = np.tile(np.linspace(0.2, 0.8, 7), 2) * np.r_[np.ones(7), -np.ones(7)]
    rng = np.random.default_rng(3)
    init = np.zeros((B, N + 1, 2 * D))
    for b in range(B):
        amp = rng.normal(0, 0.1, size=D) * (b > 0)
        for i in range(N + 1):
            init[b, i, :D] = start * (N - i) / N + end * i / N + np.sin(np.pi * i / N) * amp
        init[b, :, D:] = (end - start)[None, :] / 10.0
    z = np.zeros((B, D))
    r, s = eng.robot(model), eng.sdf(origin, cell, data)
    pl = eng.plan(r, s, st, B)
    pl.set_problem(np.repeat(start[None], B, 0), z, np.repeat(end[None], B, 0), z, init)
    pl.optimize(); pl.enable_timing(True)
    t0 = time.perf_counter(); pl.optimize(); dt = time.perf_counter() - t0
    res = pl.result()
    print(opt, f"PR2 {B / dt:.0f} traj/s ({dt * 1e3:.1f} ms per batch of {B}); iters {res['iters'].min()}..{res['iters'].max()}",
          {k: round(v['ms'] / v['launches'] * 1e3, 1) for k, v in pl.timing().items()}, flush=True)
