"""Randomised whole-plan parity sweep over robot kinds (Lie path, two-arm / lift robots, wide blocks) against the
CPU oracle.  usage: python scripts/stress_parity_robots.py [cases]"""
import sys

import numpy as np

sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import gpmp2_amd as g
from gpmp2_amd import engine, problems
from gpmp2_amd.settings import TrajOptimizerSetting
from oracle import Oracle

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
only = int(sys.argv[2]) if len(sys.argv) > 2 else None   # run just this case (e.g. under GPMP2MI_WIDE_DENSE=1)
eng, orc = engine.Engine(), Oracle()
rng = np.random.default_rng(777)
wam = g.generateArm("WAMArm")
a7 = wam.fk_model()
a2 = g.Arm(2, [0.6, 0.5], [0.0, 0.0], [0.0, 0.0])
a3 = g.Arm(3, [0.5, 0.4, 0.3], [0.0, np.pi / 2, 0.0], [0.1, 0.0, 0.05])


def simple(fk, r=0.15):
    return [g.BodySphere(l, r, (-0.05, 0.0, 0.0)) for l in range(fk.nr_links())]


def models():
    mob3 = g.Pose2MobileArm(a3, g.pose3(g.rot_yaw(0.2), (0.1, 0.0, 0.3)))
    two = g.Pose2Mobile2Arms(a2, a2, g.pose3(g.rot_yaw(0.7), (0.3, 0.2, 0.5)), g.pose3(g.rot_yaw(-0.7), (0.3, -0.2, 0.5)))
    lift = g.Pose2MobileVetLinArm(a3, g.pose3(t=(0, 0, 0.2)), g.pose3(g.rot_yaw(0.3), (0.2, 0, 0.2)), bool(rng.integers(0, 2)))
    mobw = g.Pose2MobileArm(g.Arm(7, a7.a, a7.alpha, a7.d), g.pose3(t=(0.0, 0.0, 0.3)))
    liftw = g.Pose2MobileVetLinArm(g.Arm(7, a7.a, a7.alpha, a7.d), g.pose3(t=(0, 0, 0.3)), g.pose3(t=(0.1, 0, 0.2)))
    return {
        "base": g.ArmModel(g.Pose2MobileBase(), [g.BodySphere(0, 0.3, (0.1, 0, 0)), g.BodySphere(0, 0.3, (-0.1, 0, 0))]),
        "mobile arm 3 (6)": g.ArmModel(mob3, simple(mob3)),
        "2 arms 2+2 (7)": g.ArmModel(two, simple(two)),
        "lift arm 3 (7)": g.ArmModel(lift, simple(lift)),
        "mobile WAM (10)": g.ArmModel(mobw, [g.BodySphere(0, 0.3, (0, 0, 0.15))] + [g.BodySphere(s.link_id + 1, s.radius, s.center) for s in wam.spheres]),
        "lift WAM (11)": g.ArmModel(liftw, [g.BodySphere(0, 0.3, (0, 0, 0.15)), g.BodySphere(1, 0.2, (0, 0, 0))] + [g.BodySphere(s.link_id + 2, s.radius, s.center) for s in wam.spheres]),
    }


origin, cell, data = problems.small3d_sdf(40)
origin, cell, data = list(np.array(origin) * 3), cell * 3, data * 3
bad, worst = 0, 0.0
for case in range(cases):
    ms = models()
    name = list(ms)[int(rng.integers(0, len(ms)))]
    model = ms[name]
    D = model.dof()
    N = int(rng.choice([8, 20, 40, 64]))
    inter = int(rng.choice([0, 2, 4]))
    opt = str(rng.choice(["GN", "LM", "DOGLEG"]))
    B = int(rng.choice([1, 4, 16]))
    st = TrajOptimizerSetting(D)
    st.set_total_step(N); st.set_total_time(float(rng.choice([3.0, 6.0]))); st.set_obs_check_inter(inter)
    st.set_cost_sigma(float(rng.choice([0.05, 0.2]))); st.set_epsilon(float(rng.choice([0.2, 0.4])))
    st.set_conf_prior_model(1e-3); st.set_vel_prior_model(1e-3); st.set_Qc_model(np.eye(D)); st.set_max_iter(25)
    {"GN": st.setGaussNewton, "LM": st.setLM, "DOGLEG": st.setDogleg}[opt]()
    if rng.integers(0, 2):
        st.vehicle_dynamics_sigma = 0.01
    start, end = np.zeros(D), np.zeros(D)
    start[:3] = [rng.uniform(-2, -1), rng.uniform(-1.5, 1.5), rng.uniform(-1, 1)]
    end[:3] = [rng.uniform(1, 2), rng.uniform(-1.5, 1.5), rng.uniform(-1, 1)]
    end[3:] = rng.uniform(-0.8, 0.8, size=D - 3)
    init = np.zeros((B, N + 1, 2 * D))
    for b in range(B):
        amp = rng.normal(0, 0.2, size=D) * (b > 0)
        for i in range(N + 1):
            init[b, i, :D] = start * (N - i) / N + end * i / N + np.sin(np.pi * i / N) * amp
        init[b, :, D:] = (end - start)[None, :] / st.total_time
    z = np.zeros((B, D))
    args = (np.repeat(start[None], B, 0), z, np.repeat(end[None], B, 0), z)
    if only is not None and case != only:       # (the random stream above is consumed either way)
        continue
    r, s = eng.robot(model), eng.sdf(origin, cell, data)
    ro, so = orc.robot(model), orc.sdf(origin, cell, data)
    res = eng.batch_optimize(r, s, st, *args, init)
    ref = orc.batch_optimize(ro, so, st, *args, init)
    same = list(res["iters"]) == list(ref["iters"]) and list(res["status"]) == list(ref["status"])
    dtraj = float(np.abs(res["traj"] - ref["traj"]).max())
    ok = same and dtraj < 1e-5
    bad += not ok
    worst = max(worst, dtraj)
    if only is not None:
        per = np.abs(res["traj"] - ref["traj"]).reshape(B, -1).max(1)
        print("per-trajectory max|dtraj|", [float(f"{x:.1e}") for x in per], "rel final err",
              [float(f"{abs(a / b - 1):.1e}") for a, b in zip(res["final_error"], ref["final_error"])])
    print(f"case {case:3d} {name:18s} N={N:3d} I={inter} {opt:6s} B={B:2d} iters {int(ref['iters'].min())}..{int(ref['iters'].max())} "
          f"flow {'same' if same else 'DIFF'} max|dtraj| {dtraj:.1e} {'ok' if ok else 'FAIL'}", flush=True)
print(f"{cases - bad}/{cases} cases agree (identical iteration counts and status, trajectories < 1e-5); worst {worst:.1e}")
