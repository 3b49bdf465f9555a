import sys
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
D, N, B = 10, 100, 2
st = TrajOptimizerSetting(D)
st.set_total_step(N); st.set_total_time(10.0); st.set_obs_check_inter(5); st.set_cost_sigma(0.05); st.set_epsilon(0.3)
st.set_conf_prior_model(1e-3); st.set_vel_prior_model(1e-3); st.set_Qc_model(np.eye(D)); st.set_max_iter(50)
st.setGaussNewton()
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
ro, so = orc.robot(model), orc.sdf(origin, cell, data)
a = eng.linearize(r, s, st, *args, init)
b = orc.linearize(ro, so, st, *args, init)
for name, x, y in zip(("Hd", "Ho", "g", "err"), a, b):
    print(name, "max rel diff", np.abs(x - y).max() / np.abs(y).max())
# dense reference solve from the oracle's normal equations
n = 2 * D
for bb in range(B):
    H = np.zeros(((N + 1) * n, (N + 1) * n))
    for i in range(N + 1):
        H[i*n:(i+1)*n, i*n:(i+1)*n] = b[0][bb, i]
        if i < N:
            H[(i+1)*n:(i+2)*n, i*n:(i+1)*n] = b[1][bb, i]
            H[i*n:(i+1)*n, (i+1)*n:(i+2)*n] = b[1][bb, i].T
    print("cond(H) ~ %.2e" % np.linalg.cond(H))
st.fixed_iterations = 1
res = eng.batch_optimize(r, s, st, *args, init)
ref = orc.batch_optimize(ro, so, st, *args, init)
print("one GN iteration: max traj diff", np.abs(res["traj"] - ref["traj"]).max(), "rel err trace", np.abs(res["error_trace"][:, :2] - ref["error_trace"][:, :2]).max() / np.abs(ref["error_trace"][:, :2]).max())
