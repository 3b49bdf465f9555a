import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from gpmp2_amd import engine, problems
from gpmp2_amd.trajutils import initArmTrajStraightLine
from oracle import Oracle
eng, orc = engine.Engine(), Oracle()
def run(p, tag):
    r, s = eng.robot(p.model), eng.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)
    ro, so = orc.robot(p.model), orc.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)
    a = (p.start_conf, p.start_vel, p.end_conf, p.end_vel)
    p.setting.fixed_iterations = 1
    res = eng.batch_optimize(r, s, p.setting, *a, p.init)
    ref = orc.batch_optimize(ro, so, p.setting, *a, p.init)
    d = np.abs(res["traj"] - ref["traj"])[0].max(axis=1)
    bad = np.nonzero(d > 1e-6)[0]
    print(tag, "max diff %.2e" % d.max(), "bad states", bad[:12], "..." if bad.size > 12 else "")
import gpmp2_amd as g
from gpmp2_amd.settings import TrajOptimizerSetting
from gpmp2_amd import datasets
for D, N in ((4, 2), (4, 3), (4, 4), (4, 5), (4, 6), (4, 8), (5, 2), (5, 4), (5, 8)):
    arm = g.Arm(D, [0.3] * D, [0.0] * D, [0.0] * D)
    model = g.ArmModel(arm, [g.BodySphere(l, 0.05, (-0.1, 0, 0)) for l in range(D)])
    d = datasets.generate2Ddataset("TwoObstaclesDataset")
    field = datasets.signedDistanceField2D(d.map, d.cell_size)
    st = TrajOptimizerSetting(D)
    st.set_total_step(N); st.set_total_time(3.0); st.set_obs_check_inter(2); st.set_cost_sigma(0.1); st.set_epsilon(0.2)
    st.set_conf_prior_model(1e-3); st.set_vel_prior_model(1e-3); st.set_Qc_model(np.eye(D)); st.setGaussNewton()
    start, end = np.zeros(D), np.linspace(0.3, 0.8, D)
    init = initArmTrajStraightLine(start, end, N)[None]
    z = np.zeros((1, D))
    p = problems.Problem("arm", model, [d.origin_x, d.origin_y], d.cell_size, field, st, start[None], z, end[None], z.copy(), init)
    run(p, f"planar arm D={D} N={N}")
