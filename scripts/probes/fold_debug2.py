"""N = 2 chain of a planar arm: the folded diagonal tile of block 0 against a numpy emulation of levels 1 and 2"""
import ctypes as C, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import gpmp2_amd as g
from gpmp2_amd import engine, problems, datasets
from gpmp2_amd.settings import TrajOptimizerSetting
from gpmp2_amd.trajutils import initArmTrajStraightLine
from gpmp2_amd._capi import dptr
eng = engine.Engine()
for D in (4, 5):
    n = 2 * D
    arm = g.Arm(D, [0.3] * D, [0.0] * D, [0.0] * D)
    model = g.ArmModel(arm, [g.BodySphere(l, 0.05, (-0.1, 0, 0)) for l in range(D)])
    d = datasets.generate2Ddataset("TwoObstaclesDataset")
    field = datasets.signedDistanceField2D(d.map, d.cell_size)
    N = 2
    st = TrajOptimizerSetting(D)
    st.set_total_step(N); st.set_total_time(3.0); st.set_obs_check_inter(2); st.set_cost_sigma(0.1); st.set_epsilon(0.2)
    st.set_conf_prior_model(1e-3); st.set_vel_prior_model(1e-3); st.set_Qc_model(np.eye(D)); st.setGaussNewton()
    st.fixed_iterations = 1
    start, end = np.zeros(D), np.linspace(0.3, 0.8, D)
    init = initArmTrajStraightLine(start, end, N)[None]
    z = np.zeros((1, D))
    r, s = eng.robot(model), eng.sdf([d.origin_x, d.origin_y], d.cell_size, field)
    pl = eng.plan(r, s, st, 1)
    pl.set_problem(start[None], z, end[None], z, init)
    Hd, Ho, gr, _ = pl.linearize(init)
    Hd, Ho, b = Hd[0], Ho[0], -gr[0]            # Ho[i] = block (i+1, i)
    pl.optimize()
    tiles = np.zeros((N + 1) * 256); fac = np.zeros((N + 1) * 768)
    eng._ck(eng.lib.gpmp2mi_plan_debug_read(pl.h.ptr, 0, dptr(tiles), C.c_long(tiles.size)))
    eng._ck(eng.lib.gpmp2mi_plan_debug_read(pl.h.ptr, 1, dptr(fac), C.c_long(fac.size)))
    T0 = tiles[:256].reshape(16, 16)
    # numpy: level 1 (block 1), level 2 (block 2), folded block 0
    R1 = np.linalg.cholesky(Hd[1]).T
    Wl1 = np.linalg.solve(R1.T, Ho[0]); Wr1 = np.linalg.solve(R1.T, Ho[1].T); y1 = np.linalg.solve(R1.T, b[1])
    S0 = Hd[0] - Wl1.T @ Wl1; b0 = b[0] - Wl1.T @ y1
    S2 = Hd[2] - Wr1.T @ Wr1; b2 = b[2] - Wr1.T @ y1; C2l = -Wr1.T @ Wl1
    R2 = np.linalg.cholesky(S2).T
    Wl2 = np.linalg.solve(R2.T, C2l); y2 = np.linalg.solve(R2.T, b2)
    S0f = S0 - Wl2.T @ Wl2; b0f = b0 - Wl2.T @ y2
    print(f"D={D}: |tile0 matrix - expected| {np.abs(T0[:n, :n] - S0f).max():.2e} (scale {np.abs(S0f).max():.1e}); rhs {np.abs(T0[:n, 15] - b0f).max():.2e} (scale {np.abs(b0f).max():.1e})")
    F1 = fac[768:768 * 2].reshape(3, 16, 16)
    print(f"      W_l(1) {np.abs(F1[0][:n, :n] - Wl1).max():.2e}  y1 {np.abs(F1[0][:n, 15] - y1).max():.2e}  W_r(1) {np.abs(F1[1][:n, :n] - Wr1).max():.2e}")
    F2 = fac[768 * 2:768 * 3].reshape(3, 16, 16)
    print(f"      W_l(2) {np.abs(F2[0][:n, :n] - Wl2).max():.2e}  y2 {np.abs(F2[0][:n, 15] - y2).max():.2e}")
    E = T0[:n, :n] - S0f
    if np.abs(E).max() > 1e-9 * np.abs(S0f).max():
        print("      error pattern (rows, cols with |err| > 1e-9 scale):", np.argwhere(np.abs(E) > 1e-9 * np.abs(S0f).max())[:12].tolist())
        print("      unfolded S0 err:", np.abs(T0[:n, :n] - Hd[0]).max(), " after level-1 only:", np.abs(T0[:n, :n] - S0).max())
