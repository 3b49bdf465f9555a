"""GPU parity, planner surface: the reference-shaped functions and classes of gpmp2_amd.planner /
trajutils (BatchTrajOptimize*, CollisionCost*, ISAM2TrajOptimizer*, interpolateArmTraj,
SignedDistanceField / PlanarSDF) against the CPU oracle.  The flows follow the reference's own
examples: matlab/WAMPlannerExample.m, matlab/WAMReplannerExample.m:102-126 and
gpmp2/planner/tests/testTrajUtils.cpp."""
import numpy as np
import pytest

import gpmp2_amd as g
from gpmp2_amd import problems
from helpers import vec

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------ interpolateArmTraj & friends
def test_interpolate_arm_traj_known_answer_on_gpu(engine, golden):
    d = golden["traj_utils"]                                   # testTrajUtils.cpp:26-54
    values = {("x", 0): d["x"][0], ("x", 1): d["x"][1], ("v", 0): d["v"][0], ("v", 1): d["v"][1]}
    out = g.interpolateArmTraj(values, np.array(d["Qc"]), d["delta_t"], d["inter_step"])
    assert sorted(k[1] for k in out if k[0] == "x") == list(range(6))
    for i in range(6):
        np.testing.assert_allclose(out[("x", i)], d["expected_x"][i], atol=d["tol"])
        np.testing.assert_allclose(out[("v", i)], d["expected_v"][i], atol=d["tol"])


@pytest.mark.parametrize("D,lie", [(7, False), (2, False), (5, True), (3, True)])
def test_interpolate_traj_matches_oracle(engine, oracle, D, lie):
    rng = np.random.default_rng(100 + D)
    B, N, I = 6, 11, 4
    traj = rng.normal(size=(B, N + 1, 2 * D))
    if lie:
        traj[:, :, 2] = rng.uniform(-np.pi, np.pi, size=(B, N + 1))
    for rng_ in (None, (3, 9), (0, 1), (N - 1, N)):
        a = engine.interpolate_traj(D, lie, None, 0.2, I, traj, *(rng_ or ()))
        b = oracle.interpolate_traj(D, lie, None, 0.2, I, traj, *(rng_ or ()))
        assert a.shape == b.shape
        np.testing.assert_allclose(a, b, atol=1e-12)
    # inter_step = 0 is a plain copy of the range
    np.testing.assert_array_equal(engine.interpolate_traj(D, lie, None, 0.2, 0, traj, 2, 7), traj[:, 2:8])


def test_interpolate_traj_rejects_bad_ranges(engine):
    traj = np.zeros((1, 5, 4))
    for s, e in ((3, 3), (-1, 2), (2, 5), (4, 2)):
        with pytest.raises(g.engine.Gpmp2miError):
            engine.interpolate_traj(2, False, None, 0.1, 2, traj, s, e)


# ------------------------------------------------------------------ SDF classes
def test_sdf_classes(engine, golden):
    d = golden["sdf3d"]                                        # testSignedDistanceField.cpp:25-86
    slices = np.array(d["slices"])
    sdf = g.SignedDistanceField(d["origin"], d["cell_size"], 5, 5, 3)
    for z in range(3):
        sdf.initFieldData(z, slices[z])
    for q in d["queries"]:
        assert abs(sdf.getSignedDistance(q["point"]) - q["value"]) <= q["tol"]
    with pytest.raises(g.SDFQueryOutOfRange):
        sdf.getSignedDistance([10.0, 0.0, 0.0])
    with pytest.raises(RuntimeError):
        sdf.initFieldData(3, slices[0])
    with pytest.raises(RuntimeError):
        sdf.initFieldData(0, slices[0][:4])
    d2 = golden["sdf2d"]
    psdf = g.PlanarSDF(d2["origin"], d2["cell_size"], np.array(d2["data"]))
    for q in d2["queries"]:
        assert abs(psdf.getSignedDistance(q["point"]) - q["value"]) <= q["tol"]


def test_signed_distance_field_functions_and_vol_reader(engine, tmp_path):
    d = g.generate2Ddataset("OneObstacleDataset")
    np.testing.assert_array_equal(g.signedDistanceField2D(d.map, d.cell_size), g.datasets.signedDistanceField2D(d.map, d.cell_size))
    rng = np.random.default_rng(2)
    m3 = (rng.uniform(size=(12, 9, 7)) > 0.85).astype(float)
    np.testing.assert_array_equal(g.signedDistanceField3D(m3, 0.1), g.datasets.signedDistanceField3D(m3, 0.1))
    field = g.sdf3_zyx(g.signedDistanceField3D(m3, 0.1))                  # [z][y][x]
    nz, ny, nx = field.shape
    (tmp_path / "m.vol.head").write_text(f"{nx} {ny} {nz}\n0 0 0\n0.1\n")
    (tmp_path / "m.vol.data").write_text("\n".join(repr(float(field[z, y, x])) for x in range(nx) for y in range(ny) for z in range(nz)))
    sdf = g.readSDFvolfile(tmp_path / "m")
    assert (sdf.x_count(), sdf.y_count(), sdf.z_count()) == (nx, ny, nz)
    np.testing.assert_array_equal(sdf.raw_data(), field)
    assert abs(sdf.getSignedDistance([0.3, 0.2, 0.1]) - field[1, 2, 3]) < 1e-9


# ------------------------------------------------------------------ WAMPlannerExample-like flow
def _wam(total_step=10):
    p = problems.wam_restarts(B=1, total_step=total_step, obs_check_inter=4, sdf="40")
    sdf = g.SignedDistanceField(p.sdf_origin, p.sdf_cell, p.sdf_data.shape[1], p.sdf_data.shape[2], p.sdf_data.shape[0])
    for z in range(p.sdf_data.shape[0]):
        sdf.initFieldData(z, p.sdf_data[z])
    return p, sdf


def test_batch_traj_optimize_3d_arm_values_interface(engine, oracle):
    p, sdf = _wam()
    init_values = g.values_from_traj(g.initArmTrajStraightLine(p.start_conf[0], p.end_conf[0], p.setting.total_step))
    result = g.BatchTrajOptimize3DArm(p.model, sdf, p.start_conf[0], p.start_vel[0], p.end_conf[0], p.end_vel[0],
                                      init_values, p.setting)
    assert set(result) == set(init_values)
    ro, so = oracle.robot(p.model), oracle.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)
    ref = oracle.batch_optimize(ro, so, p.setting, p.start_conf, p.start_vel, p.end_conf, p.end_vel, p.init)
    np.testing.assert_allclose(g.traj_from_values(result, p.setting.total_step), ref["traj"][0], atol=1e-6)
    cost = g.CollisionCost3DArm(p.model, sdf, result, p.setting)
    np.testing.assert_allclose(cost, oracle.collision_cost(ro, so, p.setting.total_step, ref["traj"])[0], rtol=1e-8, atol=1e-12)
    # dense up-sampling of the result for execution (WAMPlannerExample.m plot section)
    dense = g.interpolateArmTraj(result, p.setting.Qc, p.setting.total_time / p.setting.total_step, 5)
    assert len(dense) == 2 * (p.setting.total_step * 6 + 1)
    # argument checks: wrong dof raises before touching the device
    with pytest.raises(ValueError):
        g.BatchTrajOptimize3DArm(p.model, sdf, p.start_conf[0][:6], p.start_vel[0], p.end_conf[0], p.end_vel[0],
                                 init_values, p.setting)


def test_isam2_traj_optimizer_3d_arm_replan(engine, oracle):
    """WAMReplannerExample.m:102-126 with the class interface"""
    p, sdf = _wam()
    D, N = 7, p.setting.total_step
    batch = g.BatchTrajOptimize3DArm(p.model, sdf, p.start_conf[0], p.start_vel[0], p.end_conf[0], p.end_vel[0],
                                     p.init[0], p.setting)
    isam = g.ISAM2TrajOptimizer3DArm(p.model, sdf, p.setting)
    with pytest.raises(RuntimeError):
        isam.initValues(batch)
    isam.initFactorGraph(p.start_conf[0], p.start_vel[0], p.end_conf[0], p.end_vel[0])
    isam.initValues(batch)
    isam.update()
    first = isam.values()
    ro, so = oracle.robot(p.model), oracle.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)
    st = problems.wam_setting(N, 4, "GN")
    st.fixed_iterations = 1
    ref0 = oracle.batch_optimize_xp(ro, so, st, p.start_conf, p.start_vel, p.end_conf, p.end_vel, batch[None], [[]], [1])
    np.testing.assert_allclose(first, ref0["traj"][0], atol=1e-6)     # one full Gauss-Newton step from the batch answer
    goal2 = np.array([-0.6, 0.94, 0, 1.6, 0, -0.919, 1.55])
    isam.fixConfigAndVel(5, first[5, :D], first[5, D:])
    isam.changeGoalConfigAndVel(goal2, np.zeros(D))
    isam.update()
    isam.update()
    got = isam.values()
    st.fixed_iterations = 2
    w = 1.0 / st.conf_prior_sigma ** 2
    priors = [[dict(state=5, conf=first[5, :D], Wc=w * np.eye(D), vel=first[5, D:], Wv=w * np.eye(D))]]
    ref = oracle.batch_optimize_xp(ro, so, st, p.start_conf, p.start_vel, goal2[None], p.end_vel, first[None], priors, [1])
    np.testing.assert_allclose(got, ref["traj"][0], atol=1e-6)
    np.testing.assert_allclose(got[5], first[5], atol=1e-3)
    np.testing.assert_allclose(got[-1, :D], goal2, atol=1e-3)
    # noisy state estimate with covariance + free end (addStateEstimate / removeGoalConfigAndVel)
    cov = 1e-4 * np.eye(D)
    isam.addStateEstimate(7, got[7, :D] + 0.02, cov, got[7, D:], cov)
    isam.removeGoalConfigAndVel()
    isam.update()
    st.fixed_iterations = 1
    priors[0].append(dict(state=7, conf=got[7, :D] + 0.02, Wc=np.linalg.inv(cov), vel=got[7, D:], Wv=np.linalg.inv(cov)))
    ref2 = oracle.batch_optimize_xp(ro, so, st, p.start_conf, p.start_vel, goal2[None], p.end_vel, got[None], priors, [0])
    np.testing.assert_allclose(isam.values(), ref2["traj"][0], atol=1e-6)


def test_mobile_arm_planner_and_interpolation(engine, oracle):
    p = problems.mobile_arm_config5()
    sdf = g.PlanarSDF(p.sdf_origin, p.sdf_cell, p.sdf_data)
    res = g.BatchTrajOptimizePose2MobileArm2D(p.model, sdf, p.start_conf[0], p.start_vel[0], p.end_conf[0], p.end_vel[0],
                                              p.init[0], p.setting)
    ro, so = oracle.robot(p.model), oracle.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)
    ref = oracle.batch_optimize(ro, so, p.setting, p.start_conf, p.start_vel, p.end_conf, p.end_vel, p.init)
    np.testing.assert_allclose(res, ref["traj"][0], atol=1e-6)
    dt = p.setting.total_time / p.setting.total_step
    dense = g.interpolatePose2MobileArmTraj(res, p.setting.Qc, dt, 3, 0, p.setting.total_step)
    np.testing.assert_allclose(dense, oracle.interpolate_traj(5, True, None, dt, 3, ref["traj"])[0], atol=1e-6)
    # straight-line initialisation through the helper (MobileArm2FactorGraphExample init section)
    init = g.initPose2VectorTrajStraightLine(p.start_conf[0][:3], p.start_conf[0][3:], p.end_conf[0][:3],
                                             p.end_conf[0][3:], p.setting.total_step)
    np.testing.assert_allclose(init[:, :5], p.init[0][:, :5], atol=1e-12)
