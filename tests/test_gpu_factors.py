"""GPU parity, factor level: every C-ABI factor entry point against (a) the reference's own
known-answer values (tests/golden) and (b) the CPU oracle on seeded random inputs.
Tolerance: 1e-9 absolute on fp64 residuals / Jacobians (BASELINE.md parity gate); the golden
literals keep the reference's own 1e-6 / 1e-3 / 1e-9."""
import math

import numpy as np
import pytest

import gpmp2_amd as g
from gpmp2_amd import problems
from helpers import arm_from_golden, num, sdf_to_err, vec

pytestmark = pytest.mark.gpu
TOL = 1e-9


def test_library_reports_a_gpu(engine):
    assert engine.device_count() >= 1


def test_sdf3d_golden_and_oracle(engine, oracle, golden):
    d = golden["sdf3d"]
    data = np.array(d["slices"])
    s, so = engine.sdf(d["origin"], d["cell_size"], data), oracle.sdf(d["origin"], d["cell_size"], data)
    for q in d["queries"]:
        dist, _, inr = engine.sdf_query(s, [q["point"]])
        assert inr[0] == 1 and abs(dist[0] - q["value"]) <= q["tol"]
    rng = np.random.default_rng(1)
    pts = rng.uniform([-0.25, -0.25, -0.15], [0.25, 0.25, 0.15], size=(500, 3))
    pts[:5] = [[0.2, 0.2, 0.1], [-0.2, -0.2, -0.1], [0.2, 0, 0], [0, 0.2, 0.05], [0.1, 0.1, 0.1]]  # faces
    a, b = engine.sdf_query(s, pts), oracle.sdf_query(so, pts)
    assert np.array_equal(a[2], b[2])
    np.testing.assert_allclose(a[0], b[0], atol=1e-9 * 4142)   # the fixture's 4142 typo cell
    np.testing.assert_allclose(a[1], b[1], atol=1e-9 * 41420)


def test_sdf2d_golden_and_oracle(engine, oracle, golden):
    d = golden["sdf2d"]
    data = np.array(d["data"])
    s, so = engine.sdf(d["origin"], d["cell_size"], data), oracle.sdf(d["origin"], d["cell_size"], data)
    for q in d["queries"]:
        dist, _, _ = engine.sdf_query(s, [q["point"]])
        assert abs(dist[0] - q["value"]) <= q["tol"]
    pts = np.random.default_rng(2).uniform(-0.25, 0.25, size=(300, 2))
    pts[:3] = [[0.2, 0.2], [-0.2, -0.2], [0.2, -0.1]]
    a, b = engine.sdf_query(s, pts), oracle.sdf_query(so, pts)
    assert np.array_equal(a[2], b[2])
    np.testing.assert_allclose(a[0], b[0], atol=TOL)
    np.testing.assert_allclose(a[1], b[1], atol=1e-8)


def test_sdf_gtsam_layout_equals_zyx(engine):
    rng = np.random.default_rng(3)
    data = rng.normal(size=(4, 5, 6))                      # [z][y][x]
    gts = np.ascontiguousarray(np.transpose(data, (0, 2, 1)))  # [z][x][y] = column-major slices
    a = engine.sdf([0, 0, 0], 0.5, data)
    b = engine.sdf([0, 0, 0], 0.5, gts.reshape(4, 5, 6), layout=1)
    pts = rng.uniform(0, [2.5, 2.0, 1.5], size=(100, 3))
    np.testing.assert_array_equal(engine.sdf_query(a, pts)[0], engine.sdf_query(b, pts)[0])


@pytest.mark.parametrize("which", ["two_link", "three_link", "wam"])
def test_arm_fk_golden_and_oracle(engine, oracle, golden, which):
    d = golden["arm_fk"][which]
    if which == "two_link":
        arm = g.Arm(2, d["a"], d["alpha"], d["d"], g.pose3(g.rot_yaw(num(d["base_yaw"])), d["base_xyz"]))
        q0 = vec(d["cases"][1]["q"])
    else:
        arm = g.Arm(len(d["a"]), d["a"], vec(d["alpha"]), d["d"])
        q0 = vec(d["q"])
    model = g.ArmModel(arm, [])
    r, ro = engine.robot(model), oracle.robot(model)
    poses, _ = engine.forward_kinematics(r, q0)
    if which == "two_link":
        c = d["cases"][1]
        for l in range(2):
            np.testing.assert_allclose(poses[0, l], g.pose3(g.rot_yaw(num(c["yaw"][l])), c["xyz"][l]), atol=c["tol"])
    else:
        np.testing.assert_allclose(poses[0, :, :3, 3], np.array(d["xyz"]), atol=d["tol"])
    q = np.random.default_rng(4).uniform(-3, 3, size=(64, arm.dof()))
    (pa, ja), (pb, jb) = engine.forward_kinematics(r, q), oracle.forward_kinematics(ro, q)
    scale = max(1.0, np.abs(pb).max())
    np.testing.assert_allclose(pa, pb, atol=TOL * scale)
    np.testing.assert_allclose(ja, jb, atol=TOL * scale)


def test_sphere_centers_golden_and_oracle(engine, oracle, golden):
    d = golden["arm_model"]
    arm = g.Arm(2, d["a"], d["alpha"], d["d"], g.pose3(t=d["base_xyz"]))
    model = g.ArmModel(arm, [g.BodySphere(int(s[0]), s[1], s[2:5]) for s in d["spheres"]])
    r = engine.robot(model)
    for c in d["cases"]:
        ctr, _ = engine.sphere_centers(r, vec(c["q"]))
        np.testing.assert_allclose(ctr[0], np.array(c["centers"]), atol=1e-9)
    wam = g.generateArm("WAMArm")
    r, ro = engine.robot(wam), oracle.robot(wam)
    q = np.random.default_rng(5).uniform(-2.5, 2.5, size=(128, 7))
    (ca, ja), (cb, jb) = engine.sphere_centers(r, q), oracle.sphere_centers(ro, q)
    np.testing.assert_allclose(ca, cb, atol=TOL)
    np.testing.assert_allclose(ja, jb, atol=TOL)


def test_point_robot_golden_and_oracle(engine, oracle, golden):
    d = golden["point_robot"]
    model = g.generatePointRobot(1.5)
    r, ro = engine.robot(model), oracle.robot(model)
    poses, _ = engine.forward_kinematics(r, d["q"])
    np.testing.assert_allclose(poses[0, 0], g.pose3(t=d["pose_xyz"]), atol=1e-12)
    q = np.random.default_rng(6).uniform(-10, 10, size=(20, 2))
    for fa, fb in ((engine.forward_kinematics, oracle.forward_kinematics), (engine.sphere_centers, oracle.sphere_centers)):
        a, b = fa(r, q), fb(ro, q)
        np.testing.assert_allclose(a[0], b[0], atol=TOL)
        np.testing.assert_allclose(a[1], b[1], atol=TOL)


def test_obstacle_factor_golden(engine, golden):
    d = golden["obstacle_sdf_factor_arm"]
    s = engine.sdf(d["origin"], d["cell_size"], np.array(d["slices"]))
    r = engine.robot(arm_from_golden(d))
    rad, gp = d["spheres"][0][1], d["gp"]
    for c in d["unary_cases"]:
        err, _ = engine.obstacle_factor(r, s, d["epsilon"], vec(c["q"]))
        np.testing.assert_allclose(err[0], sdf_to_err(c["sdf_expected"], d["epsilon"] + rad), atol=d["tol"])
    for c in d["gp_cases"]:
        a = [vec(c[k]) for k in ("q1", "qdot1", "q2", "qdot2")]
        err, _ = engine.obstacle_gp_factor(r, s, d["epsilon"], None, gp["delta_t"], gp["tau"], *a)
        np.testing.assert_allclose(err[0], sdf_to_err(c["sdf_expected"], d["epsilon"] + rad), atol=d["tol"])


def test_obstacle_planar_factor_golden(engine, golden):
    d = golden["obstacle_planar_sdf_factor_arm"]
    s = engine.sdf(d["origin"], d["cell_size"], np.array(d["field"]))
    r = engine.robot(arm_from_golden(d))
    rad = d["spheres"][0][1]
    for c in d["cases"]:
        err, _ = engine.obstacle_factor(r, s, d["epsilon"], vec(c["q"]))
        np.testing.assert_allclose(err[0], sdf_to_err(vec(c["sdf_expected"]), d["epsilon"] + rad), atol=d["tol"])


@pytest.mark.parametrize("case", ["wam3d", "arm3_2d", "point2d"])
def test_obstacle_factors_vs_oracle(engine, oracle, case):
    rng = np.random.default_rng(7)
    if case == "wam3d":
        p = problems.wam_restarts(B=1, sdf="40")
        q = rng.uniform(-2.0, 2.0, size=(256, 7))
        dt, tau = 0.02, 0.02 / 6 * 2
    elif case == "arm3_2d":
        p = problems.arm3_planner()
        q = rng.uniform(-2.0, 2.0, size=(256, 3))
        dt, tau = 0.1, 0.025
    else:
        p = problems.point_robot_2d()
        q = rng.uniform([-19, -9], [19, 19], size=(256, 2))
        dt, tau = 0.5, 0.2
    D = p.model.dof()
    r, ro = engine.robot(p.model), oracle.robot(p.model)
    s, so = engine.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data), oracle.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)
    eps = p.setting.epsilon
    (ea, ha), (eb, hb) = engine.obstacle_factor(r, s, eps, q), oracle.obstacle_factor(ro, so, eps, q)
    assert (eb > 0).sum() > 10          # the sample actually exercises the hinge
    np.testing.assert_allclose(ea, eb, atol=TOL)
    np.testing.assert_allclose(ha, hb, atol=1e-8)
    v = rng.normal(size=(256, D))
    q2 = q + dt * v + 0.01 * rng.normal(size=(256, D))
    v2 = v + 0.1 * rng.normal(size=(256, D))
    (ea, Ha), (eb, Hb) = (f(rr, ss, eps, None, dt, tau, q, v, q2, v2) for f, rr, ss in
                          ((engine.obstacle_gp_factor, r, s), (oracle.obstacle_gp_factor, ro, so)))
    np.testing.assert_allclose(ea, eb, atol=TOL)
    for k in range(4):
        np.testing.assert_allclose(Ha[k], Hb[k], atol=1e-8)


def test_gp_prior_and_interpolator(engine, oracle, golden):
    d = golden["gp_interpolator_linear"]
    Qc = d["Qc_scale"] * np.eye(3)
    for c in d["cases"]:
        conf, _ = engine.gp_interpolate(3, False, Qc, d["delta_t"], d["tau"], c["p1"], c["v1"], c["p2"], c["v2"])
        np.testing.assert_allclose(conf[0], c["expect"], atol=d["tol"])
    for k in range(1, 5):   # gpmp2/planner/tests/testTrajUtils.cpp:26-54
        conf, vel = engine.gp_interpolate(2, False, 0.01 * np.eye(2), 0.1, 0.1 * k / 5, [0, 0], [10, 0], [1, 0], [10, 0])
        np.testing.assert_allclose(conf[0], [0.2 * k, 0], atol=1e-6)
        np.testing.assert_allclose(vel[0], [10, 0], atol=1e-6)
    rng = np.random.default_rng(8)
    a = [rng.normal(size=(50, 5)) for _ in range(4)]
    A = rng.normal(size=(5, 5))
    Q = A @ A.T + 5 * np.eye(5)
    for x, y in zip(engine.gp_interpolate(5, False, Q, 0.3, 0.11, *a), oracle.gp_interpolate(5, False, Q, 0.3, 0.11, *a)):
        np.testing.assert_allclose(x, y, atol=1e-9 * 10)
    (ea, Ha), (eb, Hb) = engine.gp_prior_factor(5, False, 0.3, *a), oracle.gp_prior_factor(5, False, 0.3, *a)
    np.testing.assert_allclose(ea, eb, atol=TOL)
    for k in range(4):
        np.testing.assert_allclose(Ha[k], Hb[k], atol=TOL)
    g0 = golden["gp_prior_linear"]
    for c in g0["zero_error_cases"]:
        err, _ = engine.gp_prior_factor(3, False, g0["delta_t"], c["p1"], c["v1"], c["p2"], c["v2"])
        np.testing.assert_allclose(err[0], 0.0, atol=g0["tol"])


def test_joint_limit_factor(engine, oracle, golden):
    d = golden["joint_limit"]
    for c in d["cases"]:
        err, _ = engine.joint_limit_factor(d["down"], d["up"], d["thresh"], c["conf"])
        np.testing.assert_allclose(err[0], c["err"], atol=d["tol"])
    x = np.random.default_rng(9).uniform(-12, 12, size=(200, 2))
    x[:4] = [[-3, -8], [3, 8], [-3 - 1e-12, 8 + 1e-12], [0, 0]]     # the strict / non-strict edges
    a, b = engine.joint_limit_factor(d["down"], d["up"], d["thresh"], x), oracle.joint_limit_factor(d["down"], d["up"], d["thresh"], x)
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])


def test_empty_batches_and_bad_arguments(engine):
    model = g.generateArm("WAMArm")
    r = engine.robot(model)
    c, J = engine.sphere_centers(r, np.zeros((0, 7)))
    assert c.shape == (0, 16, 3)
    from gpmp2_amd.engine import Gpmp2miError
    bad = g.ArmModel(g.Arm(2, [1, 1], [0, 0], [0, 0]), [g.BodySphere(0, 0.1, (0, 0, 0))])
    bad.spheres[0].link_id = 5
    with pytest.raises(Gpmp2miError):
        engine.robot(bad)


# ------------------------------------------------------------------ SDF construction on the device
@pytest.mark.parametrize("shape", [(17, 23), (6, 9, 11), (1, 7), (5, 1, 8), (3, 70, 130), (150, 3)])
def test_sdf_from_occupancy_bit_exact(engine, oracle, shape):
    rng = np.random.default_rng(sum(shape))
    occ = (rng.uniform(size=shape) > 0.9).astype(float)
    np.testing.assert_array_equal(engine.sdf_field_from_occupancy(occ, 0.05), oracle.sdf_field_from_occupancy(occ, 0.05))
    for deg in (np.zeros(shape), np.ones(shape)):
        np.testing.assert_array_equal(engine.sdf_field_from_occupancy(deg, 0.05), 1000.0 * np.ones(shape))


def test_sdf_from_occupancy_full_size_datasets(engine):
    """full-size maps of the reference's examples against the scipy transform its python utilities use"""
    import gpmp2_amd as g
    d2 = g.generate2Ddataset("MobileMap1")
    np.testing.assert_array_equal(engine.sdf_field_from_occupancy(d2.map, d2.cell_size),
                                  g.datasets.signedDistanceField2D(d2.map, d2.cell_size))
    d3 = g.generate3Ddataset("WAMDeskDataset")
    occ = g.sdf3_zyx(d3.map)
    want = g.sdf3_zyx(g.datasets.signedDistanceField3D(d3.map, d3.cell_size))
    np.testing.assert_array_equal(engine.sdf_field_from_occupancy(occ, d3.cell_size), want)
    # straight into a handle: same lookups as a handle built from the host field
    org = [d3.origin_x, d3.origin_y, d3.origin_z]
    a = engine.sdf_from_occupancy(org, d3.cell_size, occ)
    b = engine.sdf(org, d3.cell_size, want)
    pts = np.random.default_rng(0).uniform(-0.4, 0.4, size=(500, 3)) + np.array(org) + 0.5 * d3.cell_size * np.array(occ.shape[::-1])
    for x, y in zip(engine.sdf_query(a, pts), engine.sdf_query(b, pts)):
        np.testing.assert_array_equal(x, y)
    got = engine.sdf_field(a)
    assert got["dim"] == 3 and got["cell_size"] == d3.cell_size
    np.testing.assert_array_equal(got["data"], want)


def test_sdf_read_vol(engine, tmp_path):
    """readSDFvolfile (gpmp2/utils/fileUtils.cpp:17-62): head = cols rows z / origin / resolution,
    data = text with x outermost, then y, then z"""
    rng = np.random.default_rng(3)
    nx, ny, nz = 4, 3, 5
    field = rng.normal(size=(nz, ny, nx)).round(6)
    pre = tmp_path / "scene"
    (tmp_path / "scene.vol.head").write_text(f"{nx} {ny} {nz}\n-1.0 -0.5 0.25\n0.1\n")
    (tmp_path / "scene.vol.data").write_text(" ".join(f"{field[z, y, x]:.6f}" for x in range(nx) for y in range(ny) for z in range(nz)))
    h = engine.sdf_read_vol(pre)
    got = engine.sdf_field(h)
    np.testing.assert_array_equal(got["data"], field)
    np.testing.assert_array_equal(got["origin"], [-1.0, -0.5, 0.25])
    assert got["cell_size"] == 0.1
    with pytest.raises(Exception):
        engine.sdf_read_vol(tmp_path / "missing")
    (tmp_path / "short.vol.head").write_text(f"{nx} {ny} {nz}\n0 0 0\n0.1\n")
    (tmp_path / "short.vol.data").write_text("1.0 2.0")
    with pytest.raises(Exception):
        engine.sdf_read_vol(tmp_path / "short")


# ------------------------------------------------------------------ SelfCollision / goal / workspace priors
def _gold_arm(d):
    arm = g.Arm(d["arm"]["dof"], vec(d["arm"]["a"]), vec(d["arm"]["alpha"]), vec(d["arm"]["d"]), g.pose3(t=d["arm"]["base_xyz"]))
    return g.ArmModel(arm, [g.BodySphere(int(s[0]), s[1], (s[2], s[3], s[4])) for s in d["spheres"]])


def test_self_collision_factor_gpu(engine, oracle, golden):
    d = golden["self_collision"]                               # testSelfCollision.cpp:21-50
    model = _gold_arm(d)
    r, ro = engine.robot(model), oracle.robot(model)
    err, H = engine.self_collision_factor(r, d["data"], vec(d["q"]))
    np.testing.assert_allclose(err[0], d["expected"], atol=d["tol"])
    rng = np.random.default_rng(4)
    q = rng.uniform(-2, 2, size=(40, 3))
    data = [[0, 1, 2.0, 0.1], [2, 3, 5.0, 0.1], [0, 3, 0.3, 0.2], [1, 3, 0.1, 0.2]]
    a, b = engine.self_collision_factor(r, data, q), oracle.self_collision_factor(ro, data, q)
    np.testing.assert_allclose(a[0], b[0], atol=1e-12)
    np.testing.assert_allclose(a[1], b[1], atol=1e-11)
    assert (b[0] == 0).any() and (b[0] > 0).any()             # both hinge branches exercised
    # WAM sphere model: ids refer to the description's order although the engine sorts spheres by link
    wam = g.generateArm("WAMArm")
    rw, rwo = engine.robot(wam), oracle.robot(wam)
    qw = rng.uniform(-1.5, 1.5, size=(25, 7))
    dw = [[0, 15, 0.4, 0.1], [3, 12, 0.2, 0.1], [5, 9, 0.6, 0.1]]
    a, b = engine.self_collision_factor(rw, dw, qw), oracle.self_collision_factor(rwo, dw, qw)
    np.testing.assert_allclose(a[0], b[0], atol=1e-12)
    np.testing.assert_allclose(a[1], b[1], atol=1e-11)
    with pytest.raises(g.engine.Gpmp2miError):
        engine.self_collision_factor(rw, [[0, 99, 0.1, 0.1]], qw)


def test_goal_and_workspace_priors_gpu(engine, oracle, golden):
    d = golden["goal_factor_arm"]                              # testGoalFactorArm.cpp:26-70
    r = engine.robot(_gold_arm(d))
    for c in d["cases"]:
        err, _ = engine.goal_factor_arm(r, c["goal"], vec(c["q"]))
        np.testing.assert_allclose(err[0], c["expected"], atol=d["tol"])
    d = golden["workspace_pose"]                               # testGaussianPriorWorkspacePose.cpp:27-45
    r = engine.robot(_gold_arm(d))
    err, _ = engine.workspace_prior_factor(r, 2, d["joint"], np.eye(4), vec(d["q"]))
    np.testing.assert_allclose(err[0], d["expected"], atol=d["tol"])
    d = golden["workspace_orientation"]                        # ...Orientation.cpp:27-46
    z = num(d["des_rzryrx"][2])
    des = np.eye(4)
    des[:2, :2] = [[math.cos(z), -math.sin(z)], [math.sin(z), math.cos(z)]]
    err, _ = engine.workspace_prior_factor(r, 1, d["joint"], des, vec(d["q"]))
    np.testing.assert_allclose(err[0], d["expected"], atol=d["tol"])
    # batched parity on the WAM arm and on a mobile manipulator, all three modes, several links
    rng = np.random.default_rng(23)
    A = np.linalg.qr(rng.normal(size=(3, 3)))[0]
    A *= np.sign(np.linalg.det(A))
    des = np.eye(4)
    des[:3, :3] = A
    des[:3, 3] = [0.3, -0.2, 0.5]
    for model, joints in ((g.generateArm("WAMArm"), (0, 3, 6)), (g.generateMobileArm("SimpleTwoLinksArm"), (0, 2))):
        r, ro = engine.robot(model), oracle.robot(model)
        q = rng.uniform(-1.5, 1.5, size=(30, model.dof()))
        q[0] = 0.0                                              # identity-ish poses: the small-angle branches
        for mode in (0, 1, 2):
            for joint in joints:
                a = engine.workspace_prior_factor(r, mode, joint, des, q)
                b = oracle.workspace_prior_factor(ro, mode, joint, des, q)
                np.testing.assert_allclose(a[0], b[0], atol=1e-11)
                np.testing.assert_allclose(a[1], b[1], atol=1e-10)
    with pytest.raises(g.engine.Gpmp2miError):
        engine.workspace_prior_factor(r, 2, 7, des, q)


def test_vehicle_dynamics_factor_gpu(engine, oracle, golden):
    d = golden["vehicle_dynamics"]                             # testVehicleDynamics.cpp:23-169
    for lie, cases in ((True, d["lie_cases"]), (False, d["vector_cases"])):
        for c in cases:
            err, _, _ = engine.vehicle_dynamics_factor(lie, vec(c["p"]), vec(c["v"]))
            if c["expected"] is not None:
                assert abs(err[0] - c["expected"]) <= d["tol"]
        rng = np.random.default_rng(8)
        q, v = rng.normal(size=(30, 5)), rng.normal(size=(30, 5))
        for x, y in zip(engine.vehicle_dynamics_factor(lie, q, v), oracle.vehicle_dynamics_factor(lie, q, v)):
            np.testing.assert_allclose(x, y, atol=1e-13)
