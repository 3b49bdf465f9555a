"""The reference's own factor tests, re-stated against the reference-named classes of gpmp2_amd.factors
(same fixtures, same expected literals, same tolerances; GPU evaluation).  Sources:
gpmp2/obstacle/tests/testObstacleSDFFactorArm.cpp, testObstaclePlanarSDFFactorArm.cpp, testSelfCollision.cpp,
gpmp2/kinematics/tests/testGoalFactorArm.cpp, testGaussianPriorWorkspacePose.cpp,
testJointLimitFactorVector.cpp, gpmp2/gp/tests/testGaussianProcessInterpolatorLinear.cpp."""
import math

import numpy as np
import pytest

import gpmp2_amd as g
from helpers import arm_from_golden, numeric_jacobian, num, sdf_to_err, vec

pytestmark = pytest.mark.gpu


def _arm(d, spheres=True):
    arm = g.Arm(d["arm"]["dof"], vec(d["arm"]["a"]), vec(d["arm"]["alpha"]), vec(d["arm"]["d"]), g.pose3(t=d["arm"]["base_xyz"]))
    return g.ArmModel(arm, [g.BodySphere(int(s[0]), s[1], (s[2], s[3], s[4])) for s in d["spheres"]] if spheres else [])


def test_obstacle_sdf_factor_arm_and_gp(engine, golden):
    d = golden["obstacle_sdf_factor_arm"]                      # testObstacleSDFFactorArm.cpp:40-120, ...GPArm.cpp:39-149
    slices = np.array(d["slices"])
    sdf = g.SignedDistanceField(d["origin"], d["cell_size"], slices.shape[1], slices.shape[2], slices.shape[0])
    for z in range(slices.shape[0]):
        sdf.initFieldData(z, slices[z])
    arm = arm_from_golden(d)
    rad = d["spheres"][0][1]
    factor = g.ObstacleSDFFactorArm(0, arm, sdf, d["cost_sigma"], d["epsilon"])
    for c in d["unary_cases"]:
        q = vec(c["q"])
        err, H = factor.evaluateError(q, jacobians=True)
        np.testing.assert_allclose(err, sdf_to_err(c["sdf_expected"], d["epsilon"] + rad), atol=d["tol"])
        np.testing.assert_allclose(H, numeric_jacobian(lambda x: factor.evaluateError(x), q, 1e-6), atol=d["tol"])
    gp = g.ObstacleSDFFactorGPArm(0, 0, 0, 0, arm, sdf, d["cost_sigma"], d["epsilon"], np.eye(2), d["gp"]["delta_t"], d["gp"]["tau"])
    for c in d["gp_cases"]:
        args = [vec(c[k]) for k in ("q1", "qdot1", "q2", "qdot2")]
        err, H1, H2, H3, H4 = gp.evaluateError(*args, jacobians=True)
        np.testing.assert_allclose(err, sdf_to_err(c["sdf_expected"], d["epsilon"] + rad), atol=d["tol"])
        for k, H in enumerate((H1, H2, H3, H4)):
            def f(x, k=k):
                a = list(args)
                a[k] = x
                return gp.evaluateError(*a)
            np.testing.assert_allclose(H, numeric_jacobian(f, args[k], 1e-6), atol=1e-6)


def test_self_collision_arm(engine, golden):
    d = golden["self_collision"]
    factor = g.SelfCollisionArm(0, _arm(d), d["data"])
    q = vec(d["q"])
    actual, H_act = factor.evaluateError(q, jacobians=True)
    H_exp = numeric_jacobian(lambda x: factor.evaluateError(x), q, 1e-6)
    np.testing.assert_allclose(actual, d["expected"], atol=1e-6)
    np.testing.assert_allclose(H_act, H_exp, atol=1e-6)


def test_goal_factor_arm(engine, golden):
    d = golden["goal_factor_arm"]
    arm = _arm(d, spheres=False).fk_model()
    for c in d["cases"]:
        factor = g.GoalFactorArm(0, None, arm, c["goal"])
        q = vec(c["q"])
        actual, H_act = factor.evaluateError(q, jacobians=True)
        H_exp = numeric_jacobian(lambda x: factor.evaluateError(x), q, 1e-6)
        np.testing.assert_allclose(actual, c["expected"], atol=1e-6)
        np.testing.assert_allclose(H_act, H_exp, atol=1e-6)


def test_gaussian_prior_workspace_pose_arm(engine, golden):
    d = golden["workspace_pose"]
    factor = g.GaussianPriorWorkspacePoseArm(0, _arm(d, spheres=False), d["joint"], np.eye(4))
    q = vec(d["q"])
    actual, H_act = factor.evaluateError(q, jacobians=True)
    H_exp = numeric_jacobian(lambda x: factor.evaluateError(x), q, 1e-6)
    np.testing.assert_allclose(actual, d["expected"], atol=1e-6)
    np.testing.assert_allclose(H_act, H_exp, atol=1e-6)
    d = golden["workspace_orientation"]
    z = num(d["des_rzryrx"][2])
    R = np.array([[math.cos(z), -math.sin(z), 0], [math.sin(z), math.cos(z), 0], [0, 0, 1]])
    fo = g.GaussianPriorWorkspaceOrientationArm(0, _arm(d, spheres=False), d["joint"], R)
    np.testing.assert_allclose(fo.evaluateError(vec(d["q"])), d["expected"], atol=1e-6)


def test_limit_factors_and_gp_classes(engine, golden):
    d = golden["joint_limit"]                                  # testJointLimitFactorVector.cpp:25-158
    factor = g.JointLimitFactorVector(0, None, d["down"], d["up"], d["thresh"])
    for c in d["cases"]:
        conf = vec(c["conf"])
        actual, H_act = factor.evaluateError(conf, jacobians=True)
        np.testing.assert_allclose(actual, c["err"], atol=d["tol"])
        if not np.any(np.abs(np.abs(conf) - 3.0) < 1e-9):      # away from the hinge kinks
            np.testing.assert_allclose(H_act, numeric_jacobian(lambda x: factor.evaluateError(x), conf, 1e-6), atol=1e-6)
    with pytest.raises(RuntimeError):
        g.JointLimitFactorVector(0, None, [0, 0], [1, 1, 1], [0.1, 0.1])
    vf = g.VelocityLimitFactorVector(0, None, d["up"], d["thresh"])
    np.testing.assert_allclose(vf.evaluateError([10.0, -10.0]), [7.0, 2.0], atol=d["tol"])
    gi = golden["gp_interpolator_linear"]                      # testGaussianProcessInterpolatorLinear.cpp:47-158
    Qc = gi["Qc_scale"] * np.eye(gi["dof"])
    base = g.GaussianProcessInterpolatorLinear(Qc, gi["delta_t"], gi["tau"])
    for c in gi["cases"]:
        np.testing.assert_allclose(base.interpolatePose(c["p1"], c["v1"], c["p2"], c["v2"]), c["expect"], atol=gi["tol"])
    gpp = golden["gp_prior_linear"]                            # testGaussianProcessPriorLinear.cpp:29-202
    prior = g.GaussianProcessPriorLinear(0, 0, 0, 0, gpp["delta_t"], gpp["Qc_scale"] * np.eye(gpp["dof"]))
    for c in gpp["zero_error_cases"]:
        np.testing.assert_allclose(prior.evaluateError(c["p1"], c["v1"], c["p2"], c["v2"]), np.zeros(6), atol=gpp["tol"])
    c = gpp["random_case"]
    out = prior.evaluateError(c["p1"], c["v1"], c["p2"], c["v2"], jacobians=True)
    args = [vec(c[k]) for k in ("p1", "v1", "p2", "v2")]
    for k in range(4):
        def f(x, k=k):
            a = list(args)
            a[k] = x
            return prior.evaluateError(*a)
        np.testing.assert_allclose(out[1 + k], numeric_jacobian(f, args[k], 1e-6), atol=1e-6)
