"""Parity at the sizes that are timed, and off the headline path, at the contract of SURVEY.md 8(d): identical
iteration counts and status, final error 1e-9 relative, final trajectory 1e-6 absolute.  A trajectory above 1e-6 is
admitted only when the oracle's own sensitivity to a 2-ulp perturbation of its initial values explains it
(tests/parity_bound.py) -- never by a wider gate -- and the number of such cases is capped.

  * the randomised robot-kind sweep (all six Pose2 robot kinds, N 8..64, I 0..4, GN / LM / Dogleg, B 1..16; the
    generator of scripts/stress_parity_robots.py, replayed by tests/sweep_cases.py)
  * mobile base + WAM (dof 10, the robot BatchTrajOptimizePose2MobileArm exists for,
    gpmp2/planner/BatchTrajOptimizer.cpp:79-89) at N = 100, I = 5, 64 restarts, GN and LM
  * BASELINE config 4 at full size: 128 receding-horizon windows per GPU, N = 100, I = 5, Synth200 field, 3 fixed
    Gauss-Newton iterations (matlab/WAMReplannerExample.m:100-126, gpmp2/planner/ISAM2TrajOptimizer-inl.h:100-115)"""
import os

import numpy as np
import pytest

import gpmp2_amd as g
from gpmp2_amd import problems
from gpmp2_amd.settings import TrajOptimizerSetting
from parity_bound import CONTRACT, check_contract
from sweep_cases import robot_sweep_cases

pytestmark = pytest.mark.gpu

# Sweep cases with a trajectory above 1e-6: 8 of 50 in profiles/r03_parity_sensitivity.txt.  WHICH ones cross 1e-6
# changes with every change of the rounding (the column-form elimination of round 3 moved two in and two out) -- what
# does not change is that each of them is inside K_SELF x the oracle's own 2-ulp sensitivity, which check_contract
# asserts per trajectory.  The count is capped so that a systematic loss of accuracy cannot hide behind the bound.
MAX_SENSITIVE_CASES = 10


def test_robot_kind_sweep_meets_the_contract(engine, oracle):
    over = {}
    for case, name, opt, p in robot_sweep_cases(50):
        rep = check_contract(engine, oracle, p, label=f"sweep case {case} ({name}, {opt})", final_error_rtol=1e-8)
        if rep["over"].size:
            over[case] = (float(rep["d_gpu"].max()), float(rep["d_self"][rep["over"]].max()))
    print("sweep cases above 1e-6 (gpu-vs-oracle, oracle-vs-perturbed-oracle):", over)
    assert len(over) <= MAX_SENSITIVE_CASES, f"{len(over)} sweep cases above the {CONTRACT} contract: {sorted(over)}"


def mobile_wam_problem(opt, B=64, N=100, inter=5):
    """the problem of scripts/wide_time.py"""
    wam = g.generateArm("WAMArm")
    a7 = wam.fk_model()
    mob = g.Pose2MobileArm(g.Arm(7, a7.a, a7.alpha, a7.d), g.pose3(t=(0.0, 0.0, 0.3)))
    model = g.ArmModel(mob, [g.BodySphere(0, 0.3, (0, 0, 0.15))] + [g.BodySphere(s.link_id + 1, s.radius, s.center) for s in wam.spheres])
    origin, cell, data = problems.small3d_sdf(40)
    origin, cell, data = list(np.array(origin) * 3), cell * 3, data * 3
    D = 10
    st = TrajOptimizerSetting(D)
    st.set_total_step(N); st.set_total_time(10.0); st.set_obs_check_inter(inter); st.set_cost_sigma(0.05); st.set_epsilon(0.3)
    st.set_conf_prior_model(1e-3); st.set_vel_prior_model(1e-3); st.set_Qc_model(np.eye(D)); st.set_max_iter(50)
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
    return problems.Problem("mobile_wam", model, origin, cell, data, st, np.repeat(start[None], B, 0), z,
                            np.repeat(end[None], B, 0), z.copy(), init)


@pytest.mark.parametrize("opt", ["GN", "LM"])
def test_mobile_wam_full_size(engine, oracle, opt):
    rep = check_contract(engine, oracle, mobile_wam_problem(opt), label=f"mobile WAM N=100 I=5 B=64 {opt}",
                         final_error_rtol=1e-8)
    o = rep["over"]
    print(f"mobile WAM {opt}: max |dtraj| {rep['d_gpu'].max():.2e}; {o.size} of 64 trajectories above 1e-6; gpu-vs-oracle / "
          f"oracle-vs-perturbed-oracle there: {np.round(rep['d_gpu'][o] / rep['d_self'][o], 2).tolist() if o.size else '-'}")


def test_config4_full_size_windows(engine, oracle):
    """every one of the 128 windows a GPU takes in BASELINE config 4, at the size bench.py --workload windows times"""
    base = problems.wam_restarts(B=1)
    r, s = engine.robot(base.model), engine.sdf(base.sdf_origin, base.sdf_cell, base.sdf_data)
    args = lambda q: (q.start_conf, q.start_vel, q.end_conf, q.end_vel)
    sol = engine.batch_optimize(r, s, base.setting, *args(base), base.init)["traj"][0]
    p = problems.wam_windows(sol, B=128)
    res = engine.batch_optimize(r, s, p.setting, *args(p), p.init)
    ro, so = oracle.robot(p.model), oracle.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)
    ref = oracle.batch_optimize(ro, so, p.setting, *args(p), p.init, nthreads=min(os.cpu_count() or 1, 64))
    assert list(res["iters"]) == [3] * 128 == list(ref["iters"])
    assert list(res["status"]) == list(ref["status"])
    np.testing.assert_allclose(res["traj"], ref["traj"], atol=CONTRACT)
    # final error: the oracle's E at the GPU's own trajectories to 1e-9; against the oracle's own run 1e-8 (three
    # Gauss-Newton steps from a warm start turn a 1e-10 difference of the step into ~3e-9 of the error on one window)
    np.testing.assert_allclose(res["final_error"], oracle.graph_error(ro, so, p.setting, *args(p), res["traj"]), rtol=1e-9)
    np.testing.assert_allclose(res["final_error"], ref["final_error"], rtol=1e-8)
    # size-independent properties: the windows start where they were told to and three iterations lower the error
    np.testing.assert_allclose(res["traj"][:, 0, :7], p.start_conf, atol=1e-3)
    np.testing.assert_allclose(res["traj"][:, -1, :7], p.end_conf, atol=1e-3)
    e0 = engine.graph_error(r, s, p.setting, *args(p), p.init)
    assert np.all(res["final_error"] < e0)
