"""ctypes mirror of include/gpmp2mi.h: POD structs + helpers that marshal numpy arrays.

Only the structs and marshalling live here; the product library is loaded by `engine.py`.
(The test-only CPU oracle re-uses these structs through tests/oracle.py -- it is never imported
from this package.)
"""
from __future__ import annotations

import ctypes as C

import numpy as np

c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)

OPT_GAUSS_NEWTON, OPT_LM, OPT_DOGLEG = 0, 1, 2
SDF_LAYOUT_ZYX, SDF_LAYOUT_GTSAM = 0, 1
MAX_DOF = 10
MAX_SPHERES = 64

STATUS_NAMES = {0: "converged", 1: "max_iter", 2: "rolled_back", 3: "not_spd", 4: "already_optimal"}


class RobotDesc(C.Structure):
    _fields_ = [("kind", C.c_int), ("dof", C.c_int), ("arm_dof", C.c_int),
                ("a", c_double_p), ("alpha", c_double_p), ("d", c_double_p),
                ("theta_bias", c_double_p), ("base_pose", C.c_double * 16),
                ("nr_spheres", C.c_int), ("sphere_link", c_int_p),
                ("sphere_radius", c_double_p), ("sphere_center", c_double_p),
                ("arm2_dof", C.c_int), ("base_pose2", C.c_double * 16), ("base_pose3", C.c_double * 16),
                ("reverse_linact", C.c_int)]


class Settings(C.Structure):
    _fields_ = [("dof", C.c_int), ("total_step", C.c_int), ("total_time", C.c_double),
                ("conf_prior_sigma", C.c_double), ("vel_prior_sigma", C.c_double),
                ("flag_pos_limit", C.c_int), ("flag_vel_limit", C.c_int),
                ("joint_pos_limits_up", c_double_p), ("joint_pos_limits_down", c_double_p),
                ("vel_limits", c_double_p), ("pos_limit_thresh", c_double_p),
                ("vel_limit_thresh", c_double_p), ("pos_limit_sigmas", c_double_p),
                ("vel_limit_sigmas", c_double_p),
                ("epsilon", C.c_double), ("cost_sigma", C.c_double), ("obs_check_inter", C.c_int),
                ("Qc", c_double_p), ("opt_type", C.c_int), ("verbosity", C.c_int),
                ("final_iter_no_increase", C.c_int), ("rel_thresh", C.c_double),
                ("max_iter", C.c_int)]


MAX_WORKSPACE_FACTORS, MAX_SELF_COLLISION_PAIRS = 4, 16
WORKSPACE_POSITION, WORKSPACE_ORIENTATION, WORKSPACE_POSE = 0, 1, 2


class WorkspaceFactor(C.Structure):
    _fields_ = [("mode", C.c_int), ("link", C.c_int), ("first_state", C.c_int), ("last_state", C.c_int),
                ("sigma", C.c_double), ("des_pose", C.c_double * 16)]


class GraphOpts(C.Structure):
    _fields_ = [("obs_skip_first_state", C.c_int), ("vehicle_dynamics_sigma", C.c_double),
                ("lm_lambda_initial", C.c_double), ("lm_lambda_factor", C.c_double),
                ("lm_lambda_upper", C.c_double), ("lm_lambda_lower", C.c_double),
                ("lm_min_model_fidelity", C.c_double), ("dogleg_delta_initial", C.c_double),
                ("abs_error_tol", C.c_double), ("error_tol", C.c_double),
                ("fixed_iterations", C.c_int),
                ("end_conf_prior_off", C.c_int), ("n_workspace", C.c_int),
                ("workspace", WorkspaceFactor * MAX_WORKSPACE_FACTORS),
                ("n_self_collision", C.c_int), ("self_collision_first", C.c_int), ("self_collision_last", C.c_int),
                ("self_collision", (C.c_double * 4) * MAX_SELF_COLLISION_PAIRS)]


def dptr(a):
    """pointer to a C-contiguous float64 array (None -> NULL)."""
    if a is None:
        return None
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"], "need contiguous float64"
    return a.ctypes.data_as(c_double_p)


def iptr(a):
    if a is None:
        return None
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"], "need contiguous int32"
    return a.ctypes.data_as(c_int_p)


def f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


def make_robot_desc(model):
    """(RobotDesc, keepalive) from a gpmp2_amd.robots.RobotModel."""
    fl = model.flat()
    d = RobotDesc()
    d.kind, d.dof, d.arm_dof = fl["kind"], fl["dof"], fl["arm_dof"]
    d.a, d.alpha, d.d, d.theta_bias = dptr(fl["a"]), dptr(fl["alpha"]), dptr(fl["d"]), dptr(fl["theta_bias"])
    for i in range(16):
        d.base_pose[i] = float(fl["base_pose"][i])
    d.nr_spheres = len(fl["sphere_radius"])
    d.sphere_link = iptr(fl["sphere_link"])
    d.sphere_radius = dptr(fl["sphere_radius"])
    d.sphere_center = dptr(fl["sphere_center"])
    d.arm2_dof = int(fl.get("arm2_dof", 0))
    eye = np.eye(4).reshape(16)
    for i in range(16):
        d.base_pose2[i] = float(fl.get("base_pose2", eye)[i])
        d.base_pose3[i] = float(fl.get("base_pose3", eye)[i])
    d.reverse_linact = int(fl.get("reverse_linact", 0))
    return d, fl


def make_settings(setting):
    """(Settings, GraphOpts, keepalive) from a gpmp2_amd.planner.TrajOptimizerSetting."""
    s = Settings()
    keep = {}

    def vec(name, value):
        if value is None:
            return None
        arr = f64(value).reshape(-1)
        if arr.size != setting.dof:
            raise ValueError(f"[TrajOptimizerSetting] {name} dim does not fit dof")
        keep[name] = arr
        return dptr(arr)

    s.dof, s.total_step, s.total_time = setting.dof, setting.total_step, setting.total_time
    s.conf_prior_sigma, s.vel_prior_sigma = setting.conf_prior_sigma, setting.vel_prior_sigma
    s.flag_pos_limit, s.flag_vel_limit = int(setting.flag_pos_limit), int(setting.flag_vel_limit)
    s.joint_pos_limits_up = vec("joint_pos_limits_up", setting.joint_pos_limits_up)
    s.joint_pos_limits_down = vec("joint_pos_limits_down", setting.joint_pos_limits_down)
    s.vel_limits = vec("vel_limits", setting.vel_limits)
    s.pos_limit_thresh = vec("pos_limit_thresh", setting.pos_limit_thresh)
    s.vel_limit_thresh = vec("vel_limit_thresh", setting.vel_limit_thresh)
    s.pos_limit_sigmas = vec("pos_limit_sigmas", setting.pos_limit_sigmas)
    s.vel_limit_sigmas = vec("vel_limit_sigmas", setting.vel_limit_sigmas)
    s.epsilon, s.cost_sigma, s.obs_check_inter = setting.epsilon, setting.cost_sigma, setting.obs_check_inter
    if setting.Qc is not None:
        q = f64(setting.Qc)
        if q.shape != (setting.dof, setting.dof):
            raise ValueError("[TrajOptimizerSetting] Qc dim does not fit dof")
        keep["Qc"] = q
        s.Qc = dptr(q)
    s.opt_type, s.verbosity = setting.opt_type, setting.opt_verbosity
    s.final_iter_no_increase = int(setting.final_iter_no_increase)
    s.rel_thresh, s.max_iter = setting.rel_thresh, setting.max_iter
    o = GraphOpts()
    o.obs_skip_first_state = int(setting.obs_skip_first_state)
    o.vehicle_dynamics_sigma = setting.vehicle_dynamics_sigma
    o.lm_lambda_initial, o.lm_lambda_factor = setting.lm_lambda_initial, setting.lm_lambda_factor
    o.lm_lambda_upper, o.lm_lambda_lower = setting.lm_lambda_upper, setting.lm_lambda_lower
    o.lm_min_model_fidelity = setting.lm_min_model_fidelity
    o.dogleg_delta_initial = setting.dogleg_delta_initial
    o.abs_error_tol, o.error_tol = setting.abs_error_tol, setting.error_tol
    o.fixed_iterations = setting.fixed_iterations
    o.end_conf_prior_off = int(getattr(setting, "end_conf_prior_off", False))
    ws = list(getattr(setting, "workspace_factors", []) or [])
    if len(ws) > MAX_WORKSPACE_FACTORS:
        raise ValueError("too many workspace factors for one plan")
    o.n_workspace = len(ws)
    for k, w in enumerate(ws):
        o.workspace[k].mode, o.workspace[k].link = int(w["mode"]), int(w["link"])
        o.workspace[k].first_state, o.workspace[k].last_state = int(w["first_state"]), int(w["last_state"])
        o.workspace[k].sigma = float(w["sigma"])
        des = f64(w["des_pose"]).reshape(16)
        for t in range(16):
            o.workspace[k].des_pose[t] = des[t]
    sc = getattr(setting, "self_collision", None)
    if sc is not None:
        sc = f64(sc).reshape(-1, 4)
        if sc.shape[0] > MAX_SELF_COLLISION_PAIRS:
            raise ValueError("too many self-collision pairs for one plan")
        o.n_self_collision = sc.shape[0]
        rng_ = getattr(setting, "self_collision_states", None) or (0, setting.total_step)
        o.self_collision_first, o.self_collision_last = int(rng_[0]), int(rng_[1])
        for k in range(sc.shape[0]):
            for t in range(4):
                o.self_collision[k][t] = sc[k, t]
    return s, o, keep
