"""TrajOptimizerSetting -- host mirror of gpmp2/planner/TrajOptimizerSetting.h:17-100 with the
defaults of TrajOptimizerSetting(size_t) (gpmp2/planner/TrajOptimizerSetting.cpp:32-56) and the
same setter names the MATLAB/Python wrappers expose (gpmp2.h:693-735)."""
from __future__ import annotations

import numpy as np

from ._capi import OPT_DOGLEG, OPT_GAUSS_NEWTON, OPT_LM


class TrajOptimizerSetting:
    GaussNewton, LM, Dogleg = OPT_GAUSS_NEWTON, OPT_LM, OPT_DOGLEG

    def __init__(self, system_dof: int):
        self.dof = int(system_dof)
        self.total_step = 10
        self.total_time = 1.0
        self.conf_prior_sigma = 0.0001
        self.vel_prior_sigma = 0.0001
        self.flag_pos_limit = False
        self.flag_vel_limit = False
        self.joint_pos_limits_up = None      # None -> +1e6
        self.joint_pos_limits_down = None    # None -> -1e6
        self.vel_limits = None               # None -> 1e6
        self.pos_limit_thresh = None         # None -> 1e-3
        self.vel_limit_thresh = None         # None -> 1e-3
        self.pos_limit_sigmas = None         # None -> 1e-3
        self.vel_limit_sigmas = None         # None -> 1e-3
        self.epsilon = 0.2
        self.cost_sigma = 0.1
        self.obs_check_inter = 5
        self.Qc = None                       # None -> identity (noiseModel::Unit)
        self.opt_type = OPT_DOGLEG
        self.opt_verbosity = 0
        self.final_iter_no_increase = True
        self.rel_thresh = 1e-2
        self.max_iter = 50
        # --- gpmp2mi_graph_opts (not part of the reference struct; see include/gpmp2mi.h) ---
        self.obs_skip_first_state = False
        self.vehicle_dynamics_sigma = 0.0
        self.lm_lambda_initial = 100.0
        self.lm_lambda_factor = 10.0
        self.lm_lambda_upper = 1e5
        self.lm_lambda_lower = 0.0
        self.lm_min_model_fidelity = 1e-3
        self.dogleg_delta_initial = 0.2
        self.abs_error_tol = 1e-5
        self.error_tol = 0.0
        self.fixed_iterations = 0
        # extra factors of hand-built graphs, carried by the plan as data (include/gpmp2mi.h gpmp2mi_graph_opts)
        self.end_conf_prior_off = False      # no PriorFactor on x_N (a goal / workspace factor replaces it)
        self.workspace_factors = []          # dicts(mode, link, first_state, last_state, sigma, des_pose 4x4)
        self.self_collision = None           # [n][4] = sphere A, sphere B, epsilon, sigma
        self.self_collision_states = None    # (first, last) support states, default all

    # extra factors (names follow the reference's factor classes)
    def add_goal_factor_arm(self, link, dest_point, sigma, state=None):
        """GoalFactorArm (kinematics/GoalFactorArm.h:58-77) on `state` (default: the last), replacing nothing by
        itself: set end_conf_prior_off to drop the prior on x_N as matlab/Arm3GoalReachExample.m:107 does"""
        des = np.eye(4)
        des[:3, 3] = np.asarray(dest_point, dtype=np.float64)
        st = self.total_step if state is None else int(state)
        self.workspace_factors.append(dict(mode=0, link=int(link), first_state=st, last_state=st, sigma=float(sigma),
                                           des_pose=des))

    def add_workspace_prior(self, mode, link, des_pose, sigma, first_state, last_state=None):
        """GaussianPriorWorkspace{Position (0), Orientation (1), Pose (2)}<Arm> on states first_state..last_state"""
        self.workspace_factors.append(dict(mode=int(mode), link=int(link), first_state=int(first_state),
                                           last_state=int(first_state if last_state is None else last_state),
                                           sigma=float(sigma), des_pose=np.asarray(des_pose, dtype=np.float64).reshape(4, 4)))

    # traj settings
    def set_total_step(self, step): self.total_step = int(step)
    def set_total_time(self, time): self.total_time = float(time)
    def set_conf_prior_model(self, sigma): self.conf_prior_sigma = float(sigma)
    def set_vel_prior_model(self, sigma): self.vel_prior_sigma = float(sigma)
    # limits
    def set_flag_pos_limit(self, flag): self.flag_pos_limit = bool(flag)
    def set_flag_vel_limit(self, flag): self.flag_vel_limit = bool(flag)
    def set_joint_pos_limits_up(self, v): self.joint_pos_limits_up = np.asarray(v, dtype=np.float64)
    def set_joint_pos_limits_down(self, v): self.joint_pos_limits_down = np.asarray(v, dtype=np.float64)
    def set_vel_limits(self, v): self.vel_limits = np.asarray(v, dtype=np.float64)
    def set_pos_limit_thresh(self, v): self.pos_limit_thresh = np.asarray(v, dtype=np.float64)
    def set_vel_limit_thresh(self, v): self.vel_limit_thresh = np.asarray(v, dtype=np.float64)
    def set_pos_limit_model(self, v): self.pos_limit_sigmas = np.asarray(v, dtype=np.float64)
    def set_vel_limit_model(self, v): self.vel_limit_sigmas = np.asarray(v, dtype=np.float64)
    # obstacle
    def set_epsilon(self, eps): self.epsilon = float(eps)
    def set_cost_sigma(self, sigma): self.cost_sigma = float(sigma)
    def set_obs_check_inter(self, inter): self.obs_check_inter = int(inter)
    # GP
    def set_Qc_model(self, Qc): self.Qc = np.asarray(Qc, dtype=np.float64)
    # optimizer
    def setGaussNewton(self): self.opt_type = OPT_GAUSS_NEWTON
    def setLM(self): self.opt_type = OPT_LM
    def setDogleg(self): self.opt_type = OPT_DOGLEG
    def set_rel_thresh(self, thresh): self.rel_thresh = float(thresh)
    def set_max_iter(self, it): self.max_iter = int(it)
    def setVerbosityNone(self): self.opt_verbosity = 0
    def setVerbosityError(self): self.opt_verbosity = 1
    def setOptimizationNoIncrase(self, flag): self.final_iter_no_increase = bool(flag)
