"""Python binding of the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the
gpmp2_amd package never does.  Method names and array shapes mirror gpmp2_amd.engine so that a
parity test calls both with the same arguments.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from gpmp2_amd import _capi
from gpmp2_amd._capi import dptr, f64, iptr

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB = os.path.join(_ORACLE_DIR, "liboracle.so")


def build_oracle(force=False):
    if force or not os.path.exists(_LIB):
        subprocess.check_call(["make", "-C", _ORACLE_DIR, "-s"] + (["-B"] if force else []))
    return _LIB


class _Handle:
    def __init__(self, ptr, keep, destroy):
        self.ptr, self.keep, self._destroy = ptr, keep, destroy

    def __del__(self):
        try:
            if self.ptr:
                self._destroy(self.ptr)
        except Exception:
            pass


class Oracle:
    def __init__(self):
        self.lib = C.CDLL(build_oracle())
        self.lib.orc_robot_destroy.argtypes = [C.c_void_p]
        self.lib.orc_sdf_destroy.argtypes = [C.c_void_p]

    # ---------------------------------------------------------------- handles
    def robot(self, model):
        desc, keep = _capi.make_robot_desc(model)
        out = C.c_void_p()
        self.lib.orc_robot_create(C.byref(desc), C.byref(out))
        h = _Handle(out, keep, self.lib.orc_robot_destroy)
        h.dof, h.S, h.L = model.dof(), model.nr_body_spheres(), model.fk_model().nr_links()
        return h

    def sdf(self, origin, cell_size, data, layout=_capi.SDF_LAYOUT_ZYX):
        """data: [ny, nx] (planar) or [nz, ny, nx] for layout ZYX."""
        data = f64(data)
        dim = data.ndim
        if dim == 2:
            ny, nx, nz = data.shape[0], data.shape[1], 1
        else:
            nz, ny, nx = data.shape
        org = f64(list(origin) + [0.0] * (3 - len(origin)))
        out = C.c_void_p()
        self.lib.orc_sdf_create(C.c_int(dim), dptr(org), C.c_double(cell_size), nx, ny, nz,
                                dptr(data), C.c_int(layout), C.byref(out))
        h = _Handle(out, data, self.lib.orc_sdf_destroy)
        h.dim = dim
        return h

    # ---------------------------------------------------------------- factor level
    def sdf_field_from_occupancy(self, occ, cell_size):
        occ = f64(occ)
        dim = occ.ndim
        nz, ny, nx = ((1,) + occ.shape) if dim == 2 else occ.shape
        field = np.zeros_like(occ)
        self.lib.orc_sdf_field_from_occupancy(C.c_int(dim), nx, ny, nz, dptr(occ), C.c_double(cell_size), dptr(field))
        return field

    def sdf_query(self, sdf, points):
        p = f64(points).reshape(-1, sdf.dim)
        M = p.shape[0]
        dist, grad, inr = np.zeros(M), np.zeros((M, sdf.dim)), np.zeros(M, dtype=np.int32)
        self.lib.orc_sdf_query(sdf.ptr, M, dptr(p), dptr(dist), dptr(grad), iptr(inr))
        return dist, grad, inr

    def forward_kinematics(self, robot, conf):
        q = f64(conf).reshape(-1, robot.dof)
        M = q.shape[0]
        poses, J = np.zeros((M, robot.L, 4, 4)), np.zeros((M, robot.L, 6, robot.dof))
        self.lib.orc_forward_kinematics(robot.ptr, M, dptr(q), dptr(poses), dptr(J))
        return poses, J

    def sphere_centers(self, robot, conf):
        q = f64(conf).reshape(-1, robot.dof)
        M = q.shape[0]
        c, J = np.zeros((M, robot.S, 3)), np.zeros((M, robot.S, 3, robot.dof))
        self.lib.orc_sphere_centers(robot.ptr, M, dptr(q), dptr(c), dptr(J))
        return c, J

    def obstacle_factor(self, robot, sdf, epsilon, conf):
        q = f64(conf).reshape(-1, robot.dof)
        M = q.shape[0]
        err, H = np.zeros((M, robot.S)), np.zeros((M, robot.S, robot.dof))
        self.lib.orc_obstacle_factor(robot.ptr, sdf.ptr, C.c_double(epsilon), M, dptr(q), dptr(err), dptr(H))
        return err, H

    def obstacle_gp_factor(self, robot, sdf, epsilon, Qc, delta_t, tau, c1, v1, c2, v2):
        D = robot.dof
        c1, v1, c2, v2 = (f64(a).reshape(-1, D) for a in (c1, v1, c2, v2))
        M = c1.shape[0]
        Q = None if Qc is None else f64(Qc)
        err = np.zeros((M, robot.S))
        H = [np.zeros((M, robot.S, D)) for _ in range(4)]
        self.lib.orc_obstacle_gp_factor(robot.ptr, sdf.ptr, C.c_double(epsilon), dptr(Q),
                                        C.c_double(delta_t), C.c_double(tau), M, dptr(c1), dptr(v1),
                                        dptr(c2), dptr(v2), dptr(err), *[dptr(h) for h in H])
        return err, H

    def gp_prior_factor(self, dof, lie, delta_t, c1, v1, c2, v2):
        c1, v1, c2, v2 = (f64(a).reshape(-1, dof) for a in (c1, v1, c2, v2))
        M = c1.shape[0]
        err = np.zeros((M, 2 * dof))
        H = [np.zeros((M, 2 * dof, dof)) for _ in range(4)]
        self.lib.orc_gp_prior_factor(dof, int(lie), C.c_double(delta_t), M, dptr(c1), dptr(v1),
                                     dptr(c2), dptr(v2), dptr(err), *[dptr(h) for h in H])
        return err, H

    def gp_interpolate(self, dof, lie, Qc, delta_t, tau, c1, v1, c2, v2):
        c1, v1, c2, v2 = (f64(a).reshape(-1, dof) for a in (c1, v1, c2, v2))
        M = c1.shape[0]
        Q = None if Qc is None else f64(Qc)
        conf, vel = np.zeros((M, dof)), np.zeros((M, dof))
        self.lib.orc_gp_interpolate(dof, int(lie), dptr(Q), C.c_double(delta_t), C.c_double(tau), M,
                                    dptr(c1), dptr(v1), dptr(c2), dptr(v2), dptr(conf), dptr(vel))
        return conf, vel

    def gp_interpolate_jac(self, dof, lie, Qc, delta_t, tau, c1, v1, c2, v2):
        c1, v1, c2, v2 = (f64(a).reshape(-1, dof) for a in (c1, v1, c2, v2))
        M = c1.shape[0]
        Q = None if Qc is None else f64(Qc)
        H = [np.zeros((M, dof, dof)) for _ in range(4)]
        self.lib.orc_gp_interpolate_jac(dof, int(lie), dptr(Q), C.c_double(delta_t), C.c_double(tau), M,
                                        dptr(c1), dptr(v1), dptr(c2), dptr(v2), *[dptr(h) for h in H])
        return H

    def gp_matrices(self, dof, Qc, delta_t, tau):
        Q = None if Qc is None else f64(Qc)
        L, P = np.zeros((2 * dof, 2 * dof)), np.zeros((2 * dof, 2 * dof))
        self.lib.orc_gp_matrices(dof, dptr(Q), C.c_double(delta_t), C.c_double(tau), dptr(L), dptr(P))
        return L, P

    def interpolate_traj(self, dof, lie, Qc, delta_t, inter_step, traj, start_index=0, end_index=None):
        """traj [B][N+1][2D] -> [B][(end-start)*(inter_step+1)+1][2D]  (planner/TrajUtils.cpp:96-236)"""
        t = f64(traj)
        t = t.reshape(-1, t.shape[-2], 2 * dof)
        B, N = t.shape[0], t.shape[1] - 1
        end_index = N if end_index is None else int(end_index)
        Q = None if Qc is None else f64(Qc)
        out = np.zeros((B, max(end_index - start_index, 0) * (inter_step + 1) + 1, 2 * dof))
        self.lib.orc_interpolate_traj(dof, int(lie), dptr(Q), C.c_double(delta_t), int(inter_step), B, N,
                                      int(start_index), end_index, dptr(t), dptr(out))
        return out

    def workspace_prior_factor(self, robot, mode, joint, des_pose, conf, jac=True):
        """mode 0 position / 1 orientation / 2 pose; des_pose 4x4 -> err [M][3|3|6], H [M][rows][D]"""
        q = f64(conf).reshape(-1, robot.dof)
        M, rows = q.shape[0], 6 if mode == 2 else 3
        des = f64(des_pose).reshape(4, 4)
        err, H = np.zeros((M, rows)), (np.zeros((M, rows, robot.dof)) if jac else None)
        self.lib.orc_workspace_prior_factor(robot.ptr, int(mode), int(joint), dptr(des), M, dptr(q), dptr(err),
                                                    dptr(H))
        return err, H

    def self_collision_factor(self, robot, data, conf, jac=True):
        """data [n][4] = (sphere A, sphere B, epsilon, sigma) -> err [M][n], H [M][n][D]"""
        q = f64(conf).reshape(-1, robot.dof)
        d = f64(data).reshape(-1, 4)
        M, n = q.shape[0], d.shape[0]
        err, H = np.zeros((M, n)), (np.zeros((M, n, robot.dof)) if jac else None)
        self.lib.orc_self_collision_factor(robot.ptr, n, dptr(d), M, dptr(q), dptr(err), dptr(H))
        return err, H

    def vehicle_dynamics_factor(self, lie, conf, vel):
        """sliding velocity of an SE(2) base -> err [M], Hp [M][D], Hv [M][D]"""
        q, v = f64(conf), f64(vel)
        q, v = q.reshape(-1, q.shape[-1]), v.reshape(-1, v.shape[-1])
        M, D = q.shape
        err, Hp, Hv = np.zeros(M), np.zeros((M, D)), np.zeros((M, D))
        self.lib.orc_vehicle_dynamics_factor(D, int(lie), M, dptr(q), dptr(v), dptr(err), dptr(Hp), dptr(Hv))
        return err, Hp, Hv

    def joint_limit_factor(self, down, up, thresh, x):
        down, up, thresh = f64(down).reshape(-1), f64(up).reshape(-1), f64(thresh).reshape(-1)
        D = down.size
        x = f64(x).reshape(-1, D)
        err, Hd = np.zeros_like(x), np.zeros_like(x)
        self.lib.orc_joint_limit_factor(D, dptr(down), dptr(up), dptr(thresh), x.shape[0], dptr(x),
                                        dptr(err), dptr(Hd))
        return err, Hd

    # ---------------------------------------------------------------- graph level
    @staticmethod
    def _problem_arrays(setting, start_conf, start_vel, end_conf, end_vel, traj):
        D = setting.dof
        sc, sv, ec, ev = (f64(a).reshape(-1, D) for a in (start_conf, start_vel, end_conf, end_vel))
        B = sc.shape[0]
        t = f64(traj).reshape(B, setting.total_step + 1, 2 * D)
        return B, sc, sv, ec, ev, t

    def graph_error(self, robot, sdf, setting, start_conf, start_vel, end_conf, end_vel, traj):
        s, o, keep = _capi.make_settings(setting)
        B, sc, sv, ec, ev, t = self._problem_arrays(setting, start_conf, start_vel, end_conf, end_vel, traj)
        err = np.zeros(B)
        self.lib.orc_graph_error(robot.ptr, sdf.ptr, C.byref(s), C.byref(o), B, dptr(sc), dptr(sv),
                                 dptr(ec), dptr(ev), dptr(t), dptr(err))
        return err

    def linearize(self, robot, sdf, setting, start_conf, start_vel, end_conf, end_vel, traj):
        s, o, keep = _capi.make_settings(setting)
        B, sc, sv, ec, ev, t = self._problem_arrays(setting, start_conf, start_vel, end_conf, end_vel, traj)
        n, nb = 2 * setting.dof, setting.total_step + 1
        Hd, Ho = np.zeros((B, nb, n, n)), np.zeros((B, nb - 1, n, n))
        g, err = np.zeros((B, nb, n)), np.zeros(B)
        self.lib.orc_linearize(robot.ptr, sdf.ptr, C.byref(s), C.byref(o), B, dptr(sc), dptr(sv), dptr(ec),
                               dptr(ev), dptr(t), dptr(Hd), dptr(Ho), dptr(g), dptr(err))
        return Hd, Ho, g, err

    def dense_linearize(self, robot, sdf, setting, start_conf, start_vel, end_conf, end_vel, traj):
        s, o, keep = _capi.make_settings(setting)
        B, sc, sv, ec, ev, t = self._problem_arrays(setting, start_conf, start_vel, end_conf, end_vel, traj)
        assert B == 1
        rows = C.c_int(0)
        self.lib.orc_dense_linearize(robot.ptr, sdf.ptr, C.byref(s), C.byref(o), dptr(sc), dptr(sv), dptr(ec),
                                     dptr(ev), dptr(t), None, None, C.byref(rows))
        W = (setting.total_step + 1) * 2 * setting.dof
        A, r = np.zeros((rows.value, W)), np.zeros(rows.value)
        self.lib.orc_dense_linearize(robot.ptr, sdf.ptr, C.byref(s), C.byref(o), dptr(sc), dptr(sv), dptr(ec),
                                     dptr(ev), dptr(t), dptr(A), dptr(r), C.byref(rows))
        return A, r

    def block_tridiag_solve(self, Hd, Ho, b):
        Hd, Ho, b = f64(Hd), f64(Ho), f64(b)
        B, nblk, n = Hd.shape[0], Hd.shape[1], Hd.shape[2]
        x, ok = np.zeros((B, nblk, n)), np.zeros(B, dtype=np.int32)
        self.lib.orc_block_tridiag_solve(B, nblk, n, dptr(Hd), dptr(Ho), dptr(b), dptr(x), iptr(ok))
        return x, ok

    def batch_optimize(self, robot, sdf, setting, start_conf, start_vel, end_conf, end_vel, init,
                       nthreads=1):
        s, o, keep = _capi.make_settings(setting)
        B, sc, sv, ec, ev, t = self._problem_arrays(setting, start_conf, start_vel, end_conf, end_vel, init)
        out = np.zeros_like(t)
        iters, status = np.zeros(B, dtype=np.int32), np.zeros(B, dtype=np.int32)
        ferr, trace = np.zeros(B), np.zeros((B, setting.max_iter + 1))
        self.lib.orc_batch_optimize(robot.ptr, sdf.ptr, C.byref(s), C.byref(o), B, dptr(sc), dptr(sv),
                                    dptr(ec), dptr(ev), dptr(t), dptr(out), iptr(iters), dptr(ferr),
                                    iptr(status), dptr(trace), int(nthreads))
        return dict(traj=out, iters=iters, final_error=ferr, status=status, error_trace=trace)

    def dogleg_probe(self, rows=256):
        """context manager: records {gg, gHg, g.dx_n, |dx_n|^2, |dx_u|^2, dx_u.dx_n, tau, Delta, rho, new_f} of every
        Dogleg trial point of the solves run inside it (single-threaded solves only)"""
        orc = self

        class _Probe:
            def __enter__(self_):
                self_.buf = np.zeros((rows, 10))
                orc.lib.orc_set_dogleg_probe(dptr(self_.buf), rows)
                return self_

            def __exit__(self_, *a):
                self_.rows = self_.buf[: orc.lib.orc_dogleg_probe_rows()].copy()
                orc.lib.orc_set_dogleg_probe(None, 0)

        return _Probe()

    def batch_optimize_xp(self, robot, sdf, setting, start_conf, start_vel, end_conf, end_vel, init, priors, goal_on):
        """priors: per trajectory a list of dicts(state, conf, Wc, vel=None, Wv=None); goal_on: [B] ints."""
        s, o, keep = _capi.make_settings(setting)
        B, sc, sv, ec, ev, t = self._problem_arrays(setting, start_conf, start_vel, end_conf, end_vel, init)
        D, XP = setting.dof, 8
        xp_n = np.zeros(B, dtype=np.int32)
        xp_state, xp_hv = np.zeros((B, XP), dtype=np.int32), np.zeros((B, XP), dtype=np.int32)
        xp_t, xp_i = np.zeros((B, XP, 2 * D)), np.zeros((B, XP, 2, D, D))
        for b in range(B):
            for e, pr in enumerate(priors[b]):
                xp_state[b, e] = pr["state"]
                xp_t[b, e, :D] = pr["conf"]
                xp_i[b, e, 0] = pr["Wc"]
                if pr.get("vel") is not None:
                    xp_hv[b, e] = 1
                    xp_t[b, e, D:] = pr["vel"]
                    xp_i[b, e, 1] = pr["Wv"]
            xp_n[b] = len(priors[b])
        gon = np.ascontiguousarray(goal_on, dtype=np.int32)
        out = np.zeros_like(t)
        iters, status, ferr = np.zeros(B, dtype=np.int32), np.zeros(B, dtype=np.int32), np.zeros(B)
        self.lib.orc_batch_optimize_xp(robot.ptr, sdf.ptr, C.byref(s), C.byref(o), B, dptr(sc), dptr(sv), dptr(ec),
                                       dptr(ev), dptr(t), iptr(xp_n), iptr(xp_state), iptr(xp_hv), dptr(xp_t),
                                       dptr(xp_i), iptr(gon), dptr(out), iptr(iters), dptr(ferr), iptr(status))
        return dict(traj=out, iters=iters, final_error=ferr, status=status)

    def collision_cost(self, robot, sdf, total_step, traj):
        t = f64(traj).reshape(-1, total_step + 1, 2 * robot.dof)
        cost = np.zeros(t.shape[0])
        self.lib.orc_collision_cost(robot.ptr, sdf.ptr, total_step, t.shape[0], dptr(t), dptr(cost))
        return cost

    def retract(self, robot, traj, delta):
        t, d = f64(traj).reshape(-1, 2 * robot.dof), f64(delta).reshape(-1, 2 * robot.dof)
        out = np.zeros_like(t)
        self.lib.orc_retract(robot.ptr, t.shape[0], dptr(t), dptr(d), dptr(out))
        return out

    def pose2_expmap(self, v):
        v, p = f64(v), np.zeros(3)
        self.lib.orc_pose2_expmap(dptr(v), dptr(p))
        return p

    def pose2_logmap(self, p):
        p, v = f64(p), np.zeros(3)
        self.lib.orc_pose2_logmap(dptr(p), dptr(v))
        return v
