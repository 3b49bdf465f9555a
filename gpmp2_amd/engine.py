"""ctypes binding of the HIP product library (gpmp2_amd/csrc/libgpmp2mi.so, C ABI of
include/gpmp2mi.h).  There is no Python/CPU compute path here: if the shared library is missing
or no GPU is usable every call raises."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _capi
from ._capi import dptr, f64, iptr

_HERE = os.path.dirname(os.path.abspath(__file__))
# GPMP2MI_LIB: another build of the same library (diagnostic / A-B builds); the default is the in-tree product library
LIB_PATH = os.environ.get("GPMP2MI_LIB") or os.path.join(_HERE, "csrc", "libgpmp2mi.so")

ERR_NAMES = {1: "invalid argument", 2: "no usable GPU", 3: "HIP error", 4: "unsupported", 5: "allocation failed",
             6: "timed out (plan poisoned)"}


# per-trajectory status (include/gpmp2mi.h:51-59)
TRAJ_CONVERGED, TRAJ_MAX_ITER, TRAJ_ROLLED_BACK, TRAJ_NOT_SPD, TRAJ_ALREADY_OPTIMAL = range(5)


class Gpmp2miError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"gpmp2mi error {code} ({ERR_NAMES.get(code, '?')}): {msg}")
        self.code = code


def load_library(path: str = LIB_PATH) -> C.CDLL:
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no fallback implementation)")
    lib = C.CDLL(path)
    lib.gpmp2mi_last_error.restype = C.c_char_p
    lib.gpmp2mi_plan_traj_dev.restype = C.c_void_p
    for name in ("gpmp2mi_robot_destroy", "gpmp2mi_sdf_destroy", "gpmp2mi_plan_destroy"):
        getattr(lib, name).argtypes = [C.c_void_p]
        getattr(lib, name).restype = None
    return lib


class _Handle:
    def __init__(self, ptr, destroy, keep=None):
        self.ptr, self._destroy, self.keep = ptr, destroy, keep

    def close(self):
        if self.ptr:
            self._destroy(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Engine:
    """Thin object wrapper; method names mirror tests/oracle.py one-to-one."""

    def __init__(self, path: str = LIB_PATH):
        self.lib = load_library(path)

    def _ck(self, rc):
        if rc != 0:
            raise Gpmp2miError(rc, self.lib.gpmp2mi_last_error().decode())

    def device_count(self):
        return int(self.lib.gpmp2mi_device_count())

    # ---------------------------------------------------------------- handles
    def robot(self, model):
        desc, keep = _capi.make_robot_desc(model)
        out = C.c_void_p()
        self._ck(self.lib.gpmp2mi_robot_create(C.byref(desc), C.byref(out)))
        h = _Handle(out, self.lib.gpmp2mi_robot_destroy, keep)
        h.dof, h.S, h.L = model.dof(), model.nr_body_spheres(), model.fk_model().nr_links()
        return h

    def sdf(self, origin, cell_size, data, layout=_capi.SDF_LAYOUT_ZYX):
        data = f64(data)
        dim = data.ndim
        if dim == 2:
            ny, nx, nz = data.shape[0], data.shape[1], 1
        else:
            nz, ny, nx = data.shape
        org = f64(list(origin) + [0.0] * (3 - len(origin)))
        out = C.c_void_p()
        self._ck(self.lib.gpmp2mi_sdf_create(C.c_int(dim), dptr(org), C.c_double(cell_size), nx, ny, nz,
                                             dptr(data), C.c_int(layout), C.byref(out)))
        h = _Handle(out, self.lib.gpmp2mi_sdf_destroy)
        h.dim = dim
        return h

    def sdf_field_from_occupancy(self, occ, cell_size):
        """occupancy [ny][nx] or [nz][ny][nx] -> signed field of the same shape (computed on the GPU)"""
        occ = f64(occ)
        dim = occ.ndim
        nz, ny, nx = ((1,) + occ.shape) if dim == 2 else occ.shape
        field = np.zeros_like(occ)
        self._ck(self.lib.gpmp2mi_sdf_field_from_occupancy(C.c_int(dim), nx, ny, nz, dptr(occ), C.c_double(cell_size),
                                                           dptr(field)))
        return field

    def sdf_from_occupancy(self, origin, cell_size, occ, layout=_capi.SDF_LAYOUT_ZYX):
        occ = f64(occ)
        dim = occ.ndim
        nz, ny, nx = ((1,) + occ.shape) if dim == 2 else occ.shape
        org = f64(list(origin) + [0.0] * (3 - len(origin)))
        out = C.c_void_p()
        self._ck(self.lib.gpmp2mi_sdf_create_from_occupancy(C.c_int(dim), dptr(org), C.c_double(cell_size), nx, ny, nz,
                                                            dptr(occ), C.c_int(layout), C.byref(out)))
        h = _Handle(out, self.lib.gpmp2mi_sdf_destroy)
        h.dim = dim
        return h

    def sdf_read_vol(self, filename_pre):
        out = C.c_void_p()
        self._ck(self.lib.gpmp2mi_sdf_read_vol(str(filename_pre).encode(), C.byref(out)))
        h = _Handle(out, self.lib.gpmp2mi_sdf_destroy)
        h.dim = 3
        return h

    def sdf_field(self, sdf):
        """-> dict(dim, origin, cell_size, data [nz][ny][nx] or [ny][nx])"""
        dim, nx, ny, nz = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        org, cell = np.zeros(3), C.c_double()
        self._ck(self.lib.gpmp2mi_sdf_get_field(sdf.ptr, C.byref(dim), C.byref(nx), C.byref(ny), C.byref(nz), dptr(org),
                                                C.byref(cell), None))
        data = np.zeros((nz.value, ny.value, nx.value))
        self._ck(self.lib.gpmp2mi_sdf_get_field(sdf.ptr, None, None, None, None, None, None, dptr(data)))
        return dict(dim=dim.value, origin=org[:dim.value].copy(), cell_size=cell.value,
                    data=data[0] if dim.value == 2 else data)

    # ---------------------------------------------------------------- factor level
    def sdf_query(self, sdf, points):
        p = f64(points).reshape(-1, sdf.dim)
        M = p.shape[0]
        dist, grad, inr = np.zeros(M), np.zeros((M, sdf.dim)), np.zeros(M, dtype=np.int32)
        self._ck(self.lib.gpmp2mi_sdf_query(sdf.ptr, M, dptr(p), dptr(dist), dptr(grad), iptr(inr)))
        return dist, grad, inr

    def forward_kinematics(self, robot, conf):
        q = f64(conf).reshape(-1, robot.dof)
        M = q.shape[0]
        poses, J = np.zeros((M, robot.L, 4, 4)), np.zeros((M, robot.L, 6, robot.dof))
        self._ck(self.lib.gpmp2mi_forward_kinematics(robot.ptr, M, dptr(q), dptr(poses), dptr(J)))
        return poses, J

    def sphere_centers(self, robot, conf):
        q = f64(conf).reshape(-1, robot.dof)
        M = q.shape[0]
        c, J = np.zeros((M, robot.S, 3)), np.zeros((M, robot.S, 3, robot.dof))
        self._ck(self.lib.gpmp2mi_sphere_centers(robot.ptr, M, dptr(q), dptr(c), dptr(J)))
        return c, J

    def obstacle_factor(self, robot, sdf, epsilon, conf):
        q = f64(conf).reshape(-1, robot.dof)
        M = q.shape[0]
        err, H = np.zeros((M, robot.S)), np.zeros((M, robot.S, robot.dof))
        self._ck(self.lib.gpmp2mi_obstacle_factor(robot.ptr, sdf.ptr, C.c_double(epsilon), M, dptr(q),
                                                  dptr(err), dptr(H)))
        return err, H

    def obstacle_gp_factor(self, robot, sdf, epsilon, Qc, delta_t, tau, c1, v1, c2, v2):
        D = robot.dof
        c1, v1, c2, v2 = (f64(a).reshape(-1, D) for a in (c1, v1, c2, v2))
        M = c1.shape[0]
        Q = None if Qc is None else f64(Qc)
        err = np.zeros((M, robot.S))
        H = [np.zeros((M, robot.S, D)) for _ in range(4)]
        self._ck(self.lib.gpmp2mi_obstacle_gp_factor(robot.ptr, sdf.ptr, C.c_double(epsilon), dptr(Q),
                                                     C.c_double(delta_t), C.c_double(tau), M, dptr(c1),
                                                     dptr(v1), dptr(c2), dptr(v2), dptr(err),
                                                     *[dptr(h) for h in H]))
        return err, H

    def gp_prior_factor(self, dof, lie, delta_t, c1, v1, c2, v2):
        c1, v1, c2, v2 = (f64(a).reshape(-1, dof) for a in (c1, v1, c2, v2))
        M = c1.shape[0]
        err = np.zeros((M, 2 * dof))
        H = [np.zeros((M, 2 * dof, dof)) for _ in range(4)]
        self._ck(self.lib.gpmp2mi_gp_prior_factor(dof, int(lie), C.c_double(delta_t), M, dptr(c1), dptr(v1),
                                                  dptr(c2), dptr(v2), dptr(err), *[dptr(h) for h in H]))
        return err, H

    def gp_interpolate(self, dof, lie, Qc, delta_t, tau, c1, v1, c2, v2):
        c1, v1, c2, v2 = (f64(a).reshape(-1, dof) for a in (c1, v1, c2, v2))
        M = c1.shape[0]
        Q = None if Qc is None else f64(Qc)
        conf, vel = np.zeros((M, dof)), np.zeros((M, dof))
        self._ck(self.lib.gpmp2mi_gp_interpolate(dof, int(lie), dptr(Q), C.c_double(delta_t), C.c_double(tau),
                                                 M, dptr(c1), dptr(v1), dptr(c2), dptr(v2), dptr(conf), dptr(vel)))
        return conf, vel

    def interpolate_traj(self, dof, lie, Qc, delta_t, inter_step, traj, start_index=0, end_index=None):
        """traj [B][N+1][2D] -> [B][(end-start)*(inter_step+1)+1][2D]  (planner/TrajUtils.cpp:96-236)"""
        t = f64(traj)
        t = t.reshape(-1, t.shape[-2], 2 * dof)
        B, N = t.shape[0], t.shape[1] - 1
        end_index = N if end_index is None else int(end_index)
        Q = None if Qc is None else f64(Qc)
        out = np.zeros((B, max(end_index - start_index, 0) * (inter_step + 1) + 1, 2 * dof))
        self._ck(self.lib.gpmp2mi_interpolate_traj(dof, int(lie), dptr(Q), C.c_double(delta_t), int(inter_step), B, N,
                                               int(start_index), end_index, dptr(t), dptr(out)))
        return out

    def workspace_prior_factor(self, robot, mode, joint, des_pose, conf, jac=True):
        """mode 0 position / 1 orientation / 2 pose; des_pose 4x4 -> err [M][3|3|6], H [M][rows][D]"""
        q = f64(conf).reshape(-1, robot.dof)
        M, rows = q.shape[0], 6 if mode == 2 else 3
        des = f64(des_pose).reshape(4, 4)
        err, H = np.zeros((M, rows)), (np.zeros((M, rows, robot.dof)) if jac else None)
        self._ck(self.lib.gpmp2mi_workspace_prior_factor(robot.ptr, int(mode), int(joint), dptr(des), M, dptr(q), dptr(err),
                                                    dptr(H)))
        return err, H

    def goal_factor_arm(self, robot, dest_point, conf, jac=True):
        q = f64(conf).reshape(-1, robot.dof)
        M = q.shape[0]
        dest = f64(dest_point).reshape(3)
        err, H = np.zeros((M, 3)), (np.zeros((M, 3, robot.dof)) if jac else None)
        self._ck(self.lib.gpmp2mi_goal_factor_arm(robot.ptr, dptr(dest), M, dptr(q), dptr(err), dptr(H)))
        return err, H

    def self_collision_factor(self, robot, data, conf, jac=True):
        """data [n][4] = (sphere A, sphere B, epsilon, sigma) -> err [M][n], H [M][n][D]"""
        q = f64(conf).reshape(-1, robot.dof)
        d = f64(data).reshape(-1, 4)
        M, n = q.shape[0], d.shape[0]
        err, H = np.zeros((M, n)), (np.zeros((M, n, robot.dof)) if jac else None)
        self._ck(self.lib.gpmp2mi_self_collision_factor(robot.ptr, n, dptr(d), M, dptr(q), dptr(err), dptr(H)))
        return err, H

    def vehicle_dynamics_factor(self, lie, conf, vel):
        """sliding velocity of an SE(2) base -> err [M], Hp [M][D], Hv [M][D]"""
        q, v = f64(conf), f64(vel)
        q, v = q.reshape(-1, q.shape[-1]), v.reshape(-1, v.shape[-1])
        M, D = q.shape
        err, Hp, Hv = np.zeros(M), np.zeros((M, D)), np.zeros((M, D))
        self._ck(self.lib.gpmp2mi_vehicle_dynamics_factor(D, int(lie), M, dptr(q), dptr(v), dptr(err), dptr(Hp), dptr(Hv)))
        return err, Hp, Hv

    def joint_limit_factor(self, down, up, thresh, x):
        down, up, thresh = f64(down).reshape(-1), f64(up).reshape(-1), f64(thresh).reshape(-1)
        D = down.size
        x = f64(x).reshape(-1, D)
        err, Hd = np.zeros_like(x), np.zeros_like(x)
        self._ck(self.lib.gpmp2mi_joint_limit_factor(D, dptr(down), dptr(up), dptr(thresh), x.shape[0],
                                                     dptr(x), dptr(err), dptr(Hd)))
        return err, Hd

    def block_tridiag_solve(self, Hd, Ho, b):
        Hd, Ho, b = f64(Hd), f64(Ho), f64(b)
        B, nblk, n = Hd.shape[0], Hd.shape[1], Hd.shape[2]
        x, ok = np.zeros((B, nblk, n)), np.zeros(B, dtype=np.int32)
        self._ck(self.lib.gpmp2mi_block_tridiag_solve(B, nblk, n, dptr(Hd), dptr(Ho), dptr(b), dptr(x), iptr(ok)))
        return x, ok

    def collision_cost(self, robot, sdf, total_step, traj):
        t = f64(traj).reshape(-1, total_step + 1, 2 * robot.dof)
        cost = np.zeros(t.shape[0])
        self._ck(self.lib.gpmp2mi_collision_cost(robot.ptr, sdf.ptr, total_step, t.shape[0], dptr(t), dptr(cost)))
        return cost

    # ---------------------------------------------------------------- plans
    def plan(self, robot, sdf, setting, B):
        return Plan(self, robot, sdf, setting, B)

    # graph-level helpers with the oracle's call shape
    def _plan_for(self, robot, sdf, setting, start_conf, start_vel, end_conf, end_vel, traj):
        D = setting.dof
        sc = f64(start_conf).reshape(-1, D)
        pl = Plan(self, robot, sdf, setting, sc.shape[0])
        t = f64(traj).reshape(sc.shape[0], setting.total_step + 1, 2 * D)
        pl.set_problem(start_conf, start_vel, end_conf, end_vel, t)
        return pl, t

    def graph_error(self, robot, sdf, setting, start_conf, start_vel, end_conf, end_vel, traj):
        pl, t = self._plan_for(robot, sdf, setting, start_conf, start_vel, end_conf, end_vel, traj)
        return pl.graph_error(t)

    def linearize(self, robot, sdf, setting, start_conf, start_vel, end_conf, end_vel, traj):
        pl, t = self._plan_for(robot, sdf, setting, start_conf, start_vel, end_conf, end_vel, traj)
        return pl.linearize(t)

    def batch_optimize(self, robot, sdf, setting, start_conf, start_vel, end_conf, end_vel, init):
        pl, t = self._plan_for(robot, sdf, setting, start_conf, start_vel, end_conf, end_vel, init)
        pl.optimize()
        return pl.result()


class Plan:
    """gpmp2mi_plan: B trajectory problems resident on the GPU."""

    def __init__(self, eng: Engine, robot, sdf, setting, B: int):
        self.eng, self.robot, self.sdf, self.setting, self.B = eng, robot, sdf, setting, int(B)
        s, o, keep = _capi.make_settings(setting)
        self._keep = (s, o, keep)
        out = C.c_void_p()
        eng._ck(eng.lib.gpmp2mi_plan_create(robot.ptr, sdf.ptr, C.byref(s), C.byref(o), self.B, C.byref(out)))
        self.h = _Handle(out, eng.lib.gpmp2mi_plan_destroy)
        self.D, self.N = setting.dof, setting.total_step

    def close(self):
        self.h.close()

    def set_problem(self, start_conf, start_vel, end_conf, end_vel, init):
        D, B = self.D, self.B
        a = [f64(x).reshape(B, D) for x in (start_conf, start_vel, end_conf, end_vel)]
        t = f64(init).reshape(B, self.N + 1, 2 * D)
        self.eng._ck(self.eng.lib.gpmp2mi_plan_set_problem(self.h.ptr, *[dptr(x) for x in a], dptr(t)))

    def set_problem_dev(self, start_conf, start_vel, end_conf, end_vel, init, stream=None):
        """device pointers (ints, e.g. torch.Tensor.data_ptr()) of contiguous fp64 buffers."""
        args = [C.c_void_p(int(x)) for x in (start_conf, start_vel, end_conf, end_vel, init)]
        self.eng._ck(self.eng.lib.gpmp2mi_plan_set_problem_dev(self.h.ptr, *args, C.c_void_p(stream or 0)))

    def optimize(self, stream=None):
        self.eng._ck(self.eng.lib.gpmp2mi_plan_optimize(self.h.ptr, C.c_void_p(stream or 0)))

    def result(self):
        B, D, N = self.B, self.D, self.N
        traj = np.zeros((B, N + 1, 2 * D))
        iters, status = np.zeros(B, dtype=np.int32), np.zeros(B, dtype=np.int32)
        ferr, trace = np.zeros(B), np.zeros((B, self.setting.max_iter + 1))
        self.eng._ck(self.eng.lib.gpmp2mi_plan_get_result(self.h.ptr, dptr(traj), iptr(iters), dptr(ferr),
                                                          iptr(status), dptr(trace)))
        return dict(traj=traj, iters=iters, final_error=ferr, status=status, error_trace=trace)

    def result_counts(self):
        B = self.B
        iters, status, ferr = np.zeros(B, dtype=np.int32), np.zeros(B, dtype=np.int32), np.zeros(B)
        self.eng._ck(self.eng.lib.gpmp2mi_plan_get_result(self.h.ptr, None, iptr(iters), dptr(ferr), iptr(status), None))
        return iters, status, ferr

    def traj_dev_ptr(self):
        return int(self.eng.lib.gpmp2mi_plan_traj_dev(self.h.ptr))

    def graph_error(self, traj):
        t = f64(traj).reshape(self.B, self.N + 1, 2 * self.D)
        err = np.zeros(self.B)
        self.eng._ck(self.eng.lib.gpmp2mi_plan_graph_error(self.h.ptr, dptr(t), dptr(err)))
        return err

    def linearize(self, traj):
        B, n, nb = self.B, 2 * self.D, self.N + 1
        t = f64(traj).reshape(B, nb, n)
        Hd, Ho = np.zeros((B, nb, n, n)), np.zeros((B, nb - 1, n, n))
        g, err = np.zeros((B, nb, n)), np.zeros(B)
        self.eng._ck(self.eng.lib.gpmp2mi_plan_linearize(self.h.ptr, dptr(t), dptr(Hd), dptr(Ho), dptr(g), dptr(err)))
        return Hd, Ho, g, err

    # ---- incremental replanning (ISAM2TrajOptimizer's role; see include/gpmp2mi.h)
    def fix_state(self, b, state_idx, conf, vel):
        c, v = f64(conf).reshape(self.D), f64(vel).reshape(self.D)
        self.eng._ck(self.eng.lib.gpmp2mi_plan_fix_state(self.h.ptr, int(b), int(state_idx), dptr(c), dptr(v)))

    def add_state_estimate(self, b, state_idx, conf, conf_cov, vel=None, vel_cov=None):
        c, cc = f64(conf).reshape(self.D), f64(conf_cov).reshape(self.D, self.D)
        v = None if vel is None else f64(vel).reshape(self.D)
        vc = None if vel_cov is None else f64(vel_cov).reshape(self.D, self.D)
        self.eng._ck(self.eng.lib.gpmp2mi_plan_add_state_estimate(self.h.ptr, int(b), int(state_idx), dptr(c), dptr(cc),
                                                                  dptr(v), dptr(vc)))

    def change_goal(self, b, goal_conf, goal_vel):
        c, v = f64(goal_conf).reshape(self.D), f64(goal_vel).reshape(self.D)
        self.eng._ck(self.eng.lib.gpmp2mi_plan_change_goal(self.h.ptr, int(b), dptr(c), dptr(v)))

    def remove_goal(self, b):
        self.eng._ck(self.eng.lib.gpmp2mi_plan_remove_goal(self.h.ptr, int(b)))

    def clear_state_priors(self, b):
        self.eng._ck(self.eng.lib.gpmp2mi_plan_clear_state_priors(self.h.ptr, int(b)))

    def update(self, iterations=1, stream=None):
        self.eng._ck(self.eng.lib.gpmp2mi_plan_update(self.h.ptr, int(iterations), C.c_void_p(stream or 0)))

    def debug_scalars(self, b):
        out = np.zeros(17)
        self.eng._ck(self.eng.lib.gpmp2mi_plan_debug_scalars(self.h.ptr, int(b), dptr(out)))
        return dict(gd=out[0], dd=out[1], gg=out[2], ghg=out[3], gn=out[4], nn=out[5], q=out[6], xnorm=out[7], radius=out[16])

    def enable_timing(self, on=True):
        self.eng._ck(self.eng.lib.gpmp2mi_plan_enable_timing(self.h.ptr, int(on)))

    def timing(self):
        n = C.c_int(16)
        names = (C.c_char_p * 16)()
        ms = (C.c_double * 16)()
        launches = (C.c_int * 16)()
        self.eng._ck(self.eng.lib.gpmp2mi_plan_get_timing(self.h.ptr, C.byref(n), names, ms, launches))
        return {names[i].decode(): dict(ms=ms[i], launches=launches[i]) for i in range(min(n.value, 16))}
