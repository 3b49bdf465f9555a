// launch.h -- host-callable launchers implemented in the .hip translation units.
#pragma once
#include "common.h"

namespace g2 {

// sdf_kernels.hip
int launch_sdf_from_occupancy(int nx, int ny, int nz, const double* occ, double cell, int* wa, int* wb,
                              double* field, hipStream_t st);
// factor_kernels.hip
int launch_sdf_pack(const SdfDev& s, double* cells, hipStream_t st);
int launch_sdf_query(const SdfDev& s, int M, const double* pts, double* dist, double* grad, int* inr,
                     hipStream_t st);
// ld: leading dimension of conf (0 = dof; 2 dof reads the configurations out of trajectory states)
int launch_sphere_centers(const RobotDev& h, const RobotDev* R, int M, const double* conf, double* c,
                          double* J, hipStream_t st, int ld = 0);
int launch_fk(const RobotDev& h, const RobotDev* R, int M, const double* conf, double* poses, double* J,
              hipStream_t st, int ld = 0);
int launch_obstacle(const RobotDev& h, const RobotDev* R, const SdfDev& s, double eps, int M,
                    const double* conf, double* err, double* H1, hipStream_t st);
int launch_obstacle_gp(const RobotDev& h, const RobotDev* R, const SdfDev& s, double eps, const GpCoef& gc,
                       int M, const double* c1, const double* v1, const double* c2, const double* v2,
                       double* err, double* H1, double* H2, double* H3, double* H4, hipStream_t st);
int launch_gp_prior_linear(int D, double dt, int M, const double* c1, const double* v1, const double* c2,
                           const double* v2, double* err, double* H1, double* H2, double* H3, double* H4,
                           hipStream_t st);
int launch_gp_interp_linear(int D, const GpCoef& gc, int M, const double* c1, const double* v1,
                            const double* c2, const double* v2, double* conf, double* vel, hipStream_t st);
int launch_gp_prior_lie(int D, double dt, int M, const double* c1, const double* v1, const double* c2,
                        const double* v2, double* err, double* H1, double* H2, double* H3, double* H4,
                        hipStream_t st);
int launch_interpolate_traj(int D, bool lie, double dt, int inter, int B, int N, int start, int Mo,
                            const double* traj, double* out, hipStream_t st);
int launch_vehicle_dynamics(int D, int lie, int M, const double* conf, const double* vel, double* err, double* Hp,
                            double* Hv, hipStream_t st);
int launch_workspace_prior(int mode, int joint, int L, int D, int M, const double* des, const double* poses,
                           const double* Jp, double* err, double* H, hipStream_t st);
int launch_self_collision(int n, int S, int D, int M, const double* data, const double* radius, const double* c,
                          const double* Jc, double* err, double* H, hipStream_t st);
int launch_gp_interp_lie(int D, const GpCoef& gc, int M, const double* c1, const double* v1, const double* c2,
                         const double* v2, double* conf, double* vel, hipStream_t st);
int launch_joint_limit(int D, const double* down, const double* up, const double* th, int M,
                       const double* x, double* err, double* Hd, hipStream_t st);

}  // namespace g2
