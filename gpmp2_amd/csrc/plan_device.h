// plan_device.h -- device helpers shared by the plan kernels: graph-error terms and the
// gtsam::checkConvergence rule.
#pragma once
#include "device_math.h"
#include "plan.h"
#include "tiles.h"

namespace g2 {

// =============================================================================== error terms
// prior + limit + vehicle-dynamics error of one trajectory (0.5 * whitened squared residuals),
// wave-reduced.  PriorFactor (planner/BatchTrajOptimizer-inl.h:41-48), JointLimitFactorVector,
// VelocityLimitFactorVector (:50-59), VehicleDynamicsFactor (dynamics/VehicleDynamics.h:19-27).
// this thread's share when `nthr` threads split the entries (not yet reduced)
__device__ __forceinline__ double misc_error_partial(const PlanParams& P, const PlanBuffers& pb, int b,
                                                     const double* __restrict__ tr, int tid, int nthr) {
  const int D = P.D, n = P.n, N = P.N;
  double acc = 0.0;
  // without limit / dynamics factors and replanner priors only the first and the last state carry terms
  const int nxp = pb.xp_n[b];
  const bool every_state = P.flag_pos_limit || P.flag_vel_limit || P.vdyn_w > 0.0 || nxp > 0;
  const int count = every_state ? (N + 1) * n : (N > 0 ? 2 * n : n);
  for (int q = tid; q < count; q += nthr) {
    const int idx = (every_state || q < n) ? q : N * n + (q - n);
    const int i = idx / n, rho = idx - i * n;
    const int a = rho >= D, k = rho - a * D;
    const double z = tr[idx];
    if (i == 0 || (i == N && pb.goal_on[b] && (a || !P.end_conf_prior_off))) {
      const double* tg = (i == 0) ? (a ? pb.start_vel : pb.start_conf) : (a ? pb.end_vel : pb.end_conf);
      tg += (size_t)b * D;
      double d = z - tg[k];
      if (P.lie && !a && k < 3) {  // -Local(x, prior) of PriorFactor<Pose2Vector>
        const double* zs = tr + (size_t)i * n;
        const P2 bt = pose2_between(P2{zs[0], zs[1], zs[2]}, P2{tg[0], tg[1], tg[2]});
        d = -(k == 0 ? bt.x : k == 1 ? bt.y : bt.th);
      }
      acc += (a ? P.vel_prior_w : P.conf_prior_w) * d * d;
    }
    for (int e = 0; e < nxp; e++) {  // replanner state priors: r^T W r, row k's share
      const size_t xe = (size_t)b * XP_MAX + e;
      if (pb.xp_state[xe] != i || (a && !pb.xp_has_vel[xe])) continue;
      const double* Wm = pb.xp_info + (xe * 2 + a) * D * D + (size_t)k * D;
      const double* tg = pb.xp_target + xe * n + a * D;
      const double* zs = tr + (size_t)i * n;
      double wr = 0.0, rk = 0.0;
      for (int cc = 0; cc < D; cc++) {
        double rc = zs[a * D + cc] - tg[cc];
        if (P.lie && !a && cc < 3) {
          const P2 bt = pose2_between(P2{zs[0], zs[1], zs[2]}, P2{tg[0], tg[1], tg[2]});
          rc = -(cc == 0 ? bt.x : cc == 1 ? bt.y : bt.th);
        }
        wr = fma(Wm[cc], rc, wr);
        if (cc == k) rk = rc;
      }
      acc += wr * rk;
    }
    double H;
    if (!a && P.flag_pos_limit && !(P.lie && k < 3)) {
      const double e = hinge_limit(z, P.pos_lo[k], P.pos_hi[k], P.pos_th[k], H);
      acc += P.pos_w[k] * e * e;
    }
    if (a && P.flag_vel_limit) {
      const double e = hinge_limit(z, -P.vel_lim[k], P.vel_lim[k], P.vel_th[k], H);
      acc += P.vel_w[k] * e * e;
    }
    if (a && k == 1 && P.vdyn_w > 0.0) acc += P.vdyn_w * z * z;
  }
  return acc;
}
__device__ __forceinline__ double misc_error(const PlanParams& P, const PlanBuffers& pb, int b,
                                             const double* __restrict__ tr, int lane) {
  return wave_sum(misc_error_partial(P, pb, b, tr, lane, 64));
}

// total graph error of trajectory b from its point records: 0.5 * (sum e_p + sum gp energy + misc)
__device__ __forceinline__ double total_error(const PlanParams& P, const PlanBuffers& pb, int b,
                                              const double* __restrict__ tr,
                                              const double* __restrict__ rec,
                                              const double* __restrict__ gpu, int lane) {
  const double* eb = rec + (size_t)b * P.Ppad * P.RECS + (P.NG + P.D);
  double acc = 0.0;
  for (int p = lane; p < P.P; p += 64) acc += eb[(size_t)p * P.RECS];
  const double* gb = gpu + (size_t)b * P.Npad * P.GPS + P.n;
  for (int i = 1 + lane; i <= P.N; i += 64) acc += gb[(size_t)i * P.GPS];
  return 0.5 * (wave_sum(acc) + misc_error(P, pb, b, tr, lane));
}

// the same with the work split over the `nthr` threads of a workgroup: this thread's share of
// sum e_p + sum gp energy + misc (the caller reduces and halves)
__device__ __forceinline__ double total_error_partial(const PlanParams& P, const PlanBuffers& pb, int b,
                                                      const double* __restrict__ tr, const double* __restrict__ rec,
                                                      const double* __restrict__ gpu, int tid, int nthr) {
  const double* eb = rec + (size_t)b * P.Ppad * P.RECS + (P.NG + P.D);
  double acc = 0.0;
  for (int p = tid; p < P.P; p += nthr) acc += eb[(size_t)p * P.RECS];
  const double* gb = gpu + (size_t)b * P.Npad * P.GPS + P.n;
  for (int i = 1 + tid; i <= P.N; i += nthr) acc += gb[(size_t)i * P.GPS];
  return acc + misc_error_partial(P, pb, b, tr, tid, nthr);
}

__device__ __forceinline__ bool check_convergence(double rel, double abs_, double err_tol, double cur,
                                                  double nw) {
  if (nw <= err_tol) return true;
  const double abs_dec = cur - nw;
  const double rel_dec = abs_dec / cur;
  return (rel != 0.0 && rel_dec <= rel) || (abs_dec <= abs_);
}


// Called by one thread of every workgroup of the kernel that closes a pass, after its own work: the last
// workgroup to arrive publishes the pass's active count to the host-mapped flag array, so the host
// driver learns it without a copy command or an event in the stream.
__device__ __forceinline__ void publish_pass_count(const PlanBuffers& pb, int pass) {
  __threadfence();
  if (atomicAdd(pb.done + pass, 1) == (int)gridDim.x - 1) {
    const int v = atomicAdd(pb.n_active + pass, 0);
    __hip_atomic_store(pb.host_flags + pass, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

}  // namespace g2
