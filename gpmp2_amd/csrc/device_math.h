// device_math.h -- device building blocks: SDF lookup, hinge costs, DH forward kinematics with
// geometric sphere Jacobians.  Everything is fp64; dof-dependent loops are fully unrolled on the
// template parameters so that per-lane arrays stay in VGPRs (no runtime-indexed private arrays).
#pragma once
#include <type_traits>

#include "common.h"

namespace g2 {

// compile-time loop: f(std::integral_constant<int, I0>{}), ..., f(std::integral_constant<int, I1 - 1>{})
template <int I0, int I1, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I0 < I1) {
    f(std::integral_constant<int, I0>{});
    static_for<I0 + 1, I1>(f);
  }
}

// ---------------------------------------------------------------------------------------------
// Signed distance lookup.  Arithmetic follows the reference term by term
// (gpmp2/obstacle/SignedDistanceField.h:103-167, gpmp2/obstacle/PlanarSDF.h:71-116); the only
// difference is the memory layout: one 64-B (3-D) / 32-B (2-D) packed cell per lookup.
// Returns false where the reference throws SDFQueryOutOfRange.
// ---------------------------------------------------------------------------------------------
// x / cell with the stored reciprocal: q = x * (1 / cell) corrected by one residual step, q + (x - q cell) (1 / cell).
// The result is the correctly rounded quotient except in rare double-rounding cases (then 1 ulp off); it replaces the
// ~15-instruction IEEE division sequence (two of its instructions quarter rate) that every lookup paid three times.
__device__ __forceinline__ double div_cell(double x, double cell, double inv_cell) {
  const double q = x * inv_cell;
  return fma(fma(-q, cell, x), inv_cell, q);
}

__device__ __forceinline__ bool sdf3_lookup(const SdfDev& s, double px, double py, double pz,
                                            double& dist, double& gx, double& gy, double& gz) {
  if (px < s.ox || px > s.hix || py < s.oy || py > s.hiy || pz < s.oz || pz > s.hiz) return false;
  const double col = div_cell(px - s.ox, s.cell, s.inv_cell), row = div_cell(py - s.oy, s.cell, s.inv_cell),
               z = div_cell(pz - s.oz, s.cell, s.inv_cell);
  const double lr = floor(row), lc = floor(col), lz = floor(z);
  const int lri = (int)lr, lci = (int)lc, lzi = (int)lz;
#ifdef G2_SDF_PLAIN
  // A/B build only: the same lookup from the plain [z][y][x] field (8 B per voxel, 1/8 of the memory): eight 8-B
  // loads from four cache lines; upper-face neighbours clamped (their weights are exactly 0)
  const int hri = min(lri + 1, s.ny - 1), hci = min(lci + 1, s.nx - 1), hzi = min(lzi + 1, s.nz - 1);
  auto at = [&](int zz, int rr, int cc) { return s.plain[((size_t)zz * s.ny + rr) * s.nx + cc]; };
  const double v000 = at(lzi, lri, lci), v010 = at(lzi, lri, hci), v100 = at(lzi, hri, lci), v110 = at(lzi, hri, hci);
  const double v001 = at(hzi, lri, lci), v011 = at(hzi, lri, hci), v101 = at(hzi, hri, lci), v111 = at(hzi, hri, hci);
#else
  const double2* c =
      reinterpret_cast<const double2*>(s.cells + (((size_t)lzi * s.ny + lri) * s.nx + lci) * 8);
  const double2 a0 = c[0], a1 = c[1], a2 = c[2], a3 = c[3];  // 4 x 16-B loads of one 64-B cell
  // [dz][dy=row][dx=col]
  const double v000 = a0.x, v010 = a0.y, v100 = a1.x, v110 = a1.y;  // (row,col,z): vRCZ
  const double v001 = a2.x, v011 = a2.y, v101 = a3.x, v111 = a3.y;
#endif
  // Trilinear value and gradient in nested form: interpolate along the row, then the column, then z.  The reference
  // (SignedDistanceField.h:110-167) writes the same polynomial as eight weighted corners and three sums of four weighted
  // differences (76 operations); nested, the value takes 7 interpolations and the three gradients reuse their
  // differences (22 operations).  Results differ from the corner form by rounding only (a few ulp).
  const double tr = row - lr, tc = col - lc, tz = z - lz;
  const double d00 = v100 - v000, d10 = v110 - v010, d01 = v101 - v001, d11 = v111 - v011;   // d_cz = v1cz - v0cz
  const double a00 = fma(tr, d00, v000), a10 = fma(tr, d10, v010), a01 = fma(tr, d01, v001), a11 = fma(tr, d11, v011);
  const double e0 = a10 - a00, e1 = a11 - a01;                                                // along the column
  const double b0 = fma(tc, e0, a00), b1 = fma(tc, e1, a01);
  const double g_z = b1 - b0;
  dist = fma(tz, g_z, b0);
  const double g_col = fma(tz, e1 - e0, e0);
  const double r0 = fma(tc, d10 - d00, d00), r1 = fma(tc, d11 - d01, d01);
  const double g_row = fma(tz, r1 - r0, r0);
  // (g_idx / cell_size of SignedDistanceField.h:97 as a multiplication by the stored reciprocal: <= 1 ulp)
  gx = g_col * s.inv_cell;
  gy = g_row * s.inv_cell;
  gz = g_z * s.inv_cell;
  return true;
}

__device__ __forceinline__ bool sdf2_lookup(const SdfDev& s, double px, double py, double& dist,
                                            double& gx, double& gy) {
  if (px < s.ox || px > s.hix || py < s.oy || py > s.hiy) return false;
  const double col = div_cell(px - s.ox, s.cell, s.inv_cell), row = div_cell(py - s.oy, s.cell, s.inv_cell);
  const double lr = floor(row), lc = floor(col);
  const int lri = (int)lr, lci = (int)lc;
  const double2* c = reinterpret_cast<const double2*>(s.cells + ((size_t)lri * s.nx + lci) * 4);
  const double2 a0 = c[0], a1 = c[1];
  const double v00 = a0.x, v01 = a0.y, v10 = a1.x, v11 = a1.y;  // vRC
  // bilinear value and gradient in nested form (see sdf3_lookup; PlanarSDF.h:80-116 writes the four weighted corners)
  const double tr = row - lr, tc = col - lc;
  const double d0 = v10 - v00, d1 = v11 - v01;          // along the row, at column 0 / 1
  const double i0 = fma(tr, d0, v00), i1 = fma(tr, d1, v01);
  const double g_col = i1 - i0;
  dist = fma(tc, g_col, i0);
  const double g_row = fma(tc, d1 - d0, d0);
  gx = g_col * s.inv_cell;
  gy = g_row * s.inv_cell;
  return true;
}

// hingeLossObstacleCost  gpmp2/obstacle/ObstacleCost.h:26-78.  (hx,hy,hz) = d cost / d point.
template <int SDIM>
__device__ __forceinline__ double hinge_obstacle(const SdfDev& s, double px, double py, double pz,
                                                 double eps, double& hx, double& hy, double& hz) {
  double d, gx, gy, gz = 0.0;
  bool ok;
  if (SDIM == 3) ok = sdf3_lookup(s, px, py, pz, d, gx, gy, gz);
  else ok = sdf2_lookup(s, px, py, d, gx, gy);
  hx = hy = hz = 0.0;
  if (!ok) return 0.0;
  if (d > eps) return 0.0;
  hx = -gx;
  hy = -gy;
  hz = -gz;
  return eps - d;
}

// hingeLossJointLimitCost  gpmp2/kinematics/JointLimitCost.h:16-31
__device__ __forceinline__ double hinge_limit(double p, double lo, double hi, double th, double& H) {
  if (p < lo + th) {
    H = -1.0;
    return lo + th - p;
  } else if (p <= hi - th) {
    H = 0.0;
    return 0.0;
  } else {
    H = 1.0;
    return p - hi + th;
  }
}

// ---------------------------------------------------------------------------------------------
// Kinematic chain state: world rotation (columns c0,c1,c2) and origin t of the current frame.
// ---------------------------------------------------------------------------------------------
struct Frame {
  double c0[3], c1[3], c2[3], t[3];
};

__device__ __forceinline__ void frame_from_3x4(const double* M, Frame& F) {
#pragma unroll
  for (int i = 0; i < 3; i++) {
    F.c0[i] = M[i * 4 + 0];
    F.c1[i] = M[i * 4 + 1];
    F.c2[i] = M[i * 4 + 2];
    F.t[i] = M[i * 4 + 3];
  }
}

// F <- F * Rz(theta) * Tz(d) * Tx(a) * Rx(alpha)   (Arm::getJointTrans, kinematics/Arm.h:93-98,
// link_trans_notheta_ kinematics/Arm.cpp:23-27)
__device__ __forceinline__ void dh_advance_sc(Frame& F, double s, double c, double a, double d, double ca,
                                              double sa) {
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const double n0 = c * F.c0[i] + s * F.c1[i];
    const double n1 = -s * F.c0[i] + c * F.c1[i];
    const double z = F.c2[i];
    F.t[i] = F.t[i] + a * n0 + d * z;
    F.c0[i] = n0;
    F.c1[i] = ca * n1 + sa * z;
    F.c2[i] = -sa * n1 + ca * z;
  }
}
__device__ __forceinline__ void dh_advance(Frame& F, double theta, double a, double d, double ca,
                                           double sa) {
  double s, c;
  sincos(theta, &s, &c);
  dh_advance_sc(F, s, c, a, d, ca, sa);
}

// Generic visitor-style forward kinematics over a sphere model.
//   KIND     : GPMP2MI_ROBOT_*
//   AD       : number of DH joints of the (first) arm (compile time; 0 for POINT / MOBILE_BASE)
//   AD2      : number of DH joints of the second arm (two-arm robots)
// Robot = [vehicle base (Pose2)] [+ vertical lift torso] + arm 1 [+ arm 2]; configuration
// [x, y, theta, (lift), q_arm1, q_arm2]; links in the reference's order: vehicle base, (torso),
// arm-1 links, arm-2 links (kinematics/Pose2Mobile2Arms.cpp:32-108, Pose2MobileVetLinArm.cpp:31-108,
// Pose2MobileVetLin2Arms.cpp:36-114).  Sphere Jacobians are world-frame columns
// Jcol[k] = d p / d q_k (GTSAM right-perturbation for the Pose2 part): equivalent to
// RobotModel::sphereCenters (kinematics/RobotModel-inl.h:12-40) composed with the pose Jacobians of
// the FK models (kinematics/Arm.cpp:105-115): arm joint column = z_k x (p - o_k)  (SURVEY.md A.2).
//
// Column tag of a link: value = NC, columns >= NC are zero; columns [NB, first) are zero too (the
// other arm's joints); columns [first, NC) belong to joints (first - NB) .. of the DH table.
template <int NC, int FIRST>
struct ColTag {
  static constexpr int value = NC;
  static constexpr int first = FIRST;
};

template <int KIND, int AD, int AD2 = 0>
struct Kin {
  static constexpr bool MOBILE = KIND >= GPMP2MI_ROBOT_POSE2_MOBILE_BASE;
  static constexpr int BASE = MOBILE ? 3 : 0;
  static constexpr int LIFT = (KIND == GPMP2MI_ROBOT_POSE2_MOBILE_VETLIN_ARM ||
                               KIND == GPMP2MI_ROBOT_POSE2_MOBILE_VETLIN_2ARMS) ? 1 : 0;
  static constexpr int NB = BASE + LIFT;      // columns in front of the arm joints
  static constexpr int NJ = AD + AD2;         // DH joints, arm 1 first
  static constexpr int DOF = (KIND == GPMP2MI_ROBOT_POINT) ? 2 : NB + NJ;
  static constexpr int NLINKS = (KIND == GPMP2MI_ROBOT_ARM) ? AD : MOBILE ? 1 + LIFT + NJ : 1;

  // Joint axes / origins of one configuration: all a point Jacobian needs besides the point itself.
  struct Axes {
    double zax[NJ > 0 ? NJ : 1][3], org[NJ > 0 ? NJ : 1][3];
    double vt[3], bx[3], by[3];  // vehicle origin and heading columns (mobile robots)
    double lift_sign;            // +-1: d torso / d lift = lift_sign * e_z
  };

  __device__ __forceinline__ static void compose(const Frame& P, const double* M3x4, Frame& N) {
    Frame B;
    frame_from_3x4(M3x4, B);
#pragma unroll
    for (int i = 0; i < 3; i++) {
      N.c0[i] = P.c0[i] * B.c0[0] + P.c1[i] * B.c0[1] + P.c2[i] * B.c0[2];
      N.c1[i] = P.c0[i] * B.c1[0] + P.c1[i] * B.c1[1] + P.c2[i] * B.c1[2];
      N.c2[i] = P.c0[i] * B.c2[0] + P.c1[i] * B.c2[1] + P.c2[i] * B.c2[2];
      N.t[i] = P.t[i] + P.c0[i] * B.t[0] + P.c1[i] * B.t[1] + P.c2[i] * B.t[2];
    }
  }

  // Walk the kinematic tree once: f(link, F, tag) for every link in the reference's order with its
  // world frame F and column tag; fills A on the way (A.zax[k], A.org[k] valid for the joints the
  // tag covers when f is called).  POINT robots have no frame walk (see walk()).
  template <class F>
  __device__ __forceinline__ static void walk_links(const RobotDev& R, const double (&q)[DOF], Axes& A, F&& f) {
    static_assert(KIND != GPMP2MI_ROBOT_POINT, "point robots have no link frames");
    Frame Fr;
    A.vt[0] = A.vt[1] = A.vt[2] = 0.0;
    A.lift_sign = 1.0;
    if constexpr (MOBILE) {
      // computeBasePose3  kinematics/mobileBaseUtils.cpp:18-31
      double sn, c;
      sincos(q[2], &sn, &c);
      Fr.c0[0] = c; Fr.c0[1] = sn; Fr.c0[2] = 0;
      Fr.c1[0] = -sn; Fr.c1[1] = c; Fr.c1[2] = 0;
      Fr.c2[0] = 0; Fr.c2[1] = 0; Fr.c2[2] = 1;
      Fr.t[0] = q[0]; Fr.t[1] = q[1]; Fr.t[2] = 0;
      A.vt[0] = q[0]; A.vt[1] = q[1];
      A.bx[0] = c; A.bx[1] = sn; A.bx[2] = 0;
      A.by[0] = -sn; A.by[1] = c; A.by[2] = 0;
      f(0, Fr, ColTag<3, 3>{});  // link 0 = vehicle base
      if constexpr (LIFT == 1) {
        // liftBasePose3  kinematics/mobileBaseUtils.cpp:51-82: a world-z translation composed on
        // the LEFT of veh * base_T_torso
        Frame T;
        compose(Fr, R.base, T);
        A.lift_sign = R.reverse_linact ? -1.0 : 1.0;
        T.t[2] += A.lift_sign * q[3];
        Fr = T;
        f(1, Fr, ColTag<NB, NB>{});  // link 1 = torso
      }
    } else {
      frame_from_3x4(R.base, Fr);  // ARM: world_T_base
    }
    if constexpr (NJ > 0) {
      const Frame parent = Fr;
      // arm 1: computeBaseTransPose3 (mobileBaseUtils.cpp:34-48) / torso.compose(torso_T_arm)
      if constexpr (MOBILE) compose(parent, LIFT ? R.base2 : R.base, Fr);
      static_for<0, AD>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
#pragma unroll
        for (int i = 0; i < 3; i++) {
          A.zax[j][i] = Fr.c2[i];
          A.org[j][i] = Fr.t[i];
        }
        dh_advance(Fr, q[NB + j] + R.bias[j], R.a[j], R.d[j], R.ca[j], R.sa[j]);
        f((MOBILE ? 1 + LIFT : 0) + j, Fr, ColTag<NB + j + 1, NB>{});
      });
      if constexpr (AD2 > 0) {
        compose(parent, LIFT ? R.base3 : R.base2, Fr);
        static_for<0, AD2>([&](auto jc) {
          constexpr int j = AD + decltype(jc)::value;
#pragma unroll
          for (int i = 0; i < 3; i++) {
            A.zax[j][i] = Fr.c2[i];
            A.org[j][i] = Fr.t[i];
          }
          dh_advance(Fr, q[NB + j] + R.bias[j], R.a[j], R.d[j], R.ca[j], R.sa[j]);
          f(1 + LIFT + j, Fr, ColTag<NB + j + 1, NB + AD>{});
        });
      }
    }
  }

  // f(s, p, tag) for every body sphere in sorted (= link) order, with its world centre p
  // sub / nsub: visit only the spheres s with s % nsub == sub (both uniform over the wavefront).
  template <class F>
  __device__ __forceinline__ static void walk(const RobotDev& R, const double (&q)[DOF], Axes& A, F&& f,
                                              int sub = 0, int nsub = 1) {
    auto mine = [&](int s) { return nsub == 1 || (s % nsub) == sub; };
    if constexpr (KIND == GPMP2MI_ROBOT_POINT) {
      // PointRobot::forwardKinematics  kinematics/PointRobot.cpp:15-49
      for (int s = 0; s < R.nr_spheres; s++) {
        if (!mine(s)) continue;
        const double p[3] = {q[0] + R.sph_c[3 * s], q[1] + R.sph_c[3 * s + 1], R.sph_c[3 * s + 2]};
        f(s, p, ColTag<2, 0>{});
      }
    } else {
      walk_links(R, q, A, [&](int link, const Frame& Fr, auto tag) {
        for (int s = R.link_first[link]; s < R.link_first[link + 1]; s++) {
          if (!mine(s)) continue;
          double p[3];
#pragma unroll
          for (int i = 0; i < 3; i++)
            p[i] = Fr.t[i] + Fr.c0[i] * R.sph_c[3 * s] + Fr.c1[i] * R.sph_c[3 * s + 1] +
                   Fr.c2[i] * R.sph_c[3 * s + 2];
          f(s, p, tag);
        }
      });
    }
  }

  // Jacobian columns of a point p rigidly attached to a link with column tag `tag`:
  // J[k] = d p / d q_k for k < tag.value (columns >= tag.value are not touched).
  template <class Tag>
  __device__ __forceinline__ static void jacobian(const Axes& A, const double (&p)[3], Tag, double (&J)[DOF][3]) {
    constexpr int NC = Tag::value, FIRST = Tag::first;
    if constexpr (KIND == GPMP2MI_ROBOT_POINT) {
      J[0][0] = 1.0; J[0][1] = 0.0; J[0][2] = 0.0;
      J[1][0] = 0.0; J[1][1] = 1.0; J[1][2] = 0.0;
    } else {
      if constexpr (MOBILE) {
#pragma unroll
        for (int i = 0; i < 3; i++) { J[0][i] = A.bx[i]; J[1][i] = A.by[i]; }
        J[2][0] = -(p[1] - A.vt[1]);  // z x (p - t_veh)
        J[2][1] = (p[0] - A.vt[0]);
        J[2][2] = 0.0;
        if constexpr (LIFT == 1 && NC > 3) { J[3][0] = 0.0; J[3][1] = 0.0; J[3][2] = A.lift_sign; }
      }
#pragma unroll
      for (int k = NB; k < (FIRST < NC ? FIRST : NC); k++) J[k][0] = J[k][1] = J[k][2] = 0.0;  // the other arm
#pragma unroll
      for (int k = FIRST; k < NC; k++) {
        const int jn = k - NB;
        const double rx = p[0] - A.org[jn][0], ry = p[1] - A.org[jn][1], rz = p[2] - A.org[jn][2];
        J[k][0] = A.zax[jn][1] * rz - A.zax[jn][2] * ry;
        J[k][1] = A.zax[jn][2] * rx - A.zax[jn][0] * rz;
        J[k][2] = A.zax[jn][0] * ry - A.zax[jn][1] * rx;
      }
    }
  }

  // Visitor over the body spheres (sorted by link):
  //   pre(s, p) -> bool      : called with the sphere centre; return true if the Jacobian is wanted
  //   post(s, p, J, nc)      : J[k] = d p / d q_k for k < nc.value (compile time),
  //                            zero for k >= nc.value -- a sphere depends on the joints below its link only
  template <class Pre, class Post>
  __device__ __forceinline__ static void visit_spheres(const RobotDev& R, const double (&q)[DOF], Pre&& pre,
                                                       Post&& post, int sub = 0, int nsub = 1) {
    Axes A;
    double J[DOF][3];
#pragma unroll
    for (int k = 0; k < DOF; k++) J[k][0] = J[k][1] = J[k][2] = 0.0;
    walk(R, q, A, [&](int s, const double (&p)[3], auto nc) {
      if (!pre(s, p)) return;
      jacobian(A, p, nc, J);
      post(s, p, J, nc);
    }, sub, nsub);
  }

  // simple form: f(sorted_index, p[3], Jcol[DOF][3], ncols) for every sphere
  template <class F>
  __device__ __forceinline__ static void for_each_sphere(const RobotDev& R, const double (&q)[DOF], F&& f,
                                                         int sub = 0, int nsub = 1) {
    visit_spheres(R, q, [](int, const double (&)[3]) { return true; },
                  [&](int s, const double (&p)[3], const double (&J)[DOF][3], auto nc) { f(s, p, J, (int)decltype(nc)::value); },
                  sub, nsub);
  }
};

// ---------------------------------------------------------------------------------------------
// Pose2 (GTSAM semantics; restated in the oracle as well): the Lie part of
// Pose2Vector = Pose2 x R^k states [x, y, theta, q...]  (gpmp2/geometry/Pose2Vector.h:26-73).
// ---------------------------------------------------------------------------------------------
struct P2 {
  double x, y, th;
};
__device__ __forceinline__ double wrap_angle(double c, double s) { return atan2(s, c); }
// between(a, b) = a^-1 * b
__device__ __forceinline__ P2 pose2_between(const P2& a, const P2& b) {
  double sa, ca;
  sincos(a.th, &sa, &ca);
  const double dx = b.x - a.x, dy = b.y - a.y;
  double sd, cd;
  sincos(b.th - a.th, &sd, &cd);
  return P2{ca * dx + sa * dy, -sa * dx + ca * dy, wrap_angle(cd, sd)};
}
// Pose2::Logmap
__device__ __forceinline__ void pose2_logmap(const P2& p, double (&v)[3]) {
  const double w = p.th;
  if (fabs(w) < 1e-10) {
    v[0] = p.x; v[1] = p.y; v[2] = w;
  } else {
    double s, c;
    sincos(w, &s, &c);
    const double c_1 = c - 1.0, det = c_1 * c_1 + s * s;
    const double ux = c * p.x + s * p.y - p.x, uy = -s * p.x + c * p.y - p.y;
    v[0] = (w / det) * (-uy);
    v[1] = (w / det) * ux;
    v[2] = w;
  }
}
// Pose2::AdjointMap (row-major 3x3)
__device__ __forceinline__ void pose2_adjoint(const P2& p, double (&A)[9]) {
  double s, c;
  sincos(p.th, &s, &c);
  A[0] = c; A[1] = -s; A[2] = p.y; A[3] = s; A[4] = c; A[5] = -p.x; A[6] = 0; A[7] = 0; A[8] = 1;
}
__device__ __forceinline__ P2 pose2_inverse(const P2& a) {
  double s, c;
  sincos(a.th, &s, &c);
  return P2{-(c * a.x + s * a.y), -(-s * a.x + c * a.y), wrap_angle(c, -s)};
}
// Pose2::LogmapDerivative
__device__ __forceinline__ void pose2_logmap_derivative(const P2& p, double (&J)[9]) {
  double v[3];
  pose2_logmap(p, v);
  const double alpha = v[2];
  if (fabs(alpha) > 1e-5) {
    const double ai = 1 / alpha, hc = 0.5 * sin(alpha) / (1 - cos(alpha));
    J[0] = alpha * hc; J[1] = -0.5 * alpha; J[2] = v[0] * ai - v[0] * hc + 0.5 * v[1];
    J[3] = 0.5 * alpha; J[4] = alpha * hc; J[5] = v[1] * ai - 0.5 * v[0] - v[1] * hc;
    J[6] = 0; J[7] = 0; J[8] = 1;
  } else {
    J[0] = 1; J[1] = 0; J[2] = 0.5 * v[1]; J[3] = 0; J[4] = 1; J[5] = -0.5 * v[0]; J[6] = 0; J[7] = 0; J[8] = 1;
  }
}
// Pose2::Expmap
__device__ __forceinline__ P2 pose2_expmap(const double (&v)[3]) {
  const double w = v[2];
  if (fabs(w) < 1e-10) return P2{v[0], v[1], v[2]};
  double s, c;
  sincos(w, &s, &c);
  const double ox = -v[1], oy = v[0];
  const double rx = c * ox - s * oy, ry = s * ox + c * oy;
  return P2{(ox - rx) / w, (oy - ry) / w, wrap_angle(c, s)};
}
// Pose2::ExpmapDerivative
__device__ __forceinline__ void pose2_expmap_derivative(const double (&v)[3], double (&J)[9]) {
  const double alpha = v[2];
  if (fabs(alpha) > 1e-5) {
    const double sZ = sin(alpha) / alpha, c1Z = (cos(alpha) - 1) / alpha;
    const double v1Z = v[0] / alpha, v2Z = v[1] / alpha;
    J[0] = sZ; J[1] = -c1Z; J[2] = v1Z + v2Z * c1Z - v1Z * sZ;
    J[3] = c1Z; J[4] = sZ; J[5] = -v1Z * c1Z + v2Z - v2Z * sZ;
    J[6] = 0; J[7] = 0; J[8] = 1;
  } else {
    J[0] = 1; J[1] = 0; J[2] = -0.5 * v[1]; J[3] = 0; J[4] = 1; J[5] = 0.5 * v[0]; J[6] = 0; J[7] = 0; J[8] = 1;
  }
}
__device__ __forceinline__ P2 pose2_compose(const P2& a, const P2& b) {
  double s, c, sb, cb;
  sincos(a.th, &s, &c);
  sincos(b.th, &sb, &cb);
  return P2{a.x + c * b.x - s * b.y, a.y + s * b.x + c * b.y, wrap_angle(c * cb - s * sb, s * cb + c * sb)};
}
__device__ __forceinline__ void mat3_mul(const double (&A)[9], const double (&B)[9], double (&C)[9]) {
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}
// GaussianProcessInterpolatorLie<Pose2Vector>::interpolatePose  gp/GaussianProcessInterpolatorLie.h:64-100
// for states [x, y, theta, q...].  Lambda / Psi are (2x2) (x) I, so the arm part is the linear
// interpolation and only the 3x3 pose blocks M1..M4 of the four Jacobians are non-trivial:
//   Hint_k = diag(M_k, s_k I),  s = (l11, l12, p11, p12).
template <int D>
__device__ __forceinline__ void lie_interpolate(const GpCoef& c, const double (&x0)[D], const double (&v0)[D],
                                                const double (&x1)[D], const double (&v1)[D], double (&q)[D],
                                                double (*M)[9] /* [4][9] or nullptr */) {
  const P2 p1{x0[0], x0[1], x0[2]}, p2{x1[0], x1[1], x1[2]};
  const P2 bt = pose2_between(p1, p2);
  double lg[3];
  pose2_logmap(bt, lg);
  double xi[3];
#pragma unroll
  for (int k = 0; k < 3; k++) xi[k] = c.l12 * v0[k] + c.p11 * lg[k] + c.p12 * v1[k];
  const P2 e = pose2_expmap(xi);
  const P2 p = pose2_compose(p1, e);
  q[0] = p.x; q[1] = p.y; q[2] = p.th;
#pragma unroll
  for (int k = 3; k < D; k++) q[k] = x0[k] + (c.l12 * v0[k] + c.p11 * (x1[k] - x0[k]) + c.p12 * v1[k]);
  if (!M) return;
  double E[9], L[9], Hinv[9], Hc1[9], Hc21[9], T[9], U[9];
  pose2_expmap_derivative(xi, E);
  pose2_logmap_derivative(bt, L);
  pose2_adjoint(p1, Hinv);                  // Inverse: -Ad(p1)
  pose2_adjoint(pose2_inverse(p2), Hc1);    // Compose H1 = Ad(p2^-1)
  pose2_adjoint(pose2_inverse(e), Hc21);    // Compose(p1, e) H1 = Ad(e^-1)
  mat3_mul(E, L, T);                        // Hexp * Hlog
  mat3_mul(T, Hc1, U);
  mat3_mul(U, Hinv, T);                     // Hexp Hlog Hcomp1 Ad(p1)  (sign applied below)
  double EL[9];
  mat3_mul(E, L, EL);
#pragma unroll
  for (int k = 0; k < 9; k++) {
    M[0][k] = Hc21[k] - c.p11 * T[k];
    M[1][k] = c.l12 * E[k];
    M[2][k] = c.p11 * EL[k];
    M[3][k] = c.p12 * E[k];
  }
}

// Values::retract of one state component: Pose2 first-order chart on the first three coordinates
// (gtsam Pose2::ChartAtOrigin::Retract, non-SLOW build) composed on the right, '+' elsewhere
// (gpmp2/geometry/ProductDynamicLieGroup.h:84-90).  z = the state's first three coordinates.
__device__ __forceinline__ double retract_coord(bool lie, int rho, const double* z, const double* dz) {
  if (!lie || rho > 2) return z[rho] + dz[rho];
  double s, c;
  sincos(z[2], &s, &c);
  if (rho == 0) return z[0] + c * dz[0] - s * dz[1];
  if (rho == 1) return z[1] + s * dz[0] + c * dz[1];
  double sd, cd;
  sincos(dz[2], &sd, &cd);
  return wrap_angle(c * cd - s * sd, s * cd + c * sd);
}

// stage the robot model into LDS (every thread of the block must call this)
__device__ __forceinline__ void stage_robot(RobotDev* dst_lds, const RobotDev* __restrict__ src) {
  const int n = sizeof(RobotDev) / 4;
  const int* s = reinterpret_cast<const int*>(src);
  int* d = reinterpret_cast<int*>(dst_lds);
  for (int i = threadIdx.x; i < n; i += blockDim.x) d[i] = s[i];
  __syncthreads();
}

}  // namespace g2
