// oracle_core.cpp -- CPU restatement of the GPMP2 hot path (TEST INFRASTRUCTURE ONLY).
// See oracle_core.h for the role of this code.  Citations are path:line in ori-drs/gpmp2.
#include "oracle_core.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <limits>
#include <stdexcept>

namespace orc {

// =============================================================================== dense helpers
Mat matmul(const Mat& A, const Mat& B) {
  Mat C(A.r, B.c);
  for (int i = 0; i < A.r; i++)
    for (int k = 0; k < A.c; k++) {
      const double aik = A(i, k);
      if (aik == 0.0) continue;
      for (int j = 0; j < B.c; j++) C(i, j) += aik * B(k, j);
    }
  return C;
}
Mat transpose(const Mat& A) {
  Mat T(A.c, A.r);
  for (int i = 0; i < A.r; i++)
    for (int j = 0; j < A.c; j++) T(j, i) = A(i, j);
  return T;
}
Mat operator+(const Mat& A, const Mat& B) {
  Mat C = A;
  for (size_t i = 0; i < C.a.size(); i++) C.a[i] += B.a[i];
  return C;
}
Mat operator-(const Mat& A, const Mat& B) {
  Mat C = A;
  for (size_t i = 0; i < C.a.size(); i++) C.a[i] -= B.a[i];
  return C;
}
Mat operator*(double s, const Mat& A) {
  Mat C = A;
  for (auto& x : C.a) x *= s;
  return C;
}
Mat inverse(const Mat& A) {
  const int n = A.r;
  Mat M = A, I = Mat::identity(n);
  for (int c = 0; c < n; c++) {
    int p = c;
    for (int i = c + 1; i < n; i++)
      if (std::fabs(M(i, c)) > std::fabs(M(p, c))) p = i;
    if (M(p, c) == 0.0) throw std::runtime_error("singular matrix");
    if (p != c)
      for (int j = 0; j < n; j++) {
        std::swap(M(p, j), M(c, j));
        std::swap(I(p, j), I(c, j));
      }
    const double inv = 1.0 / M(c, c);
    for (int j = 0; j < n; j++) {
      M(c, j) *= inv;
      I(c, j) *= inv;
    }
    for (int i = 0; i < n; i++) {
      if (i == c) continue;
      const double f = M(i, c);
      if (f == 0.0) continue;
      for (int j = 0; j < n; j++) {
        M(i, j) -= f * M(c, j);
        I(i, j) -= f * I(c, j);
      }
    }
  }
  return I;
}
Mat chol_upper(const Mat& W) {
  const int n = W.r;
  Mat L(n, n);
  for (int j = 0; j < n; j++) {
    double s = W(j, j);
    for (int k = 0; k < j; k++) s -= L(j, k) * L(j, k);
    if (!(s > 0.0)) throw std::runtime_error("chol_upper: not SPD");
    L(j, j) = std::sqrt(s);
    for (int i = j + 1; i < n; i++) {
      double t = W(i, j);
      for (int k = 0; k < j; k++) t -= L(i, k) * L(j, k);
      L(i, j) = t / L(j, j);
    }
  }
  return transpose(L);
}

// =============================================================================== GP constants
// gpmp2/gp/GPutils.h:25-59
static void set_block(Mat& M, int bi, int bj, int d, double s, const Mat& B) {
  for (int i = 0; i < d; i++)
    for (int j = 0; j < d; j++) M(bi * d + i, bj * d + j) = s * B(i, j);
}
Mat calcQ(const Mat& Qc, double tau) {  // GPutils.h:25-30
  const int d = Qc.r;
  Mat Q(2 * d, 2 * d);
  set_block(Q, 0, 0, d, 1.0 / 3 * std::pow(tau, 3.0), Qc);
  set_block(Q, 0, 1, d, 1.0 / 2 * std::pow(tau, 2.0), Qc);
  set_block(Q, 1, 0, d, 1.0 / 2 * std::pow(tau, 2.0), Qc);
  set_block(Q, 1, 1, d, tau, Qc);
  return Q;
}
Mat calcQ_inv(const Mat& Qc, double tau) {  // GPutils.h:33-39
  const int d = Qc.r;
  const Mat Qi = inverse(Qc);
  Mat Q(2 * d, 2 * d);
  set_block(Q, 0, 0, d, 12.0 * std::pow(tau, -3.0), Qi);
  set_block(Q, 0, 1, d, -6.0 * std::pow(tau, -2.0), Qi);
  set_block(Q, 1, 0, d, -6.0 * std::pow(tau, -2.0), Qi);
  set_block(Q, 1, 1, d, 4.0 * std::pow(tau, -1.0), Qi);
  return Q;
}
Mat calcPhi(int d, double tau) {  // GPutils.h:42-46
  Mat P = Mat::identity(2 * d);
  for (int i = 0; i < d; i++) P(i, d + i) = tau;
  return P;
}
Mat calcPsi(const Mat& Qc, double delta_t, double tau) {  // GPutils.h:56-59
  return matmul(matmul(calcQ(Qc, tau), transpose(calcPhi(Qc.r, delta_t - tau))),
                calcQ_inv(Qc, delta_t));
}
Mat calcLambda(const Mat& Qc, double delta_t, double tau) {  // GPutils.h:49-53
  return calcPhi(Qc.r, tau) - matmul(calcPsi(Qc, delta_t, tau), calcPhi(Qc.r, delta_t));
}

// =============================================================================== Pose2 (GTSAM)
// Upstream GTSAM semantics (gtsam/geometry/Pose2.cpp) -- un-vendored.
static inline double wrap_theta(double c, double s) { return std::atan2(s, c); }
Pose2 pose2_compose(const Pose2& a, const Pose2& b) {
  const double c = std::cos(a.th), s = std::sin(a.th);
  Pose2 r;
  r.x = a.x + c * b.x - s * b.y;
  r.y = a.y + s * b.x + c * b.y;
  // Rot2 product keeps (c,s); theta() = atan2
  const double cb = std::cos(b.th), sb = std::sin(b.th);
  r.th = wrap_theta(c * cb - s * sb, s * cb + c * sb);
  return r;
}
Pose2 pose2_inverse(const Pose2& a) {
  const double c = std::cos(a.th), s = std::sin(a.th);
  Pose2 r;
  r.x = -(c * a.x + s * a.y);
  r.y = -(-s * a.x + c * a.y);
  r.th = wrap_theta(c, -s);
  return r;
}
Pose2 pose2_between(const Pose2& a, const Pose2& b) { return pose2_compose(pose2_inverse(a), b); }
void pose2_logmap(const Pose2& p, double v[3]) {
  const double c = std::cos(p.th), s = std::sin(p.th);
  const double w = wrap_theta(c, s);
  if (std::fabs(w) < 1e-10) {
    v[0] = p.x;
    v[1] = p.y;
    v[2] = w;
  } else {
    const double c_1 = c - 1.0, det = c_1 * c_1 + s * s;
    // R.unrotate(t) - t, then rotate by +90 deg
    const double ux = c * p.x + s * p.y - p.x, uy = -s * p.x + c * p.y - p.y;
    const double px = -uy, py = ux;
    v[0] = (w / det) * px;
    v[1] = (w / det) * py;
    v[2] = w;
  }
}
Pose2 pose2_expmap(const double v[3]) {
  const double w = v[2];
  Pose2 r;
  if (std::fabs(w) < 1e-10) {
    r.x = v[0];
    r.y = v[1];
    r.th = v[2];
  } else {
    const double c = std::cos(w), s = std::sin(w);
    const double ox = -v[1], oy = v[0];  // v_ortho = R_PI_2 * v
    const double rx = c * ox - s * oy, ry = s * ox + c * oy;
    r.x = (ox - rx) / w;
    r.y = (oy - ry) / w;
    r.th = wrap_theta(c, s);
  }
  return r;
}
Pose2 pose2_retract(const Pose2& p, const double v[3]) {
  Pose2 d;
  d.x = v[0];
  d.y = v[1];
  d.th = v[2];
  return pose2_compose(p, d);
}
void pose2_adjoint(const Pose2& p, double A[9]) {
  const double c = std::cos(p.th), s = std::sin(p.th);
  const double M[9] = {c, -s, p.y, s, c, -p.x, 0, 0, 1};
  std::memcpy(A, M, sizeof(M));
}
void pose2_expmap_derivative(const double v[3], double J[9]) {
  const double alpha = v[2];
  if (std::fabs(alpha) > 1e-5) {
    const double sZ = std::sin(alpha) / alpha, c1Z = (std::cos(alpha) - 1) / alpha;
    const double v1Z = v[0] / alpha, v2Z = v[1] / alpha;
    const double M[9] = {sZ, -c1Z, v1Z + v2Z * c1Z - v1Z * sZ, c1Z, sZ,
                         -v1Z * c1Z + v2Z - v2Z * sZ, 0, 0, 1};
    std::memcpy(J, M, sizeof(M));
  } else {
    const double M[9] = {1, 0, -0.5 * v[1], 0, 1, 0.5 * v[0], 0, 0, 1};
    std::memcpy(J, M, sizeof(M));
  }
}
void pose2_logmap_derivative(const Pose2& p, double J[9]) {
  double v[3];
  pose2_logmap(p, v);
  const double alpha = v[2];
  if (std::fabs(alpha) > 1e-5) {
    const double ai = 1 / alpha, hc = 0.5 * std::sin(alpha) / (1 - std::cos(alpha));
    const double v1 = v[0], v2 = v[1];
    const double M[9] = {alpha * hc, -0.5 * alpha, v1 * ai - v1 * hc + 0.5 * v2,
                         0.5 * alpha, alpha * hc, v2 * ai - 0.5 * v1 - v2 * hc, 0, 0, 1};
    std::memcpy(J, M, sizeof(M));
  } else {
    const double M[9] = {1, 0, 0.5 * v[1], 0, 1, -0.5 * v[0], 0, 0, 1};
    std::memcpy(J, M, sizeof(M));
  }
}

// =============================================================================== SO(3) / SE(3) logs
static void skew(const double w[3], double W[9]) {
  W[0] = 0; W[1] = -w[2]; W[2] = w[1];
  W[3] = w[2]; W[4] = 0; W[5] = -w[0];
  W[6] = -w[1]; W[7] = w[0]; W[8] = 0;
}
static void m3mul(const double* A, const double* B, double* C) {
  double T[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) T[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
  std::memcpy(C, T, sizeof(T));
}
// gtsam SO3::Logmap (4.0): trace based, special branch near pi, series near 0
void rot3_logmap(const double R[9], double w[3]) {
  const double R11 = R[0], R12 = R[1], R13 = R[2], R21 = R[3], R22 = R[4], R23 = R[5], R31 = R[6], R32 = R[7], R33 = R[8];
  const double tr = R11 + R22 + R33;
  if (std::fabs(tr + 1.0) < 1e-10) {
    if (std::fabs(R33 + 1.0) > 1e-10) {
      const double k = M_PI / std::sqrt(2.0 + 2.0 * R33);
      w[0] = k * R13; w[1] = k * R23; w[2] = k * (1.0 + R33);
    } else if (std::fabs(R22 + 1.0) > 1e-10) {
      const double k = M_PI / std::sqrt(2.0 + 2.0 * R22);
      w[0] = k * R12; w[1] = k * (1.0 + R22); w[2] = k * R32;
    } else {
      const double k = M_PI / std::sqrt(2.0 + 2.0 * R11);
      w[0] = k * (1.0 + R11); w[1] = k * R21; w[2] = k * R31;
    }
  } else {
    double magnitude;
    const double tr_3 = tr - 3.0;
    if (tr_3 < -1e-7) {
      const double theta = std::acos((tr - 1.0) / 2.0);
      magnitude = theta / (2.0 * std::sin(theta));
    } else {
      magnitude = 0.5 - tr_3 * tr_3 / 12.0;
    }
    w[0] = magnitude * (R32 - R23);
    w[1] = magnitude * (R13 - R31);
    w[2] = magnitude * (R21 - R12);
  }
}
// gtsam SO3::LogmapDerivative
void rot3_logmap_derivative(const double w[3], double H[9]) {
  const double theta2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  for (int i = 0; i < 9; i++) H[i] = (i % 4 == 0) ? 1.0 : 0.0;
  if (theta2 <= std::numeric_limits<double>::epsilon()) return;
  const double theta = std::sqrt(theta2);
  double W[9], WW[9];
  skew(w, W);
  m3mul(W, W, WW);
  const double k = 1.0 / (theta * theta) - (1.0 + std::cos(theta)) / (2.0 * theta * std::sin(theta));
  for (int i = 0; i < 9; i++) H[i] += 0.5 * W[i] + k * WW[i];
}
// gtsam Pose3::Logmap: xi = [omega; u]
void pose3_logmap(const double R[9], const double T[3], double xi[6]) {
  double w[3];
  rot3_logmap(R, w);
  const double t = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  xi[0] = w[0]; xi[1] = w[1]; xi[2] = w[2];
  if (t < 1e-10) {
    xi[3] = T[0]; xi[4] = T[1]; xi[5] = T[2];
    return;
  }
  const double wn[3] = {w[0] / t, w[1] / t, w[2] / t};
  double W[9];
  skew(wn, W);
  const double Tan = std::tan(0.5 * t);
  double WT[3], WWT[3];
  for (int i = 0; i < 3; i++) WT[i] = W[i * 3] * T[0] + W[i * 3 + 1] * T[1] + W[i * 3 + 2] * T[2];
  for (int i = 0; i < 3; i++) WWT[i] = W[i * 3] * WT[0] + W[i * 3 + 1] * WT[1] + W[i * 3 + 2] * WT[2];
  for (int i = 0; i < 3; i++) xi[3 + i] = T[i] - (0.5 * t) * WT[i] + (1.0 - t / (2.0 * Tan)) * WWT[i];
}
// gtsam Pose3::computeQforExpmapDerivative (Barfoot eq. 102 with the sign convention of gtsam)
static void pose3_Q(const double xi[6], double Q[9]) {
  const double* w = xi;
  const double* v = xi + 3;
  double V[9], W[9];
  skew(v, V);
  skew(w, W);
  double WV[9], VW[9], WVW[9], WWV[9], VWW[9], WVWW[9], WWVW[9], WW[9];
  m3mul(W, V, WV);
  m3mul(V, W, VW);
  m3mul(WV, W, WVW);
  m3mul(W, W, WW);
  m3mul(WW, V, WWV);
  m3mul(VW, W, VWW);
  m3mul(WVW, W, WVWW);
  m3mul(W, WVW, WWVW);
  const double phi = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  double c1, c2, c3;
  if (std::fabs(phi) > 1e-5) {
    const double s = std::sin(phi), c = std::cos(phi);
    const double phi2 = phi * phi, phi3 = phi2 * phi, phi4 = phi3 * phi, phi5 = phi4 * phi;
    c1 = (phi - s) / phi3;
    c2 = (1.0 - phi2 / 2.0 - c) / phi4;
    c3 = -0.5 * ((1.0 - phi2 / 2.0 - c) / phi4 - 3.0 * (phi - s - phi3 / 6.0) / phi5);
  } else {
    c1 = 1.0 / 6.0;
    c2 = 1.0 / 24.0;
    c3 = -0.5 * (1.0 / 24.0 + 3.0 / 120.0);
  }
  for (int i = 0; i < 9; i++)
    Q[i] = -0.5 * V[i] + c1 * (WV[i] + VW[i] - WVW[i]) + c2 * (WWV[i] + VWW[i] - 3.0 * WVW[i]) + c3 * (WVWW[i] + WWVW[i]);
}
// gtsam Pose3::LogmapDerivative: [Jw 0; -Jw Q Jw, Jw]
void pose3_logmap_derivative(const double R[9], const double T[3], double H[36]) {
  double xi[6], Jw[9], Q[9], Q2[9];
  pose3_logmap(R, T, xi);
  rot3_logmap_derivative(xi, Jw);
  pose3_Q(xi, Q);
  m3mul(Jw, Q, Q2);
  m3mul(Q2, Jw, Q2);
  for (int i = 0; i < 36; i++) H[i] = 0.0;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      H[i * 6 + j] = Jw[i * 3 + j];
      H[(3 + i) * 6 + 3 + j] = Jw[i * 3 + j];
      H[(3 + i) * 6 + j] = -Q2[i * 3 + j];
    }
}

void workspace_prior_factor(const Robot& R, int mode, int joint, const double des[16], const double* conf,
                            double* err, double* H) {
  const int D = R.dof, L = R.nr_links();
  std::vector<double> poses(16 * L), Jp((size_t)L * 6 * D);
  forward_kinematics(R, conf, poses.data(), Jp.data());
  const double* T = &poses[16 * joint];
  const double* J6 = &Jp[(size_t)joint * 6 * D];
  double Rm[9], t[3], Rd[9], td[3];
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) {
      Rm[i * 3 + j] = T[i * 4 + j];
      Rd[i * 3 + j] = des[i * 4 + j];
    }
    t[i] = T[i * 4 + 3];
    td[i] = des[i * 4 + 3];
  }
  if (mode == WS_POSITION) {
    // Pose3::translation(H) = [0 R]
    for (int i = 0; i < 3; i++) err[i] = t[i] - td[i];
    if (H)
      for (int i = 0; i < 3; i++)
        for (int k = 0; k < D; k++) {
          double a = 0;
          for (int m = 0; m < 3; m++) a += Rm[i * 3 + m] * J6[(3 + m) * D + k];
          H[i * D + k] = a;
        }
    return;
  }
  // between(des, pose): R_rel = Rd^T R, t_rel = Rd^T (t - td)
  double Rrel[9], trel[3];
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) {
      double a = 0;
      for (int m = 0; m < 3; m++) a += Rd[m * 3 + i] * Rm[m * 3 + j];
      Rrel[i * 3 + j] = a;
    }
    trel[i] = Rd[0 * 3 + i] * (t[0] - td[0]) + Rd[1 * 3 + i] * (t[1] - td[1]) + Rd[2 * 3 + i] * (t[2] - td[2]);
  }
  if (mode == WS_ORIENTATION) {
    rot3_logmap(Rrel, err);
    if (H) {
      double Her[9];
      rot3_logmap_derivative(err, Her);  // Pose3::rotation(H) = [I 0]
      for (int i = 0; i < 3; i++)
        for (int k = 0; k < D; k++) {
          double a = 0;
          for (int m = 0; m < 3; m++) a += Her[i * 3 + m] * J6[m * D + k];
          H[i * D + k] = a;
        }
    }
    return;
  }
  pose3_logmap(Rrel, trel, err);
  if (H) {
    double Hep[36];
    pose3_logmap_derivative(Rrel, trel, Hep);
    for (int i = 0; i < 6; i++)
      for (int k = 0; k < D; k++) {
        double a = 0;
        for (int m = 0; m < 6; m++) a += Hep[i * 6 + m] * J6[m * D + k];
        H[i * D + k] = a;
      }
  }
}

void self_collision_factor(const Robot& R, int n_pairs, const double* data, const double* conf, double* err,
                           double* H) {
  const int D = R.dof, S = R.nr_spheres();
  std::vector<double> c(3 * S), J(H ? (size_t)S * 3 * D : 0);
  sphere_centers(R, conf, c.data(), H ? J.data() : nullptr);
  for (int i = 0; i < n_pairs; i++) {
    const int a = (int)data[i * 4 + 0], b = (int)data[i * 4 + 1];
    const double eps = R.sph_r[a] + R.sph_r[b] + data[i * 4 + 2];
    const double dx = c[3 * a] - c[3 * b], dy = c[3 * a + 1] - c[3 * b + 1], dz = c[3 * a + 2] - c[3 * b + 2];
    const double dist = std::sqrt(dx * dx + dy * dy + dz * dz);  // gtsam::distance3: H_A = d^T/|d|, H_B = -H_A
    if (H)
      for (int k = 0; k < D; k++) H[(size_t)i * D + k] = 0.0;
    if (dist > eps) {
      err[i] = 0.0;
      continue;
    }
    err[i] = eps - dist;
    if (H) {
      const double n[3] = {dx / dist, dy / dist, dz / dist};
      for (int k = 0; k < D; k++) {
        double v = 0;
        for (int m = 0; m < 3; m++) v += -n[m] * J[((size_t)a * 3 + m) * D + k] + n[m] * J[((size_t)b * 3 + m) * D + k];
        H[(size_t)i * D + k] = v;
      }
    }
  }
}

// =============================================================================== SDF construction
// matlab/+gpmp2/signedDistanceField3D.m:16-34 (bwdist) / gpmp2_python/utils/signedDistanceField3D.py:22-42
// (scipy.ndimage.distance_transform_edt): exact Euclidean distance to the nearest cell of the
// other class.  Restated as the textbook separable minimisation of integer squared distances
//   d2(p) = min_q |p - q|^2  =  min_z' ( min_y' ( min_x' [q in set] ) + (y-y')^2 ) + (z-z')^2
// evaluated by plain loops (test sizes only).
static void edt_axis(std::vector<long long>& v, size_t outer, size_t len, size_t inner) {
  std::vector<long long> line(len);
  for (size_t o = 0; o < outer; o++)
    for (size_t in = 0; in < inner; in++) {
      long long* g = v.data() + o * len * inner + in;
      for (size_t j = 0; j < len; j++) line[j] = g[j * inner];
      for (size_t i = 0; i < len; i++) {
        long long best = line[i];
        for (size_t j = 0; j < len; j++) {
          const long long dd = (long long)i - (long long)j;
          best = std::min(best, line[j] + dd * dd);
        }
        g[i * inner] = best;
      }
    }
}
void sdf_from_occupancy(int nx, int ny, int nz, const double* occ, double cell, double* field) {
  const size_t n = (size_t)nx * ny * nz;
  const long long INF = 1ll << 40;
  std::vector<long long> a(n), b(n);
  bool any_obst = false, any_free = false;
  for (size_t i = 0; i < n; i++) {
    const bool obst = occ[i] > 0.75;
    a[i] = obst ? 0 : INF;
    b[i] = obst ? INF : 0;
    any_obst |= obst;
    any_free |= !obst;
  }
  if (!any_obst || !any_free) {  // bwdist gives Inf -> "limit inf" branch, signedDistanceField3D.m:30-33
    for (size_t i = 0; i < n; i++) field[i] = 1000.0;
    return;
  }
  for (auto* v : {&a, &b}) {
    edt_axis(*v, (size_t)ny * nz, nx, 1);
    edt_axis(*v, nz, ny, nx);
    edt_axis(*v, 1, nz, (size_t)nx * ny);
  }
  for (size_t i = 0; i < n; i++) {
    const double map_dist = std::sqrt((double)a[i]), inv_map_dist = std::sqrt((double)b[i]);
    field[i] = (map_dist - inv_map_dist) * cell;
  }
}

// =============================================================================== 4x4 helpers
static void m4mul(const double* A, const double* B, double* C) {
  double T[16];
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      double s = 0;
      for (int k = 0; k < 4; k++) s += A[i * 4 + k] * B[k * 4 + j];
      T[i * 4 + j] = s;
    }
  std::memcpy(C, T, sizeof(T));
}
static void m4ident(double* A) {
  std::memset(A, 0, 16 * sizeof(double));
  A[0] = A[5] = A[10] = A[15] = 1.0;
}
static void m4inv_rigid(const double* A, double* B) {
  // inverse of [R t; 0 1] (the reference uses Eigen's general 4x4 inverse, Arm.cpp:59,80)
  double T[16];
  m4ident(T);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) T[i * 4 + j] = A[j * 4 + i];
  for (int i = 0; i < 3; i++)
    T[i * 4 + 3] = -(T[i * 4 + 0] * A[3] + T[i * 4 + 1] * A[7] + T[i * 4 + 2] * A[11]);
  std::memcpy(B, T, sizeof(T));
}

// link_trans_notheta_[i] = Trans(0,0,d) * Trans(a,0,0) * Rx(alpha)   kinematics/Arm.cpp:23-27
static void dh_const(double a, double alpha, double d, double* C) {
  const double ca = std::cos(alpha), sa = std::sin(alpha);
  const double M[16] = {1, 0, 0, a, 0, ca, -sa, 0, 0, sa, ca, d, 0, 0, 0, 1};
  std::memcpy(C, M, sizeof(M));
}

// Arm::forwardKinematics, pose part   kinematics/Arm.cpp:31-143 (jv == none)
// poses [dof][16]; Jp [dof][6][ldJ] written into columns [col0, col0+dof)
// joints [j0, j0 + n) of the DH table (n < 0: all of them)
static void arm_fk(const Robot& R, const double* base, const double* q, double* poses, double* Jp,
                   int ldJ, int col0, int j0 = 0, int n = -1) {
  if (n < 0) n = R.arm_dof;
  std::vector<double> H(16 * n), dH(16 * n), Ho(16 * (n + 1)), Hoinv(16 * (n + 1));
  std::memcpy(&Ho[0], base, 16 * sizeof(double));
  m4inv_rigid(&Ho[0], &Hoinv[0]);
  for (int i = 1; i <= n; i++) {
    double C[16];
    dh_const(R.a[j0 + i - 1], R.alpha[j0 + i - 1], R.d[j0 + i - 1], C);
    const double th = q[i - 1] + (R.bias.empty() ? 0.0 : R.bias[j0 + i - 1]);
    const double c = std::cos(th), s = std::sin(th);
    const double Rz[16] = {c, -s, 0, 0, s, c, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    m4mul(Rz, C, &H[16 * (i - 1)]);                       // getH        Arm.h:101-103
    m4mul(&Ho[16 * (i - 1)], &H[16 * (i - 1)], &Ho[16 * i]);  // Arm.cpp:67
    if (Jp) {
      const double dR[16] = {-s, -c, 0, 0, c, -s, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      m4mul(dR, C, &dH[16 * (i - 1)]);                    // getdH       Arm.h:106-114
      m4inv_rigid(&Ho[16 * i], &Hoinv[16 * i]);
    }
  }
  for (int i = 0; i < n; i++) {
    std::memcpy(poses + 16 * i, &Ho[16 * (i + 1)], 16 * sizeof(double));
    if (!Jp) continue;
    for (int j = 0; j <= i; j++) {
      double dHo[16], T[16];
      m4mul(&Ho[16 * j], &dH[16 * j], dHo);               // Arm.cpp:85-92
      if (i > j) {
        m4mul(dHo, &Hoinv[16 * (j + 1)], T);
        m4mul(T, &Ho[16 * (i + 1)], dHo);
      }
      double S[16];
      m4mul(&Hoinv[16 * (i + 1)], dHo, S);                // Arm.cpp:105-115
      double* J = Jp + (size_t)i * 6 * ldJ;
      J[0 * ldJ + col0 + j] = S[2 * 4 + 1];
      J[1 * ldJ + col0 + j] = S[0 * 4 + 2];
      J[2 * ldJ + col0 + j] = S[1 * 4 + 0];
      J[3 * ldJ + col0 + j] = S[0 * 4 + 3];
      J[4 * ldJ + col0 + j] = S[1 * 4 + 3];
      J[5 * ldJ + col0 + j] = S[2 * 4 + 3];
    }
  }
}

// gtsam::Pose3::AdjointMap, tangent order [omega; v]
static void pose3_adjoint(const double* T, double* Ad /*6x6*/) {
  std::memset(Ad, 0, 36 * sizeof(double));
  const double t[3] = {T[3], T[7], T[11]};
  const double Sk[9] = {0, -t[2], t[1], t[2], 0, -t[0], -t[1], t[0], 0};
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      Ad[i * 6 + j] = T[i * 4 + j];
      Ad[(i + 3) * 6 + (j + 3)] = T[i * 4 + j];
      double s = 0;
      for (int k = 0; k < 3; k++) s += Sk[i * 3 + k] * T[k * 4 + j];
      Ad[(i + 3) * 6 + j] = s;
    }
}

// computeBasePose3   kinematics/mobileBaseUtils.cpp:18-31
static void base_pose3(const double* p2, double* T, double* J /*6x3 or null*/) {
  const double c = std::cos(p2[2]), s = std::sin(p2[2]);
  const double M[16] = {c, -s, 0, p2[0], s, c, 0, p2[1], 0, 0, 1, 0, 0, 0, 0, 1};
  std::memcpy(T, M, sizeof(M));
  if (J) {
    std::memset(J, 0, 18 * sizeof(double));
    J[2 * 3 + 2] = 1.0;  // Rot3::ExpmapDerivative((0,0,theta)).col(2) = e_z
    J[3 * 3 + 0] = 1.0;
    J[4 * 3 + 1] = 1.0;
  }
}

// H_out (6 x nb) = Ad(T^-1) * H_in : Jacobian of parent.compose(T) w.r.t. the parent's parameters
static void compose_jac(const double* T, const double* Hin, int nb, double* Hout) {
  double inv[16], Ad[36];
  m4inv_rigid(T, inv);
  pose3_adjoint(inv, Ad);
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < nb; j++) {
      double s = 0;
      for (int k = 0; k < 6; k++) s += Ad[i * 6 + k] * Hin[k * nb + j];
      Hout[i * nb + j] = s;
    }
}

// vehicle base [+ vertical lift torso] + one or two arms
static void tree_fk(const Robot& R, const double* conf, double* poses, double* Jpose) {
  const int D = R.dof;
  const bool lift = R.has_lift();
  const int nb = lift ? 4 : 3;  // parameters the arm bases depend on
  const int A2 = (R.kind == MOBILE_VETLIN_ARM) ? 0 : R.arm2_dof, A1 = R.arm_dof - A2;
  double veh[16], Hveh[18];
  base_pose3(conf, veh, Hveh);
  std::memcpy(poses, veh, sizeof(veh));
  if (Jpose)
    for (int i = 0; i < 6; i++)
      for (int j = 0; j < 3; j++) Jpose[i * D + j] = Hveh[i * 3 + j];
  double parent[16], Hparent[24];
  int first_arm_link = 1;
  if (lift) {
    // liftBasePose3  kinematics/mobileBaseUtils.cpp:51-82: lift_pose.compose(veh.compose(base_T_torso))
    double armbase[16], Harmbase[18];
    m4mul(veh, R.base, armbase);
    compose_jac(R.base, Hveh, 3, Harmbase);
    const double lz = R.reverse_linact ? -conf[3] : conf[3];
    std::memcpy(parent, armbase, sizeof(armbase));
    parent[11] += lz;  // pure world-z translation on the left
    // Hcomp2 = I ; Hcomp1 = Ad(armbase^-1), its column 5 = d/d(lift z)
    double inv[16], Ad[36];
    m4inv_rigid(armbase, inv);
    pose3_adjoint(inv, Ad);
    for (int i = 0; i < 6; i++) {
      for (int j = 0; j < 3; j++) Hparent[i * 4 + j] = Harmbase[i * 3 + j];
      Hparent[i * 4 + 3] = R.reverse_linact ? -Ad[i * 6 + 5] : Ad[i * 6 + 5];
    }
    std::memcpy(poses + 16, parent, sizeof(parent));
    if (Jpose)
      for (int i = 0; i < 6; i++)
        for (int j = 0; j < 4; j++) Jpose[(size_t)6 * D + i * D + j] = Hparent[i * 4 + j];
    first_arm_link = 2;
  } else {
    std::memcpy(parent, veh, sizeof(veh));
    for (int i = 0; i < 18; i++) Hparent[i] = Hveh[i];
  }
  const double* T_arm[2] = {lift ? R.base2 : R.base, lift ? R.base3 : R.base2};
  const int na[2] = {A1, A2}, j0[2] = {0, A1};
  int link = first_arm_link;
  for (int a = 0; a < 2; a++) {
    if (na[a] == 0) continue;
    double armb[16], Harm[24];
    m4mul(parent, T_arm[a], armb);
    compose_jac(T_arm[a], Hparent, nb, Harm);
    arm_fk(R, armb, conf + nb + j0[a], poses + 16 * link, Jpose ? Jpose + (size_t)link * 6 * D : nullptr, D,
           nb + j0[a], j0[a], na[a]);
    if (Jpose)
      for (int l = 0; l < na[a]; l++) {  // "see compose's jacobian": Ad(link^-1 * arm_base) * Harm_base
        double inv[16], T[16], Ad[36];
        m4inv_rigid(poses + 16 * (link + l), inv);
        m4mul(inv, armb, T);
        pose3_adjoint(T, Ad);
        double* J = Jpose + (size_t)(link + l) * 6 * D;
        for (int i = 0; i < 6; i++)
          for (int j = 0; j < nb; j++) {
            double s = 0;
            for (int k = 0; k < 6; k++) s += Ad[i * 6 + k] * Harm[k * nb + j];
            J[i * D + j] = s;
          }
      }
    link += na[a];
  }
}

void forward_kinematics(const Robot& R, const double* conf, double* poses, double* Jpose) {
  const int D = R.dof, L = R.nr_links();
  if (Jpose) std::memset(Jpose, 0, sizeof(double) * (size_t)L * 6 * D);
  switch (R.kind) {
    case ARM:
      arm_fk(R, R.base, conf, poses, Jpose, D, 0);
      break;
    case POINT: {  // kinematics/PointRobot.cpp:15-49
      m4ident(poses);
      poses[3] = conf[0];
      poses[7] = conf[1];
      if (Jpose) {
        Jpose[3 * D + 0] = 1.0;  // Pose3::Create H2 = [0; R] with R = I
        Jpose[4 * D + 1] = 1.0;
      }
    } break;
    case MOBILE_BASE: {  // kinematics/Pose2MobileBase.cpp:20-55
      double J[18];
      base_pose3(conf, poses, Jpose ? J : nullptr);
      if (Jpose)
        for (int i = 0; i < 6; i++)
          for (int j = 0; j < 3; j++) Jpose[i * D + j] = J[i * 3 + j];
    } break;
    case MOBILE_ARM: {  // kinematics/Pose2MobileArm.cpp:30-108
      double Hveh[18], veh[16], armb[16];
      base_pose3(conf, veh, Hveh);
      m4mul(veh, R.base, armb);  // computeBaseTransPose3  mobileBaseUtils.cpp:34-48
      double Harm[18];
      {
        double inv[16], Ad[36];
        m4inv_rigid(R.base, inv);
        pose3_adjoint(inv, Ad);  // compose Jacobian wrt first = Ad(base_T_arm^-1)
        for (int i = 0; i < 6; i++)
          for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 6; k++) s += Ad[i * 6 + k] * Hveh[k * 3 + j];
            Harm[i * 3 + j] = s;
          }
      }
      std::memcpy(poses, veh, sizeof(veh));
      if (Jpose)
        for (int i = 0; i < 6; i++)
          for (int j = 0; j < 3; j++) Jpose[i * D + j] = Hveh[i * 3 + j];
      arm_fk(R, armb, conf + 3, poses + 16, Jpose ? Jpose + 6 * D : nullptr, D, 3);
      if (Jpose)
        for (int l = 0; l < R.arm_dof; l++) {  // Pose2MobileArm.cpp:97-102
          double inv[16], T[16], Ad[36];
          m4inv_rigid(poses + 16 * (l + 1), inv);
          m4mul(inv, armb, T);
          pose3_adjoint(T, Ad);
          double* J = Jpose + (size_t)(l + 1) * 6 * D;
          for (int i = 0; i < 6; i++)
            for (int j = 0; j < 3; j++) {
              double s = 0;
              for (int k = 0; k < 6; k++) s += Ad[i * 6 + k] * Harm[k * 3 + j];
              J[i * D + j] = s;
            }
        }
    } break;
    case MOBILE_2ARMS:         // kinematics/Pose2Mobile2Arms.cpp:32-108
    case MOBILE_VETLIN_ARM:    // kinematics/Pose2MobileVetLinArm.cpp:31-108
    case MOBILE_VETLIN_2ARMS:  // kinematics/Pose2MobileVetLin2Arms.cpp:36-114
      tree_fk(R, conf, poses, Jpose);
      break;
  }
}

// RobotModel<FK>::sphereCenters   kinematics/RobotModel-inl.h:12-40
void sphere_centers(const Robot& R, const double* conf, double* centers, double* J) {
  const int D = R.dof, L = R.nr_links(), S = R.nr_spheres();
  std::vector<double> poses(16 * L), Jp(J ? (size_t)L * 6 * D : 0);
  forward_kinematics(R, conf, poses.data(), J ? Jp.data() : nullptr);
  for (int s = 0; s < S; s++) {
    const double* T = &poses[16 * R.sph_link[s]];
    const double* c = &R.sph_c[3 * s];
    for (int i = 0; i < 3; i++)
      centers[3 * s + i] = T[i * 4 + 0] * c[0] + T[i * 4 + 1] * c[1] + T[i * 4 + 2] * c[2] +
                           T[i * 4 + 3];
    if (!J) continue;
    // Pose3::transform_from Dpose = R * [-[c]x , I]   
    double Dp[18];
    const double Sk[9] = {0, -c[2], c[1], c[2], 0, -c[0], -c[1], c[0], 0};
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        double a = 0;
        for (int k = 0; k < 3; k++) a += T[i * 4 + k] * Sk[k * 3 + j];
        Dp[i * 6 + j] = -a;
        Dp[i * 6 + 3 + j] = T[i * 4 + j];
      }
    const double* Jl = &Jp[(size_t)R.sph_link[s] * 6 * D];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < D; j++) {
        double a = 0;
        for (int k = 0; k < 6; k++) a += Dp[i * 6 + k] * Jl[k * D + j];
        J[((size_t)s * 3 + i) * D + j] = a;
      }
  }
}

// =============================================================================== SDF + hinge
// SignedDistanceField::getSignedDistance  obstacle/SignedDistanceField.h:93-167
// PlanarSDF::getSignedDistance            obstacle/PlanarSDF.h:61-116
bool sdf_query(const Sdf& s, const double* p, double* dist, double* grad) {
  if (s.dim == 3) {
    if (p[0] < s.origin[0] || p[0] > (s.origin[0] + (s.nx - 1.0) * s.cell) ||
        p[1] < s.origin[1] || p[1] > (s.origin[1] + (s.ny - 1.0) * s.cell) ||
        p[2] < s.origin[2] || p[2] > (s.origin[2] + (s.nz - 1.0) * s.cell))
      return false;
    const double col = (p[0] - s.origin[0]) / s.cell, row = (p[1] - s.origin[1]) / s.cell,
                 z = (p[2] - s.origin[2]) / s.cell;
    const double lr = std::floor(row), lc = std::floor(col), lz = std::floor(z);
    const double hr = lr + 1.0, hc = lc + 1.0, hz = lz + 1.0;
    const int lri = (int)lr, lci = (int)lc, lzi = (int)lz;
    // quirk A.4: at the upper face hri == n indexes one past the end with weight 0; clamp the
    // index, keep the weight.
    const int hri = std::min((int)hr, s.ny - 1), hci = std::min((int)hc, s.nx - 1),
              hzi = std::min((int)hz, s.nz - 1);
    const double v000 = s.at(lri, lci, lzi), v100 = s.at(hri, lci, lzi), v010 = s.at(lri, hci, lzi),
                 v110 = s.at(hri, hci, lzi), v001 = s.at(lri, lci, hzi), v101 = s.at(hri, lci, hzi),
                 v011 = s.at(lri, hci, hzi), v111 = s.at(hri, hci, hzi);
    *dist = (hr - row) * (hc - col) * (hz - z) * v000 + (row - lr) * (hc - col) * (hz - z) * v100 +
            (hr - row) * (col - lc) * (hz - z) * v010 + (row - lr) * (col - lc) * (hz - z) * v110 +
            (hr - row) * (hc - col) * (z - lz) * v001 + (row - lr) * (hc - col) * (z - lz) * v101 +
            (hr - row) * (col - lc) * (z - lz) * v011 + (row - lr) * (col - lc) * (z - lz) * v111;
    if (grad) {
      const double g_row = (hc - col) * (hz - z) * (v100 - v000) + (col - lc) * (hz - z) * (v110 - v010) +
                           (hc - col) * (z - lz) * (v101 - v001) + (col - lc) * (z - lz) * (v111 - v011);
      const double g_col = (hr - row) * (hz - z) * (v010 - v000) + (row - lr) * (hz - z) * (v110 - v100) +
                           (hr - row) * (z - lz) * (v011 - v001) + (row - lr) * (z - lz) * (v111 - v101);
      const double g_z = (hr - row) * (hc - col) * (v001 - v000) + (row - lr) * (hc - col) * (v101 - v100) +
                         (hr - row) * (col - lc) * (v011 - v010) + (row - lr) * (col - lc) * (v111 - v110);
      grad[0] = g_col / s.cell;  // SignedDistanceField.h:97
      grad[1] = g_row / s.cell;
      grad[2] = g_z / s.cell;
    }
    return true;
  }
  // planar
  if (p[0] < s.origin[0] || p[0] > (s.origin[0] + (s.nx - 1.0) * s.cell) || p[1] < s.origin[1] ||
      p[1] > (s.origin[1] + (s.ny - 1.0) * s.cell))
    return false;
  const double col = (p[0] - s.origin[0]) / s.cell, row = (p[1] - s.origin[1]) / s.cell;
  const double lr = std::floor(row), lc = std::floor(col), hr = lr + 1.0, hc = lc + 1.0;
  const int lri = (int)lr, lci = (int)lc;
  const int hri = std::min((int)hr, s.ny - 1), hci = std::min((int)hc, s.nx - 1);
  const double v00 = s.at(lri, lci, 0), v10 = s.at(hri, lci, 0), v01 = s.at(lri, hci, 0),
               v11 = s.at(hri, hci, 0);
  *dist = (hr - row) * (hc - col) * v00 + (row - lr) * (hc - col) * v10 +
          (hr - row) * (col - lc) * v01 + (row - lr) * (col - lc) * v11;
  if (grad) {
    const double g_row = (hc - col) * (v10 - v00) + (col - lc) * (v11 - v01);
    const double g_col = (hr - row) * (v01 - v00) + (row - lr) * (v11 - v10);
    grad[0] = g_col / s.cell;  // PlanarSDF.h:66
    grad[1] = g_row / s.cell;
  }
  return true;
}

// hingeLossObstacleCost   obstacle/ObstacleCost.h:26-50 (3-D), :54-78 (2-D)
double hinge_obstacle(const Sdf& s, const double* p, double eps, double* Hp) {
  double d, g[3] = {0, 0, 0};
  if (!sdf_query(s, p, &d, g)) {
    if (Hp)
      for (int i = 0; i < s.dim; i++) Hp[i] = 0.0;
    return 0.0;
  }
  if (d > eps) {
    if (Hp)
      for (int i = 0; i < s.dim; i++) Hp[i] = 0.0;
    return 0.0;
  }
  if (Hp)
    for (int i = 0; i < s.dim; i++) Hp[i] = -g[i];
  return eps - d;
}

// hingeLossJointLimitCost   kinematics/JointLimitCost.h:16-31
double hinge_limit(double p, double lo, double hi, double th, double* H) {
  if (p < lo + th) {
    if (H) *H = -1.0;
    return lo + th - p;
  } else if (p <= hi - th) {
    if (H) *H = 0.0;
    return 0.0;
  } else {
    if (H) *H = 1.0;
    return p - hi + th;
  }
}

// =============================================================================== factors
// ObstacleSDFFactor::evaluateError          obstacle/ObstacleSDFFactor-inl.h:18-56
// ObstaclePlanarSDFFactor::evaluateError    obstacle/ObstaclePlanarSDFFactor-inl.h:18-58
void obstacle_factor(const Robot& R, const Sdf& s, double eps, const double* conf, double* err,
                     double* H1) {
  const int D = R.dof, S = R.nr_spheres();
  std::vector<double> c(3 * S), J(H1 ? (size_t)S * 3 * D : 0);
  sphere_centers(R, conf, c.data(), H1 ? J.data() : nullptr);
  for (int k = 0; k < S; k++) {
    const double total_eps = R.sph_r[k] + eps;
    double Hp[3];
    err[k] = hinge_obstacle(s, &c[3 * k], total_eps, H1 ? Hp : nullptr);
    if (H1)
      for (int j = 0; j < D; j++) {
        double a = 0;
        for (int i = 0; i < s.dim; i++) a += Hp[i] * J[((size_t)k * 3 + i) * D + j];
        H1[(size_t)k * D + j] = a;
      }
  }
}

GPInterp::GPInterp(int dof_, bool lie_, const Mat& Qc_, double dt, double tau_)
    : dof(dof_), lie(lie_), delta_t(dt), tau(tau_), Qc(Qc_) {
  // gp/GaussianProcessInterpolatorLinear.h:48-55, gp/GaussianProcessInterpolatorLie.h:50-58
  Lambda = calcLambda(Qc, delta_t, tau);
  Psi = calcPsi(Qc, delta_t, tau);
}

static Mat block(const Mat& M, int i0, int j0, int r, int c) {
  Mat B(r, c);
  for (int i = 0; i < r; i++)
    for (int j = 0; j < c; j++) B(i, j) = M(i0 + i, j0 + j);
  return B;
}

// --- Pose2Vector = Pose2 x R^k as flat [x,y,theta,q...]  geometry/ProductDynamicLieGroup.h
struct LieBetween {
  std::vector<double> r;  // Logmap(x1^-1 x2)
  Mat Hinv, Hcomp1, Hlog;  // Hcomp2 = I
};
static LieBetween lie_between_log(int d, const double* x1, const double* x2, bool jac) {
  LieBetween o;
  o.r.resize(d);
  Pose2 p1{x1[0], x1[1], x1[2]}, p2{x2[0], x2[1], x2[2]};
  const Pose2 p1i = pose2_inverse(p1);
  const Pose2 b = pose2_compose(p1i, p2);
  pose2_logmap(b, o.r.data());
  for (int i = 3; i < d; i++) o.r[i] = (-x1[i]) + x2[i];
  if (jac) {
    o.Hinv = Mat(d, d);
    o.Hcomp1 = Mat(d, d);
    o.Hlog = Mat(d, d);
    double A[9], A2[9], L[9];
    pose2_adjoint(p1, A);                 // Inverse: H = -Ad(p1)
    pose2_adjoint(pose2_inverse(p2), A2); // Compose(a,b): H1 = Ad(b^-1)
    pose2_logmap_derivative(b, L);
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        o.Hinv(i, j) = -A[i * 3 + j];
        o.Hcomp1(i, j) = A2[i * 3 + j];
        o.Hlog(i, j) = L[i * 3 + j];
      }
    for (int i = 3; i < d; i++) {
      o.Hinv(i, i) = -1.0;  // vector-space traits: Inverse H = -I, Compose H1 = I, Logmap H = I
      o.Hcomp1(i, i) = 1.0;
      o.Hlog(i, i) = 1.0;
    }
  }
  return o;
}

void GPInterp::interpolate_pose(const double* c1, const double* v1, const double* c2,
                                const double* v2, double* conf, Mat* H1, Mat* H2, Mat* H3,
                                Mat* H4) const {
  const int d = dof;
  const bool jac = H1 || H2 || H3 || H4;
  if (!lie) {
    // GaussianProcessInterpolatorLinear::interpolatePose  gp/GaussianProcessInterpolatorLinear.h:62-84
    for (int i = 0; i < d; i++) {
      double a = 0;
      for (int j = 0; j < d; j++)
        a += Lambda(i, j) * c1[j] + Lambda(i, d + j) * v1[j] + Psi(i, j) * c2[j] + Psi(i, d + j) * v2[j];
      conf[i] = a;
    }
    if (H1) *H1 = block(Lambda, 0, 0, d, d);
    if (H2) *H2 = block(Lambda, 0, d, d, d);
    if (H3) *H3 = block(Psi, 0, 0, d, d);
    if (H4) *H4 = block(Psi, 0, d, d, d);
    return;
  }
  // GaussianProcessInterpolatorLie<Pose2Vector>::interpolatePose  gp/GaussianProcessInterpolatorLie.h:64-100
  LieBetween lb = lie_between_log(d, c1, c2, jac);
  std::vector<double> xi(d);
  for (int i = 0; i < d; i++) {
    double a = 0;
    for (int j = 0; j < d; j++)
      a += Lambda(i, d + j) * v1[j] + Psi(i, j) * lb.r[j] + Psi(i, d + j) * v2[j];
    xi[i] = a;
  }
  const Pose2 e = pose2_expmap(xi.data());
  const Pose2 p1{c1[0], c1[1], c1[2]};
  const Pose2 p = pose2_compose(p1, e);
  conf[0] = p.x;
  conf[1] = p.y;
  conf[2] = p.th;
  for (int i = 3; i < d; i++) conf[i] = c1[i] + xi[i];
  if (!jac) return;
  Mat Hcomp21(d, d), Hexp(d, d);
  double A[9], E[9];
  pose2_adjoint(pose2_inverse(e), A);
  pose2_expmap_derivative(xi.data(), E);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      Hcomp21(i, j) = A[i * 3 + j];
      Hexp(i, j) = E[i * 3 + j];
    }
  for (int i = 3; i < d; i++) {
    Hcomp21(i, i) = 1.0;
    Hexp(i, i) = 1.0;
  }
  const Mat Hexpr1 = Hexp;  // Hcomp22 = I
  const Mat P11 = block(Psi, 0, 0, d, d);
  if (H1) *H1 = Hcomp21 + matmul(matmul(matmul(matmul(Hexpr1, P11), lb.Hlog), lb.Hcomp1), lb.Hinv);
  if (H2) *H2 = matmul(Hexpr1, block(Lambda, 0, d, d, d));
  if (H3) *H3 = matmul(matmul(Hexpr1, P11), lb.Hlog);  // Hcomp12 = I
  if (H4) *H4 = matmul(Hexpr1, block(Psi, 0, d, d, d));
}

void GPInterp::interpolate_velocity(const double* c1, const double* v1, const double* c2,
                                    const double* v2, double* vel) const {
  const int d = dof;
  if (!lie) {  // gp/GaussianProcessInterpolatorLinear.h:100-122
    for (int i = 0; i < d; i++) {
      double a = 0;
      for (int j = 0; j < d; j++)
        a += Lambda(d + i, j) * c1[j] + Lambda(d + i, d + j) * v1[j] + Psi(d + i, j) * c2[j] +
             Psi(d + i, d + j) * v2[j];
      vel[i] = a;
    }
    return;
  }
  LieBetween lb = lie_between_log(d, c1, c2, false);  // gp/GaussianProcessInterpolatorLie.h:114-146
  for (int i = 0; i < d; i++) {
    double a = 0;
    for (int j = 0; j < d; j++)
      a += Lambda(d + i, d + j) * v1[j] + Psi(d + i, j) * lb.r[j] + Psi(d + i, d + j) * v2[j];
    vel[i] = a;
  }
}

// ObstacleSDFFactorGP::evaluateError        obstacle/ObstacleSDFFactorGP-inl.h:18-76
// ObstaclePlanarSDFFactorGP::evaluateError  obstacle/ObstaclePlanarSDFFactorGP-inl.h:19-79
void obstacle_gp_factor(const Robot& R, const Sdf& s, double eps, const GPInterp& gp,
                        const double* c1, const double* v1, const double* c2, const double* v2,
                        double* err, double* H1, double* H2, double* H3, double* H4) {
  const int D = R.dof, S = R.nr_spheres();
  const bool useH = H1 || H2 || H3 || H4;
  std::vector<double> conf(D);
  Mat J1, J2, J3, J4;
  gp.interpolate_pose(c1, v1, c2, v2, conf.data(), useH ? &J1 : nullptr, useH ? &J2 : nullptr,
                      useH ? &J3 : nullptr, useH ? &J4 : nullptr);
  std::vector<double> Jerr(useH ? (size_t)S * D : 0);
  obstacle_factor(R, s, eps, conf.data(), err, useH ? Jerr.data() : nullptr);
  if (!useH) return;
  // GPBase::updatePoseJacobians  gp/GaussianProcessInterpolatorLinear.h:88-96
  auto chain = [&](const Mat& Hint, double* H) {
    if (!H) return;
    for (int k = 0; k < S; k++)
      for (int j = 0; j < D; j++) {
        double a = 0;
        for (int i = 0; i < D; i++) a += Jerr[(size_t)k * D + i] * Hint(i, j);
        H[(size_t)k * D + j] = a;
      }
  };
  chain(J1, H1);
  chain(J2, H2);
  chain(J3, H3);
  chain(J4, H4);
}

// GaussianProcessPriorLinear::evaluateError  gp/GaussianProcessPriorLinear.h:57-83
// GaussianProcessPriorLie::evaluateError     gp/GaussianProcessPriorLie.h:61-86
void gp_prior_factor(int d, bool lie, double dt, const double* c1, const double* v1,
                     const double* c2, const double* v2, double* err, Mat* H1, Mat* H2, Mat* H3,
                     Mat* H4) {
  if (!lie) {
    for (int i = 0; i < d; i++) {
      err[i] = c1[i] + dt * v1[i] - c2[i];
      err[d + i] = v1[i] - v2[i];
    }
    if (H1) { *H1 = Mat(2 * d, d); for (int i = 0; i < d; i++) (*H1)(i, i) = 1.0; }
    if (H2) { *H2 = Mat(2 * d, d); for (int i = 0; i < d; i++) { (*H2)(i, i) = dt; (*H2)(d + i, i) = 1.0; } }
    if (H3) { *H3 = Mat(2 * d, d); for (int i = 0; i < d; i++) (*H3)(i, i) = -1.0; }
    if (H4) { *H4 = Mat(2 * d, d); for (int i = 0; i < d; i++) (*H4)(d + i, i) = -1.0; }
    return;
  }
  const bool jac = H1 || H2 || H3 || H4;
  LieBetween lb = lie_between_log(d, c1, c2, jac);
  for (int i = 0; i < d; i++) {
    err[i] = lb.r[i] - v1[i] * dt;
    err[d + i] = v2[i] - v1[i];
  }
  if (H1) {
    const Mat T = matmul(matmul(lb.Hlog, lb.Hcomp1), lb.Hinv);
    *H1 = Mat(2 * d, d);
    for (int i = 0; i < d; i++)
      for (int j = 0; j < d; j++) (*H1)(i, j) = T(i, j);
  }
  if (H2) { *H2 = Mat(2 * d, d); for (int i = 0; i < d; i++) { (*H2)(i, i) = -dt; (*H2)(d + i, i) = -1.0; } }
  if (H3) {
    *H3 = Mat(2 * d, d);
    for (int i = 0; i < d; i++)
      for (int j = 0; j < d; j++) (*H3)(i, j) = lb.Hlog(i, j);
  }
  if (H4) { *H4 = Mat(2 * d, d); for (int i = 0; i < d; i++) (*H4)(d + i, i) = 1.0; }
}

// =============================================================================== graph
void Problem::prepare() {
  const int d = set.dof;
  if (set.Qc.r != d) set.Qc = Mat::identity(d);
  // planner/BatchTrajOptimizer-inl.h:30-31
  delta_t = set.total_time / static_cast<double>(set.total_step);
  const double inter_dt = delta_t / static_cast<double>(set.obs_check_inter + 1);
  interp.clear();
  for (int j = 1; j <= set.obs_check_inter; j++)
    interp.emplace_back(d, robot->is_lie(), set.Qc, delta_t, inter_dt * static_cast<double>(j));
  // GaussianProcessPriorLinear ctor: Gaussian::Covariance(calcQ(Qc, delta_t))
  // gp/GaussianProcessPriorLinear.h:40-45 -> information = Q^-1, R^T R = Q^-1
  Qinv = inverse(calcQ(set.Qc, delta_t));
  Rgp = chol_upper(Qinv);
  auto fill = [&](std::vector<double>& v, double x) {
    if ((int)v.size() != d) v.assign(d, x);
  };
  fill(set.pos_up, 1e6);
  fill(set.pos_down, -1e6);
  fill(set.vel_limits, 1e6);
  fill(set.pos_thresh, 1e-3);
  fill(set.vel_thresh, 1e-3);
  fill(set.pos_sigmas, 1e-3);
  fill(set.vel_sigmas, 1e-3);
}

void Problem::retract(const double* traj, const double* delta, double* out) const {
  const int d = set.dof, n = 2 * d;
  for (int i = 0; i < nstates(); i++) {
    const double* z = traj + (size_t)i * n;
    const double* dz = delta + (size_t)i * n;
    double* o = out + (size_t)i * n;
    if (robot->is_lie()) {  // ProductDynamicLieGroup::retract  geometry/ProductDynamicLieGroup.h:84-90
      const Pose2 p = pose2_retract(Pose2{z[0], z[1], z[2]}, dz);
      o[0] = p.x;
      o[1] = p.y;
      o[2] = p.th;
      for (int k = 3; k < d; k++) o[k] = z[k] + dz[k];
    } else {
      for (int k = 0; k < d; k++) o[k] = z[k] + dz[k];
    }
    for (int k = d; k < n; k++) o[k] = z[k] + dz[k];
  }
}

// NonlinearFactorGraph::linearize / ::error over the graph of
// internal::BatchTrajOptimize  planner/BatchTrajOptimizer-inl.h:21-84 (+ script variants 3.3)
double Problem::linearize(const double* traj, std::vector<LinFactor>* F) const {
  const int d = set.dof, n = 2 * d, N = set.total_step, S = robot->nr_spheres();
  const bool lie = robot->is_lie();
  double total = 0.0;
  if (F) F->clear();
  auto push = [&](LinFactor&& f) {
    double e = 0;
    for (double x : f.r) e += x * x;
    total += 0.5 * e;
    if (F) F->push_back(std::move(f));
  };
  for (int i = 0; i <= N; i++) {
    const double* x = traj + (size_t)i * n;
    const double* v = x + d;
    // PriorFactor on start / end  (BatchTrajOptimizer-inl.h:41-48)
    if (i == 0 || (i == N && set.goal_on)) {
      const double* pc = (i == 0) ? start_conf.data() : end_conf.data();
      const double* pv = (i == 0) ? start_vel.data() : end_vel.data();
      const bool conf_prior = !(i == N && set.end_conf_prior_off);  // a goal factor may stand in for it
      LinFactor f;
      f.s0 = i; f.ns = 1; f.m = d; f.r.assign(d, 0.0);
      if (F) f.A.assign((size_t)d * n, 0.0);
      if (!conf_prior) {
        // no factor (rows stay zero: contributes nothing to the error or the normal equations)
      } else if (lie) {
        // gtsam 4.0.x PriorFactor<T>::evaluateError: H = Identity, error = -Local(x, prior)
        // (ProductDynamicLieGroup::localCoordinates throws when Jacobians are requested,
        // geometry/ProductDynamicLieGroup.h:92-101, so only this PriorFactor form can work with
        // Pose2Vector).  Local = first-order Pose2 chart of between(x, prior).
        const Pose2 b = pose2_between(Pose2{x[0], x[1], x[2]}, Pose2{pc[0], pc[1], pc[2]});
        f.r[0] = -b.x / set.conf_prior_sigma;
        f.r[1] = -b.y / set.conf_prior_sigma;
        f.r[2] = -b.th / set.conf_prior_sigma;
        for (int k = 3; k < d; k++) f.r[k] = -(pc[k] - x[k]) / set.conf_prior_sigma;
        if (F)
          for (int k = 0; k < d; k++) f.A[(size_t)k * n + k] = 1.0 / set.conf_prior_sigma;
      } else {
        for (int k = 0; k < d; k++) f.r[k] = (x[k] - pc[k]) / set.conf_prior_sigma;
        if (F)
          for (int k = 0; k < d; k++) f.A[(size_t)k * n + k] = 1.0 / set.conf_prior_sigma;
      }
      push(std::move(f));
      LinFactor g;
      g.s0 = i; g.ns = 1; g.m = d; g.r.assign(d, 0.0);
      if (F) g.A.assign((size_t)d * n, 0.0);
      for (int k = 0; k < d; k++) g.r[k] = (v[k] - pv[k]) / set.vel_prior_sigma;
      if (F)
        for (int k = 0; k < d; k++) g.A[(size_t)k * n + d + k] = 1.0 / set.vel_prior_sigma;
      push(std::move(g));
    }
    // replanner priors: fixConfigAndVel / addPoseEstimate / addStateEstimate
    // (planner/ISAM2TrajOptimizer-inl.h:159-195): PriorFactor with a Gaussian (full information) model
    for (const auto& sp : set.state_priors) {
      if (sp.state != i) continue;
      for (int part = 0; part < (sp.has_vel ? 2 : 1); part++) {
        const Mat Rm = chol_upper(part ? sp.Wv : sp.Wc);
        std::vector<double> r(d);
        const double* tg = part ? sp.vel.data() : sp.conf.data();
        const double* zz = part ? v : x;
        for (int k = 0; k < d; k++) r[k] = zz[k] - tg[k];
        if (lie && !part) {
          const Pose2 b = pose2_between(Pose2{x[0], x[1], x[2]}, Pose2{tg[0], tg[1], tg[2]});
          r[0] = -b.x; r[1] = -b.y; r[2] = -b.th;
        }
        LinFactor f;
        f.s0 = i; f.ns = 1; f.m = d; f.r.assign(d, 0.0);
        if (F) f.A.assign((size_t)d * n, 0.0);
        for (int a = 0; a < d; a++) {
          double s = 0;
          for (int k = 0; k < d; k++) s += Rm(a, k) * r[k];
          f.r[a] = s;
          if (F)
            for (int k = 0; k < d; k++) f.A[(size_t)a * n + part * d + k] = Rm(a, k);
        }
        push(std::move(f));
      }
    }
    // workspace priors / goal factor on this state (isotropic noise)
    for (const auto& w : set.workspace) {
      if (i < w.first_state || i > w.last_state) continue;
      const int rows = (w.mode == WS_POSE) ? 6 : 3;
      std::vector<double> e(rows), H((size_t)rows * d);
      workspace_prior_factor(*robot, w.mode, w.link, w.des, x, e.data(), H.data());
      LinFactor f;
      f.s0 = i; f.ns = 1; f.m = rows; f.r.assign(rows, 0.0);
      if (F) f.A.assign((size_t)rows * n, 0.0);
      for (int a = 0; a < rows; a++) {
        f.r[a] = e[a] / w.sigma;
        if (F)
          for (int k = 0; k < d; k++) f.A[(size_t)a * n + k] = H[(size_t)a * d + k] / w.sigma;
      }
      push(std::move(f));
    }
    // self collision (Diagonal::Sigmas(data.col(3)))
    if (!set.self_collision.empty() && i >= set.self_collision_first && i <= set.self_collision_last) {
      const int np = (int)set.self_collision.size() / 4;
      std::vector<double> e(np), H((size_t)np * d);
      self_collision_factor(*robot, np, set.self_collision.data(), x, e.data(), H.data());
      LinFactor f;
      f.s0 = i; f.ns = 1; f.m = np; f.r.assign(np, 0.0);
      if (F) f.A.assign((size_t)np * n, 0.0);
      for (int a = 0; a < np; a++) {
        const double sg = set.self_collision[(size_t)a * 4 + 3];
        f.r[a] = e[a] / sg;
        if (F)
          for (int k = 0; k < d; k++) f.A[(size_t)a * n + k] = H[(size_t)a * d + k] / sg;
      }
      push(std::move(f));
    }
    // joint / velocity limits  (BatchTrajOptimizer-inl.h:50-59)
    if (set.flag_pos_limit) {
      LinFactor f;
      f.s0 = i; f.ns = 1; f.m = d; f.r.assign(d, 0.0);
      if (F) f.A.assign((size_t)d * n, 0.0);
      for (int k = (lie ? 3 : 0); k < d; k++) {  // JointLimitFactorPose2Vector.h:66-91
        double H;
        const double e = hinge_limit(x[k], set.pos_down[k], set.pos_up[k], set.pos_thresh[k], &H);
        f.r[k] = e / set.pos_sigmas[k];
        if (F) f.A[(size_t)k * n + k] = H / set.pos_sigmas[k];
      }
      push(std::move(f));
    }
    if (set.flag_vel_limit) {
      LinFactor f;
      f.s0 = i; f.ns = 1; f.m = d; f.r.assign(d, 0.0);
      if (F) f.A.assign((size_t)d * n, 0.0);
      for (int k = 0; k < d; k++) {
        double H;
        const double e = hinge_limit(v[k], -set.vel_limits[k], set.vel_limits[k], set.vel_thresh[k], &H);
        f.r[k] = e / set.vel_sigmas[k];
        if (F) f.A[(size_t)k * n + d + k] = H / set.vel_sigmas[k];
      }
      push(std::move(f));
    }
    // vehicle dynamics (hand-built graphs only, matlab/MobileArm2FactorGraphExample.m:122-126)
    if (set.vehicle_dynamics_sigma > 0) {  // dynamics/VehicleDynamics.h:19-27
      LinFactor f;
      f.s0 = i; f.ns = 1; f.m = 1; f.r.assign(1, v[1] / set.vehicle_dynamics_sigma);
      if (F) {
        f.A.assign(n, 0.0);
        f.A[d + 1] = 1.0 / set.vehicle_dynamics_sigma;
      }
      push(std::move(f));
    }
    // unary obstacle factor (BatchTrajOptimizer-inl.h:62)
    if (!(set.obs_skip_first && i == 0)) {
      LinFactor f;
      f.s0 = i; f.ns = 1; f.m = S; f.r.assign(S, 0.0);
      std::vector<double> H(F ? (size_t)S * d : 0);
      obstacle_factor(*robot, *sdf, set.epsilon, x, f.r.data(), F ? H.data() : nullptr);
      for (auto& e : f.r) e /= set.cost_sigma;
      if (F) {
        f.A.assign((size_t)S * n, 0.0);
        for (int k = 0; k < S; k++)
          for (int j = 0; j < d; j++) f.A[(size_t)k * n + j] = H[(size_t)k * d + j] / set.cost_sigma;
      }
      push(std::move(f));
    }
    if (i == 0) continue;
    const double* x0 = traj + (size_t)(i - 1) * n;
    const double* v0 = x0 + d;
    // interpolated obstacle factors (BatchTrajOptimizer-inl.h:69-75)
    for (int j = 0; j < set.obs_check_inter; j++) {
      LinFactor f;
      f.s0 = i - 1; f.ns = 2; f.m = S; f.r.assign(S, 0.0);
      std::vector<double> H1, H2, H3, H4;
      if (F) {
        H1.resize((size_t)S * d); H2.resize((size_t)S * d); H3.resize((size_t)S * d); H4.resize((size_t)S * d);
      }
      obstacle_gp_factor(*robot, *sdf, set.epsilon, interp[j], x0, v0, x, v, f.r.data(),
                         F ? H1.data() : nullptr, F ? H2.data() : nullptr, F ? H3.data() : nullptr,
                         F ? H4.data() : nullptr);
      for (auto& e : f.r) e /= set.cost_sigma;
      if (F) {
        f.A.assign((size_t)S * 2 * n, 0.0);
        for (int k = 0; k < S; k++)
          for (int c = 0; c < d; c++) {
            f.A[(size_t)k * 2 * n + c] = H1[(size_t)k * d + c] / set.cost_sigma;
            f.A[(size_t)k * 2 * n + d + c] = H2[(size_t)k * d + c] / set.cost_sigma;
            f.A[(size_t)k * 2 * n + n + c] = H3[(size_t)k * d + c] / set.cost_sigma;
            f.A[(size_t)k * 2 * n + n + d + c] = H4[(size_t)k * d + c] / set.cost_sigma;
          }
      }
      push(std::move(f));
    }
    // GP prior (BatchTrajOptimizer-inl.h:78-79)
    {
      LinFactor f;
      f.s0 = i - 1; f.ns = 2; f.m = n;
      std::vector<double> e(n);
      Mat H1, H2, H3, H4;
      gp_prior_factor(d, lie, delta_t, x0, v0, x, v, e.data(), F ? &H1 : nullptr, F ? &H2 : nullptr,
                      F ? &H3 : nullptr, F ? &H4 : nullptr);
      f.r.assign(n, 0.0);
      for (int r = 0; r < n; r++) {
        double a = 0;
        for (int k = 0; k < n; k++) a += Rgp(r, k) * e[k];
        f.r[r] = a;
      }
      if (F) {
        Mat H(n, 2 * n);
        for (int r = 0; r < n; r++)
          for (int c = 0; c < d; c++) {
            H(r, c) = H1(r, c);
            H(r, d + c) = H2(r, c);
            H(r, n + c) = H3(r, c);
            H(r, n + d + c) = H4(r, c);
          }
        const Mat W = matmul(Rgp, H);
        f.A = W.a;
      }
      push(std::move(f));
    }
  }
  return total;
}

// =============================================================================== normal equations
void NormalEq::assemble(const std::vector<LinFactor>& F, int nblk_, int n_) {
  nblk = nblk_;
  n = n_;
  D.assign((size_t)nblk * n * n, 0.0);
  O.assign((size_t)(nblk - 1) * n * n, 0.0);
  g.assign((size_t)nblk * n, 0.0);
  for (const auto& f : F) {
    const int w = f.ns * n;
    for (int r = 0; r < f.m; r++) {
      const double* a = &f.A[(size_t)r * w];
      const double rr = f.r[r];
      for (int p = 0; p < w; p++) {
        if (a[p] == 0.0) continue;
        const int bp = f.s0 + p / n, ip = p % n;
        g[(size_t)bp * n + ip] += a[p] * rr;
        for (int q = 0; q < w; q++) {
          if (a[q] == 0.0) continue;
          const int bq = f.s0 + q / n, iq = q % n;
          if (bp == bq)
            D[((size_t)bp * n + ip) * n + iq] += a[p] * a[q];
          else if (bp == bq + 1)
            O[((size_t)bq * n + ip) * n + iq] += a[p] * a[q];
        }
      }
    }
  }
}

bool NormalEq::solve(double lambda, double* x) const {
  // block-tridiagonal Cholesky  H = L L^T,  then  L y = -g,  L^T x = y
  std::vector<double> Ld((size_t)nblk * n * n), Lo((size_t)std::max(nblk - 1, 0) * n * n), y((size_t)nblk * n);
  std::vector<double> S(n * n);
  for (int b = 0; b < nblk; b++) {
    for (int i = 0; i < n * n; i++) S[i] = D[(size_t)b * n * n + i];
    for (int i = 0; i < n; i++) S[i * n + i] += lambda;
    if (b > 0) {
      const double* L = &Lo[(size_t)(b - 1) * n * n];  // L_{b,b-1}
      for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
          double a = 0;
          for (int k = 0; k < n; k++) a += L[i * n + k] * L[j * n + k];
          S[i * n + j] -= a;
        }
    }
    double* Lb = &Ld[(size_t)b * n * n];
    for (int j = 0; j < n; j++) {
      double s = S[j * n + j];
      for (int k = 0; k < j; k++) s -= Lb[j * n + k] * Lb[j * n + k];
      if (!(s > 0.0)) return false;
      Lb[j * n + j] = std::sqrt(s);
      for (int i = j + 1; i < n; i++) {
        double t = S[i * n + j];
        for (int k = 0; k < j; k++) t -= Lb[i * n + k] * Lb[j * n + k];
        Lb[i * n + j] = t / Lb[j * n + j];
      }
    }
    if (b + 1 < nblk) {  // L_{b+1,b} = O_b L_bb^-T
      const double* Ob = &O[(size_t)b * n * n];
      double* Ln = &Lo[(size_t)b * n * n];
      for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
          double t = Ob[i * n + j];
          for (int k = 0; k < j; k++) t -= Ln[i * n + k] * Lb[j * n + k];
          Ln[i * n + j] = t / Lb[j * n + j];
        }
    }
    // forward substitution
    for (int i = 0; i < n; i++) {
      double t = -g[(size_t)b * n + i];
      if (b > 0) {
        const double* L = &Lo[(size_t)(b - 1) * n * n];
        for (int k = 0; k < n; k++) t -= L[i * n + k] * y[(size_t)(b - 1) * n + k];
      }
      for (int k = 0; k < i; k++) t -= Lb[i * n + k] * y[(size_t)b * n + k];
      y[(size_t)b * n + i] = t / Lb[i * n + i];
    }
  }
  for (int b = nblk - 1; b >= 0; b--) {
    const double* Lb = &Ld[(size_t)b * n * n];
    for (int i = n - 1; i >= 0; i--) {
      double t = y[(size_t)b * n + i];
      if (b + 1 < nblk) {
        const double* Ln = &Lo[(size_t)b * n * n];
        for (int k = 0; k < n; k++) t -= Ln[k * n + i] * x[(size_t)(b + 1) * n + k];
      }
      for (int k = i + 1; k < n; k++) t -= Lb[k * n + i] * x[(size_t)b * n + k];
      x[(size_t)b * n + i] = t / Lb[i * n + i];
    }
  }
  return true;
}

void NormalEq::times(const double* x, double* y) const {
  for (int b = 0; b < nblk; b++)
    for (int i = 0; i < n; i++) {
      double a = 0;
      for (int k = 0; k < n; k++) a += D[((size_t)b * n + i) * n + k] * x[(size_t)b * n + k];
      if (b > 0)
        for (int k = 0; k < n; k++) a += O[((size_t)(b - 1) * n + i) * n + k] * x[(size_t)(b - 1) * n + k];
      if (b + 1 < nblk)
        for (int k = 0; k < n; k++) a += O[((size_t)b * n + k) * n + i] * x[(size_t)(b + 1) * n + k];
      y[(size_t)b * n + i] = a;
    }
}

double NormalEq::quad(const double* x) const {
  std::vector<double> y((size_t)nblk * n);
  times(x, y.data());
  double a = 0;
  for (size_t i = 0; i < y.size(); i++) a += g[i] * x[i] + 0.5 * x[i] * y[i];
  return a;
}

// =============================================================================== optimizers
// GTSAM semantics, SURVEY.md appendix B ("parity unpinned": GTSAM is not vendored).
static bool check_convergence(double rel, double abs_, double err_tol, double cur, double nw) {
  if (nw <= err_tol) return true;
  const double abs_dec = cur - nw;
  const double rel_dec = abs_dec / cur;
  return (rel != 0.0 && rel_dec <= rel) || (abs_dec <= abs_);
}

namespace {
struct State {
  std::vector<double> values;
  double error = 0;
  int iterations = 0;
  double lambda = 0;  // LM
  double delta = 0;   // Dogleg trust region
  bool not_spd = false;
};
}  // namespace

static void iterate_gn(const Problem& P, State& st) {
  // GaussNewtonOptimizer::iterate: linearize, solve, retract, error
  std::vector<LinFactor> F;
  P.linearize(st.values.data(), &F);
  NormalEq ne;
  ne.assemble(F, P.nstates(), P.n());
  std::vector<double> dx(st.values.size());
  if (!ne.solve(0.0, dx.data())) {
    st.not_spd = true;
    st.iterations++;
    return;
  }
  std::vector<double> nv(st.values.size());
  P.retract(st.values.data(), dx.data(), nv.data());
  st.error = P.error(nv.data());
  st.values.swap(nv);
  st.iterations++;
}

static void iterate_lm(const Problem& P, State& st) {
  // LevenbergMarquardtOptimizer::iterate -> tryLambda loop (fixed lambda factor, no diagonal damping)
  const Settings& s = P.set;
  std::vector<LinFactor> F;
  P.linearize(st.values.data(), &F);
  NormalEq ne;
  ne.assemble(F, P.nstates(), P.n());
  std::vector<double> dx(st.values.size()), nv(st.values.size());
  for (;;) {
    bool step_ok = false, stop = false;
    double new_err = std::numeric_limits<double>::infinity(), fidelity = 0;
    const bool solved = ne.solve(st.lambda, dx.data());
    if (solved) {
      const double old_lin = st.error;           // linear.error(0) == nonlinear error (no robust)
      const double new_lin = st.error + ne.quad(dx.data());
      const double lin_change = old_lin - new_lin;
      if (lin_change >= 0) {
        P.retract(st.values.data(), dx.data(), nv.data());
        new_err = P.error(nv.data());
        const double cost_change = st.error - new_err;
        if (lin_change > std::numeric_limits<double>::epsilon() * old_lin) {
          fidelity = cost_change / lin_change;
          step_ok = fidelity > s.lm_min_model_fidelity;
        }
        const double min_abs = s.rel_thresh * st.error;
        if (std::fabs(cost_change) < min_abs) stop = true;
      }
    }
    if (step_ok) {
      st.values = nv;
      st.error = new_err;
      st.lambda = std::max(s.lm_lambda_lower, st.lambda / s.lm_lambda_factor);
      st.iterations++;
      return;
    } else if (!stop) {
      st.lambda *= s.lm_lambda_factor;
      if (st.lambda >= s.lm_lambda_upper) return;  // give up, state unchanged
    } else {
      return;
    }
  }
}

// test probe: when set, every Dogleg trial point appends {gg, gHg, g.dx_n, |dx_n|^2, |dx_u|^2, dx_u.dx_n, tau,
// Delta, rho, new_f} (tau = -1 outside the blend branch).  Single-threaded use only (tests/, scripts/).
static double* g_dl_probe = nullptr;
static int g_dl_probe_cap = 0, g_dl_probe_n = 0;
void set_dogleg_probe(double* buf, int cap_rows) {
  g_dl_probe = buf;
  g_dl_probe_cap = cap_rows;
  g_dl_probe_n = 0;
}
int dogleg_probe_rows() { return g_dl_probe_n; }

static void iterate_dogleg(const Problem& P, State& st) {
  // DoglegOptimizer::iterate + DoglegOptimizerImpl::Iterate(ONE_STEP_PER_ITERATION)
  std::vector<LinFactor> F;
  P.linearize(st.values.data(), &F);
  NormalEq ne;
  ne.assemble(F, P.nstates(), P.n());
  const size_t m = st.values.size();
  std::vector<double> dx_n(m), dx_u(m), Hg(m), dx_d(m), nv(m);
  if (!ne.solve(0.0, dx_n.data())) {
    st.not_spd = true;
    st.iterations++;
    return;
  }
  // steepest descent point: dx_u = -(g^T g / g^T H g) g   (optimizeGradientSearch)
  ne.times(ne.g.data(), Hg.data());
  double gg = 0, gHg = 0;
  for (size_t i = 0; i < m; i++) {
    gg += ne.g[i] * ne.g[i];
    gHg += ne.g[i] * Hg[i];
  }
  const double step = -gg / gHg;
  for (size_t i = 0; i < m; i++) dx_u[i] = step * ne.g[i];
  double delta = st.delta;
  const double f_error = st.error, M_error = st.error;
  double new_f = f_error;
  bool stay = true;
  while (stay) {
    // ComputeDoglegPoint
    double uu = 0, nn = 0, un = 0;
    for (size_t i = 0; i < m; i++) {
      uu += dx_u[i] * dx_u[i];
      nn += dx_n[i] * dx_n[i];
      un += dx_u[i] * dx_n[i];
    }
    const double DeltaSq = delta * delta;
    double tau_used = -1.0;
    if (DeltaSq < uu) {
      const double k = std::sqrt(DeltaSq / uu);
      for (size_t i = 0; i < m; i++) dx_d[i] = k * dx_u[i];
    } else if (DeltaSq < nn) {
      const double a = uu - 2. * un + nn, b = 2. * (un - uu), c = uu - delta * delta;
      const double sq = std::sqrt(b * b - 4 * a * c);
      const double tau1 = (-b + sq) / (2. * a), tau2 = (-b - sq) / (2. * a);
      const double tau = (0.0 <= tau1 && tau1 <= 1.0) ? tau1 : tau2;
      tau_used = tau;
      for (size_t i = 0; i < m; i++) dx_d[i] = (1. - tau) * dx_u[i] + tau * dx_n[i];
    } else {
      dx_d = dx_n;
    }
    P.retract(st.values.data(), dx_d.data(), nv.data());
    new_f = P.error(nv.data());
    const double new_M = M_error + ne.quad(dx_d.data());
    const double rho = (std::fabs(f_error - new_f) < 1e-15 || std::fabs(M_error - new_M) < 1e-15)
                           ? 0.5
                           : (f_error - new_f) / (M_error - new_M);
    if (g_dl_probe && g_dl_probe_n < g_dl_probe_cap) {
      double gn = 0;
      for (size_t i = 0; i < m; i++) gn += ne.g[i] * dx_n[i];
      double* row = g_dl_probe + (size_t)g_dl_probe_n * 10;
      const double vals[10] = {gg, gHg, gn, nn, uu, un, tau_used, delta, rho, new_f};
      for (int k = 0; k < 10; k++) row[k] = vals[k];
      g_dl_probe_n++;
    }
    if (rho >= 0.75) {
      double dn = 0;
      for (size_t i = 0; i < m; i++) dn += dx_d[i] * dx_d[i];
      delta = std::max(delta, 3.0 * std::sqrt(dn));
      stay = false;
    } else if (rho >= 0.25) {
      stay = false;
    } else if (rho >= 0.0) {
      if (delta > 1e-5) delta = 0.5 * delta;
      stay = false;
    } else {
      if (delta > 1e-5) {
        delta *= 0.5;
        stay = true;
      } else {
        std::fill(dx_d.begin(), dx_d.end(), 0.0);
        P.retract(st.values.data(), dx_d.data(), nv.data());
        new_f = f_error;
        stay = false;
      }
    }
  }
  st.values = nv;
  st.error = new_f;
  st.delta = delta;
  st.iterations++;
}

// gpmp2::optimize   planner/BatchTrajOptimizer.cpp:212-308
OptResult optimize(const Problem& P, const double* init, double* out) {
  const Settings& s = P.set;
  const size_t m = (size_t)P.nstates() * P.n();
  State st;
  st.values.assign(init, init + m);
  st.error = P.error(init);
  st.lambda = s.lm_lambda_initial;
  st.delta = s.dogleg_delta_initial;
  OptResult res;
  res.trace.push_back(st.error);
  auto finish = [&](const std::vector<double>& v, double err, int status) {
    std::memcpy(out, v.data(), m * sizeof(double));
    res.final_error = err;
    res.status = status;
    res.iterations = st.iterations;
    return res;
  };
  if (s.fixed_iterations > 0) {
    for (int k = 0; k < s.fixed_iterations && !st.not_spd; k++) {
      if (s.opt_type == 0) iterate_gn(P, st);
      else if (s.opt_type == 1) iterate_lm(P, st);
      else iterate_dogleg(P, st);
      res.trace.push_back(st.error);
    }
    return finish(st.values, st.error, st.not_spd ? 3 : 1);
  }
  double current = st.error;
  if (current <= s.error_tol) return finish(st.values, st.error, 4);  // :250-255
  if (st.iterations >= s.max_iter) return finish(st.values, st.error, 1);  // :264-268
  std::vector<double> last_values;
  double last_error = current;
  do {  // :273-286
    current = st.error;
    last_values = st.values;
    last_error = st.error;
    if (s.opt_type == 0) iterate_gn(P, st);
    else if (s.opt_type == 1) iterate_lm(P, st);
    else iterate_dogleg(P, st);
    if (s.verbosity) std::printf("newError: %.17g\n", st.error);
    if (st.not_spd) return finish(last_values, last_error, 3);
    res.trace.push_back(st.error);
  } while (st.iterations < s.max_iter &&
           !check_convergence(s.rel_thresh, s.abs_error_tol, s.error_tol, current, st.error));
  const bool conv = check_convergence(s.rel_thresh, s.abs_error_tol, s.error_tol, current, st.error);
  if (st.error > current) {  // :297-307
    if (s.final_iter_no_increase) return finish(last_values, last_error, 2);
    return finish(st.values, st.error, conv ? 0 : 1);
  }
  return finish(st.values, st.error, conv ? 0 : 1);
}

}  // namespace orc
