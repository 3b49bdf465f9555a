// common.h -- shared host/device definitions of the gfx950 engine (product code).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <string>

#include "../../include/gpmp2mi.h"

namespace g2 {

constexpr int MAXJ = 14;                    // DH joints a kernel is instantiated for (two 7-joint arms)
constexpr int MAXS = GPMP2MI_MAX_SPHERES;   // sphere model staged on chip
constexpr int MAXD = GPMP2MI_MAX_DOF;       // max total dof (3 base + 1 lift + 7 + 7 arm joints: the PR2 model)
constexpr int REC_G_MAX = MAXD * (MAXD + 1) / 2;

// ---------------------------------------------------------------- error plumbing
void set_error(const std::string& msg);
// diagnostic -DG2_STAMPS builds: every kernel stamps only while the trajectory is in this iteration (its second), so the
// 64 slots of a trajectory hold ONE pass of every kernel and differences between slots never mix passes
#ifndef G2_STAMP_ITER
#define G2_STAMP_ITER 1
#endif
#define G2_HIP(call)                                                                      \
  do {                                                                                    \
    hipError_t e_ = (call);                                                               \
    if (e_ != hipSuccess) {                                                               \
      g2::set_error(std::string(#call) + ": " + hipGetErrorString(e_));                   \
      return (e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice ||                    \
              e_ == hipErrorNoBinaryForGpu) ? GPMP2MI_ERR_NO_DEVICE : GPMP2MI_ERR_HIP;    \
    }                                                                                     \
  } while (0)
#define G2_CHECK(cond, code, msg)   \
  do {                              \
    if (!(cond)) {                  \
      g2::set_error(msg);           \
      return code;                  \
    }                               \
  } while (0)

#define G2_TRY(expr)             \
  do {                           \
    int rc_ = (expr);            \
    if (rc_ != GPMP2MI_OK) return rc_; \
  } while (0)

// ---------------------------------------------------------------- device-side model data
// Robot model as kernels see it.  Lives in HBM once per robot handle; every workgroup stages it
// into LDS (3 KB) before use so per-sphere constants are broadcast LDS reads.
// Spheres are sorted by link id on the host (sph_orig keeps the caller's order).
struct RobotDev {
  int kind, dof, arm_dof, nr_links, nr_spheres, base_dof, pad0, pad1;
  double ca[MAXJ], sa[MAXJ], a[MAXJ], d[MAXJ], bias[MAXJ];
  double base[12];           // 3x4 row-major: ARM world_T_base, MOBILE_ARM base_T_arm, 2ARMS base_T_arm1,
                             // VETLIN_* base_T_torso
  double base2[12];          // 2ARMS base_T_arm2; VETLIN_ARM torso_T_arm; VETLIN_2ARMS torso_T_arm1
  double base3[12];          // VETLIN_2ARMS torso_T_arm2
  int arm2_dof, reverse_linact;
  int sph_link[MAXS];        // ascending
  int sph_orig[MAXS];        // index in the caller's BodySphereVector
  int link_first[MAXJ + 4];  // first sorted sphere of each link (vehicle base + torso + MAXJ arm links), [nr_links] = nr_spheres
  double sph_r[MAXS];
  double sph_c[MAXS * 3];
};

// Signed distance field as kernels see it.  `cells` is the cell-packed layout: for every cell
// (z, y, x) the 2^dim corner values are stored contiguously (64 B in 3-D, 32 B in 2-D) so one
// trilinear lookup touches one 64-B sector instead of 4 cache lines.  Corner order (3-D):
// [dz][dy][dx] = v000(lo row, lo col, lo z) ... with index  dz*4 + dy*2 + dx.
struct SdfDev {
  int dim, nx, ny, nz;
  double ox, oy, oz, cell, inv_cell;
  double hix, hiy, hiz;      // origin + (n-1)*cell, computed exactly as the reference does
  const double* cells;       // [nz][ny][nx][2^dim]
  const double* plain;       // [nz][ny][nx] (kept for the A/B layout experiment)
};

// 8 scalars of Lambda / Psi per GP sub-step (both are (2x2) (x) I_D).
struct GpCoef {
  double l11, l12, l21, l22, p11, p12, p21, p22;
};

}  // namespace g2
