// dispatch.h -- runtime (kind, arm_dof) -> compile-time template instantiation.
// Every kernel that unrolls over the kinematic chain is instantiated for this table only;
// anything else returns GPMP2MI_ERR_UNSUPPORTED.
#pragma once
#include "common.h"

#define G2_CASE2_(K, A, A2, kind, ad, ad2, STMT)                  \
  if (!done_ && (kind) == (K) && (ad) == (A) && (ad2) == (A2)) {  \
    constexpr int KIND_ = (K);                                    \
    constexpr int AD_ = (A);                                      \
    constexpr int AD2_ = (A2);                                    \
    STMT;                                                         \
    done_ = true;                                                 \
  }
#define G2_CASE_(K, A, kind, ad, STMT) G2_CASE2_(K, A, 0, kind, ad, 0, STMT)

// (kind, joints of arm 1, joints of arm 2) of a robot handle `h` (RobotDev)
#define G2_DISPATCH_ROBOT_H(h, STMT) G2_DISPATCH_ROBOT2((h).kind, (h).arm_dof - (h).arm2_dof, (h).arm2_dof, STMT)
#define G2_DISPATCH_ROBOT(kind, ad, STMT) G2_DISPATCH_ROBOT2(kind, ad, 0, STMT)
#define G2_DISPATCH_ROBOT2(kind, ad, ad2, STMT)                                           \
  do {                                                                                \
    bool done_ = false;                                                               \
    G2_CASE_(GPMP2MI_ROBOT_ARM, 1, kind, ad, STMT)                                    \
    G2_CASE_(GPMP2MI_ROBOT_ARM, 2, kind, ad, STMT)                                    \
    G2_CASE_(GPMP2MI_ROBOT_ARM, 3, kind, ad, STMT)                                    \
    G2_CASE_(GPMP2MI_ROBOT_ARM, 4, kind, ad, STMT)                                    \
    G2_CASE_(GPMP2MI_ROBOT_ARM, 5, kind, ad, STMT)                                    \
    G2_CASE_(GPMP2MI_ROBOT_ARM, 6, kind, ad, STMT)                                    \
    G2_CASE_(GPMP2MI_ROBOT_ARM, 7, kind, ad, STMT)                                    \
    G2_CASE_(GPMP2MI_ROBOT_ARM, 8, kind, ad, STMT)                                    \
    G2_CASE_(GPMP2MI_ROBOT_POINT, 0, kind, ad, STMT)                                  \
    G2_CASE_(GPMP2MI_ROBOT_POSE2_MOBILE_BASE, 0, kind, ad, STMT)                      \
    G2_CASE_(GPMP2MI_ROBOT_POSE2_MOBILE_ARM, 1, kind, ad, STMT)                       \
    G2_CASE_(GPMP2MI_ROBOT_POSE2_MOBILE_ARM, 2, kind, ad, STMT)                       \
    G2_CASE_(GPMP2MI_ROBOT_POSE2_MOBILE_ARM, 3, kind, ad, STMT)                       \
    G2_CASE_(GPMP2MI_ROBOT_POSE2_MOBILE_ARM, 4, kind, ad, STMT)                       \
    G2_CASE_(GPMP2MI_ROBOT_POSE2_MOBILE_ARM, 5, kind, ad, STMT)                       \
    G2_CASE_(GPMP2MI_ROBOT_POSE2_MOBILE_ARM, 6, kind, ad, STMT)                       \
    G2_CASE_(GPMP2MI_ROBOT_POSE2_MOBILE_ARM, 7, kind, ad, STMT)                       \
    G2_CASE2_(GPMP2MI_ROBOT_POSE2_MOBILE_2ARMS, 1, 1, kind, ad, ad2, STMT)            \
    G2_CASE2_(GPMP2MI_ROBOT_POSE2_MOBILE_2ARMS, 2, 2, kind, ad, ad2, STMT)            \
    G2_CASE2_(GPMP2MI_ROBOT_POSE2_MOBILE_2ARMS, 3, 3, kind, ad, ad2, STMT)            \
    G2_CASE2_(GPMP2MI_ROBOT_POSE2_MOBILE_2ARMS, 4, 4, kind, ad, ad2, STMT)            \
    G2_CASE2_(GPMP2MI_ROBOT_POSE2_MOBILE_VETLIN_ARM, 1, 0, kind, ad, ad2, STMT)       \
    G2_CASE2_(GPMP2MI_ROBOT_POSE2_MOBILE_VETLIN_ARM, 2, 0, kind, ad, ad2, STMT)       \
    G2_CASE2_(GPMP2MI_ROBOT_POSE2_MOBILE_VETLIN_ARM, 3, 0, kind, ad, ad2, STMT)       \
    G2_CASE2_(GPMP2MI_ROBOT_POSE2_MOBILE_VETLIN_ARM, 7, 0, kind, ad, ad2, STMT)       \
    G2_CASE2_(GPMP2MI_ROBOT_POSE2_MOBILE_VETLIN_2ARMS, 2, 2, kind, ad, ad2, STMT)     \
    G2_CASE2_(GPMP2MI_ROBOT_POSE2_MOBILE_VETLIN_2ARMS, 3, 3, kind, ad, ad2, STMT)     \
    G2_CASE2_(GPMP2MI_ROBOT_POSE2_MOBILE_VETLIN_2ARMS, 7, 7, kind, ad, ad2, STMT)     \
    if (!done_) {                                                                     \
      g2::set_error("robot kind / dof combination is not instantiated");              \
      return GPMP2MI_ERR_UNSUPPORTED;                                                 \
    }                                                                                 \
  } while (0)

// fixed-base arms only (kernels that exist in an extra variant for them): AD_ = number of joints
#define G2_DISPATCH_ROBOT_ARM_ONLY(ad, STMT)                                \
  do {                                                                      \
    switch (ad) {                                                           \
      case 1: { constexpr int AD_ = 1; STMT; } break;                       \
      case 2: { constexpr int AD_ = 2; STMT; } break;                       \
      case 3: { constexpr int AD_ = 3; STMT; } break;                       \
      case 4: { constexpr int AD_ = 4; STMT; } break;                       \
      case 5: { constexpr int AD_ = 5; STMT; } break;                       \
      case 6: { constexpr int AD_ = 6; STMT; } break;                       \
      case 7: { constexpr int AD_ = 7; STMT; } break;                       \
      case 8: { constexpr int AD_ = 8; STMT; } break;                       \
      default:                                                              \
        g2::set_error("arm dof is not instantiated");                       \
        return GPMP2MI_ERR_UNSUPPORTED;                                     \
    }                                                                       \
  } while (0)
