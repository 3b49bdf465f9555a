// dispatch.h -- runtime (kind, arm_dof) -> compile-time template instantiation.
// Every kernel that unrolls over the kinematic chain is instantiated for this table only;
// anything else returns GPMP2MI_ERR_UNSUPPORTED.
#pragma once
#include "common.h"

#define G2_CASE_(K, A, kind, ad, STMT)                 \
  if (!done_ && (kind) == (K) && (ad) == (A)) {        \
    constexpr int KIND_ = (K);                         \
    constexpr int AD_ = (A);                           \
    STMT;                                              \
    done_ = true;                                      \
  }

#define G2_DISPATCH_ROBOT(kind, ad, STMT)                                             \
  do {                                                                                \
    bool done_ = false;                                                               \
    G2_CASE_(GPMP2MI_ROBOT_ARM, 1, kind, ad, STMT)                                    \
    G2_CASE_(GPMP2MI_ROBOT_ARM, 2, kind, ad, STMT)                                    \
    G2_CASE_(GPMP2MI_ROBOT_ARM, 3, kind, ad, STMT)                                    \
    G2_CASE_(GPMP2MI_ROBOT_ARM, 4, kind, ad, STMT)                                    \
    G2_CASE_(GPMP2MI_ROBOT_ARM, 5, kind, ad, STMT)                                    \
    G2_CASE_(GPMP2MI_ROBOT_ARM, 6, kind, ad, STMT)                                    \
    G2_CASE_(GPMP2MI_ROBOT_ARM, 7, kind, ad, STMT)                                    \
    G2_CASE_(GPMP2MI_ROBOT_POINT, 0, kind, ad, STMT)                                  \
    G2_CASE_(GPMP2MI_ROBOT_POSE2_MOBILE_BASE, 0, kind, ad, STMT)                      \
    G2_CASE_(GPMP2MI_ROBOT_POSE2_MOBILE_ARM, 2, kind, ad, STMT)                       \
    G2_CASE_(GPMP2MI_ROBOT_POSE2_MOBILE_ARM, 3, kind, ad, STMT)                       \
    if (!done_) {                                                                     \
      g2::set_error("robot kind / dof combination is not instantiated");              \
      return GPMP2MI_ERR_UNSUPPORTED;                                                 \
    }                                                                                 \
  } while (0)
