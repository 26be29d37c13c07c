/*
 * oracle/exact_cg_norm.h -- TEST INFRASTRUCTURE ONLY.
 * Force-included when src/scs.c is compiled for the "exactcg" flavour of the
 * reference: glbopts.h:253-255 lets a build override CG_NORM, the norm that
 * project_lin_sys (src/scs.c:745-762) feeds into the CG tolerance schedule.
 * With a norm that is identically zero the schedule collapses to its floor,
 * tol = CG_BEST_TOL = 1e-12, on every iteration -- i.e. the reference's indirect
 * solver with (numerically) exact linear solves.  Only scs.c is built with this;
 * the backend (linsys/cpu/indirect/private.c) keeps the real norm for its own
 * stopping test.  Used to compare ADMM trajectories without CG's well-known
 * sensitivity to summation order at loose tolerances (parity).
 */
#ifndef SCS_AMD_ORACLE_EXACT_CG_NORM_H
#define SCS_AMD_ORACLE_EXACT_CG_NORM_H
#include "scs_types.h"
static inline scs_float scs_amd_zero_norm(const scs_float *v, scs_int len) {
  (void)v;
  (void)len;
  return 0;
}
#define CG_NORM scs_amd_zero_norm
#endif
