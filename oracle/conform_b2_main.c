/*
 * oracle/conform_b2_main.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Runs the reference's OWN test problems (the headers under <reference>/test/problems,
 * included from where they lie -- nothing is copied) against THIS repo's whole-solve
 * library: scs_init / scs_solve / scs_update / scs_finish / scs resolve to
 * scs_amd/lib/libscsamd.so, while the verification helpers the test headers use
 * (test/problem_utils.h: _scs_proj_dual_cone, _scs_accum_by_a, _scs_dot, ...) come from
 * the reference's objects.  Unlike test/run_tests.c it does not stop at the first
 * failure and it leaves out the tests that need cones / features this backend
 * announces as out of scope (spectral cones) or that poke reference-internal structs.
 */
#include <stdio.h>

#include "minunit.h"
#include "problem_utils.h"
#include "scs.h"

#include "problems/degenerate.h"
#include "problems/dense_qp.h"
#include "problems/hs21_tiny_qp.h"
#include "problems/infeasible_lp.h"
#include "problems/infeasible_socp.h"
#include "problems/infeasible_tiny_qp.h"
#include "problems/lp_update.h"
#include "problems/qafiro_tiny_qp.h"
#include "problems/small_lp.h"
#include "problems/small_qp.h"
#include "problems/test_dual_exp_cone.h"
#include "problems/test_exp_cone.h"
#include "problems/test_power_cone.h"
#include "problems/complex_PSD.h"
#include "problems/sd_and_complex_sd.h"
#include "problems/rob_gauss_cov_est.h"
#include "problems/hs21_tiny_qp_rw.h"
#include "problems/max_ent.h"
#include "problems/mpc_bug.h"
#include "problems/random_prob.h"
#include "problems/test_inaccurate.h"
#include "problems/test_mixed_cones.h"
#include "problems/test_soc_sizes.h"
#include "problems/test_box_cone.h"
#include "problems/test_psd_n1.h"
#include "problems/test_solver_options.h"
#include "problems/test_zero_cone.h"
#include "problems/unbounded_lp.h"
#include "problems/unbounded_socp.h"
#include "problems/unbounded_tiny_qp.h"
#include "problems/test_validation.h"

int tests_run = 0;
static int n_fail = 0;

#define RUN(t)                                                                 \
  do {                                                                         \
    const char *msg_;                                                          \
    printf("*********************************************************\n");    \
    printf("Running test: %s\n", #t);                                          \
    msg_ = t();                                                                \
    tests_run++;                                                               \
    if (msg_) {                                                                \
      n_fail++;                                                                \
      printf("CONFORM FAIL %s : %s\n", #t, msg_);                              \
    } else {                                                                   \
      printf("CONFORM PASS %s\n", #t);                                         \
    }                                                                          \
    fflush(stdout);                                                            \
  } while (0)

int main(void) {
  RUN(test_validation);
  RUN(degenerate);
  RUN(dense_qp);
  RUN(small_lp);
  RUN(small_qp);
  RUN(lp_update);
  RUN(hs21_tiny_qp);
  RUN(qafiro_tiny_qp);
  RUN(infeasible_tiny_qp);
  RUN(infeasible_lp);
  RUN(infeasible_socp);
  RUN(unbounded_tiny_qp);
  RUN(unbounded_lp);
  RUN(unbounded_socp);
  RUN(test_exp_cone);
  RUN(test_dual_exp_cone);
  RUN(test_power_cone);
  RUN(test_power_cone_p09);
  RUN(test_dual_power_cone);
  RUN(test_multi_power);
  RUN(test_power_cone_infeasible);
  RUN(complex_PSD);
  RUN(sd_and_complex_sd);
  RUN(rob_gauss_cov_est);
  RUN(hs21_tiny_qp_rw);
  RUN(max_ent);
  RUN(mpc_bug);
  RUN(random_prob);
  RUN(test_soc_size1);
  RUN(test_soc_size2);
  RUN(test_soc_size3);
  RUN(test_soc_size5);
  RUN(test_multi_soc);
  RUN(test_zero_cone);
  RUN(test_box_cone_lp);
  RUN(test_psd_n1);
  RUN(test_solved_inaccurate);
  RUN(test_infeasible_inaccurate);
  RUN(test_unbounded_inaccurate);
  RUN(test_max_iters_1);
  RUN(test_adaptive_scale);
  RUN(test_no_acceleration);
  RUN(test_type2_acceleration);
  RUN(test_aa_relaxation_sweep);
  RUN(test_aa_regularization_sweep);
  RUN(test_negative_lookback_rejected);
  RUN(test_invalid_aa_relaxation_rejected);
  RUN(test_invalid_aa_regularization_rejected);
  RUN(test_normalize_off);
  RUN(test_scs_version);
  RUN(test_time_limit_secs);
  RUN(test_warm_start);
  RUN(test_mixed_cones);
  printf("CONFORM SUMMARY: %d run, %d failed\n", tests_run, n_fail);
  return n_fail ? 1 : 0;
}
