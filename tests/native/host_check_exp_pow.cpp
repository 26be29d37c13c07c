// CPU build of the exponential / power cone arithmetic of scs_amd/csrc/cones_exp_pow.h (SCSAMD_EXPPOW_HOST_CHECK:
// the same functions, one lane at a time, wave votes reduced to the lane's own predicate) so that the `-m "not gpu"`
// suite can pin it to the reference's golden vectors without a GPU.  Test infrastructure only.
#include <cmath>
#ifdef XP_CHECK_FLOAT // the -DSFLOAT arithmetic
typedef float scs_float;
#else
typedef double scs_float;
#endif
#define SCSAMD_EXPPOW_HOST_CHECK 1
using std::exp;
using std::pow;
using std::sqrt;
#include "../../scs_amd/csrc/cones_exp_pow.h"

extern "C" void xp_check_project_exp(scs_float *v, int dual) {
  scsamd::Triple t{v[0], v[1], v[2]};
  const scsamd::Triple r = scsamd::xp::project_exp(t, true, dual != 0);
  v[0] = r.u; v[1] = r.w; v[2] = r.t;
}
extern "C" void xp_check_project_pow(scs_float *v, scs_float a) {
  scsamd::Triple t{v[0], v[1], v[2]};
  const scsamd::Triple r = scsamd::xp::project_pow(t, a, true);
  v[0] = r.u; v[1] = r.w; v[2] = r.t;
}
// evaluations of F (one exp each) the root search spent since the last call; resets the counter
extern "C" long xp_check_take_eval_count(void) {
  const long c = scsamd::xp::xp_eval_count;
  scsamd::xp::xp_eval_count = 0;
  return c;
}
