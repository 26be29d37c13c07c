// aa_host.cpp -- host-side Anderson acceleration (stays on the host by design,
// BASELINE.json north_star).  Placeholder until the restatement of reference
// src/aa.c:657-1000 lands: aa_host_init returns NULL, which the driver treats
// exactly like the reference treats a NULL aa_init (src/scs.c:1097-1109, the
// no-LAPACK build): it warns once and runs plain Douglas-Rachford.
#include "scs_host.h"

namespace scsamd {

struct AaHost {
  int dummy;
};

AaHost *aa_host_init(int, int, int, int, real, real, real, real, int) { return nullptr; }
real aa_host_apply(real *, const real *, AaHost *) { return 0; }
int aa_host_safeguard(real *, real *, AaHost *) { return 0; }
void aa_host_reset(AaHost *) {}
void aa_host_finish(AaHost *a) { delete a; }
void aa_host_stats(const AaHost *, AaStats *) {}

} // namespace scsamd
