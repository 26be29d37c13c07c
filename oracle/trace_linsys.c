/*
 * oracle/trace_linsys.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A pass-through implementation of the reference's linear-system plugin ABI
 * (include/linsys.h:25-71) that forwards every call to the reference's own CPU
 * indirect backend (compiled with its five entry points renamed to ref_*) and
 * records what crossed the boundary.  Linked with the reference's scs.c it shows,
 * per scs_solve_lin_sys call: tol, |b|_inf on entry, |s|_inf, |[x;y]|_inf on exit;
 * with SCS_TRACE_DUMP=<dir> and SCS_TRACE_CALLS=i,j,k it also dumps the full
 * (b, s, tol) -> (x, y) vectors of those calls as raw little-endian doubles --
 * the golden vectors under tests/golden/ were produced this way.
 *
 * It also reports the reference's OWN CG iteration counts: the backend keeps a
 * running total in its private workspace (linsys/cpu/indirect/private.h:29,
 * incremented at private.c:318); the shim includes that header where it lies and
 * differences the total around every call.  oracle_trace_cg_its() / _calls() /
 * _cg_its_of_call() expose them so that bench.py's CPU window can be priced in
 * the reference's own seconds per CG iteration.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "scs.h"
#include "private.h" /* the reference backend's own workspace struct: tot_cg_its */

ScsLinSysWork *ref_scs_init_lin_sys_work(const ScsMatrix *A, const ScsMatrix *P, const scs_float *diag_r);
scs_int ref_scs_solve_lin_sys(ScsLinSysWork *w, scs_float *b, const scs_float *s, scs_float tol);
scs_int ref_scs_update_lin_sys_diag_r(ScsLinSysWork *w, const scs_float *new_diag_r);
void ref_scs_free_lin_sys_work(ScsLinSysWork *w);
const char *ref_scs_get_lin_sys_method(void);

static long g_call = 0;
static int g_n = 0, g_m = 0;
static FILE *g_log = NULL;
/* per-call CG iteration counts of the most recent workspace (ring of the first TRACE_MAX_CALLS calls) */
#define TRACE_MAX_CALLS 65536
static int g_cg_of_call[TRACE_MAX_CALLS];
static signed char g_warm_of_call[TRACE_MAX_CALLS]; /* 1: s != NULL (an ADMM iteration's solve), 0: the g solve */
static long long g_cg_total = 0;

long long oracle_trace_cg_its(void) { return g_cg_total; }
long oracle_trace_calls(void) { return g_call; }
int oracle_trace_cg_its_of_call(long call) { return call >= 0 && call < g_call && call < TRACE_MAX_CALLS ? g_cg_of_call[call] : -1; }
int oracle_trace_call_had_warm_start(long call) { return call >= 0 && call < g_call && call < TRACE_MAX_CALLS ? g_warm_of_call[call] : -1; }

static double amax(const scs_float *v, long len) {
  double mx = 0;
  long i;
  for (i = 0; i < len; ++i) {
    double a = fabs((double)v[i]);
    if (a > mx) mx = a;
  }
  return mx;
}

static int want_dump(long call) {
  const char *c = getenv("SCS_TRACE_CALLS");
  char buf[256], *tok;
  if (!c || !getenv("SCS_TRACE_DUMP")) return 0;
  strncpy(buf, c, sizeof buf - 1);
  buf[sizeof buf - 1] = 0;
  for (tok = strtok(buf, ","); tok; tok = strtok(NULL, ","))
    if (atol(tok) == call) return 1;
  return 0;
}

static void dump(const char *what, long call, const scs_float *v, long len) {
  char path[512];
  FILE *f;
  snprintf(path, sizeof path, "%s/call%ld_%s.bin", getenv("SCS_TRACE_DUMP"), call, what);
  f = fopen(path, "wb");
  if (f) {
    fwrite(v, sizeof(scs_float), (size_t)len, f);
    fclose(f);
  }
}

ScsLinSysWork *scs_init_lin_sys_work(const ScsMatrix *A, const ScsMatrix *P, const scs_float *diag_r) {
  const char *lf = getenv("SCS_TRACE_FILE");
  g_n = A->n;
  g_m = A->m;
  g_call = 0;
  g_cg_total = 0;
  if (g_log && g_log != stderr) fclose(g_log);
  g_log = lf ? fopen(lf, "w") : NULL;
  if (getenv("SCS_TRACE_DUMP")) {
    char path[512];
    FILE *f;
    snprintf(path, sizeof path, "%s/diag_r_init.bin", getenv("SCS_TRACE_DUMP"));
    f = fopen(path, "wb");
    if (f) {
      fwrite(diag_r, sizeof(scs_float), (size_t)(A->n + A->m), f);
      fclose(f);
    }
    snprintf(path, sizeof path, "%s/A_x.bin", getenv("SCS_TRACE_DUMP"));
    f = fopen(path, "wb");
    if (f) {
      fwrite(A->x, sizeof(scs_float), (size_t)A->p[A->n], f);
      fclose(f);
    }
  }
  return ref_scs_init_lin_sys_work(A, P, diag_r);
}

scs_int scs_solve_lin_sys(ScsLinSysWork *w, scs_float *b, const scs_float *s, scs_float tol) {
  const long call = g_call++;
  const int d = want_dump(call);
  /* the two extra norm passes only run when a log is requested: the timed CPU window pays nothing for them */
  const double nb = g_log ? amax(b, (long)g_n + g_m) : 0.0, ns = g_log ? (s ? amax(s, g_n) : -1.0) : 0.0;
  scs_int rc;
  if (d) {
    dump("b", call, b, (long)g_n + g_m);
    if (s) dump("s", call, s, g_n);
  }
  const long long cg0 = (long long)w->tot_cg_its;
  rc = ref_scs_solve_lin_sys(w, b, s, tol);
  const int cg = (int)((long long)w->tot_cg_its - cg0);
  g_cg_total += cg;
  if (call < TRACE_MAX_CALLS) {
    g_cg_of_call[call] = cg;
    g_warm_of_call[call] = s ? 1 : 0;
  }
  if (d) dump("xy", call, b, (long)g_n + g_m);
  if (g_log) {
    fprintf(g_log, "%ld tol=%.17g nb=%.17g ns=%.17g nout=%.17g x0=%.17g cg=%d\n", call, (double)tol, nb, ns,
            amax(b, (long)g_n + g_m), (double)b[0], cg);
    fflush(g_log);
  }
  return rc;
}

scs_int scs_update_lin_sys_diag_r(ScsLinSysWork *w, const scs_float *new_diag_r) {
  if (g_log) fprintf(g_log, "update_diag_r ry_last=%.17g\n", (double)new_diag_r[g_n + g_m - 1]);
  return ref_scs_update_lin_sys_diag_r(w, new_diag_r);
}

void scs_free_lin_sys_work(ScsLinSysWork *w) {
  ref_scs_free_lin_sys_work(w);
  if (g_log) {
    fclose(g_log);
    g_log = NULL;
  }
}

const char *scs_get_lin_sys_method(void) { return ref_scs_get_lin_sys_method(); }
