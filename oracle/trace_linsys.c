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
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "scs.h"

ScsLinSysWork *ref_scs_init_lin_sys_work(const ScsMatrix *A, const ScsMatrix *P, const scs_float *diag_r);
scs_int ref_scs_solve_lin_sys(ScsLinSysWork *w, scs_float *b, const scs_float *s, scs_float tol);
scs_int ref_scs_update_lin_sys_diag_r(ScsLinSysWork *w, const scs_float *new_diag_r);
void ref_scs_free_lin_sys_work(ScsLinSysWork *w);
const char *ref_scs_get_lin_sys_method(void);

static long g_call = 0;
static int g_n = 0, g_m = 0;
static FILE *g_log = NULL;

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
  const double nb = amax(b, (long)g_n + g_m), ns = s ? amax(s, g_n) : -1.0;
  scs_int rc;
  if (d) {
    dump("b", call, b, (long)g_n + g_m);
    if (s) dump("s", call, s, g_n);
  }
  rc = ref_scs_solve_lin_sys(w, b, s, tol);
  if (d) dump("xy", call, b, (long)g_n + g_m);
  if (g_log) {
    fprintf(g_log, "%ld tol=%.17g nb=%.17g ns=%.17g nout=%.17g x0=%.17g\n", call, (double)tol, nb, ns,
            amax(b, (long)g_n + g_m), (double)b[0]);
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
