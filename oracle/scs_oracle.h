/* oracle/scs_oracle.h -- TEST INFRASTRUCTURE ONLY (see scs_oracle.c). */
#ifndef SCS_ORACLE_H
#define SCS_ORACLE_H
#ifdef __cplusplus
extern "C" {
#endif
typedef struct OrLinSys OrLinSys;
typedef struct OrCone OrCone;
typedef struct {
  int normalize, adaptive_scale, max_iters;
  double scale, rho_x, eps_abs, eps_rel, eps_infeas, alpha;
  double cg_tol_override; /* > 0: every per-iteration solve uses this tolerance */
} OrSettings;
typedef struct {
  int iter, status_val, scale_updates;
  double pobj, dobj, res_pri, res_dual, gap, scale, cg_its;
} OrInfo;
void or_accum_by_atrans(int n, const int *Ap, const int *Ai, const double *Ax, const double *x, double *y);
void or_accum_by_a(int n, const int *Ap, const int *Ai, const double *Ax, const double *x, double *y);
OrLinSys *or_linsys_init(int m, int n, const int *Ap, const int *Ai, const double *Ax, const double *diag_r);
void or_linsys_update_diag_r(OrLinSys *w, const double *diag_r);
int or_linsys_solve(OrLinSys *w, double *b, const double *s, double tol);
long or_linsys_tot_cg_its(const OrLinSys *w);
void or_linsys_free(OrLinSys *w);
OrCone *or_cone_init(int m, int z, int l, int bsize, const double *bl, const double *bu, int qsize, const int *q,
                     int ssize, const int *s, const double *D);
void or_cone_set_extra(OrCone *c, int cssize, const int *cs, int ep, int ed, int psize, const double *pw);
void or_cone_proj_dual(OrCone *c, double *x, const double *r_y);
void or_cone_free(OrCone *c);
int or_solve(int m, int n, const int *Ap, const int *Ai, const double *Ax, const double *b, const double *c, int z,
             int l, int bsize, const double *bl, const double *bu, int qsize, const int *q, int ssize, const int *s,
             const OrSettings *st, double *x, double *y, double *s_out, OrInfo *info);
int or_solve_ext(int m, int n, const int *Ap, const int *Ai, const double *Ax, const double *b, const double *c,
                 int z, int l, int bsize, const double *bl, const double *bu, int qsize, const int *q, int ssize,
                 const int *s, int cssize, const int *cs, int ep, int ed, int psize, const double *pw,
                 const OrSettings *st, double *x, double *y, double *s_out, OrInfo *info);
#ifdef __cplusplus
}
#endif
#endif
