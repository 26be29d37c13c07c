/*
 * scs_amd.h -- C ABI of the MI355X-native ADMM hot path for SCS.
 *
 * One shared library (scs_amd/lib/libscsamd.so, fp64; libscsamd_f32.so when
 * built with -DSFLOAT) exports three concentric boundaries.  Every entry point
 * below is bound exactly as the reference (cvxgrp/scs v3.2.11) binds the one it
 * replaces; citations are <file>:<line> relative to the reference tree.
 *
 *   B2  whole-solve API             reference include/scs.h:271-338
 *         scs_init  scs_update  scs_solve  scs_finish  scs
 *         scs_set_default_settings  scs_version
 *       The iterate vectors live in HBM for the whole scs_solve; the host only
 *       sees scalars (residual norms, tau, CG counts) and, every
 *       acceleration_interval iterations, the vector v for Anderson
 *       acceleration (host side by design).
 *
 *   B1  linear-system plugin        reference include/linsys.h:25-71
 *         scs_init_lin_sys_work  scs_solve_lin_sys  scs_update_lin_sys_diag_r
 *         scs_free_lin_sys_work  scs_get_lin_sys_method
 *       Host pointers in/out, identical contract to linsys/cpu/indirect
 *       (private.c:221-349): `b` is overwritten by [x; y], `s` may be NULL,
 *       0 == success, NULL on init failure.  These five symbols are also
 *       exported by scs_amd/lib/libscsamd_linsys.so (besides scs_amd_* helpers,
 *       nothing else in it), which is what a reference build links instead of
 *       linsys/<backend>/private.o.  Beside them, on DEVICE pointers, the pieces
 *       of `mat_vec` (private.c:106-119) for a caller that splits one system by
 *       rows of A across GPUs: scs_amd_linsys_mat_vec_dev / _mul_a_dev /
 *       _mul_at_dev / _sync (declared with the instrumentation below).
 *
 *   B1' cone projection             reference include/cones.h:80-90
 *         scs_amd_cone_init  scs_amd_cone_proj_dual  scs_amd_cone_finish
 *       (the reference has no plugin API for cones; INTEGRATION.md shows the
 *       three-line shim that maps _scs_proj_dual_cone onto these).
 *
 * Plain C types only: pointers, sizes, the structs below.  No HIP or torch type
 * crosses this boundary.  Struct layouts are ABI facts of the reference and are
 * restated field-for-field (include/scs.h:47-244, include/aa_stats.h:21-42,
 * include/scs_types.h:13-32).  scs_int is 32-bit by default and 64-bit when the
 * header is compiled with -DDLONG, exactly like the reference's own switch; the
 * matching library is libscsamd_dlong.so (same entry points, 64-bit indices and
 * sizes at this boundary; on the device row / column indices stay 32-bit --
 * m + n + 1 < 2^31, refused loudly otherwise -- and entry positions are 64-bit
 * in that build, so nnz(A) >= 2^31 is accepted).
 */
#ifndef SCS_AMD_H
#define SCS_AMD_H

#ifdef __cplusplus
extern "C" {
#endif

/* ---- primitive types (reference include/scs_types.h:13-32) ------------- */
#ifdef DLONG
typedef long long scs_int; /* reference -DDLONG, include/scs_types.h:13-20 */
#else
typedef int scs_int;
#endif
#ifndef SFLOAT
typedef double scs_float;
#else
typedef float scs_float;
#endif

/* ---- exit flags (reference include/scs.h:33-42) ------------------------ */
#define SCS_INFEASIBLE_INACCURATE (-7)
#define SCS_UNBOUNDED_INACCURATE (-6)
#define SCS_SIGINT (-5)
#define SCS_FAILED (-4)
#define SCS_INDETERMINATE (-3)
#define SCS_INFEASIBLE (-2)
#define SCS_UNBOUNDED (-1)
#define SCS_UNFINISHED (0)
#define SCS_SOLVED (1)
#define SCS_SOLVED_INACCURATE (2)

/* ---- opaque workspaces -------------------------------------------------- */
typedef struct SCS_WORK ScsWork;                /* B2 */
typedef struct SCS_LIN_SYS_WORK ScsLinSysWork;  /* B1 */
typedef struct SCS_AMD_CONE_WORK ScsAmdConeWork; /* B1' */

/* ---- data structs ------------------------------------------------------- */
/* CSC, zero based (reference include/scs.h:47-58). */
typedef struct {
  scs_float *x; /* values, nnz                */
  scs_int *i;   /* row indices, nnz           */
  scs_int *p;   /* column pointers, n + 1     */
  scs_int m;    /* rows                       */
  scs_int n;    /* columns                    */
} ScsMatrix;

/* reference include/scs.h:61-101 */
typedef struct {
  scs_int normalize;
  scs_float scale;
  scs_int adaptive_scale;
  scs_float rho_x;
  scs_int max_iters;
  scs_float eps_abs;
  scs_float eps_rel;
  scs_float eps_infeas;
  scs_float alpha;
  scs_float time_limit_secs;
  scs_int verbose;
  scs_int warm_start;
  scs_int acceleration_lookback;
  scs_int acceleration_interval;
  scs_int acceleration_type_1;
  scs_float acceleration_regularization;
  scs_float acceleration_relaxation;
  const char *write_data_filename;
  const char *log_csv_filename;
} ScsSettings;

/* reference include/scs.h:104-119 */
typedef struct {
  scs_int m;
  scs_int n;
  ScsMatrix *A; /* m x n                               */
  ScsMatrix *P; /* n x n upper triangle, or NULL       */
  scs_float *b; /* m                                   */
  scs_float *c; /* n                                   */
} ScsData;

/* reference include/scs.h:122-172 (spectral-cone members are compiled out of
 * the default reference build, scs.mk:195, and are not part of this ABI). */
typedef struct {
  scs_int z;      /* zero cone rows                                  */
  scs_int l;      /* nonnegative orthant rows                        */
  scs_float *bu;  /* box upper bounds, bsize - 1                     */
  scs_float *bl;  /* box lower bounds, bsize - 1                     */
  scs_int bsize;  /* box cone length including t                     */
  scs_int *q;     /* second-order cone sizes                         */
  scs_int qsize;
  scs_int *s;     /* PSD cone matrix dimensions                      */
  scs_int ssize;
  scs_int *cs;    /* complex PSD cone matrix dimensions              */
  scs_int cssize;
  scs_int ep;     /* primal exponential cones (3 rows each)          */
  scs_int ed;     /* dual exponential cones                          */
  scs_float *p;   /* power cone parameters in [-1,1], <0 = dual cone  */
  scs_int psize;
} ScsCone;

/* reference include/scs.h:180-187 */
typedef struct {
  scs_float *x;
  scs_float *y;
  scs_float *s;
} ScsSolution;

/* reference include/aa_stats.h:21-42 */
typedef struct {
  scs_int iter;
  scs_int n_accept;
  scs_int n_reject_lapack;
  scs_int n_reject_rank0;
  scs_int n_reject_nonfinite;
  scs_int n_reject_weight_cap;
  scs_int n_safeguard_reject;
  scs_int last_rank;
  scs_float last_aa_norm;
  scs_float last_regularization;
} AaStats;

/* reference include/scs.h:190-244 */
typedef struct {
  scs_int iter;
  char status[128];
  char lin_sys_solver[128];
  scs_int status_val;
  scs_int scale_updates;
  scs_float pobj;
  scs_float dobj;
  scs_float res_pri;
  scs_float res_dual;
  scs_float gap;
  scs_float res_infeas;
  scs_float res_unbdd_a;
  scs_float res_unbdd_p;
  scs_float setup_time; /* ms */
  scs_float solve_time; /* ms */
  scs_float scale;
  scs_float comp_slack;
  scs_int rejected_accel_steps;
  scs_int accepted_accel_steps;
  AaStats aa_stats;
  scs_float lin_sys_time; /* ms */
  scs_float cone_time;    /* ms */
  scs_float accel_time;   /* ms */
} ScsInfo;

/* ======================= B2: whole-solve API ============================== */
/* replaces src/scs.c:1245 (scs_init)   -- validates, deep-copies, equilibrates
 * on the host, then uploads A (both orientations), b, c, D, E once. */
ScsWork *scs_init(const ScsData *d, const ScsCone *k, const ScsSettings *stgs);
/* replaces src/scs.c:1287 */
scs_int scs_update(ScsWork *w, scs_float *b, scs_float *c);
/* replaces src/scs.c:1327 -- the device-resident ADMM loop */
scs_int scs_solve(ScsWork *w, ScsSolution *sol, ScsInfo *info,
                  scs_int warm_start);
/* replaces src/scs.c:1486 */
void scs_finish(ScsWork *w);
/* replaces src/scs.c:1538 */
scs_int scs(const ScsData *d, const ScsCone *k, const ScsSettings *stgs,
            ScsSolution *sol, ScsInfo *info);
/* replaces src/util.c:158 */
void scs_set_default_settings(ScsSettings *stgs);
/* replaces src/scs_version.c */
const char *scs_version(void);

/* ======================= B1: linear-system plugin ========================= */
/* replaces linsys/cpu/indirect/private.c:225 (callers src/scs.c:1092) */
ScsLinSysWork *scs_init_lin_sys_work(const ScsMatrix *A, const ScsMatrix *P,
                                     const scs_float *diag_r);
/* replaces private.c:284 (callers src/scs.c:763,1127) */
scs_int scs_solve_lin_sys(ScsLinSysWork *w, scs_float *b, const scs_float *s,
                          scs_float tol);
/* replaces private.c:327 (caller src/scs.c:1220) */
scs_int scs_update_lin_sys_diag_r(ScsLinSysWork *w,
                                  const scs_float *new_diag_r);
/* replaces private.c:333 (caller src/scs.c:1493) */
void scs_free_lin_sys_work(ScsLinSysWork *w);
/* replaces private.c:221 (callers src/scs.c:127,1346) */
const char *scs_get_lin_sys_method(void);

/* ======================= B1': cone projection ============================= */
/* replaces src/cones.c:1498 (_scs_init_cone).  `k` is read, never mutated: the
 * box bounds are copied and the lazy D-normalisation of cones.c:1161-1177 is
 * applied to the copy when D != NULL.  D may be NULL (un-normalised cones). */
ScsAmdConeWork *scs_amd_cone_init(const ScsCone *k, scs_int m,
                                  const scs_float *D);
/* replaces src/cones.c:1552 (_scs_proj_dual_cone): x (host, length m) is
 * overwritten by its projection onto the DUAL cone under the r_y metric;
 * r_y may be NULL (Euclidean).  Returns <0 on failure, like the reference. */
scs_int scs_amd_cone_proj_dual(ScsAmdConeWork *c, scs_float *x,
                               const scs_float *r_y);
/* replaces src/cones.c:338 (_scs_finish_cone) */
void scs_amd_cone_finish(ScsAmdConeWork *c);

/* ======================= instrumentation ================================== */
/* Not in the reference (its `tot_cg_its`, private.h:28, is never surfaced).
 * Counters accumulate per workspace since init / since the last reset.  Kernel
 * times come from HIP events recorded on the library's own stream. */
typedef struct {
  long long cg_iters;        /* PCG iterations, all solves                   */
  long long lin_sys_solves;  /* scs_solve_lin_sys calls                      */
  long long mat_vecs;        /* applications of R_x + P + A' R_y^-1 A        */
  long long spmv_launches;   /* CSR SpMV kernel launches (both orientations) */
  double spmv_ms;            /* summed HIP-event time of those launches      */
  double cg_ms;              /* summed HIP-event time of whole PCG solves    */
  double cone_ms;            /* summed HIP-event time of cone projections    */
  long long cone_projs;
  long long nnz;             /* nnz(A)                                       */
  long long spmv_bytes;      /* algorithmic bytes of ONE mat_vec (2 SpMV)    */
  long long psd_unconverged; /* PSD block projections whose Jacobi eigensolve hit the sweep cap (the
                              * reference's LAPACK info > 0 case: reported, not fatal, src/cones.c:1031) */
} ScsAmdStats;

void scs_amd_linsys_get_stats(const ScsLinSysWork *w, ScsAmdStats *out);
void scs_amd_linsys_set_profiling(ScsLinSysWork *w, scs_int on);
/* Pieces of the operator of linsys/cpu/indirect/private.c:106-119 (`mat_vec`) on DEVICE pointers, for a caller that splits
 * ONE linear system by rows of A across GPUs (scs_amd/shard.py): the workspace is created by
 * scs_init_lin_sys_work on a row slab A_r (m_r x n) with diag_r = [R_x / ranks ; R_y of the slab], so that the sum over
 * ranks of mat_vec is R_x x + A' R_y^-1 A x.  Work is enqueued on the workspace's stream; _sync waits for it.
 *   mat_vec_dev   y(n)   = diag_r[0..n) .* x + A_r' R_r^-1 A_r x
 *   mul_a_dev     y(m_r) = A_r x           (src/scs.c:559 uses the same product for the residuals)
 *   mul_at_dev    x(n)   = A_r' y
 * CONTRACT (the entries take no lengths and no stream):
 *   - buffers: x / y are device pointers on the workspace's device (the one selected by scs_amd_set_device when the workspace
 *     was created) holding at least n resp. m_r scs_float of THIS library's precision; they are not validated;
 *   - ordering: the work is enqueued on the workspace's PRIVATE non-blocking stream, which has no implicit ordering with any
 *     stream of the caller (e.g. torch's current stream).  The caller must (1) make its writes to the input visible before the
 *     call -- synchronise its own stream (or the device) first -- and (2) call scs_amd_linsys_sync(w) before it reads the
 *     output or reuses the input; entries issued back to back on one workspace run in issue order;
 *   - threads: one host thread per workspace at a time (as for the five plugin functions, include/linsys.h).            */
scs_int scs_amd_linsys_mat_vec_dev(ScsLinSysWork *w, const scs_float *x_dev, scs_float *y_dev);
scs_int scs_amd_linsys_mul_a_dev(ScsLinSysWork *w, const scs_float *x_dev, scs_float *y_dev);
scs_int scs_amd_linsys_mul_at_dev(ScsLinSysWork *w, const scs_float *y_dev, scs_float *x_dev);
scs_int scs_amd_linsys_sync(ScsLinSysWork *w);
void scs_amd_get_stats(const ScsWork *w, ScsAmdStats *out);
void scs_amd_set_profiling(ScsWork *w, scs_int on);
/* scs_solve split in three so a harness can time / inspect an exact range of ADMM
 * iterations (bench.py's timed region, trajectory tests):
 *   scs_amd_solve_begin(w, sol_or_NULL, warm)   == everything scs_solve does before
 *                                                  the loop (src/scs.c:1341-1354)
 *   scs_amd_solve_steps(w, k)                   runs up to k more iterations, returns
 *                                                  the iteration counter, stream idle
 *   scs_amd_solve_end(w, sol, info)             == finalize + timings (:1457-1484)   */
scs_int scs_amd_solve_begin(ScsWork *w, const ScsSolution *sol, scs_int warm_start);
scs_int scs_amd_solve_steps(ScsWork *w, scs_int steps);
scs_int scs_amd_solve_converged(const ScsWork *w);
scs_int scs_amd_solve_end(ScsWork *w, ScsSolution *sol, ScsInfo *info);
/* test hook: every per-iteration linear solve uses this tolerance instead of the
 * schedule of src/scs.c:745-762 (0 restores the schedule) */
void scs_amd_set_cg_tol_override(ScsWork *w, double tol);
/* What scs_init decided about its internal numbering (scs_amd/csrc/reorder.h: variables, the rows of the zero / nonnegative
 * cones and -- round 6 -- the rows behind the first one of a second-order cone may be renumbered so that the gathers of the CSR products of linsys/scs_matrix.c:161-186 share cache lines; callers never see
 * it -- b, c, warm starts and the returned (x, y, s) are mapped at this boundary).  out[0] = 1 if renumbered, out[1], out[2] =
 * distinct 128-byte lines per gathered entry of the A / A' product as given, out[3], out[4] = after, out[5] = seconds spent. */
void scs_amd_get_reorder_info(const ScsWork *w, double *out);
/* how scs_init laid out A (out[0..2]) and A' (out[3..5]): wave-owned-rows layout built (0 / 1), built on the device (0 / 1),
 * distinct 128-byte lines per gathered entry (scs_amd/csrc/spmv_wave.h, spmv_wave_build.h) */
void scs_amd_get_layout_info(const ScsWork *w, double *out);
/* the SpMV kernel scs_init chose for A (which = 0) / A' (which = 1): the template's name as rocprofv3 lists it, e.g.
 * "csr_wave_lockstep_kernel<EPI,16,4>", "csr_wave_kernel<EPI,0>", "csr_stream_kernel<EPI>".  Returns the length needed. */
scs_int scs_amd_get_spmv_kernel_name(const ScsWork *w, scs_int which, char *buf, scs_int cap);
/* Test hook, host code only: the renumbering decision scs_init would take for this matrix and cone (no device needed).
 * col_new2old (n) and row_new2old (m) receive new index -> caller's index (identity when nothing is kept); info (6 doubles, may be
 * NULL) as scs_amd_get_reorder_info.  Returns 1 if a renumbering is kept, 0 if not, < 0 on error. */
scs_int scs_amd_plan_reorder(const ScsMatrix *A, const ScsCone *k, scs_int *col_new2old, scs_int *row_new2old, double *info);
/* measurement hook: recompute the residuals after every ADMM iteration, where the reference does when
 * `log_csv_filename` is set (src/scs.c:1449-1454).  Those norms feed the next iteration's CG tolerance
 * (src/scs.c:745-762): a logged reference run follows a tighter schedule than an unlogged one, and this puts
 * the solve on that schedule without writing a log (bench.py times the CPU window's schedule with it). */
void scs_amd_set_residuals_every_iter(ScsWork *w, scs_int on);
/* ---- B1', drop-in form: the reference's internal cone interface -------------------
 * The nine symbols of include/cones.h:80-90 (prefix `_scs_` = glbopts.h's SCS(x)), exported
 * by libscsamd_cones.so so that a reference build links it IN PLACE OF src/cones.o (the
 * reference has no plugin API for cones).  Same argument meaning and return conventions:
 * init returns NULL on failure, proj_dual_cone returns <0 on failure (the reference then
 * aborts with SCS_FAILED, src/scs.c:1389), deep_copy_cone returns 1 on success,
 * get_cone_header returns a malloc'd string the caller frees.  Vectors are HOST pointers
 * (x of length m is projected in place).  `struct SCS_CONE_WORK` is opaque here; the
 * reference only looks inside it under USE_SPECTRAL_CONES, which this backend does not
 * carry.  The device side is created at the first projection (that is when the caller says,
 * through `scal`, whether box bounds are normalised by D: src/cones.c:1557-1565). */
typedef struct SCS_CONE_WORK ScsConeWork;
typedef struct { /* reference include/scs_work.h:24-29 */
  scs_float *D, *E;
  scs_int m, n;
  scs_float primal_scale, dual_scale;
} ScsScaling;
ScsConeWork *_scs_init_cone(ScsCone *k, scs_int m);                      /* src/cones.c:1498 */
scs_int _scs_proj_dual_cone(scs_float *x, ScsConeWork *c, const ScsScaling *scal,
                            scs_float *r_y);                             /* src/cones.c:1552 */
void _scs_finish_cone(ScsConeWork *c);                                   /* src/cones.c:284  */
void _scs_set_r_y(const ScsConeWork *c, scs_float scale, scs_float *r_y);/* src/cones.c:349  */
void _scs_enforce_cone_boundaries(const ScsConeWork *c, scs_float *vec,
                                  scs_float (*f)(const scs_float *, scs_int)); /* :366 */
scs_int _scs_validate_cones(const ScsData *d, const ScsCone *k);         /* src/cones.c:583  */
char *_scs_get_cone_header(const ScsCone *k);                            /* src/cones.c:565  */
scs_int _scs_deep_copy_cone(ScsCone *dest, const ScsCone *src);          /* src/cones.c:154  */
void _scs_free_cone(ScsCone *k);                                         /* src/cones.c:122  */

/* Test hook: the data equilibration scs_init performs (linsys/scs_matrix.c:433-496, 25 Ruiz
 * + 1 L2 pass) on caller-owned CSC arrays.  A->x (and P->x, P may be NULL) are overwritten
 * with the equilibrated values, D (m) and E (n) receive the scalings.  where = 0: host
 * code, 1: the device kernels (bit-identical by construction).  0 on success. */
scs_int scs_amd_equilibrate(ScsMatrix *A, ScsMatrix *P, const ScsCone *k, scs_float *D, scs_float *E,
                            scs_int where);
/* Problem files in the reference's binary layout (src/rw.c:574-705): replaces
 * _scs_write_data / _scs_read_data; a file written by either side is read by the other.
 * `write_data_filename` in ScsSettings makes scs_init write one (src/scs.c:1272-1275).
 * scs_amd_read_data allocates with malloc; release with scs_amd_free_data. */
scs_int scs_amd_write_data(const ScsData *d, const ScsCone *k, const ScsSettings *stgs, const char *filename);
scs_int scs_amd_read_data(const char *filename, ScsData **d, ScsCone **k, ScsSettings **stgs);
void scs_amd_free_data(ScsData *d, ScsCone *k, ScsSettings *stgs);
/* Host-side Anderson acceleration, as used inside scs_solve; same contract as the
 * reference's aa_init / aa_apply / aa_safeguard / aa_reset / aa_finish
 * (include/aa.h:66-143).  Pure host code (exposed so it can be tested without a GPU). */
void *scs_amd_aa_init(scs_int dim, scs_int mem, scs_int min_len, scs_int type1,
                      scs_float regularization, scs_float relaxation,
                      scs_float safeguard_factor, scs_float max_weight_norm,
                      scs_int ir_max_steps);
scs_float scs_amd_aa_apply(scs_float *f, const scs_float *x, void *a);
scs_int scs_amd_aa_safeguard(scs_float *f_new, scs_float *x_new, void *a);
void scs_amd_aa_reset(void *a);
void scs_amd_aa_finish(void *a);
void scs_amd_aa_get_stats(const void *a, AaStats *out);
/* Device-resident Anderson acceleration (what scs_solve uses once n+m+1 >= 32768; env
 * SCS_AMD_AA=host|dev forces either): same contract again, replacing aa_init / aa_apply /
 * aa_safeguard / aa_reset / aa_finish of include/aa.h:66-143.  These wrappers take HOST
 * pointers and stage them through HBM so the device path can be pinned against the
 * reference's AA on identical sequences; inside scs_solve the iterates never leave HBM.
 * init returns NULL on bad parameters or any HIP failure; apply returns NaN on a HIP failure. */
void *scs_amd_aa_dev_init(scs_int dim, scs_int mem, scs_int min_len, scs_int type1,
                          scs_float regularization, scs_float relaxation,
                          scs_float safeguard_factor, scs_float max_weight_norm,
                          scs_int ir_max_steps);
scs_float scs_amd_aa_dev_apply(scs_float *f, const scs_float *x, void *a);
scs_int scs_amd_aa_dev_safeguard(scs_float *f_new, scs_float *x_new, void *a);
void scs_amd_aa_dev_reset(void *a);
void scs_amd_aa_dev_finish(void *a);
void scs_amd_aa_dev_get_stats(const void *a, AaStats *out);
/* ---- ONE linear system split by rows of A across GPUs, native form -------------------------------------------
 * The operator of linsys/cpu/indirect/private.c:106-119 is a sum over row slabs: G = R_x + A' R_y^-1 A = sum_r (R_x / N + A_r' R_r^-1 A_r).
 * Rank r creates a workspace on ITS slab (rows [r0, r1) of A, all n columns, CSC) with diag_r_local = [R_x / N (n) ; R_y of the
 * slab (m_r)]; the PCG of private.c:133-217 then runs device-controlled on every rank with ONE all-reduce of an n-vector per
 * iteration, enqueued on the solver's stream (RCCL, opened with dlopen at the first call: no link-time dependency), and the O(n)
 * part replicated.  All calls on a group of workspaces are collective (every rank calls them in the same order).
 *   scs_amd_shard_unique_id     rank 0 fills 128 opaque bytes; the launcher hands them to every rank (file, socket, ...)
 *   scs_amd_shard_init_rccl     one process per GPU (device = scs_amd_set_device); NULL on failure
 *   scs_amd_shard_solve         b_local = [r_x (n, the same on every rank) ; r_y of the slab] -> [x ; y of the slab], in place;
 *                               s = warm start (n) or NULL; tolerance / return value as scs_solve_lin_sys
 *   scs_amd_shard_group_create / _init_threads / _group_free: a TEST DOUBLE of the collective -- the ranks are host threads of one
 *                               process sharing one GPU -- so that the N = 2 algebra runs on a single-GPU box (RCCL refuses two
 *                               ranks on one device); at most 8 ranks
 *   scs_amd_shard_get_stats     out[0] PCG iterations, out[1] all-reduces enqueued, out[2] all-reduces timed, out[3] their mean us
 *                               (HIP events on the solver's stream; scs_amd_shard_set_profiling(h, 1) switches the sampling on),
 *                               out[4] solves                                                                                     */
typedef struct SCS_AMD_SHARD ScsAmdShard;
scs_int scs_amd_shard_unique_id(char *out128);
ScsAmdShard *scs_amd_shard_init_rccl(const ScsMatrix *A_slab, const scs_float *diag_r_local, scs_int world, scs_int rank, const char *id128);
void *scs_amd_shard_group_create(scs_int world);
void scs_amd_shard_group_free(void *group);
ScsAmdShard *scs_amd_shard_init_threads(const ScsMatrix *A_slab, const scs_float *diag_r_local, void *group, scs_int rank);
scs_int scs_amd_shard_solve(ScsAmdShard *h, scs_float *b_local, const scs_float *s, scs_float tol);
scs_int scs_amd_shard_update_diag_r(ScsAmdShard *h, const scs_float *diag_r_local);
void scs_amd_shard_get_stats(ScsAmdShard *h, double *out);
void scs_amd_shard_set_profiling(ScsAmdShard *h, scs_int on);
void scs_amd_shard_free(ScsAmdShard *h);
/* Test hook for the failure convention (src/scs.c:361-371, :1381-1384, include/linsys.h:25-71): the k-th HIP runtime call
 * the library checks from now on is reported as failed although it succeeded (k <= 0 disarms; SCS_AMD_FAIL_AT=k in the
 * environment arms it at load).  What must follow: scs_init / scs_init_lin_sys_work return NULL with nothing leaked,
 * scs_solve returns SCS_FAILED with a NaN-filled solution and the SIGINT handler restored, scs_solve_lin_sys returns
 * non-zero, and the library stays usable.  Returns the countdown that was armed before the call. */
long long scs_amd_test_fail_at(long long k);
/* ---- options (scs_amd/csrc/options.h holds the table; INTEGRATION.md section 5 prints it) ---------------------------------
 * The reference steers everything through ScsSettings (include/scs.h:61-101) and reads no environment variable.  This
 * library's own switches -- which implementation (host / device) of a step, which kernel schedule -- go through ONE entry:
 *   scs_amd_set_option("reorder", "0")   process-wide; takes effect for workspaces created AFTERWARDS (scs_init,
 *                                        scs_init_lin_sys_work, _scs_init_cone); value NULL = back to the default.
 *                                        Returns 0, or -1 for a key that is not in the table.
 *   scs_amd_get_option("reorder")        the value in force, NULL at the default (or for an unknown key)
 *   scs_amd_list_options(buf, cap)       the table as text, one row per line (key, class, numerics, values, meaning)
 * Environment fallback SCS_AMD_<KEY>: honoured for rows of class `supported` and `diag` only; rows of class `ab` (measurement
 * variants) and `test` (test hooks) are reachable from the environment only when SCS_AMD_ALLOW_ENV_HOOKS=1 is also set. */
scs_int scs_amd_set_option(const char *key, const char *value);
const char *scs_amd_get_option(const char *key);
scs_int scs_amd_list_options(char *buf, scs_int cap);
/* free device memory (bytes) on the selected device after a device-wide synchronise, < 0 on failure */
long long scs_amd_device_free_bytes(void);
/* number of visible HIP devices, or <0 with no usable runtime (never throws) */
scs_int scs_amd_device_count(void);
/* select the device used by subsequently created workspaces (default 0) */
scs_int scs_amd_set_device(scs_int dev);

#ifdef __cplusplus
}
#endif
#endif /* SCS_AMD_H */
