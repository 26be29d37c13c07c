/*
 * oracle/blas_shim.h -- TEST INFRASTRUCTURE ONLY (not part of the product).
 *
 * The image has no system BLAS/LAPACK; scipy ships an LP64 OpenBLAS whose
 * Fortran symbols carry a `scipy_` prefix.  The reference mangles names as
 * d<name>_ (include/scs_blas.h:41-50); this header, force-included when the
 * reference sources are compiled into oracle/_ref/, redirects each mangled
 * name to the scipy-prefixed symbol.  Written for this repo; no reference code.
 */
#ifndef SCS_AMD_ORACLE_BLAS_SHIM_H
#define SCS_AMD_ORACLE_BLAS_SHIM_H
#define SHIM2(p, n) p##n##_
/* double */
#define daxpy_ scipy_daxpy_
#define ddot_ scipy_ddot_
#define dgemv_ scipy_dgemv_
#define dgeqp3_ scipy_dgeqp3_
#define dgesv_ scipy_dgesv_
#define dgesvd_ scipy_dgesvd_
#define dgetrs_ scipy_dgetrs_
#define dlange_ scipy_dlange_
#define dnrm2_ scipy_dnrm2_
#define dormqr_ scipy_dormqr_
#define dscal_ scipy_dscal_
#define dsyevr_ scipy_dsyevr_
#define dsyrk_ scipy_dsyrk_
#define dtrmv_ scipy_dtrmv_
#define dtrsv_ scipy_dtrsv_
#define idamax_ scipy_idamax_
#define idlange_ scipy_idlange_
#define zheevr_ scipy_zheevr_
#define zherk_ scipy_zherk_
#define zscal_ scipy_zscal_
/* single (SFLOAT=1 build) */
#define saxpy_ scipy_saxpy_
#define sdot_ scipy_sdot_
#define sgemv_ scipy_sgemv_
#define sgeqp3_ scipy_sgeqp3_
#define sgesv_ scipy_sgesv_
#define sgesvd_ scipy_sgesvd_
#define sgetrs_ scipy_sgetrs_
#define slange_ scipy_slange_
#define snrm2_ scipy_snrm2_
#define sormqr_ scipy_sormqr_
#define sscal_ scipy_sscal_
#define ssyevr_ scipy_ssyevr_
#define ssyrk_ scipy_ssyrk_
#define strmv_ scipy_strmv_
#define strsv_ scipy_strsv_
#define isamax_ scipy_isamax_
#define islange_ scipy_islange_
#define cheevr_ scipy_cheevr_
#define cherk_ scipy_cherk_
#define cscal_ scipy_cscal_
#endif
