// cones.h -- device-resident projection onto K = zero x pos x box x SOC^q x PSD^s
// (reference src/cones.c:1340-1394 `proj_cone`, wrapped by the Moreau identity of
// src/cones.c:1552-1596 `proj_dual_cone`).
#pragma once
#include "common.h"

namespace scsamd {

typedef scs_float real;

struct BigPsd; // psd_big.h

struct ConeDev {
  ~ConeDev();
  ConeDev() = default;
  ConeDev(const ConeDev &) = delete;
  ConeDev &operator=(const ConeDev &) = delete;
  int m = 0;
  hipStream_t stream = nullptr;
  // layout (row offsets into the length-m cone vector)
  int z = 0, l = 0, bsize = 0, box_off = 0;
  // box cone
  DevBuf<real> bl, bu;      // bsize-1 each (already D-normalised, +-inf applied)
  DevBuf<real> box_t;       // [0] Newton warm start (reference c->box_t_warm_start)
  int psd_pipe = 1;         // step of k_psd_jacobi for orders <= 72 (option psd_pipe, read in init): 0 two-phase, 1 look-ahead, 2 signal form
  bool box_multi = false;   // large box: Newton steps as chip-wide launches (k_box_step), else one workgroup (k_box)
  DevBuf<real> box_part, box_ctl;
  // second-order cones
  int n_tiny = 0;           // cones with q <= SOC_TINY_MAX: one lane per cone
  DevBuf<int> tiny_off, tiny_len;
  int n_big = 0, n_tiles = 0; // remaining cones: tiled three-pass reduction
  DevBuf<int> big_off, big_len, big_tile0; // per big cone (+1 sentinel in big_tile0)
  DevBuf<int> tile_cone, tile_off, tile_len;
  DevBuf<real> tile_part;   // per-tile sum of squares of the tail
  DevBuf<real> big_coef;    // per big cone: [head, tail multiplier]
  // PSD cones
  int n_psd = 0, psd_kmax = 0, psd_lds_kmax = 0;
  DevBuf<int> psd_off, psd_k;
  BigPsd *psd_big = nullptr; // blocks of order > PSD_LDS_KMAX: chip-wide Jacobi steps
  DevBuf<real> psd_vprev;   // per block: eigenbasis of the previous projection (warm start of the LDS kernel, every order it handles)
  DevBuf<real> psd_tscratch; // per block: T = A Vp of the warm start when three matrices do not fit LDS (orders 73..92)
  long long psd_calls = 0;  // projections since the last cold start
  void reset_warm_start();
  // exponential (primal, dual) and power cones: 3 rows each, after the PSD blocks
  int ep = 0, ed = 0, psize = 0, exp_off = 0;
  DevBuf<real> pow_a;       // psize power-cone parameters (negative = dual cone)
  DevBuf<int> status;       // [0] = PSD block projections that hit the Jacobi sweep cap since the last take_status()
  PinnedBuf<int> hstatus;   // its host copy (pinned: take_status runs at every residual evaluation)
  int take_status(hipStream_t st);

  // host staging for the B1' boundary
  DevBuf<real> x_stage, s_stage, r_stage;

  void init(const ScsCone *k, int m, const real *D_host, hipStream_t s);
  // cw (device, length m) <- Proj_K(cw), projection under the diag(r_y)^-1 metric
  // (only the box cone looks at r_y; nullptr = Euclidean).
  void proj_primal(real *cw, const real *r_y);
  void proj_exp_pow(real *cw);
  // x (device, length m) <- Proj_{K*}^{R}(x) via Moreau; `scratch` is a length-m
  // device buffer that receives the saved input.
  void proj_dual(real *x, real *scratch, const real *r_y);
  int rows() const { return m; }
};

// validation shared by scs_init and scs_amd_cone_init; returns <0 when the cone
// description is inconsistent with m or uses a cone this backend does not carry.
int validate_cone(const ScsCone *k, int m, bool verbose);
long long cone_total_rows(const ScsCone *k);

} // namespace scsamd
