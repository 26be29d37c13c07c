// spmv.h -- CSR sparse mat-vec for CDNA4, "stream" formulation.
//
// Replaces SCS(accum_by_atrans) (reference linsys/scs_matrix.c:161-186), which
// is a row-parallel CSR product on the transpose, and the cuSPARSE calls of the
// reference's CUDA backend (linsys/gpu/gpu.c:14-28).  Both orientations of A
// are stored as CSR (CSC(A) *is* CSR(A')), so every product here is
// "y_r (op)= sum_k val[k] * x[idx[k]]" over a row.
//
// Why not wave-per-row: rows are short (10 nnz in CSR(A'), Poisson(5) in CSR(A)
// on the headline config), so a 64-lane wave per row idles >80 % of its lanes
// and issues 40-byte loads.  Instead a workgroup owns a contiguous run of rows
// holding <= NNZ_PER_BLOCK entries:
//   phase 1  all 256 lanes stream val/idx with unit stride (fully coalesced,
//            8 independent loads in flight per lane), gather x[idx] (served by
//            L2 / Infinity Cache: x is 8-16 MB), and park the products in LDS;
//   phase 2  one lane per row walks its products in LDS in index order -- the
//            same summation order as the reference's scalar loop -- and applies
//            the fused epilogue.
// A row longer than NNZ_PER_BLOCK gets a workgroup to itself (tree reduction).
// The grid is capped and strides over row-blocks so a fused dot product leaves
// at most SPMV_MAX_GRID partials.
#pragma once
#include "common.h"

namespace scsamd {

typedef scs_float real;

constexpr int NNZ_PER_BLOCK = 2048; // products staged in LDS per row-block (16 KB fp64)
constexpr int ROWS_PER_BLOCK_MAX = 2048;
constexpr int SPMV_MAX_GRID = 4096;
constexpr int SPMV_UNROLL = NNZ_PER_BLOCK / SCSAMD_BLOCK; // 8

struct CsrView {
  int rows, cols, nblk;
  const eoff *ptr;   // rows + 1 : entry positions (64-bit in the DLONG build)
  const int *idx;    // nnz
  const real *val;   // nnz
  const int *rowblk; // nblk + 1 : first row of each row-block
  const eoff *blkptr; // nblk + 1 : ptr[rowblk[b]] (first entry of each row-block: saves a dependent read)
};

// Epilogues (what happens to the row sum `acc`):
enum {
  EPI_PLAIN = 0,  // y[r] = acc
  EPI_DIV = 1,    // y[r] = acc / d[r]                      (z = R_y^-1 A p)
  EPI_GP = 2,     // y[r] = (y0[r] + acc) + d[r] * xin[r];  partial dot xin.y
  EPI_ACC = 3,    // y[r] = y[r] + acc  (sum starts at y[r], like the reference)
  EPI_NEGDIV = 4, // y[r] = (-y[r] + acc) / d[r]            (y = R_y^-1 (A x - r_y))
  EPI_GP3 = 5,    // EPI_GP plus the partial sums of (M r) . y and y . (M y): the three-kernel PCG iteration (linsys.hip, k_cg3_update)
};

struct EpiArgs {
  const real *d;   // divisor (R_y) or multiplier (R_x)
  const real *xin; // EPI_GP: the vector being multiplied (p), length rows
  const real *y0;  // EPI_GP: optional initial value (P p) or nullptr
  real *partial;   // EPI_GP: per-workgroup partial of xin . y, or nullptr
  const real *mv = nullptr;  // EPI_GP3: the Jacobi preconditioner M (length rows)
  const real *rv = nullptr;  // EPI_GP3: the residual r
  real *partial2 = nullptr;  // EPI_GP3: per-workgroup partial of sum M r y
  real *partial3 = nullptr;  // EPI_GP3: per-workgroup partial of sum M y y
};

#ifdef __HIPCC__
template <int EPI>
__device__ __forceinline__ real epi_init(const EpiArgs &e, const real *y, int r) {
  if (EPI == EPI_GP || EPI == EPI_GP3) return e.y0 ? e.y0[r] : (real)0;
  if (EPI == EPI_ACC) return y[r];
  if (EPI == EPI_NEGDIV) return -y[r];
  return (real)0;
}
template <int EPI>
__device__ __forceinline__ real epi_apply(const EpiArgs &e, real *y, int r, real acc, real &dot) {
  real out = acc;
  if (EPI == EPI_DIV || EPI == EPI_NEGDIV) out = acc / e.d[r];
  if (EPI == EPI_GP) {
    const real xr = e.xin[r];
    out = acc + e.d[r] * xr;
    dot += xr * out;
  }
  y[r] = out;
  return out;
}

// EPI_GP3: y[r] = acc + d[r] xin[r] as EPI_GP, and the three dot products the fused update needs:
//   dot += xin y (= p'Gp),  d1 += (M r) y (= z'Gp),  d2 += y (M y)
__device__ __forceinline__ void epi_apply3(const EpiArgs &e, real *y, int r, real acc, real &dot, real &d1, real &d2) {
  const real xr = e.xin[r];
  const real out = acc + e.d[r] * xr;
  const real t = e.mv[r] * out;
  dot += xr * out;
  d1 += e.rv[r] * t;
  d2 += out * t;
  y[r] = out;
}

// epilogue with the row's operands already in registers (same arithmetic as epi_apply)
template <int EPI>
__device__ __forceinline__ void epi_apply_pre(real *y, int r, real acc, real dr, real xr, real &dot) {
  real out = acc;
  if (EPI == EPI_DIV || EPI == EPI_NEGDIV) out = acc / dr;
  if (EPI == EPI_GP) {
    out = acc + dr * xr;
    dot += xr * out;
  }
  y[r] = out;
}

// Small systems are bound by chains of dependent reads (every kernel's first touch of another kernel's output
// comes from the Infinity Cache / HBM, ~1 us each), not by bandwidth: the loads that do not depend on each other
// are therefore issued together -- {skip flag, row-block table entry}, then {entries, the lane's row pointers, its
// epilogue operands}, then the gathers -- three round trips instead of seven.  Same arithmetic, same order.
// the row-block loop of the stream product (shared by csr_stream_kernel and the small-system CG kernel of linsys.hip,
// which gathers from an LDS copy of x): workgroup `b0` of `nb` strides over the row-blocks
template <int EPI>
__device__ __forceinline__ void csr_stream_blocks(const CsrView &A, const real *x, real *y, const EpiArgs &e, real *prod,
                                                  real *red, int b0, int nb, real &dot) {
  const int tid = threadIdx.x;
  int b = b0;
  int r0 = 0, r1 = 0;
  eoff k0 = 0, k1 = 0;
  if (b < A.nblk) {
    r0 = A.rowblk[b];
    r1 = A.rowblk[b + 1];
    k0 = A.blkptr[b];
    k1 = A.blkptr[b + 1];
  }
  while (b < A.nblk) {
    const eoff cnt_all = k1 - k0;
    const int cnt = cnt_all > NNZ_PER_BLOCK ? NNZ_PER_BLOCK + 1 : (int)cnt_all;
    if (cnt > NNZ_PER_BLOCK) {
      // one long row: strided partial sums, workgroup tree reduction
      real acc = 0;
      for (eoff k = k0 + tid; k < k1; k += SCSAMD_BLOCK) acc += A.val[k] * x[A.idx[k]];
      acc = block_sum(acc, red);
      if (tid == 0) {
        acc += epi_init<EPI>(e, y, r0);
        epi_apply<EPI>(e, y, r0, acc, dot);
      }
    } else {
      // phase 1: coalesced stream + gather, products to LDS
      int ii[SPMV_UNROLL];
      real vv[SPMV_UNROLL];
#pragma unroll
      for (int j = 0; j < SPMV_UNROLL; ++j) {
        const int k = tid + j * SCSAMD_BLOCK;
        const bool ok = k < cnt;
        ii[j] = ok ? A.idx[k0 + k] : 0;
        vv[j] = ok ? A.val[k0 + k] : (real)0;
      }
      // the lane's first row: pointers and epilogue operands travel with the entries
      const int rf = r0 + tid;
      const bool has = rf < r1;
      int a0 = 0, z0 = 0;
      real init0 = 0, d0 = 1, x0 = 0;
      if (has) {
        a0 = (int)(A.ptr[rf] - k0);
        z0 = (int)(A.ptr[rf + 1] - k0);
        init0 = epi_init<EPI>(e, y, rf);
        if (EPI == EPI_DIV || EPI == EPI_NEGDIV || EPI == EPI_GP) d0 = e.d[rf];
        if (EPI == EPI_GP) x0 = e.xin[rf];
      }
      real xx[SPMV_UNROLL];
#pragma unroll
      for (int j = 0; j < SPMV_UNROLL; ++j) xx[j] = x[ii[j]];
#pragma unroll
      for (int j = 0; j < SPMV_UNROLL; ++j) {
        const int k = tid + j * SCSAMD_BLOCK;
        if (k < cnt) prod[k] = vv[j] * xx[j];
      }
      __syncthreads();
      // phase 2: one lane per row, sequential (reference order) sum out of LDS
      if (has) {
        real acc = init0;
        for (int k = a0; k < z0; ++k) acc += prod[k];
        epi_apply_pre<EPI>(y, rf, acc, d0, x0, dot);
      }
      for (int r = rf + SCSAMD_BLOCK; r < r1; r += SCSAMD_BLOCK) {
        const int a = (int)(A.ptr[r] - k0), z = (int)(A.ptr[r + 1] - k0);
        real acc = epi_init<EPI>(e, y, r);
        for (int k = a; k < z; ++k) acc += prod[k];
        epi_apply<EPI>(e, y, r, acc, dot);
      }
      __syncthreads();
    }
    b += nb;
    if (b < A.nblk) {
      r0 = A.rowblk[b];
      r1 = A.rowblk[b + 1];
      k0 = A.blkptr[b];
      k1 = A.blkptr[b + 1];
    }
  }
}

template <int EPI>
__global__ __launch_bounds__(SCSAMD_BLOCK) void csr_stream_kernel(CsrView A, const real *__restrict__ x,
                                                                  real *y, EpiArgs e,
                                                                  const int *skip) {
  __shared__ real prod[NNZ_PER_BLOCK];
  __shared__ real red[SCSAMD_BLOCK / SCSAMD_WAVE];
  // the skip flag and the first row-block's table entries are requested together (csr_stream_blocks starts with them)
  const int sk = skip ? *skip : 0;
  if ((int)blockIdx.x < A.nblk) { // in flight together with the flag: csr_stream_blocks then finds the lines in L1
    const int touch = A.rowblk[blockIdx.x + 1] ^ (int)A.blkptr[blockIdx.x + 1];
    asm volatile("" ::"v"(touch));
  }
  if (sk) return;
  real dot = 0;
  csr_stream_blocks<EPI>(A, x, y, e, prod, red, blockIdx.x, gridDim.x, dot);
  if (EPI == EPI_GP && e.partial) {
    dot = block_sum(dot, red);
    if (threadIdx.x == 0) e.partial[blockIdx.x] = dot;
  }
}
#endif // __HIPCC__

// ---- device-resident CSR with its row-block table ---------------------------
struct WaveRowsDev;
struct CsrDev {
  WaveRowsDev *wave = nullptr; // owned; built by LinSys::init when WaveRowsDev::wanted()
  ~CsrDev();
  CsrDev() = default;
  CsrDev(const CsrDev &) = delete;
  int rows = 0, cols = 0, nblk = 0;
  long long nnz = 0;
  DevBuf<eoff> ptr, blkptr;
  DevBuf<int> idx, rowblk;
  DevBuf<real> val;
  // test hook of the DLONG build (SCS_AMD_TEST_OFFSET_BIAS, apply_offset_bias): every stored entry position carries +bias and the
  // index / value arrays are handed to the kernels shifted by -bias, so a kernel that narrows a position to 32 bits anywhere reads
  // the wrong entry -- 64-bit entry positions exercised without a 26 GB matrix
  long long bias = 0;
  CsrView view() const { return CsrView{rows, cols, nblk, ptr.p, idx.p - bias, val.p - bias, rowblk.p, blkptr.p}; }
  int max_grid = SPMV_MAX_GRID; // tests shrink it (SCS_AMD_SPMV_MAX_GRID) to force grid-striding
  int grid() const { return nblk < max_grid ? (nblk > 0 ? nblk : 1) : max_grid; }
  // algorithmic bytes of one product with this matrix:
  // nnz*(sf+si) + (rows+1)*si + cols*sf + rows*sf
  long long algorithmic_bytes() const {
    return nnz * (long long)(sizeof(real) + sizeof(int)) + (long long)(rows + 1) * sizeof(int) +
           (long long)cols * sizeof(real) + (long long)rows * sizeof(real);
  }
  // host CSR arrays -> device, plus the row-block table
  void upload(int rows_, int cols_, const eoff *hptr, const int *hidx, const real *hval,
              hipStream_t s) {
    rows = rows_;
    cols = cols_;
    nnz = hptr[rows_];
    ptr.alloc((size_t)rows + 1);
    idx.alloc((size_t)nnz);
    val.alloc((size_t)nnz);
    ptr.upload(hptr, (size_t)rows + 1, s);
    if (nnz) {
      idx.upload(hidx, (size_t)nnz, s);
      val.upload(hval, (size_t)nnz, s);
    }
    build_row_blocks(hptr, s);
  }
  // (round 5) CSR arrays that are ALREADY in HBM (left there by the device equilibration): ownership moves here, nothing is copied;
  // hptr = the host copy of the row pointers (the row-block table is cut from it)
  void adopt(int rows_, int cols_, const eoff *hptr, DevBuf<eoff> &dptr, DevBuf<int> &didx, DevBuf<real> &dval, hipStream_t s) {
    rows = rows_;
    cols = cols_;
    nnz = hptr[rows_];
    ptr.take(dptr);
    idx.take(didx);
    val.take(dval);
    build_row_blocks(hptr, s);
  }
  void build_row_blocks(const eoff *hptr, hipStream_t s) {
    if (const char *e = opt_get("spmv_max_grid")) {
      int g = atoi(e);
      if (g >= 1 && g <= SPMV_MAX_GRID) max_grid = g;
    }
    std::vector<int> rb;
    rb.push_back(0);
    int r = 0;
    while (r < rows) {
      int start = r;
      long long acc = 0;
      while (r < rows && (r - start) < ROWS_PER_BLOCK_MAX) {
        long long rn = (long long)hptr[r + 1] - hptr[r];
        if (acc + rn > NNZ_PER_BLOCK) break;
        acc += rn;
        ++r;
      }
      if (r == start) ++r; // a single row longer than the LDS tile: own block
      rb.push_back(r);
    }
    nblk = (int)rb.size() - 1;
    rowblk.alloc(rb.size());
    rowblk.upload(rb.data(), rb.size(), s);
    std::vector<eoff> bp(rb.size());
    for (size_t i = 0; i < rb.size(); ++i) bp[i] = hptr[rb[i]];
    blkptr.alloc(bp.size());
    blkptr.upload(bp.data(), bp.size(), s);
    HIP_CHECK(hipStreamSynchronize(s)); // rb is a local
  }
};

} // namespace scsamd
