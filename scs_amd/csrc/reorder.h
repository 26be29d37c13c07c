// reorder.h -- locality by construction: a cheap, optional renumbering of variables (columns of A) and of
// the rows inside the zero / nonnegative cones, tried once in scs_init when the gathers of the two CSR products of
// linsys/scs_matrix.c:161-186 share few cache lines in the order the caller gave.
//
// Why: the wave-owned-rows product (spmv_wave.h) reaches 0.41 - 0.49 of the HBM roofline when the gathers of a unit of
// consecutive rows fall into a narrow window of the gathered vector, and 0.28 when every gather is its own 128-byte line.
// Real SOCPs come out of modelling layers with that structure present but hidden: variables and the rows of the
// elementwise cones are numbered in whatever order the model was written.  The row order inside second-order, PSD, box,
// exponential and power cones is part of the cone's definition (include/scs.h:121-172) and is never touched; rows of the zero
// and nonnegative cones are interchangeable, and so are variables.
//
// What: (a) anchors -- rows that must stay where they are (everything behind the nonnegative cone) already carry an order;
// a column is keyed by the mean position of its anchored entries, the free rows by the mean new position of their columns;
// (b) without enough anchors (pure LPs) a breadth-first Cuthill-McKee numbering of the bipartite row / column graph from
// a pseudo-peripheral start.  The candidate is kept only if the MEASURED line sharing (distinct 128-byte lines of the
// gathered vector per entry, per unit of rows as spmv_wave.h cuts them) improves by 20 % or more; a uniformly random
// matrix (the headline benchmark) is recognised from the spread of its anchors in one pass over the pattern and goes to (c).
// (c) round 6 -- "chain + home" for patterns WITHOUT hidden locality (what rounds 4-5 skipped): nothing can make most gathers of a
// uniformly random matrix share lines, but a fixed share can be made local by construction -- columns numbered along greedy walks in
// which neighbours share a row (one line of x serves both entries of that row; the neighbouring columns ask for the same y entry),
// and every movable row keyed by one of its columns, the rows of a cone's range ordered by that key in line-sized blocks that are
// dealt wide (so that no unit of consecutive rows piles its entries on one spot of x).  Movable here also means the TAIL of a
// second-order cone: |x|_2 does not depend on the order of x (src/cones.c:1247-1279), D is constant inside a cone
// (linsys/scs_matrix.c:257,329) and R_y too (src/cones.c:349-363); the cone's first row t stays.  ~3 of 10 entries per column turn
// local on the headline family: 0.96 / 0.98 -> 0.77 / 0.72 distinct lines per gathered entry (kept from 15 % on); both CG products
// 65 - 67 -> 56 us (profiles/r6_chain_home.md).
// The solve then runs entirely in the new numbering; scs_update / warm starts / the returned (x, y, s) are mapped at the
// API boundary (admm.hip), so callers never see it.  P != NULL disables it (a symmetric permutation of the upper triangle
// is not implemented).  SCS_AMD_REORDER=0 switches it off, =1 forces the attempt on small problems too (tests).
#pragma once
#include "scs_host.h"
#include <vector>

namespace scsamd {

struct Reorder {
  bool active = false;
  std::vector<int> col_new2old, row_new2old; // new index -> caller's index
  double before[2] = {1, 1}, after[2] = {1, 1}; // lines per entry: [0] product with A (gathers from x), [1] with A' (gathers from y)
  double seconds = 0;
  const char *method = "none";
  const char *why = "not attempted";
  // the renumbered matrix, when plan_reorder already built it (beside the measurement of the candidate, on its own threads):
  // apply_reorder then only moves it into place
  mutable HostCsc ready;
  mutable bool have_ready = false;
};

// decides (and fills R); A is the caller's matrix (m x n CSC), k the cone
void plan_reorder(const HostCsc &A, const ScsCone *k, bool has_P, Reorder &R);
// A <- A[row_new2old, col_new2old], row indices sorted inside every column
void apply_reorder(HostCsc &A, const Reorder &R);
// distinct 128-byte lines of the gathered vector per entry, over units of consecutive rows of ~ nnz / 2048 entries
double lines_per_entry(const eoff *ptr, const int *idx, int rows, int cols, size_t elem_bytes);

} // namespace scsamd
