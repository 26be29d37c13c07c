// cones_shim.cpp -- boundary B1': the reference's internal cone interface
// (include/cones.h:80-90; the `_scs_` prefix is glbopts.h's SCS(x) macro) exported by
// libscsamd_cones.so, so that a reference build links THIS in place of src/cones.o and
// every cone projection of its ADMM loop runs on the MI355X.  The reference has no
// plugin API for cones; the nine symbols below are everything the rest of its tree
// (src/scs.c, linsys/scs_matrix.c, test/problem_utils.h) calls.
//
//   _scs_init_cone               src/cones.c:1498-1538   workspace (device side built lazily)
//   _scs_proj_dual_cone          src/cones.c:1552-1596   Moreau-wrapped projection, in place, host vector
//   _scs_finish_cone             src/cones.c:284-338
//   _scs_set_r_y                 src/cones.c:349-363
//   _scs_enforce_cone_boundaries src/cones.c:366-379     (used by the reference's equilibration)
//   _scs_validate_cones          src/cones.c:583-760
//   _scs_get_cone_header         src/cones.c:565-581     (caller frees)
//   _scs_deep_copy_cone / _scs_free_cone  src/cones.c:122-275
//
// `struct SCS_CONE_WORK` is ours to define: outside src/cones.c the reference only
// touches its fields under USE_SPECTRAL_CONES (src/scs.c:943-944, src/rw.c:854-858),
// which is off by default and not carried here.  Host memory handed back to the
// reference comes from calloc/malloc, matching its scs_calloc/scs_free (glbopts.h:131-137).
// The device work is created at the first projection, because only then does the caller
// say whether the box bounds are to be normalised by the equilibration's D
// (reference: lazy normalize_box_cone, cones.c:1557-1565).  Unlike the reference the
// caller's ScsCone is never mutated (the normalised bounds live in HBM).
#include "cones.h"
#include "scs_host.h"

using namespace scsamd;

struct SCS_CONE_WORK {
  const ScsCone *k = nullptr; // the caller's (deep-copied) cone description; outlives us
  scs_int m = 0;
  std::vector<int> seg;       // cone_boundaries of src/cones.c:386-424
  ScsAmdConeWork *dev = nullptr;
};

extern "C" {

void _scs_free_cone(ScsCone *k) {
  if (!k) return;
  free(k->bu);
  free(k->bl);
  free(k->q);
  free(k->s);
  free(k->cs);
  free(k->p);
  free(k);
}

// returns 1 on success, 0 if an allocation failed (dest is left freeable either way)
scs_int _scs_deep_copy_cone(ScsCone *dest, const ScsCone *src) {
  memset(dest, 0, sizeof *dest);
  dest->z = src->z;
  dest->l = src->l;
  dest->bsize = src->bsize;
  dest->qsize = src->qsize;
  dest->ssize = src->ssize;
  dest->cssize = src->cssize;
  dest->ep = src->ep;
  dest->ed = src->ed;
  dest->psize = src->psize;
  auto dup = [](const void *from, size_t count, size_t elem) -> void * {
    void *p = calloc(count, elem);
    if (p) memcpy(p, from, count * elem);
    return p;
  };
  if (src->bsize > 1) {
    dest->bu = (scs_float *)dup(src->bu, (size_t)src->bsize - 1, sizeof(scs_float));
    dest->bl = (scs_float *)dup(src->bl, (size_t)src->bsize - 1, sizeof(scs_float));
    if (!dest->bu || !dest->bl) return 0;
  }
  if (src->qsize > 0 && !(dest->q = (scs_int *)dup(src->q, (size_t)src->qsize, sizeof(scs_int)))) return 0;
  if (src->ssize > 0 && !(dest->s = (scs_int *)dup(src->s, (size_t)src->ssize, sizeof(scs_int)))) return 0;
  if (src->cssize > 0 && !(dest->cs = (scs_int *)dup(src->cs, (size_t)src->cssize, sizeof(scs_int)))) return 0;
  if (src->psize > 0 && !(dest->p = (scs_float *)dup(src->p, (size_t)src->psize, sizeof(scs_float)))) return 0;
  return 1;
}

scs_int _scs_validate_cones(const ScsData *d, const ScsCone *k) {
  if (!d || !k) return -1;
  return validate_cone(k, d->m, true) < 0 ? -1 : 0;
}

char *_scs_get_cone_header(const ScsCone *k) {
  std::string h = "cones: ";
  char b[96];
  auto add = [&](const char *fmt, long a, long c) {
    snprintf(b, sizeof b, fmt, a, c);
    h += b;
  };
  if (k->z) add("\t  z: primal zero / dual free vars: %li\n", (long)k->z, 0);
  if (k->l) add("\t  l: linear vars: %li\n", (long)k->l, 0);
  if (k->bsize) add("\t  b: box cone vars: %li\n", (long)k->bsize, 0);
  long rows = 0;
  for (int i = 0; i < k->qsize; ++i) rows += k->q[i];
  if (k->qsize) add("\t  q: soc vars: %li, qsize: %li\n", rows, (long)k->qsize);
  rows = 0;
  for (int i = 0; i < k->ssize; ++i) rows += (long)k->s[i] * (k->s[i] + 1) / 2;
  if (k->ssize) add("\t  s: psd vars: %li, ssize: %li\n", rows, (long)k->ssize);
  rows = 0;
  for (int i = 0; i < k->cssize; ++i) rows += (long)k->cs[i] * k->cs[i];
  if (k->cssize) add("\t  cs: complex psd vars: %li, cssize: %li\n", rows, (long)k->cssize);
  if (k->ep || k->ed) add("\t  e: exp vars: %li, dual exp vars: %li\n", 3L * k->ep, 3L * k->ed);
  if (k->psize) add("\t  p: primal + dual power vars: %li\n", 3L * k->psize, 0);
  h += "\t  (projections on MI355X: libscsamd_cones)\n";
  char *out = (char *)malloc(h.size() + 1);
  if (out) memcpy(out, h.c_str(), h.size() + 1);
  return out;
}

ScsConeWork *_scs_init_cone(ScsCone *k, scs_int m) {
  if (!k || validate_cone(k, m, true) < 0) return nullptr;
  ScsConeWork *c = new (std::nothrow) ScsConeWork();
  if (!c) return nullptr;
  c->k = k;
  c->m = m;
  c->seg = cone_segments(k);
  return c;
}

void _scs_finish_cone(ScsConeWork *c) {
  if (!c) return;
  if (c->dev) scs_amd_cone_finish(c->dev);
  delete c;
}

void _scs_set_r_y(const ScsConeWork *c, scs_float scale, scs_float *r_y) {
  const scs_int z = c->k->z;
  for (scs_int i = 0; i < z; ++i) r_y[i] = (scs_float)1.0 / ((scs_float)1000. * scale);
  for (scs_int i = z; i < c->m; ++i) r_y[i] = (scs_float)1.0 / scale;
}

// every cone after the first (row-wise) block gets the single value f(its slice)
void _scs_enforce_cone_boundaries(const ScsConeWork *c, scs_float *vec,
                                  scs_float (*f)(const scs_float *, scs_int)) {
  size_t pos = (size_t)c->seg[0];
  for (size_t s = 1; s < c->seg.size(); ++s) {
    const scs_int len = c->seg[s];
    const scs_float w = f(vec + pos, len);
    for (scs_int j = 0; j < len; ++j) vec[pos + j] = w;
    pos += (size_t)len;
  }
}

// x <- Proj_{K*}^{R}(x); returns <0 on failure (the reference aborts the solve, src/scs.c:1389)
scs_int _scs_proj_dual_cone(scs_float *x, ScsConeWork *c, const ScsScaling *scal, scs_float *r_y) {
  if (!c || !x) return -1;
  if (!c->dev) {
    c->dev = scs_amd_cone_init(c->k, c->m, scal ? scal->D : nullptr);
    if (!c->dev) {
      fprintf(stderr, "scs_amd: cone projection needs a HIP device -- this backend has no CPU fallback\n");
      return -1;
    }
  }
  return scs_amd_cone_proj_dual(c->dev, x, r_y);
}

} // extern "C"
