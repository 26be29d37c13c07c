// problem_io.cpp -- reading / writing the reference's binary problem dump (host side).
//
// the data format either side of the path.  Byte layout
// is that of reference src/rw.c (non-spectral build): header :580-590 (sizeof(scs_int),
// sizeof(scs_float), version string), cone :100-123, data :405-422 with matrices :381-395,
// settings :250-283 (warm_start always written as 0; the two filenames are not written),
// extension block "SCSE" v1 :456-492 (complex-PSD sizes, four zero counts for the spectral
// cones, time_limit_secs).  A file written here is read back by the reference's
// SCS(read_data) and vice versa (tests/test_problem_io.py); `write_data_filename` in
// ScsSettings makes scs_init dump the problem exactly like src/scs.c:1272-1275.
#include "scs_host.h"
#include <cstdint>
#include <cstdio>
#include <cstring>

namespace scsamd {

static const char *kFileVersion = "3.2.11"; // include/glbopts.h:26 -- the format revision we write
static const uint32_t kExtMagic = 0x53435345u, kExtVersion = 1u;

namespace {
struct Writer {
  FILE *f;
  bool ok = true;
  template <typename T> void put(const T &v) { ok = ok && fwrite(&v, sizeof(T), 1, f) == 1; }
  template <typename T> void arr(const T *p, long n) {
    if (n > 0) ok = ok && p && fwrite(p, sizeof(T), (size_t)n, f) == (size_t)n;
  }
};
struct Reader {
  FILE *f;
  size_t int_sz = 4;
  bool ok = true;
  bool raw(void *p, size_t sz, size_t n) {
    if (n == 0) return true;
    ok = ok && fread(p, sz, n, f) == n;
    return ok;
  }
  long long getl() {
    if (int_sz == 4) {
      int32_t v = 0;
      raw(&v, 4, 1);
      return (long long)v;
    }
    long long v = 0;
    raw(&v, 8, 1);
    return v;
  }
  int geti() { return (int)getl(); }
  real getf() {
    real v = 0;
    raw(&v, sizeof(real), 1);
    return v;
  }
  // an index array of the file (4- or 8-byte integers, whatever wrote it) as this build's scs_int
  scs_int *ints(long n) {
    if (n <= 0) return nullptr;
    scs_int *p = (scs_int *)calloc((size_t)n, sizeof(scs_int));
    if (int_sz == sizeof(scs_int)) raw(p, sizeof(scs_int), (size_t)n);
    else
      for (long i = 0; i < n && ok; ++i) p[i] = (scs_int)getl();
    return p;
  }
  real *floats(long n) {
    if (n <= 0) return nullptr;
    real *p = (real *)calloc((size_t)n, sizeof(real));
    raw(p, sizeof(real), (size_t)n);
    return p;
  }
  void skip_ints(long n) {
    for (long i = 0; i < n && ok; ++i) (void)geti();
  }
};

void write_matrix(Writer &w, const ScsMatrix *A) {
  w.put<scs_int>(A->m);
  w.put<scs_int>(A->n);
  w.arr(A->p, (long)A->n + 1);
  w.arr(A->x, A->p[A->n]);
  w.arr(A->i, A->p[A->n]);
}
ScsMatrix *read_matrix(Reader &r) {
  ScsMatrix *A = (ScsMatrix *)calloc(1, sizeof(ScsMatrix));
  A->m = r.geti();
  A->n = r.geti();
  if (!r.ok || A->m < 0 || A->n < 0) {
    r.ok = false;
    return A;
  }
  A->p = r.ints((long)A->n + 1);
  const long nnz = (r.ok && A->p) ? A->p[A->n] : 0;
  if (nnz < 0) r.ok = false;
  if (r.ok) {
    A->x = r.floats(nnz);
    A->i = r.ints(nnz);
  }
  return A;
}
void free_matrix(ScsMatrix *A) {
  if (!A) return;
  free(A->p);
  free(A->x);
  free(A->i);
  free(A);
}
} // namespace

int write_problem(const ScsData *d, const ScsCone *k, const ScsSettings *s, const char *filename) {
  FILE *f = fopen(filename, "wb");
  if (!f) {
    printf("Error: could not open %s for writing\n", filename);
    return -1;
  }
  Writer w{f};
  w.put<uint32_t>((uint32_t)sizeof(scs_int));
  w.put<uint32_t>((uint32_t)sizeof(scs_float));
  w.put<uint32_t>((uint32_t)strlen(kFileVersion));
  w.arr(kFileVersion, (long)strlen(kFileVersion));
  // cone
  const long box_len = k->bsize > 1 ? k->bsize - 1 : 0;
  w.put<scs_int>(k->z);
  w.put<scs_int>(k->l);
  w.put<scs_int>(k->bsize);
  w.arr(k->bl, box_len);
  w.arr(k->bu, box_len);
  w.put<scs_int>(k->qsize);
  w.arr(k->q, k->qsize);
  w.put<scs_int>(k->ssize);
  w.arr(k->s, k->ssize);
  w.put<scs_int>(k->ep);
  w.put<scs_int>(k->ed);
  w.put<scs_int>(k->psize);
  w.arr(k->p, k->psize);
  // data
  w.put<scs_int>(d->m);
  w.put<scs_int>(d->n);
  w.arr(d->b, d->m);
  w.arr(d->c, d->n);
  write_matrix(w, d->A);
  w.put<scs_int>(d->P ? 1 : 0);
  if (d->P) write_matrix(w, d->P);
  // settings
  w.put<scs_int>(s->normalize);
  w.put<scs_float>(s->scale);
  w.put<scs_float>(s->rho_x);
  w.put<scs_int>(s->max_iters);
  w.put<scs_float>(s->eps_abs);
  w.put<scs_float>(s->eps_rel);
  w.put<scs_float>(s->eps_infeas);
  w.put<scs_float>(s->alpha);
  w.put<scs_int>(s->verbose);
  w.put<scs_int>(0); // warm_start
  w.put<scs_int>(s->acceleration_lookback);
  w.put<scs_int>(s->acceleration_interval);
  w.put<scs_int>(s->acceleration_type_1);
  w.put<scs_float>(s->acceleration_regularization);
  w.put<scs_float>(s->acceleration_relaxation);
  w.put<scs_int>(s->adaptive_scale);
  // extension block
  w.put<uint32_t>(kExtMagic);
  w.put<uint32_t>(kExtVersion);
  w.put<scs_int>(k->cssize);
  w.arr(k->cs, k->cssize);
  for (int i = 0; i < 4; ++i) w.put<scs_int>(0); // logdet, nuclear, ell1, sum-largest counts
  w.put<scs_float>(s->time_limit_secs);
  const bool ok = w.ok;
  if (fclose(f) != 0 || !ok) {
    printf("Error: failed writing SCS data to %s\n", filename);
    return -1;
  }
  return 0;
}

void free_problem(ScsData *d, ScsCone *k, ScsSettings *s) {
  if (d) {
    free(d->b);
    free(d->c);
    free_matrix(d->A);
    free_matrix(d->P);
    free(d);
  }
  if (k) {
    free(k->bl);
    free(k->bu);
    free(k->q);
    free(k->s);
    free(k->cs);
    free(k->p);
    free(k);
  }
  free(s);
}

int read_problem(const char *filename, ScsData **dout, ScsCone **kout, ScsSettings **sout) {
  *dout = nullptr;
  *kout = nullptr;
  *sout = nullptr;
  FILE *f = fopen(filename, "rb");
  if (!f) {
    printf("Error reading file %s\n", filename);
    return -1;
  }
  Reader r{f};
  uint32_t isz = 0, fsz = 0, vsz = 0;
  char ver[16] = {0};
  r.raw(&isz, 4, 1);
  r.raw(&fsz, 4, 1);
  r.raw(&vsz, 4, 1);
  if (!r.ok || (isz != 4 && isz != 8) || fsz != sizeof(scs_float) || vsz >= sizeof ver) {
    printf("Error: %s is not an SCS data file for this build (int %u B, float %u B)\n", filename, isz, fsz);
    fclose(f);
    return -1;
  }
  r.int_sz = isz;
  r.raw(ver, 1, vsz);
  const bool legacy = strcmp(ver, kFileVersion) != 0; // src/rw.c:663
  ScsCone *k = (ScsCone *)calloc(1, sizeof(ScsCone));
  ScsData *d = (ScsData *)calloc(1, sizeof(ScsData));
  ScsSettings *s = (ScsSettings *)calloc(1, sizeof(ScsSettings));
  scs_set_default_settings(s);
  k->z = r.geti();
  k->l = r.geti();
  k->bsize = r.geti();
  if (r.ok && k->bsize < 0) r.ok = false;
  const long box_len = k->bsize > 1 ? k->bsize - 1 : 0;
  if (r.ok) {
    k->bl = r.floats(box_len);
    k->bu = r.floats(box_len);
    k->qsize = r.geti();
    if (k->qsize < 0) r.ok = false;
  }
  if (r.ok) {
    k->q = r.ints(k->qsize);
    k->ssize = r.geti();
    if (k->ssize < 0) r.ok = false;
  }
  if (r.ok) {
    k->s = r.ints(k->ssize);
    k->ep = r.geti();
    k->ed = r.geti();
    k->psize = r.geti();
    if (k->psize < 0) r.ok = false;
  }
  if (r.ok) k->p = r.floats(k->psize);
  if (r.ok) {
    d->m = r.geti();
    d->n = r.geti();
    if (d->m < 0 || d->n < 0) r.ok = false;
  }
  if (r.ok) {
    d->b = r.floats(d->m);
    d->c = r.floats(d->n);
    d->A = read_matrix(r);
    const int has_p = r.ok ? r.geti() : 0;
    if (r.ok && has_p) d->P = read_matrix(r);
  }
  if (r.ok) {
    s->normalize = r.geti();
    s->scale = r.getf();
    s->rho_x = r.getf();
    s->max_iters = r.geti();
    s->eps_abs = r.getf();
    s->eps_rel = r.getf();
    s->eps_infeas = r.getf();
    s->alpha = r.getf();
    s->verbose = r.geti();
    s->warm_start = r.geti();
    s->acceleration_lookback = r.geti();
    s->acceleration_interval = r.geti();
    if (legacy) {
      s->adaptive_scale = r.geti();
    } else {
      s->acceleration_type_1 = r.geti();
      s->acceleration_regularization = r.getf();
      s->acceleration_relaxation = r.getf();
      s->adaptive_scale = r.geti();
    }
  }
  if (r.ok) { // optional extension block (src/rw.c:494-560)
    uint32_t magic = 0;
    const size_t got = fread(&magic, 1, 4, f);
    if (got == 4 && magic == kExtMagic) {
      uint32_t v = 0;
      r.raw(&v, 4, 1);
      if (r.ok && v == kExtVersion) {
        k->cssize = r.geti();
        if (k->cssize < 0) r.ok = false;
        if (r.ok) k->cs = r.ints(k->cssize);
        const int dsize = r.ok ? r.geti() : 0;
        r.skip_ints(dsize);
        const int nuc = r.ok ? r.geti() : 0;
        r.skip_ints(2L * nuc);
        const int ell1 = r.ok ? r.geti() : 0;
        r.skip_ints(ell1);
        const int sl = r.ok ? r.geti() : 0;
        r.skip_ints(2L * sl);
        if (r.ok) s->time_limit_secs = r.getf();
      } else {
        r.ok = false;
      }
    } else if (got != 0 && got != 4) {
      r.ok = false;
    }
  }
  fclose(f);
  if (!r.ok) {
    printf("Error: failed reading SCS data from %s\n", filename);
    free_problem(d, k, s);
    return -1;
  }
  *dout = d;
  *kout = k;
  *sout = s;
  return 0;
}

} // namespace scsamd

extern "C" {
// replaces _scs_write_data (src/rw.c:574) / _scs_read_data (:605) / the three frees of test/run_from_file.c
scs_int scs_amd_write_data(const ScsData *d, const ScsCone *k, const ScsSettings *stgs, const char *filename) {
  if (!d || !k || !stgs || !filename || !d->A) return -1;
  return scsamd::write_problem(d, k, stgs, filename);
}
scs_int scs_amd_read_data(const char *filename, ScsData **d, ScsCone **k, ScsSettings **stgs) {
  if (!filename || !d || !k || !stgs) return -1;
  return scsamd::read_problem(filename, d, k, stgs);
}
void scs_amd_free_data(ScsData *d, ScsCone *k, ScsSettings *stgs) { scsamd::free_problem(d, k, stgs); }
}
