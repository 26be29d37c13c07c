// shard_native.cpp -- ONE linear system split by rows of A across GPUs, native form.
//
// What is split: the operator of reference linsys/cpu/indirect/private.c:106-119 inside the solve of :133-324.  Rank r holds the
// row slab A_r (m_r x n) and its part of R_y; G = R_x + A' R_y^-1 A = sum_r (R_x / N + A_r' R_r^-1 A_r).  The PCG loop is the
// device-controlled loop of linsys.hip (LinSys with a ShardHook): per iteration the two slab products, ONE all-reduce of the
// n-vector G p enqueued on the solver's own stream, and the level-1 part replicated on every rank.  No host read-back per
// iteration: the host reads one 48-byte control block per batch of iterations, exactly as in the unsplit solve; every rank
// enqueues the same sequence, so the collectives match up without any handshake.
//
// Collectives: (a) RCCL through its C API (ncclAllReduce on the solver's stream).  librccl is opened at run time with dlopen --
// the product libraries carry no link-time dependency on it, and a process that already has an RCCL loaded (torch) shares it.
// (b) "threads": the ranks are host threads of ONE process on ONE GPU and the all-reduce is two kernels between event
// hand-shakes -- a test double, so that the N = 2 algebra of the native path runs on a box with a single GPU (RCCL refuses two
// ranks on one device).
#include "linsys.h"
#include <condition_variable>
#include <dlfcn.h>
#include <mutex>

// The five RCCL entry points used here, declared locally: the product libraries must build on a ROCm install without the
// RCCL development headers -- the library itself is only ever dlopen'ed.  Values are those of the NCCL C API
// (nccl.h: ncclUniqueId is 128 opaque bytes; ncclSuccess = 0; ncclFloat = 7, ncclDouble = 8; ncclSum = 0, ncclMax = 2).
extern "C" {
typedef struct ncclComm *ncclComm_t;
#define NCCL_UNIQUE_ID_BYTES 128
typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;
// (fixed underlying type: RCCL returns result codes 1..7 through these function pointers, which an enum that lists only 0 could not
// formally hold -- with -fstrict-enums `rc != ncclSuccess` could be folded away)
typedef enum : int { ncclSuccess = 0 } ncclResult_t;
typedef enum : int { ncclFloat = 7, ncclDouble = 8 } ncclDataType_t;
typedef enum : int { ncclSum = 0, ncclMax = 2 } ncclRedOp_t;
}

namespace scsamd {

// ---- RCCL, resolved at run time ----------------------------------------------------------------------------------------
struct RcclApi {
  void *lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};
static RcclApi &rccl() {
  static RcclApi api = [] {
    RcclApi a;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      a.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (a.lib) break;
    }
    if (!a.lib) return a;
    a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(a.lib, "ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(a.lib, "ncclCommInitRank"));
    a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(dlsym(a.lib, "ncclAllReduce"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(a.lib, "ncclCommDestroy"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(a.lib, "ncclGetErrorString"));
    a.ok = a.GetUniqueId && a.CommInitRank && a.AllReduce && a.CommDestroy;
    return a;
  }();
  return api;
}

// ---- "threads": ranks = host threads of one process sharing one GPU -------------------------------------------------
constexpr int SHARD_THREADS_MAX = 8;
struct ShardPtrs { // by value in the kernel arguments: the host threads run far ahead of the device, a shared table would be overwritten
  const real *p[SHARD_THREADS_MAX];
};
__global__ __launch_bounds__(SCSAMD_BLOCK) void k_shard_combine(real *out, ShardPtrs bufs, int world, size_t count, int op) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
    real acc = bufs.p[0][i];
    for (int r = 1; r < world; ++r) { // rank order: the same bits on every rank
      const real v = bufs.p[r][i];
      acc = op == 0 ? acc + v : (v > acc ? v : acc);
    }
    out[i] = acc;
  }
}
struct ThreadGroup {
  int world = 0;
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0;
  long long generation = 0;
  std::vector<real *> buf;           // the buffer every rank wants reduced
  std::vector<hipEvent_t> ev_in, ev_mid;
  void barrier() {
    std::unique_lock<std::mutex> lk(mu);
    const long long gen = generation;
    if (++arrived == world) {
      arrived = 0;
      ++generation;
      cv.notify_all();
    } else {
      cv.wait(lk, [&] { return generation != gen; });
    }
  }
};

struct ShardWork {
  LinSys ls;
  ShardHook hook;
  int device = 0;
  // rccl
  ncclComm_t comm = nullptr;
  // threads
  ThreadGroup *group = nullptr;
  DevBuf<real> scratch;
  ~ShardWork() {
    if (comm && rccl().ok) (void)rccl().CommDestroy(comm);
  }
};

static int allreduce_rccl(void *ctx, real *buf, size_t count, int op, hipStream_t st) {
  ShardWork *w = static_cast<ShardWork *>(ctx);
  const ncclResult_t rc = rccl().AllReduce(buf, buf, count, sizeof(real) == 8 ? ncclDouble : ncclFloat, op == 0 ? ncclSum : ncclMax, w->comm, st);
  if (rc != ncclSuccess) {
    fprintf(stderr, "scs_amd: ncclAllReduce failed: %s\n", rccl().GetErrorString ? rccl().GetErrorString(rc) : "?");
    return -1;
  }
  return 0;
}

// every rank: publish my buffer, wait until all inputs are complete, combine all of them (rank order) into my scratch, wait until
// everybody has read my buffer, copy the result over it
static int allreduce_threads(void *ctx, real *buf, size_t count, int op, hipStream_t st) {
  ShardWork *w = static_cast<ShardWork *>(ctx);
  ThreadGroup *g = w->group;
  const int r = w->hook.rank;
  try {
    if (w->scratch.n < count) throw HipError("scs_amd: shard scratch too small");
    g->buf[r] = buf;
    HIP_CHECK(hipEventRecord(g->ev_in[r], st));
    g->barrier();
    ShardPtrs ptrs;
    for (int q = 0; q < SHARD_THREADS_MAX; ++q) ptrs.p[q] = q < g->world ? g->buf[q] : nullptr;
    for (int q = 0; q < g->world; ++q)
      if (q != r) HIP_CHECK(hipStreamWaitEvent(st, g->ev_in[q], 0));
    const int grid = (int)std::max<size_t>(1, std::min<size_t>((count + SCSAMD_BLOCK - 1) / SCSAMD_BLOCK, 1024));
    hipLaunchKernelGGL(k_shard_combine, dim3(grid), dim3(SCSAMD_BLOCK), 0, st, w->scratch.p, ptrs, g->world, count, op);
    HIP_CHECK(hipEventRecord(g->ev_mid[r], st));
    g->barrier();
    for (int q = 0; q < g->world; ++q)
      if (q != r) HIP_CHECK(hipStreamWaitEvent(st, g->ev_mid[q], 0));
    HIP_CHECK(hipMemcpyAsync(buf, w->scratch.p, count * sizeof(real), hipMemcpyDeviceToDevice, st));
    g->barrier(); // nobody re-publishes a pointer (or reuses an event) before everybody has enqueued this round
  } catch (const std::exception &ex) {
    fprintf(stderr, "%s\n", ex.what());
    return -1;
  }
  return 0;
}

static ShardWork *shard_create(const ScsMatrix *A_slab, const scs_float *diag_r_local) {
  ShardWork *w = new ShardWork();
  w->device = selected_device();
  HIP_CHECK(hipSetDevice(w->device));
  CscArg a(A_slab);
  w->ls.init(a.ptr(), nullptr, nullptr);
  w->ls.b_stage.alloc((size_t)A_slab->n + A_slab->m);
  w->ls.s_stage.alloc((size_t)A_slab->n);
  (void)diag_r_local;
  return w;
}

} // namespace scsamd

using namespace scsamd;

extern "C" {

// 128 opaque bytes that rank 0 creates and hands to every rank (by whatever channel the launcher has: a file, a socket,
// torch.distributed's object broadcast); returns 0 on success, < 0 if RCCL cannot be loaded
scs_int scs_amd_shard_unique_id(char *out128) {
  if (!out128 || !rccl().ok) return -1;
  ncclUniqueId id;
  if (rccl().GetUniqueId(&id) != ncclSuccess) return -2;
  memcpy(out128, id.internal, NCCL_UNIQUE_ID_BYTES);
  return 0;
}

// A_slab: rows [r0, r1) of A (m_r x n, CSC); diag_r_local = [R_x / world (n) ; R_y of the slab (m_r)].  One process per GPU
// (the device chosen by scs_amd_set_device); collective = RCCL.  NULL on failure.
ScsAmdShard *scs_amd_shard_init_rccl(const ScsMatrix *A_slab, const scs_float *diag_r_local, scs_int world, scs_int rank, const char *id128) {
  if (!A_slab || !diag_r_local || !id128 || world < 1 || rank < 0 || rank >= world) return nullptr;
  ShardWork *w = nullptr;
  try {
    if (!rccl().ok) throw HipError("scs_amd: librccl could not be loaded");
    w = shard_create(A_slab, diag_r_local);
    ncclUniqueId id;
    memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
    if (rccl().CommInitRank(&w->comm, (int)world, id, (int)rank) != ncclSuccess) throw HipError("scs_amd: ncclCommInitRank failed");
    w->hook.ctx = w;
    w->hook.world = (int)world;
    w->hook.rank = (int)rank;
    w->hook.allreduce = allreduce_rccl;
    w->ls.set_shard(&w->hook);
    w->ls.set_diag_r_host(diag_r_local);
    HIP_CHECK(hipStreamSynchronize(w->ls.stream));
  } catch (const std::exception &ex) {
    fprintf(stderr, "%s\n", ex.what());
    delete w;
    return nullptr;
  }
  return reinterpret_cast<ScsAmdShard *>(w);
}

// test double: the ranks are host threads of this process on the selected GPU.  group_create once, then every thread calls
// init_threads with its rank (collectively: init all-reduces the preconditioner), solves collectively, frees; group_free last.
void *scs_amd_shard_group_create(scs_int world) {
  if (world < 1 || world > SHARD_THREADS_MAX) return nullptr;
  ThreadGroup *g = nullptr;
  try {
    HIP_CHECK(hipSetDevice(selected_device()));
    g = new ThreadGroup();
    g->world = (int)world;
    g->buf.assign((size_t)world, nullptr);
    g->ev_in.resize((size_t)world);
    g->ev_mid.resize((size_t)world);
    for (int r = 0; r < world; ++r) {
      HIP_CHECK(hipEventCreateWithFlags(&g->ev_in[r], hipEventDisableTiming));
      HIP_CHECK(hipEventCreateWithFlags(&g->ev_mid[r], hipEventDisableTiming));
    }
  } catch (const std::exception &ex) {
    fprintf(stderr, "%s\n", ex.what());
    delete g;
    return nullptr;
  }
  return g;
}
void scs_amd_shard_group_free(void *group) {
  ThreadGroup *g = static_cast<ThreadGroup *>(group);
  if (!g) return;
  for (auto e : g->ev_in) (void)hipEventDestroy(e);
  for (auto e : g->ev_mid) (void)hipEventDestroy(e);
  delete g;
}
ScsAmdShard *scs_amd_shard_init_threads(const ScsMatrix *A_slab, const scs_float *diag_r_local, void *group, scs_int rank) {
  ThreadGroup *g = static_cast<ThreadGroup *>(group);
  if (!A_slab || !diag_r_local || !g || rank < 0 || rank >= g->world) return nullptr;
  ShardWork *w = nullptr;
  try {
    w = shard_create(A_slab, diag_r_local);
    w->group = g;
    w->scratch.alloc(std::max<size_t>((size_t)A_slab->n, 4096));
    w->hook.ctx = w;
    w->hook.world = g->world;
    w->hook.rank = (int)rank;
    w->hook.allreduce = allreduce_threads;
    w->ls.set_shard(&w->hook);
    w->ls.set_diag_r_host(diag_r_local);
    HIP_CHECK(hipStreamSynchronize(w->ls.stream));
  } catch (const std::exception &ex) {
    fprintf(stderr, "%s\n", ex.what());
    delete w;
    return nullptr;
  }
  return reinterpret_cast<ScsAmdShard *>(w);
}

// b_local = [r_x (n, the same on every rank) ; r_y of the slab (m_r)] -> [x (n) ; y of the slab (m_r)], in place; s: warm start (n)
// or NULL; same meaning and return convention as scs_solve_lin_sys (include/linsys.h:25-71), called collectively by all ranks
scs_int scs_amd_shard_solve(ScsAmdShard *h, scs_float *b_local, const scs_float *s, scs_float tol) {
  ShardWork *w = reinterpret_cast<ShardWork *>(h);
  if (!w || !b_local) return -1;
  try {
    HIP_CHECK(hipSetDevice(w->device));
    LinSys &ls = w->ls;
    const size_t n = ls.n, m = ls.m;
    ls.b_stage.upload(b_local, n + m, ls.stream);
    if (s) ls.s_stage.upload(s, n, ls.stream);
    ls.solve_dev(ls.b_stage.p, s ? ls.s_stage.p : nullptr, tol);
    ls.b_stage.download(b_local, n + m, ls.stream);
    HIP_CHECK(hipStreamSynchronize(ls.stream));
  } catch (const std::exception &ex) {
    fprintf(stderr, "%s\n", ex.what());
    return -1;
  }
  return 0;
}

scs_int scs_amd_shard_update_diag_r(ScsAmdShard *h, const scs_float *diag_r_local) {
  ShardWork *w = reinterpret_cast<ShardWork *>(h);
  if (!w || !diag_r_local) return -1;
  try {
    HIP_CHECK(hipSetDevice(w->device));
    w->ls.set_diag_r_host(diag_r_local);
    HIP_CHECK(hipStreamSynchronize(w->ls.stream));
  } catch (const std::exception &ex) {
    fprintf(stderr, "%s\n", ex.what());
    return -1;
  }
  return 0;
}

// out[0] = PCG iterations so far, out[1] = all-reduces enqueued, out[2] = sampled all-reduces, out[3] = their mean time in us
// (HIP events on the solver's stream around the collective, 1 in 4 sampled while profiling is on), out[4] = solves
void scs_amd_shard_get_stats(ScsAmdShard *h, double *out) {
  ShardWork *w = reinterpret_cast<ShardWork *>(h);
  if (!w || !out) return;
  (void)hipSetDevice(w->device);
  (void)hipStreamSynchronize(w->ls.stream);
  w->ls.ar_timer.harvest();
  out[0] = (double)w->ls.tot_cg_its;
  out[1] = (double)w->ls.n_allreduce;
  out[2] = (double)w->ls.ar_timer.samples;
  out[3] = w->ls.ar_timer.samples ? 1e3 * w->ls.ar_timer.total_ms / (double)w->ls.ar_timer.samples : 0.0;
  out[4] = (double)w->ls.n_solves;
}
void scs_amd_shard_set_profiling(ScsAmdShard *h, scs_int on) {
  ShardWork *w = reinterpret_cast<ShardWork *>(h);
  if (w) w->ls.profiling = on != 0;
}

void scs_amd_shard_free(ScsAmdShard *h) {
  ShardWork *w = reinterpret_cast<ShardWork *>(h);
  if (!w) return;
  (void)hipSetDevice(w->device);
  delete w;
}

} // extern "C"
