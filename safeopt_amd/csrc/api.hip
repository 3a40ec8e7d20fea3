// C ABI of libsafeopt_hip.so -- see include/safeopt_hip.h for the contract and
// the reference call site each entry point replaces.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <cmath>
#include <limits>
#include <new>

#include <atomic>
#include <chrono>

#include "common.h"
#include "sets_front.h"
#include "small_path.h"

namespace {
std::string g_err;  // errors raised without a context

struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t,
                            ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t,
                            ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
};
Rccl g_rccl;

// RCCL is loaded lazily: single-GPU runs never touch it.
int load_rccl(sgp_ctx* ctx) {
  if (g_rccl.lib) return 0;
  const char* names[] = {"librccl.so.1", "librccl.so",
                         "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
  for (const char* nm : names) {
    g_rccl.lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
    if (g_rccl.lib) break;
  }
  if (!g_rccl.lib) {
    sgp_set_error(ctx, "cannot dlopen librccl.so: %s", dlerror());
    return -3;
  }
#define SYM(field, name)                                                       \
  g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(                     \
      dlsym(g_rccl.lib, name));                                                \
  if (!g_rccl.field) {                                                         \
    sgp_set_error(ctx, "librccl.so lacks %s", name);                           \
    return -3;                                                                 \
  }
  SYM(GetUniqueId, "ncclGetUniqueId")
  SYM(CommInitRank, "ncclCommInitRank")
  SYM(CommDestroy, "ncclCommDestroy")
  SYM(AllReduce, "ncclAllReduce")
  SYM(AllGather, "ncclAllGather")
  SYM(GetErrorString, "ncclGetErrorString")
  SYM(CommCount, "ncclCommCount")
#undef SYM
  return 0;
}

#define SGP_NCCL(ctx, call)                                                    \
  do {                                                                         \
    ncclResult_t r_ = (call);                                                  \
    if (r_ != ncclSuccess) {                                                   \
      sgp_set_error((ctx), "%s:%d %s -> %s", __FILE__, __LINE__, #call,        \
                    g_rccl.GetErrorString(r_));                                \
      return -4;                                                               \
    }                                                                          \
  } while (0)

int fill_kern(sgp_ctx* ctx, KernDesc* kd, int d, int n_parts, const int* kinds,
              const double* variances, const double* inv_ls) {
  SGP_CHECK(ctx, d >= 1 && d <= SGP_MAX_D, "input dimension %d not in 1..%d", d,
            SGP_MAX_D);
  SGP_CHECK(ctx, n_parts >= 1 && n_parts <= SGP_MAX_PARTS,
            "kernel has %d parts, supported 1..%d", n_parts, SGP_MAX_PARTS);
  memset(kd, 0, sizeof(*kd));
  kd->d = d;
  kd->n_parts = n_parts;
  kd->kdiag = 1.0;
  for (int p = 0; p < n_parts; ++p) {
    SGP_CHECK(ctx, kinds[p] >= SGP_RBF && kinds[p] <= SGP_MATERN52,
              "unknown kernel kind %d", kinds[p]);
    kd->kind[p] = kinds[p];
    kd->variance[p] = variances[p];
    kd->kdiag *= variances[p];
    for (int k = 0; k < d; ++k) {
      kd->inv_ls[p][k] = inv_ls[p * d + k];
      const double w = kd->inv_ls[p][k] * kern_unit(kinds[p]);
      kd->wsq[p][k] = w * w;
    }
  }
  for (int k = 0; k < d; ++k)
    kd->scale0[k] = kd->inv_ls[0][k] * kern_unit(kd->kind[0]);
  return 0;
}

int collect_gps(sgp_ctx* ctx, sgp_gp* const* gps, int G, int d, GpDev* host) {
  SGP_CHECK(ctx, G >= 1 && G <= SGP_MAX_GPS, "%d GPs, supported 1..%d", G,
            SGP_MAX_GPS);
  for (int g = 0; g < G; ++g) {
    SGP_CHECK(ctx, gps[g] && gps[g]->n > 0, "GP %d has no data", g);
    SGP_CHECK(ctx, gps[g]->ctx == ctx,
              "GP %d lives in another context (device %d) than the grid "
              "(device %d): no stream ordering, foreign device pointers",
              g, gps[g]->ctx ? gps[g]->ctx->device : -1, ctx->device);
    SGP_CHECK(ctx, gps[g]->kern.d == d, "GP %d input_dim %d != %d", g,
              gps[g]->kern.d, d);
    host[g] = gps[g]->dev;
    // same inputs, kernel, noise and jitter as the GP in front: same factor (the
    // factorisation is deterministic), only alpha differs
    host[g].share = -1;
    if (ctx->share_factors && g > 0) {
      const sgp_gp *a = gps[g - 1], *b = gps[g];
      if (a != b && a->n == b->n && !a->xhash.empty() && a->xhash.size() == size_t(a->n) &&
          b->xhash.size() == size_t(b->n) && a->xhash.back() == b->xhash.back() &&
          a->prov == b->prov &&
          memcmp(&a->kern, &b->kern, sizeof(KernDesc)) == 0 &&
          a->noise_var == b->noise_var && a->jitter == b->jitter &&
          // (the hashes only nominate: a collision must not hand one GP the other's
          // |L^-1 k|^2 -- the rows themselves decide, a few KB of memcmp per launch)
          a->xhost.size() == b->xhost.size() &&
          memcmp(a->xhost.data(), b->xhost.data(), a->xhost.size() * sizeof(double)) == 0)
        host[g].share = host[g - 1].share >= 0 ? host[g - 1].share : g - 1;
    }
  }
  return 0;
}

// FNV-1a over the bytes of training rows, chained row by row (sgp_gp::xhash)
uint64_t hash_rows(uint64_t h, const double* rows, size_t count) {
  const unsigned char* b = reinterpret_cast<const unsigned char*>(rows);
  for (size_t i = 0; i < count * sizeof(double); ++i) {
    h ^= b[i];
    h *= 1099511628211ull;
  }
  return h;
}
}  // namespace

void sgp_set_error(sgp_ctx* ctx, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (ctx) ctx->err = buf;
  g_err = buf;
}

// SGP_POISON=1 (debugging): every fresh device allocation is filled with 0xFF bytes -- NaN as
// doubles, -1 as integers -- so that a kernel that reads what nobody wrote shows in the results
// instead of depending on what an earlier call left at that address.
int sgp_poison(sgp_ctx* ctx, void* p, size_t bytes) {
  static const bool on = getenv("SGP_POISON") && atoi(getenv("SGP_POISON")) != 0;
  if (!on) return 0;
  SGP_HIP(ctx, hipMemsetAsync(p, 0xFF, bytes, ctx->stream));    // (in stream order: in front of every use)
  return 0;
}

int sgp_reserve(sgp_ctx* ctx, DevBuf* b, size_t bytes) {
  if (bytes <= b->cap && b->p) return 0;
  size_t cap = bytes < 256 ? 256 : bytes;
  if (b->p) {
    // a buffer that grows once usually grows again (one observation per BO
    // iteration): leave room, every growth is a hipMalloc and a stream sync
    if (cap < b->cap + b->cap / 2) cap = b->cap + b->cap / 2;
    SGP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    SGP_HIP(ctx, hipFree(b->p));
    b->p = nullptr;
    b->cap = 0;
  }
  SGP_HIP(ctx, hipMalloc(&b->p, cap));
  b->cap = cap;
  ++ctx->n_allocs;
  SGP_TRY(sgp_poison(ctx, b->p, cap));
  return 0;
}

void* sgp_scratch(sgp_ctx* ctx, int slot, size_t bytes) {
  if (sgp_reserve(ctx, &ctx->scratch[slot], bytes) != 0) return nullptr;
  // SGP_POISON=2: ... and a scratch slot every time it is asked for (what a call finds there
  // from the call before is as good as unwritten), except the slots that are asked for again
  // with their contents in place: the candidate staged in front of enqueue_expander (7), the
  // selection carried from one call of a pass to the next (8), the operand blocks (9-11)
  static const bool every = getenv("SGP_POISON") && atoi(getenv("SGP_POISON")) == 2;
  if (every && (slot < 7 || slot > 11) &&
      hipMemsetAsync(ctx->scratch[slot].p, 0xFF, bytes, ctx->stream) != hipSuccess)
    return nullptr;
  return ctx->scratch[slot].p;
}

// Host <-> device copies.  Small transfers go through the pinned staging
// buffer so the async copy really is asynchronous w.r.t. pageable memory.
int sgp_h2d(sgp_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (bytes == 0) return 0;
  SGP_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice,
                              ctx->stream));
  // the source is a borrowed (pageable) host buffer: complete before return
  SGP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return 0;
}

int sgp_d2h(sgp_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (bytes == 0) return 0;
  if (bytes <= ctx->pinned_cap) {
    SGP_HIP(ctx, hipMemcpyAsync(ctx->pinned, src, bytes, hipMemcpyDeviceToHost,
                                ctx->stream));
    SGP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(dst, ctx->pinned, bytes);
  } else {
    SGP_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost,
                                ctx->stream));
    SGP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  return 0;
}

extern "C" {

// ---- context ------------------------------------------------------------------
int sgp_device_count(int* n) {
  *n = 0;
  hipError_t e = hipGetDeviceCount(n);
  if (e != hipSuccess) {
    *n = 0;
    sgp_set_error(nullptr, "hipGetDeviceCount -> %s", hipGetErrorString(e));
    (void)hipGetLastError();
    return -1;
  }
  return 0;
}

int sgp_create(int device, sgp_ctx** out) {
  *out = nullptr;
  sgp_ctx* ctx = new (std::nothrow) sgp_ctx();
  if (!ctx) return -1;
  ctx->device = device;
#define CREATE_HIP(call)                                                       \
  do {                                                                         \
    hipError_t e_ = (call);                                                    \
    if (e_ != hipSuccess) {                                                    \
      sgp_set_error(nullptr, "%s -> %s", #call, hipGetErrorString(e_));        \
      delete ctx;                                                              \
      return -1;                                                               \
    }                                                                          \
  } while (0)
  CREATE_HIP(hipSetDevice(device));
  CREATE_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
  ctx->pinned_cap = 1 << 16;
  CREATE_HIP(hipHostMalloc(&ctx->pinned, ctx->pinned_cap, hipHostMallocDefault));
  CREATE_HIP(hipEventCreate(&ctx->ev0));
  CREATE_HIP(hipEventCreate(&ctx->ev1));
  hipDeviceProp_t prop;
  CREATE_HIP(hipGetDeviceProperties(&prop, device));
  ctx->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
#undef CREATE_HIP
  *out = ctx;
  return 0;
}

void sgp_destroy(sgp_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->comm && g_rccl.CommDestroy)
    g_rccl.CommDestroy(static_cast<ncclComm_t>(ctx->comm));
  for (auto& b : ctx->scratch)
    if (b.p) (void)hipFree(b.p);
  if (ctx->stage_tab.p) (void)hipFree(ctx->stage_tab.p);
  if (ctx->pstage_tab.p) (void)hipFree(ctx->pstage_tab.p);
  if (ctx->pair_split.p) (void)hipFree(ctx->pair_split.p);
  if (ctx->pair_post.p) (void)hipFree(ctx->pair_post.p);
  for (auto e : ctx->prof_events) (void)hipEventDestroy(e);
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  if (ctx->pinned) (void)hipHostFree(ctx->pinned);
  if (ctx->step_host) (void)hipHostFree(ctx->step_host);
  if (ctx->sets_host) (void)hipHostFree(ctx->sets_host);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

const char* sgp_last_error(sgp_ctx* ctx) {
  return ctx ? ctx->err.c_str() : g_err.c_str();
}

int sgp_sync(sgp_ctx* ctx) {
  SGP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return 0;
}

// ---- GP -----------------------------------------------------------------------
int sgp_gp_create(sgp_ctx* ctx, int d, int n_parts, const int* kinds,
                  const double* variances, const double* inv_ls,
                  double noise_var, sgp_gp** out) {
  *out = nullptr;
  KernDesc kd;
  SGP_TRY(fill_kern(ctx, &kd, d, n_parts, kinds, variances, inv_ls));
  sgp_gp* gp = new (std::nothrow) sgp_gp();
  SGP_CHECK(ctx, gp != nullptr, "out of host memory");
  gp->ctx = ctx;
  static uint64_t next_serial = 1;
  gp->serial = next_serial++;
  gp->kern = kd;
  gp->noise_var = noise_var;
  *out = gp;
  return 0;
}

void sgp_gp_destroy(sgp_gp* gp) {
  if (!gp) return;
  sgp_ctx* ctx = gp->ctx;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  DevBuf* bufs[] = {&gp->X, &gp->Y, &gp->Xpad, &gp->Xs, &gp->XA, &gp->alpha, &gp->Apack,
                    &gp->Linv, &gp->Kmat, &gp->work, &gp->tvec, &gp->updw,
                    &gp->upd};
  for (DevBuf* b : bufs)
    if (b->p) (void)hipFree(b->p);
  delete gp;
}

int sgp_gp_set_data(sgp_gp* gp, const double* X, const double* Y, int64_t n,
                    int* chol_info, double* jitter_used) {
  sgp_ctx* ctx = gp->ctx;
  SGP_HIP(ctx, hipSetDevice(ctx->device));
  SGP_CHECK(ctx, n >= 1 && n <= 16384, "n = %lld training points unsupported",
            (long long)n);
  const int d = gp->kern.d;
  ++gp->data_version;
  gp->n = n;
  gp->n_pad = int((n + 15) / 16) * 16;
  gp->n_f = int((n + 31) / 32) * 32;
  // row capacity: room for 64+ one-row appends before the next reallocation
  if (gp->ld < gp->n_f) gp->ld = int((n + 64 + 63) / 64) * 64;
  SGP_TRY(sgp_reserve(ctx, &gp->X, size_t(gp->ld) * d * sizeof(double)));
  SGP_TRY(sgp_reserve(ctx, &gp->Y, size_t(gp->ld) * sizeof(double)));
  SGP_TRY(sgp_h2d(ctx, gp->X.p, X, size_t(n) * d * sizeof(double)));
  SGP_TRY(sgp_h2d(ctx, gp->Y.p, Y, size_t(n) * sizeof(double)));
  gp->xhost.assign(X, X + size_t(n) * d);
  gp->xhash.resize(size_t(n));
  {
    uint64_t h = 14695981039346656037ull;
    for (int64_t i = 0; i < n; ++i) {
      h = hash_rows(h, X + i * d, size_t(d));
      gp->xhash[size_t(i)] = h;
    }
    gp->prov = 0x9e3779b97f4a7c15ull ^ uint64_t(n);
  }
  // GPy util.linalg.jitchol: plain attempt, then jitter = mean(diag)*1e-6,
  // *10 per retry, at most 5 retries.
  gp->jitter = 0.0;
  int info = 0;
  SGP_TRY(factor_gp(gp, &info));
  if (info != 0) {
    const double diag_mean = gp->kern.kdiag + gp->noise_var + 1e-8;
    double jitter = diag_mean * 1e-6;
    for (int t = 0; t < 5 && info != 0 && std::isfinite(jitter); ++t) {
      gp->jitter = jitter;
      SGP_TRY(factor_gp(gp, &info));
      jitter *= 10.0;
    }
  }
  if (chol_info) *chol_info = info;
  if (jitter_used) *jitter_used = gp->jitter;
  if (info != 0) {
    gp->n = 0;
    sgp_set_error(ctx, "not positive definite, even with jitter (pivot %d)",
                  info);
    return info > 0 ? info : -2;
  }
  return 0;
}

int sgp_gp_append(sgp_gp* gp, const double* x, double y, int* info) {
  sgp_ctx* ctx = gp->ctx;
  SGP_HIP(ctx, hipSetDevice(ctx->device));
  SGP_CHECK(ctx, gp->n > 0, "GP has no data");
  *info = -1;
  if (gp->n + 1 > gp->ld) return 0;  // no room: caller refits with set_data
  const int d = gp->kern.d;
  double* X = static_cast<double*>(gp->X.p);
  double* Y = static_cast<double*>(gp->Y.p);
  SGP_TRY(sgp_h2d(ctx, X + size_t(gp->n) * d, x, size_t(d) * sizeof(double)));
  SGP_TRY(sgp_h2d(ctx, Y + gp->n, &y, sizeof(double)));
  const int64_t n0 = gp->n;
  ++gp->data_version;
  SGP_TRY(append_gp(gp, y, info));
  if (gp->n == n0 + 1) {
    gp->xhost.resize(size_t(n0) * d);
    gp->xhost.insert(gp->xhost.end(), x, x + d);
    gp->xhash.resize(size_t(n0));
    gp->xhash.push_back(hash_rows(n0 > 0 ? gp->xhash.back() : 14695981039346656037ull, x,
                                  size_t(d)));
    gp->prov = gp->prov * 1099511628211ull + 2;
  }
  return 0;
}

int sgp_gp_pop(sgp_gp* gp) {
  sgp_ctx* ctx = gp->ctx;
  SGP_HIP(ctx, hipSetDevice(ctx->device));
  SGP_CHECK(ctx, gp->n > 1, "cannot remove the only training point");
  ++gp->data_version;
  SGP_TRY(pop_gp(gp));
  gp->xhost.resize(size_t(gp->n) * gp->kern.d);
  gp->xhash.resize(size_t(gp->n));
  gp->prov = gp->prov * 1099511628211ull + 3;
  return 0;
}

int sgp_gp_predict(sgp_gp* gp, const double* Xnew, int64_t N,
                   int64_t stride_row, int64_t stride_col, double* mean,
                   double* var) {
  sgp_ctx* ctx = gp->ctx;
  SGP_HIP(ctx, hipSetDevice(ctx->device));
  SGP_CHECK(ctx, gp->n > 0, "GP has no data");
  if (N <= 0) return 0;
  const int d = gp->kern.d;
  // stage the rows through a dense row-major device copy
  double* stage = static_cast<double*>(
      sgp_scratch(ctx, 3, size_t(N) * d * sizeof(double)));
  double* pts = static_cast<double*>(
      sgp_scratch(ctx, 4, size_t(N) * (d + 2) * sizeof(double)));
  GpDev* gdev = static_cast<GpDev*>(sgp_scratch(ctx, 5, sizeof(GpDev)));
  SGP_CHECK(ctx, stage && pts && gdev, "device allocation failed: %s",
            ctx->err.c_str());
  if (small_path_pays(gp, N)) {
    // a few points: triangular multi-RHS products instead of one sweep tile
    std::vector<double> tmp(size_t(N) * d);
    for (int64_t r = 0; r < N; ++r)
      for (int k = 0; k < d; ++k)
        tmp[size_t(r) * d + k] = Xnew[r * stride_row + k * stride_col];
    SGP_TRY(sgp_h2d(ctx, stage, tmp.data(), tmp.size() * sizeof(double)));
    double* mvs = pts;                        // [mean N | var N]
    SGP_TRY(sgp_h2d(ctx, gdev, &gp->dev, sizeof(GpDev)));
    SmallBufs sb;
    SGP_CHECK(ctx, small_reserve(ctx, &gp->dev, 1, int(N), &sb) == 0,
              "device allocation failed: %s", ctx->err.c_str());
    SGP_TRY(posterior_small_all(ctx, gdev, &gp->dev, 1, stage, int(N), sb, mvs, mvs + N));
    SGP_TRY(sgp_d2h(ctx, mean, mvs, size_t(N) * sizeof(double)));
    return sgp_d2h(ctx, var, mvs + N, size_t(N) * sizeof(double));
  }
  if (stride_col == 1 && stride_row == d) {
    SGP_TRY(sgp_h2d(ctx, stage, Xnew, size_t(N) * d * sizeof(double)));
    SGP_TRY(launch_import_points(ctx, stage, N, d, d, 1, pts));
  } else if (stride_row == 1 && stride_col == N) {
    SGP_TRY(sgp_h2d(ctx, pts, Xnew, size_t(N) * d * sizeof(double)));
  } else {
    // generic strides: gather on the host into pinned-free scratch
    std::vector<double> tmp(size_t(N) * d);
    for (int64_t r = 0; r < N; ++r)
      for (int k = 0; k < d; ++k)
        tmp[size_t(k) * N + r] = Xnew[r * stride_row + k * stride_col];
    SGP_TRY(sgp_h2d(ctx, pts, tmp.data(), tmp.size() * sizeof(double)));
  }
  SGP_TRY(sgp_h2d(ctx, gdev, &gp->dev, sizeof(GpDev)));
  double* mv = pts + size_t(N) * d;
  SweepPoints sp{pts, N, 1, N};
  ConfOut co{};
  co.Q = nullptr;
  co.mean = mv;
  co.var = mv + N;
  co.S = nullptr;
  co.partial = nullptr;
  co.beta = 0.0;
  SGP_TRY(launch_sweep_conf(ctx, gdev, &gp->dev, 1, d, sp, co));
  SGP_TRY(sgp_d2h(ctx, mean, mv, size_t(N) * sizeof(double)));
  SGP_TRY(sgp_d2h(ctx, var, mv + N, size_t(N) * sizeof(double)));
  return 0;
}

int sgp_gp_get_factor(sgp_gp* gp, double* Linv, double* alpha) {
  sgp_ctx* ctx = gp->ctx;
  SGP_HIP(ctx, hipSetDevice(ctx->device));
  SGP_CHECK(ctx, gp->n > 0, "GP has no data");
  const int64_t n = gp->n;
  if (Linv) {
    SGP_HIP(ctx, hipMemcpy2DAsync(Linv, n * sizeof(double), gp->Linv.p,
                                  size_t(gp->ld) * sizeof(double),
                                  n * sizeof(double), n, hipMemcpyDeviceToHost,
                                  ctx->stream));
    SGP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  if (alpha) SGP_TRY(sgp_d2h(ctx, alpha, gp->alpha.p, n * sizeof(double)));
  return 0;
}

int sgp_kern_K(sgp_ctx* ctx, int d, int n_parts, const int* kinds,
               const double* variances, const double* inv_ls, const double* X1,
               int64_t n1, const double* X2, int64_t n2, double* out) {
  SGP_HIP(ctx, hipSetDevice(ctx->device));
  KernDesc kd;
  SGP_TRY(fill_kern(ctx, &kd, d, n_parts, kinds, variances, inv_ls));
  if (n1 <= 0 || n2 <= 0) return 0;
  const size_t b1 = size_t(n1) * d * sizeof(double);
  const size_t b2 = size_t(n2) * d * sizeof(double);
  const size_t bo = size_t(n1) * n2 * sizeof(double);
  double* x1 = static_cast<double*>(sgp_scratch(ctx, 3, b1 + b2));
  double* o = static_cast<double*>(sgp_scratch(ctx, 4, bo));
  SGP_CHECK(ctx, x1 && o, "device allocation failed: %s", ctx->err.c_str());
  double* x2 = x1 + size_t(n1) * d;
  SGP_TRY(sgp_h2d(ctx, x1, X1, b1));
  SGP_TRY(sgp_h2d(ctx, x2, X2, b2));
  SGP_TRY(launch_kernel_matrix(ctx, kd, x1, n1, x2, n2, o, n2, 0, 0.0,
                               std::numeric_limits<int64_t>::max()));
  SGP_TRY(sgp_d2h(ctx, out, o, bo));
  return 0;
}

// ---- grid ---------------------------------------------------------------------
int sgp_grid_create(sgp_ctx* ctx, const double* base, int64_t N, int d,
                    int64_t stride_row_B, int64_t stride_col_B, int G,
                    int64_t global_offset, sgp_grid** out) {
  *out = nullptr;
  SGP_HIP(ctx, hipSetDevice(ctx->device));
  SGP_CHECK(ctx, N >= 1, "empty grid");
  SGP_CHECK(ctx, d >= 1 && d <= SGP_MAX_D, "input dimension %d not in 1..%d", d,
            SGP_MAX_D);
  SGP_CHECK(ctx, G >= 1 && G <= SGP_MAX_GPS, "%d GPs, supported 1..%d", G,
            SGP_MAX_GPS);
  SGP_CHECK(ctx, stride_row_B % 8 == 0 && stride_col_B % 8 == 0,
            "grid strides must be multiples of 8 bytes");
  sgp_grid* g = new (std::nothrow) sgp_grid();
  SGP_CHECK(ctx, g != nullptr, "out of host memory");
  g->ctx = ctx;
  g->N = N;
  g->d = d;
  g->G = G;
  g->goff = global_offset;
  const size_t nd = size_t(N) * sizeof(double);
  g->partial_cap = N / 16 + 32 + 2048;   // (+ the split launches of sweep_pair.hip)
  struct {
    void** p;
    size_t bytes;
  } allocs[] = {{(void**)&g->pts, nd * d},      {(void**)&g->Q, nd * 2 * G},
                {(void**)&g->mean, nd * G},     {(void**)&g->var, nd * G},
                {(void**)&g->S, size_t(N)},     {(void**)&g->M, size_t(N)},
                {(void**)&g->Gm, size_t(N)},    {(void**)&g->cand, size_t(N)},
                {(void**)&g->w, nd},
                {(void**)&g->partial, size_t(g->partial_cap) * sizeof(double)},
                {(void**)&g->gpdev, sizeof(GpDev) * SGP_MAX_GPS},
                {(void**)&g->scal, 64}};
  for (auto& a : allocs) {
    hipError_t e = hipMalloc(a.p, a.bytes);
    if (e != hipSuccess) {
      sgp_set_error(ctx, "hipMalloc(%zu) -> %s", a.bytes, hipGetErrorString(e));
      sgp_grid_destroy(g);
      return -1;
    }
    ++ctx->n_allocs;
    SGP_TRY(sgp_poison(ctx, *a.p, a.bytes));
  }
  SGP_HIP(ctx, hipMemsetAsync(g->S, 0, N, ctx->stream));
  SGP_HIP(ctx, hipMemsetAsync(g->M, 0, N, ctx->stream));
  SGP_HIP(ctx, hipMemsetAsync(g->Gm, 0, N, ctx->stream));
  SGP_HIP(ctx, hipMemsetAsync(g->cand, 0, N, ctx->stream));
  // defined contents before the first sweep (zero mean, zero variance, Q = 0)
  SGP_HIP(ctx, hipMemsetAsync(g->mean, 0, nd * G, ctx->stream));
  SGP_HIP(ctx, hipMemsetAsync(g->var, 0, nd * G, ctx->stream));
  SGP_HIP(ctx, hipMemsetAsync(g->Q, 0, nd * 2 * G, ctx->stream));
  // upload the rows; the resident layout is SoA [d][N] (= the F-ordered array
  // linearly_spaced_combinations returns, so the common case is one memcpy)
  const int64_t sr = stride_row_B / 8, sc = stride_col_B / 8;
  if (sr == 1 && sc == N) {
    SGP_TRY(sgp_h2d(ctx, g->pts, base, nd * d));
  } else if (sc == 1 && sr == d) {
    double* stage = static_cast<double*>(sgp_scratch(ctx, 3, nd * d));
    SGP_CHECK(ctx, stage, "device allocation failed: %s", ctx->err.c_str());
    SGP_TRY(sgp_h2d(ctx, stage, base, nd * d));
    SGP_TRY(launch_import_points(ctx, stage, N, d, d, 1, g->pts));
    SGP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  } else {
    std::vector<double> tmp(size_t(N) * d);
    for (int64_t r = 0; r < N; ++r)
      for (int k = 0; k < d; ++k) tmp[size_t(k) * N + r] = base[r * sr + k * sc];
    SGP_TRY(sgp_h2d(ctx, g->pts, tmp.data(), nd * d));
  }
  *out = g;
  return 0;
}

void sgp_grid_destroy(sgp_grid* g) {
  if (!g) return;
  sgp_ctx* ctx = g->ctx;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  void* ptrs[] = {g->pts, g->Q,    g->mean, g->var,     g->S,    g->M,
                  g->Gm,  g->cand, g->w,    g->partial, g->gpdev, g->scal};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  if (g->ax_vals.p) (void)hipFree(g->ax_vals.p);
  for (DevBuf& b : g->sep_tab)
    if (b.p) (void)hipFree(b.p);
  delete g;
}

int sgp_grid_set_context(sgp_grid* g, const double* c, int nc) {
  sgp_ctx* ctx = g->ctx;
  SGP_HIP(ctx, hipSetDevice(ctx->device));
  SGP_CHECK(ctx, nc >= 0 && nc <= g->d && nc <= SGP_MAX_GPS,
            "bad number of context columns %d", nc);
  if (nc == 0) return 0;
  if (g->axes_valid) {
    // the context columns are constant columns of the tensor grid: new axis values
    for (int i = 0; i < nc; ++i) {
      const int k = g->d - nc + i;
      if (g->ax_count[k] != 1) {
        g->axes_valid = false;      // (declared otherwise: the declaration is void)
        break;
      }
      g->ax_host[size_t(g->ax_off[k])] = c[i];
    }
    if (g->axes_valid) {
      SGP_TRY(sgp_h2d(ctx, g->ax_vals.p, g->ax_host.data(), g->ax_host.size() * sizeof(double)));
      ++g->ax_version;
    }
  }
  return launch_fill_cols(g, c, nc);
}

// SafeOpt's parameter_set is a tensor grid (linearly_spaced_combinations,
// utilities.py:21-54) followed by constant context columns (gp_opt.py:439-451): global
// row i has column k equal to values_k[(i / stride[k]) % count[k]].  The declaration is
// checked against the resident rows bit for bit; *ok = 1 when it holds (the sweeps
// then take per-axis factor tables for RBF kernels), 0 when not (nothing changes).
int sgp_grid_set_axes(sgp_grid* g, int d, const int64_t* count, const int64_t* stride,
                      const double* values, int* ok) {
  sgp_ctx* ctx = g->ctx;
  SGP_HIP(ctx, hipSetDevice(ctx->device));
  *ok = 0;
  g->axes_valid = false;
  SGP_CHECK(ctx, d == g->d, "grid has %d columns, %d axes declared", g->d, d);
  int64_t total = 1, nvals = 0;
  for (int k = 0; k < d; ++k) {
    SGP_CHECK(ctx, count[k] >= 1 && stride[k] >= 1, "axis %d: count %lld, stride %lld", k,
              (long long)count[k], (long long)stride[k]);
    if (count[k] > 1) total *= count[k];
    nvals += count[k];
    if (total >= (int64_t(1) << 31) || count[k] * stride[k] >= (int64_t(1) << 31)) return 0;
  }
  // (32-bit index arithmetic in the sweep)
  if (g->goff + g->N > total || g->goff + g->N >= (int64_t(1) << 31)) return 0;
  g->ax_host.assign(values, values + nvals);
  int off = 0;
  for (int k = 0; k < d; ++k) {
    g->ax_count[k] = uint32_t(count[k]);
    g->ax_stride[k] = uint32_t(stride[k]);
    g->ax_off[k] = off;
    off += int(count[k]);
  }
  SGP_TRY(sgp_reserve(ctx, &g->ax_vals, size_t(nvals) * sizeof(double)));
  SGP_TRY(sgp_h2d(ctx, g->ax_vals.p, g->ax_host.data(), size_t(nvals) * sizeof(double)));
  int* mism = static_cast<int*>(sgp_scratch(ctx, 2, 256));
  SGP_CHECK(ctx, mism, "device allocation failed: %s", ctx->err.c_str());
  SGP_HIP(ctx, hipMemsetAsync(mism, 0, sizeof(int), ctx->stream));
  SGP_TRY(launch_verify_axes(g, mism));
  int bad = 0;
  SGP_TRY(sgp_d2h(ctx, &bad, mism, sizeof(int)));
  if (bad != 0) return 0;
  // the strides must chain: sorted by stride, each is the product of the counts below
  {
    int ord[SGP_MAX_D], na = 0;
    for (int k = 0; k < d; ++k)
      if (count[k] > 1) ord[na++] = k;
    std::sort(ord, ord + na, [&](int a, int b) { return stride[a] < stride[b]; });
    int64_t expect = 1;
    for (int i = 0; i < na; ++i) {
      if (stride[ord[i]] != expect) return 0;
      expect *= count[ord[i]];
    }
  }
  g->axes_valid = true;
  ++g->ax_version;
  *ok = 1;
  return 0;
}

// The tensor-grid description of a sweep over grid g with the GPs of `host`, or
// nullptr when the grid is not one / a kernel is not a product of RBF parts / the
// switch is off.  Builds (or reuses) the factor tables of every GP.
static const SepLaunch* sep_launch(sgp_grid* g, sgp_gp* const* gps, const GpDev* host, int G,
                                   SepLaunch* sl) {
  sgp_ctx* ctx = g->ctx;
  static const bool off = getenv("SGP_NO_SEP") != nullptr;
  if (off || !g->axes_valid || (ctx->sweep_choice & 8)) return nullptr;
  for (int i = 0; i < G; ++i)
    for (int p = 0; p < host[i].kern.n_parts; ++p)
      if (host[i].kern.kind[p] != SGP_RBF) return nullptr;
  int cols[SGP_MAX_D], na = 0;
  for (int k = 0; k < g->d; ++k)
    if (g->ax_count[k] > 1) cols[na++] = k;
  if (na < 1 || na > 3) return nullptr;      // (4 axes: evaluated, launch_posterior)
  // (GPs with few observations are swept by the VALU kernel, which evaluates: no tables to
  // build or refresh at every append)
  if (tiny_sweep_wanted(ctx, host, G, g->N, /*rows_sharded=*/true)) return nullptr;
  // A table is nblk x count x 128 bytes per axis -- with ONE long axis that is the whole
  // n_pad x N covariance matrix (a 1-D grid of 1e6 points, n = 544: 4.3 GB) -- and the sweeps
  // address it with 32-bit offsets.  Beyond a budget that keeps tables an L2 / Infinity-Cache
  // affair (and far below the 4 GB where the offsets would wrap) the covariances are
  // evaluated instead.
  constexpr size_t kSepBudgetBytes = size_t(256) << 20;       // per GP, all axes
  // The paired kernel (factors of more than 256 rows) streams 2-33 MB of packed factor per
  // GP through the 4-MB L2 of an XCD once per tile: tables that do not fit NEXT to that
  // stream are evicted between two uses of a line and every read becomes a fabric request
  // (config 4: 3 x 1.6 MB of tables, 43 GB per 8e6-row launch against 3.6 GB evaluated --
  // 126 x the algorithmic bytes for 2-3 % of time; profiles/r05/ab_tables.txt).  There the
  // tables are used only while all of them fit half an L2 (SGP_SEP_PAIR=1: always).
  static const bool pair_always = getenv("SGP_SEP_PAIR") && atoi(getenv("SGP_SEP_PAIR")) == 1;
  const bool paired = pair_sweep_wanted(ctx, host, G);
  size_t all_bytes = 0;
  for (int i = 0; i < G; ++i) {
    size_t bytes = 0;
    for (int a = 0; a < na; ++a)
      bytes += sep_table_doubles(host[i], g->ax_count[cols[a]]) * sizeof(double);
    if (bytes > kSepBudgetBytes) return nullptr;
    if (host[i].share < 0) all_bytes += bytes;
  }
  if (paired && !pair_always && all_bytes > (size_t(2) << 20)) return nullptr;
  std::sort(cols, cols + na, [&](int a, int b) { return g->ax_stride[a] < g->ax_stride[b]; });
  sl->naxes = na;
  sl->goff = g->goff;
  for (int a = 0; a < na; ++a) sl->count[a] = g->ax_count[cols[a]];
  for (int i = 0; i < G; ++i) {
    // (a follower of a shared factor has its leader's inputs and kernel: same tables)
    const int src = host[i].share >= 0 ? host[i].share : i;
    if (src != i) {
      for (int a = 0; a < na; ++a) sl->tab[i][a] = sl->tab[src][a];
      continue;
    }
    size_t need = 0, offa[4];
    for (int a = 0; a < na; ++a) {
      offa[a] = need;
      need += sep_table_doubles(host[i], sl->count[a]);
    }
    const uint64_t key[3] = {gps[i]->serial, gps[i]->data_version, g->ax_version};
    const bool fresh = g->sep_tab[i].p && g->sep_tab[i].cap >= need * sizeof(double) &&
                       memcmp(g->sep_key[i], key, sizeof(key)) == 0;
    // (room for the appends to come: a growing table must not reallocate every step)
    if (!g->sep_tab[i].p || g->sep_tab[i].cap < need * sizeof(double))
      if (sgp_reserve(ctx, &g->sep_tab[i], (need + need / 4) * sizeof(double)) != 0)
        return nullptr;
    double* out[4];
    for (int a = 0; a < na; ++a) {
      out[a] = static_cast<double*>(g->sep_tab[i].p) + offa[a];
      sl->tab[i][a] = out[a];
    }
    if (!fresh) {
      if (launch_sep_tables(ctx, host[i], g->d, g->ax_count,
                            static_cast<const double*>(g->ax_vals.p), g->ax_off, na, cols,
                            out) != 0)
        return nullptr;
      memcpy(g->sep_key[i], key, sizeof(key));
    }
  }
  return sl;
}

// max l0 over S ends up in g->scal[0]; out2 == nullptr defers the read-back
// (sgp_grid_sets_fused picks the value up on the device and reports it).
static int finish_safe_partials(sgp_grid* g, int nblocks, double* out2) {
  sgp_ctx* ctx = g->ctx;
  g->l0_pending = 0;
  if (!out2 && nblocks <= 4096) {
    // no read-back: the consumer (sgp_grid_sets_fused) folds the partials
    // itself; anything else that needs scal[0] calls settle_max_l first
    g->l0_pending = nblocks;
    return 0;
  }
  SGP_TRY(launch_reduce_max(ctx, g->partial, nblocks, g->scal));
  if (!out2) return 0;
  double m = 0.0;
  SGP_TRY(sgp_d2h(ctx, &m, g->scal, sizeof(double)));
  out2[0] = m;
  out2[1] = (m > -INFINITY) ? 1.0 : 0.0;
  return 0;
}

// scal[0] = max l0[S] when a deferred confidence pass left it as partials
static int settle_max_l(sgp_grid* g) {
  if (g->l0_pending > 0) {
    SGP_TRY(launch_reduce_max(g->ctx, g->partial, g->l0_pending, g->scal));
    g->l0_pending = 0;
  }
  return 0;
}

// GP descriptors -> g->gpdev through a fixed slot of the pinned staging block
// (no stream sync: the slot always holds the descriptors of this very call or
// of an identical earlier one that may still be in flight).
static int stage_gpdev(sgp_grid* g, const GpDev* host, int G) {
  sgp_ctx* ctx = g->ctx;
  // (the set passes of a step follow its confidence pass with the same GPs)
  if (G == g->gpdev_count && memcmp(g->gpdev_host, host, sizeof(GpDev) * G) == 0)
    return 0;
  memcpy(g->gpdev_host, host, sizeof(GpDev) * G);
  g->gpdev_count = G;
  char* slot = static_cast<char*>(ctx->pinned) + ctx->pinned_cap -
               sizeof(GpDev) * SGP_MAX_GPS;
  memcpy(slot, host, sizeof(GpDev) * G);
  SGP_HIP(ctx, hipMemcpyAsync(g->gpdev, slot, sizeof(GpDev) * G,
                              hipMemcpyHostToDevice, ctx->stream));
  return 0;
}

int sgp_grid_confidence(sgp_grid* g, sgp_gp* const* gps, int G, double beta,
                        const double* fmin, double* out2) {
  sgp_ctx* ctx = g->ctx;
  SGP_HIP(ctx, hipSetDevice(ctx->device));
  SGP_CHECK(ctx, G == g->G, "grid was created for %d GPs, got %d", g->G, G);
  GpDev host[SGP_MAX_GPS];
  SGP_TRY(collect_gps(ctx, gps, G, g->d, host));
  SGP_TRY(stage_gpdev(g, host, G));
  SweepPoints sp{g->pts, g->N, 1, g->N};
  ConfOut co{};
  co.Q = g->Q;
  co.mean = g->mean;
  co.var = g->var;
  co.S = g->S;
  co.partial = g->partial;
  co.beta = beta;
  for (int i = 0; i < SGP_MAX_GPS; ++i) co.fmin[i] = (i < G) ? fmin[i] : -INFINITY;
  SepLaunch sl;
  SGP_TRY(launch_sweep_conf(ctx, g->gpdev, host, G, g->d, sp, co,
                            sep_launch(g, gps, host, G, &sl), /*rows_sharded=*/true));
  return finish_safe_partials(g, sweep_num_partials(ctx, g->N), out2);
}

int sgp_grid_posterior(sgp_grid* g, sgp_gp* const* gps, int G) {
  sgp_ctx* ctx = g->ctx;
  SGP_HIP(ctx, hipSetDevice(ctx->device));
  SGP_CHECK(ctx, G == g->G, "grid was created for %d GPs, got %d", g->G, G);
  GpDev host[SGP_MAX_GPS];
  SGP_TRY(collect_gps(ctx, gps, G, g->d, host));
  SGP_TRY(stage_gpdev(g, host, G));
  SweepPoints sp{g->pts, g->N, 1, g->N};
  ConfOut co{};            // Q, S, partial stay null: mean / var only
  co.mean = g->mean;
  co.var = g->var;
  co.beta = 0.0;
  for (int i = 0; i < SGP_MAX_GPS; ++i) co.fmin[i] = -INFINITY;
  SepLaunch sl;
  return launch_sweep_conf(ctx, g->gpdev, host, G, g->d, sp, co,
                           sep_launch(g, gps, host, G, &sl), /*rows_sharded=*/true);
}

int sgp_grid_rank1_update(sgp_grid* g, sgp_gp* const* gps, int G,
                          const int* which, double beta, const double* fmin,
                          double* out2) {
  sgp_ctx* ctx = g->ctx;
  SGP_HIP(ctx, hipSetDevice(ctx->device));
  SGP_CHECK(ctx, G == g->G, "grid was created for %d GPs, got %d", g->G, G);
  GpDev host[SGP_MAX_GPS];
  SGP_TRY(collect_gps(ctx, gps, G, g->d, host));
  Rank1Args ra{};
  for (int i = 0; i < SGP_MAX_GPS; ++i) {
    ra.fmin[i] = (i < G) ? fmin[i] : -INFINITY;
    ra.which[i] = (i < G) ? which[i] : 0;
    if (i < G && which[i])
      SGP_CHECK(ctx, gps[i]->upd_valid,
                "GP %d has no append record for a rank-1 update", i);
  }
  SGP_TRY(stage_gpdev(g, host, G));
  ra.Q = g->Q;
  ra.mean = g->mean;
  ra.var = g->var;
  ra.S = g->S;
  ra.partial = g->partial;
  ra.beta = beta;
  SweepPoints sp{g->pts, g->N, 1, g->N};
  SGP_TRY(launch_rank1(ctx, g->gpdev, G, g->d, sp, ra));
  return finish_safe_partials(g, rank1_num_blocks(g->N), out2);
}

int sgp_grid_upload_Q(sgp_grid* g, const double* Q, const double* fmin,
                      double* out2) {
  sgp_ctx* ctx = g->ctx;
  SGP_HIP(ctx, hipSetDevice(ctx->device));
  SGP_TRY(sgp_h2d(ctx, g->Q, Q, size_t(g->N) * 2 * g->G * sizeof(double)));
  SGP_TRY(launch_safe_set(g, fmin));
  return finish_safe_partials(g, int((g->N + 255) / 256), out2);
}

int sgp_grid_maximizers(sgp_grid* g, double max_l, double* out) {
  sgp_ctx* ctx = g->ctx;
  SGP_HIP(ctx, hipSetDevice(ctx->device));
  SGP_TRY(launch_maximizers(g, max_l));
  double* red = static_cast<double*>(sgp_scratch(ctx, 1, 64));
  SGP_CHECK(ctx, red, "device allocation failed: %s", ctx->err.c_str());
  SGP_TRY(launch_reduce_max(ctx, g->partial, (g->N + 255) / 256, red));
  return sgp_d2h(ctx, out, red, sizeof(double));
}

int sgp_grid_candidates(sgp_grid* g, double max_var, const double* scaling,
                        const double* thr_beta, int full_sets,
                        int64_t* counts) {
  sgp_ctx* ctx = g->ctx;
  SGP_HIP(ctx, hipSetDevice(ctx->device));
  unsigned long long* cd =
      static_cast<unsigned long long*>(sgp_scratch(ctx, 1, 64));
  SGP_CHECK(ctx, cd, "device allocation failed: %s", ctx->err.c_str());
  SGP_TRY(launch_candidates(g, max_var, nullptr, scaling, thr_beta, full_sets,
                            cd));
  return sgp_d2h(ctx, counts, cd, 2 * sizeof(int64_t));
}

int sgp_grid_topk(sgp_grid* g, int mode, double cut_w, int64_t cut_idx, int k,
                  double* w_out, int64_t* gidx_out, int* n_out) {
  sgp_ctx* ctx = g->ctx;
  SGP_HIP(ctx, hipSetDevice(ctx->device));
  SGP_CHECK(ctx, k >= 1 && k <= 64, "k = %d not in 1..64", k);
  char* res = static_cast<char*>(sgp_scratch(ctx, 1, 2048));
  SGP_CHECK(ctx, res, "device allocation failed: %s", ctx->err.c_str());
  double* wd = reinterpret_cast<double*>(res);
  int64_t* id = reinterpret_cast<int64_t*>(res + 512);
  int* nd = reinterpret_cast<int*>(res + 1024);
  if (mode == 1) cut_w = (cut_idx < 0) ? INFINITY : -double(cut_idx);
  SGP_TRY(launch_topk(g, mode, cut_w, cut_idx, k, wd, id, nd));
  char host[1024 + 8];
  SGP_TRY(sgp_d2h(ctx, host, res, 1024 + 8));
  memcpy(w_out, host, size_t(k) * sizeof(double));
  memcpy(gidx_out, host + 512, size_t(k) * sizeof(int64_t));
  memcpy(n_out, host + 1024, sizeof(int));
  return 0;
}

static int upload_local_idx(sgp_grid* g, const int64_t* gidx, int m,
                            int64_t** dev) {
  sgp_ctx* ctx = g->ctx;
  SGP_CHECK(ctx, m >= 1 && m <= 4096, "m = %d rows out of range", m);
  std::vector<int64_t> li(m);
  for (int j = 0; j < m; ++j) {
    li[j] = gidx[j] - g->goff;
    SGP_CHECK(ctx, li[j] >= 0 && li[j] < g->N,
              "global index %lld is not owned by this shard",
              (long long)gidx[j]);
  }
  int64_t* d = static_cast<int64_t*>(sgp_scratch(ctx, 6, size_t(m) * 8));
  SGP_CHECK(ctx, d, "device allocation failed: %s", ctx->err.c_str());
  SGP_TRY(sgp_h2d(ctx, d, li.data(), size_t(m) * 8));
  *dev = d;
  return 0;
}

int sgp_grid_gather_rows(sgp_grid* g, const int64_t* gidx, int m, double* x,
                         double* mean, double* var, double* Q) {
  sgp_ctx* ctx = g->ctx;
  SGP_HIP(ctx, hipSetDevice(ctx->device));
  int64_t* li = nullptr;
  SGP_TRY(upload_local_idx(g, gidx, m, &li));
  const int d = g->d, G = g->G;
  const size_t per = size_t(d) + 4 * size_t(G);
  double* o = static_cast<double*>(sgp_scratch(ctx, 7, size_t(m) * per * 8));
  SGP_CHECK(ctx, o, "device allocation failed: %s", ctx->err.c_str());
  double* ox = o;
  double* om = ox + size_t(m) * d;
  double* ov = om + size_t(m) * G;
  double* oq = ov + size_t(m) * G;
  SGP_TRY(launch_gather_rows(g, li, m, ox, om, ov, oq));
  std::vector<double> host(size_t(m) * per);
  SGP_TRY(sgp_d2h(ctx, host.data(), o, host.size() * 8));
  memcpy(x, host.data(), size_t(m) * d * 8);
  memcpy(mean, host.data() + size_t(m) * d, size_t(m) * G * 8);
  memcpy(var, host.data() + size_t(m) * (d + G), size_t(m) * G * 8);
  memcpy(Q, host.data() + size_t(m) * (d + 2 * G), size_t(m) * 2 * G * 8);
  return 0;
}

int sgp_grid_mark_expanders(sgp_grid* g, const int64_t* gidx, int m) {
  sgp_ctx* ctx = g->ctx;
  SGP_HIP(ctx, hipSetDevice(ctx->device));
  if (m <= 0) return 0;
  int64_t* li = nullptr;
  SGP_TRY(upload_local_idx(g, gidx, m, &li));
  return launch_mark(g, li, m, 1);
}

int sgp_grid_unmark_expanders(sgp_grid* g, const int64_t* gidx, int m) {
  sgp_ctx* ctx = g->ctx;
  SGP_HIP(ctx, hipSetDevice(ctx->device));
  if (m <= 0) return 0;
  int64_t* li = nullptr;
  SGP_TRY(upload_local_idx(g, gidx, m, &li));
  return launch_mark(g, li, m, 0);
}

int sgp_grid_lipschitz_check(sgp_grid* g, int G, const double* fmin,
                             const double* lipschitz, int m, const double* xc,
                             const double* u_c, int32_t* flags) {
  sgp_ctx* ctx = g->ctx;
  SGP_HIP(ctx, hipSetDevice(ctx->device));
  SGP_CHECK(ctx, G == g->G, "grid was created for %d GPs, got %d", g->G, G);
  SGP_CHECK(ctx, m >= 1 && m <= SGP_TOPK, "m = %d not in 1..%d", m, SGP_TOPK);
  const int d = g->d;
  const size_t bx = size_t(m) * d * 8, bu = size_t(m) * G * 8,
               bf = size_t(m) * G * 4;
  char* buf = static_cast<char*>(sgp_scratch(ctx, 7, bx + bu + bf + 64));
  SGP_CHECK(ctx, buf, "device allocation failed: %s", ctx->err.c_str());
  double* dx = reinterpret_cast<double*>(buf);
  double* du = reinterpret_cast<double*>(buf + bx);
  int32_t* df = reinterpret_cast<int32_t*>(buf + bx + bu);
  SGP_TRY(sgp_h2d(ctx, dx, xc, bx));
  SGP_TRY(sgp_h2d(ctx, du, u_c, bu));
  SGP_HIP(ctx, hipMemsetAsync(df, 0, bf, ctx->stream));
  SGP_TRY(launch_lipschitz(g, G, fmin, lipschitz, m, dx, du, df));
  return sgp_d2h(ctx, flags, df, bf);
}

int sgp_grid_argmax(sgp_grid* g, int mode, const double* scaling, double* value,
                    int64_t* gidx) {
  sgp_ctx* ctx = g->ctx;
  SGP_HIP(ctx, hipSetDevice(ctx->device));
  char* res = static_cast<char*>(sgp_scratch(ctx, 1, 64));
  SGP_CHECK(ctx, res, "device allocation failed: %s", ctx->err.c_str());
  SGP_TRY(launch_argmax(g, mode, scaling, reinterpret_cast<double*>(res),
                        reinterpret_cast<int64_t*>(res + 8)));
  char host[16];
  SGP_TRY(sgp_d2h(ctx, host, res, 16));
  memcpy(value, host, 8);
  memcpy(gidx, host + 8, 8);
  return 0;
}

int sgp_grid_upload_mask(sgp_grid* g, int what, const uint8_t* mask) {
  sgp_ctx* ctx = g->ctx;
  SGP_HIP(ctx, hipSetDevice(ctx->device));
  uint8_t* dst = what == SGP_S ? g->S : what == SGP_M ? g->M : what == SGP_G ? g->Gm : nullptr;
  SGP_CHECK(ctx, dst, "sgp_grid_upload_mask: selector %d is not SGP_S / SGP_M / SGP_G", what);
  return sgp_h2d(ctx, dst, mask, size_t(g->N));
}

int sgp_grid_download(sgp_grid* g, int what, void* out) {
  sgp_ctx* ctx = g->ctx;
  SGP_HIP(ctx, hipSetDevice(ctx->device));
  const size_t N = size_t(g->N), G = size_t(g->G);
  switch (what) {
    case SGP_Q: return sgp_d2h(ctx, out, g->Q, N * 2 * G * 8);
    case SGP_S: return sgp_d2h(ctx, out, g->S, N);
    case SGP_M: return sgp_d2h(ctx, out, g->M, N);
    case SGP_G: return sgp_d2h(ctx, out, g->Gm, N);
    case SGP_MEAN: return sgp_d2h(ctx, out, g->mean, N * G * 8);
    case SGP_VAR: return sgp_d2h(ctx, out, g->var, N * G * 8);
    case SGP_CAND: return sgp_d2h(ctx, out, g->cand, N);     // expander candidates
    case SGP_WIDTH: return sgp_d2h(ctx, out, g->w, N * 8);   // max_i (u_i - l_i)
  }
  sgp_set_error(ctx, "unknown array selector %d", what);
  return -2;
}

// Enqueue the expander test (operands + scan); flags stay on the device.
// `top` != nullptr: the single candidate is already on the device (result
// block of the front half: x | mean | q) and xc / mu_c / u_c are ignored.
struct ExpanderBufs {   // device layout: xc | resid[G][16] | delta | inv_s2 | tn2 | flags | W
  double *xc, *resid, *delta, *inv_s2, *tn2, *W;
  int32_t* flags;
  size_t bx, bv, bf;
  int64_t wstride;
};

static int expander_bufs(sgp_grid* g, const GpDev* host, int G, ExpanderBufs* b) {
  sgp_ctx* ctx = g->ctx;
  int np_max = 0;
  for (int i = 0; i < G; ++i) np_max = host[i].n_pad > np_max ? host[i].n_pad : np_max;
  b->wstride = int64_t(np_max / 4) * 64;
  b->bx = size_t(SGP_TOPK) * g->d * 8;
  b->bv = size_t(G) * 16 * 8;
  b->bf = size_t(SGP_TOPK) * G * 4 + 64;
  const size_t total = b->bx + 4 * b->bv + b->bf + size_t(G) * b->wstride * 8;
  char* buf = static_cast<char*>(sgp_scratch(ctx, 7, total));
  SGP_CHECK(ctx, buf, "device allocation failed: %s", ctx->err.c_str());
  b->xc = reinterpret_cast<double*>(buf);
  b->resid = reinterpret_cast<double*>(buf + b->bx);
  b->delta = reinterpret_cast<double*>(buf + b->bx + b->bv);
  b->inv_s2 = reinterpret_cast<double*>(buf + b->bx + 2 * b->bv);
  b->tn2 = reinterpret_cast<double*>(buf + b->bx + 3 * b->bv);
  b->flags = reinterpret_cast<int32_t*>(buf + b->bx + 4 * b->bv);
  b->W = reinterpret_cast<double*>(buf + b->bx + 4 * b->bv + b->bf);
  return 0;
}

// `staged`: xc / resid already hold the (single) candidate and the flags are
// zeroed (k_front_final did both on the device); xc / mu_c / u_c are ignored.
static int enqueue_expander(sgp_grid* g, sgp_gp* const* gps, int G, double beta,
                            const double* fmin, int m, const double* xc,
                            const double* mu_c, const double* u_c,
                            double near_frac, int32_t** flags_dev,
                            const double* top = nullptr, bool staged = false,
                            const FrontArgs* fold = nullptr) {
  sgp_ctx* ctx = g->ctx;
  SGP_CHECK(ctx, G == g->G, "grid was created for %d GPs, got %d", g->G, G);
  SGP_CHECK(ctx, m >= 1 && m <= SGP_TOPK, "m = %d not in 1..%d", m, SGP_TOPK);
  GpDev host[SGP_MAX_GPS];
  SGP_TRY(collect_gps(ctx, gps, G, g->d, host));
  const int d = g->d;
  ExpanderBufs eb;
  SGP_TRY(expander_bufs(g, host, G, &eb));
  const size_t bx = eb.bx, bv = eb.bv, bf = eb.bf;
  const int64_t wstride = eb.wstride;
  double *dxc = eb.xc, *dres = eb.resid, *ddel = eb.delta, *dis2 = eb.inv_s2,
         *dtn2 = eb.tn2, *dW = eb.W;
  int32_t* dfl = eb.flags;
  // pinned staging block: xc | resid (descriptors have their own slot)
  const size_t hb = bx + bv + sizeof(GpDev) * SGP_MAX_GPS;
  SGP_CHECK(ctx, hb <= ctx->pinned_cap / 2, "staging buffer too small");
  char* stage = static_cast<char*>(ctx->pinned) + ctx->pinned_cap / 2;
  // previous users of this block have completed (every call that writes it
  // syncs before returning)
  if (staged) {
    // nothing to stage
  } else if (top) {
    SGP_HIP(ctx, hipMemsetAsync(dxc, 0, bx + bv, ctx->stream));
    SGP_TRY(launch_stage_top(g, top, top + d, top + d + G, dxc, dres));
  } else {
    memset(stage, 0, bx + bv);
    memcpy(stage, xc, size_t(m) * d * 8);
    double* resid = reinterpret_cast<double*>(stage + bx);
    for (int c = 0; c < m; ++c)
      for (int i = 0; i < G; ++i)
        resid[size_t(i) * 16 + c] = u_c[c * G + i] - mu_c[c * G + i];
    SGP_HIP(ctx, hipMemcpyAsync(dxc, stage, bx + bv, hipMemcpyHostToDevice,
                                ctx->stream));
  }
  SGP_TRY(stage_gpdev(g, host, G));
  if (!staged) SGP_HIP(ctx, hipMemsetAsync(dfl, 0, bf, ctx->stream));
  ExpanderArgs ea{};
  for (int i = 0; i < SGP_MAX_GPS; ++i) {
    ea.fmin[i] = (i < G) ? fmin[i] : -INFINITY;
    ea.active[i] = (i < G) && (fmin[i] != -INFINITY);
  }
  ExpanderOps ops{};
  ops.xc = dxc;
  ops.resid = dres;
  ops.Wpack = dW;
  ops.delta = ddel;
  ops.inv_s2 = dis2;
  ops.tn2 = dtn2;
  ops.wstride = wstride;
  ops.m = m;
  for (int i = 0; i < SGP_MAX_GPS; ++i) ops.active[i] = ea.active[i];
  SGP_TRY(expander_operands_all(ctx, g->gpdev, host, G, d, ops, fold));
  ea.Wpack = dW;
  ea.xc = dxc;
  ea.delta = ddel;
  ea.inv_s2 = dis2;
  ea.tn2 = dtn2;
  ea.m = m;
  ea.beta = beta;
  ea.S = g->S;
  ea.mean = g->mean;
  ea.var = g->var;
  ea.flags = dfl;
  ea.wstride = wstride;
  ea.near_frac = near_frac;
  // single candidate: pre-filter + listed contraction; the group counter sits
  // in the slack behind the flags (zeroed with them), the list in scratch
  ea.count = reinterpret_cast<int*>(reinterpret_cast<char*>(dfl) +
                                    size_t(SGP_TOPK) * G * 4 + 32);
  ea.list = (m == 1) ? static_cast<int*>(sgp_scratch(
                           ctx, 0, (size_t(g->N) + 64) * sizeof(int)))
                     : nullptr;
  SweepPoints sp{g->pts, g->N, 1, g->N};
  SGP_TRY(launch_expander_check(ctx, g->gpdev, host, G, d, sp, ea));
  *flags_dev = dfl;
  return 0;
}

int sgp_grid_expander_check(sgp_grid* g, sgp_gp* const* gps, int G, double beta,
                            const double* fmin, int m, const double* xc,
                            const double* mu_c, const double* u_c,
                            double near_frac, int32_t* flags) {
  sgp_ctx* ctx = g->ctx;
  SGP_HIP(ctx, hipSetDevice(ctx->device));
  int32_t* dfl = nullptr;
  SGP_TRY(enqueue_expander(g, gps, G, beta, fmin, m, xc, mu_c, u_c, near_frac,
                           &dfl));
  std::vector<int32_t> fl(size_t(SGP_TOPK) * G);
  SGP_TRY(sgp_d2h(ctx, fl.data(), dfl, fl.size() * 4));
  memcpy(flags, fl.data(), size_t(m) * G * 4);
  return 0;
}

// One pass of the expander loop (gp_opt.py:557-612) over the next k candidates in visiting
// order with ONE stream synchronisation: top-k behind the cut -> their rows staged on the
// device as the operands of the test -> the exact scan over the unsafe rows -> widths,
// global indices, count and flags in one read-back.  (The step-by-step entry points --
// sgp_grid_topk, _gather_rows, _expander_check -- need a host round trip each; a grid
// without expanders, the usual state of a converged run, visits ALL its candidates every
// iteration.)  One rank: on N ranks the candidates of the shards are merged on the host.
int sgp_grid_expander_batch(sgp_grid* g, sgp_gp* const* gps, int G, double beta,
                            const double* fmin, int mode, double cut_w, int64_t cut_idx,
                            int k, double* w_out, int64_t* gidx_out, int* n_out,
                            int32_t* flags) {
  sgp_ctx* ctx = g->ctx;
  SGP_HIP(ctx, hipSetDevice(ctx->device));
  SGP_CHECK(ctx, k >= 1 && k <= SGP_TOPK, "k = %d not in 1..%d", k, SGP_TOPK);
  SGP_CHECK(ctx, G == g->G, "grid was created for %d GPs, got %d", g->G, G);
  char* res = static_cast<char*>(sgp_scratch(ctx, 1, 2048));
  SGP_CHECK(ctx, res, "device allocation failed: %s", ctx->err.c_str());
  double* wd = reinterpret_cast<double*>(res);
  int64_t* id = reinterpret_cast<int64_t*>(res + 512);
  int* nd = reinterpret_cast<int*>(res + 1024);
  if (mode == 1) cut_w = (cut_idx < 0) ? INFINITY : -double(cut_idx);
  SGP_TRY(launch_topk(g, mode, cut_w, cut_idx, k, wd, id, nd));
  GpDev ghost[SGP_MAX_GPS];
  SGP_TRY(collect_gps(ctx, gps, G, g->d, ghost));
  ExpanderBufs eb;
  SGP_TRY(expander_bufs(g, ghost, G, &eb));
  SGP_TRY(launch_stage_batch(g, id, nd, k, eb.xc, int((eb.bx + eb.bv) / 8), eb.flags,
                             int(eb.bf / 4)));
  int32_t* dfl = nullptr;
  SGP_TRY(enqueue_expander(g, gps, G, beta, fmin, k, nullptr, nullptr, nullptr, 0.0, &dfl,
                           nullptr, true));
  // flags behind the top-k block of the scratch: one read-back for both
  SGP_HIP(ctx, hipMemcpyAsync(res + 1032, dfl, size_t(SGP_TOPK) * G * 4,
                              hipMemcpyDeviceToDevice, ctx->stream));
  char host[2048];
  const size_t span = 1032 + size_t(SGP_TOPK) * G * 4;
  SGP_TRY(sgp_d2h(ctx, host, res, span));
  memcpy(w_out, host, size_t(k) * sizeof(double));
  memcpy(gidx_out, host + 512, size_t(k) * sizeof(int64_t));
  memcpy(n_out, host + 1024, sizeof(int));
  memcpy(flags, host + 1032, size_t(k) * G * 4);
  return 0;
}

// A pass of the expander loop over MANY candidates (gp_opt.py:557-612 where the loop has to
// go far: no expander among the first candidates -- or none at all, the state of a converged
// run -- and full_sets, which visits every safe row).  The next ~`want` candidates behind
// the cut (visiting order: key descending -- the width, or minus the row index in full_sets
// mode -- then index descending), chosen by a histogram of the keys instead of a sort, are
// ALL tested in one scan of the unsafe rows (k_expander_many); two stream synchronisations
// (the size of the pass, its result) whatever the number of candidates.
//   mode 0: out6 = { candidates tested, expanders among them, key and global row of the
//           FIRST expander in visiting order (nothing is marked: the caller settles exact
//           ties and marks), key below which the candidates are still untested (-inf: none
//           left), 0 }
//   mode 1: every expander of the pass is marked in G; out6[2..3] unused.
// key_lo / key_hi: range of the keys still behind the cut (histogram range; mode 0:
// 0 .. the width of the cut).  One rank.
// the result of a pass and, behind it in the same read-back, the arg-max of the step
// (gp_opt.py:631-641 over M | G) for the case that the pass ends the loop without a hit
static int pass_result_and_argmax(sgp_grid* g, const int* list, int count, const int32_t* dfl,
                                  const double* fmin, int mode, const double* scaling,
                                  double* res, double* out6) {
  sgp_ctx* ctx = g->ctx;
  SGP_TRY(launch_pass_result(g, list, count, dfl, fmin, mode, res));
  const bool spec = scaling != nullptr && mode == 0;
  if (spec)
    SGP_TRY(launch_argmax(g, SGP_ARGMAX_MG_WIDTH, scaling, res + 3,
                          reinterpret_cast<int64_t*>(res + 4)));
  double hr[5];
  SGP_TRY(sgp_d2h(ctx, hr, res, sizeof(hr)));
  out6[1] = hr[0];
  out6[2] = hr[1];
  int64_t bi;
  memcpy(&bi, &hr[2], 8);
  out6[3] = double(bi);
  int64_t ai = -1;
  if (spec) memcpy(&ai, &hr[4], 8);
  out6[5] = double(ai);
  return 0;
}

int sgp_grid_expander_pass(sgp_grid* g, sgp_gp* const* gps, int G, double beta,
                           const double* fmin, int mode, double cut_w, int64_t cut_idx,
                           double key_lo, double key_hi, int want, const double* scaling,
                           double* out6) {
  sgp_ctx* ctx = g->ctx;
  SGP_HIP(ctx, hipSetDevice(ctx->device));
  SGP_CHECK(ctx, G == g->G, "grid was created for %d GPs, got %d", g->G, G);
  SGP_CHECK(ctx, want >= 1, "want = %d", want);
  SGP_CHECK(ctx, key_hi > key_lo, "empty key range %g .. %g", key_lo, key_hi);
  SGP_CHECK(ctx, g->N < (int64_t(1) << 31), "%lld rows", (long long)g->N);
  for (int i = 0; i < 6; ++i) out6[i] = 0.0;
  // selection
  const size_t nl = (size_t(g->N) * 4 + 63) & ~size_t(63);      // list | histogram | sel | counts
  char* sb = static_cast<char*>(sgp_scratch(ctx, 8, nl + 16384 + 256 + (size_t(g->N) / 256 + 2) * 4));
  SGP_CHECK(ctx, sb, "device allocation failed: %s", ctx->err.c_str());
  int* list = reinterpret_cast<int*>(sb);
  unsigned* hist = reinterpret_cast<unsigned*>(sb + nl);
  char* sel = sb + nl + 16384;
  int* counts = reinterpret_cast<int*>(sb + nl + 16384 + 256);
  SGP_TRY(launch_pass_select(g, mode, cut_w, cut_idx, key_lo, key_hi, want, sel, list, hist, counts));
  struct { double thr; int count, est; } hs;
  SGP_TRY(sgp_d2h(ctx, &hs, sel, sizeof(hs)));
  const int count = hs.count;
  out6[0] = double(count);
  out6[4] = hs.thr;
  if (count == 0) {
    out6[4] = -INFINITY;
    return 0;
  }
  // operands
  GpDev host[SGP_MAX_GPS];
  SGP_TRY(collect_gps(ctx, gps, G, g->d, host));
  const int d = g->d;
  int np_max = 0;
  for (int i = 0; i < G; ++i) np_max = host[i].n_pad > np_max ? host[i].n_pad : np_max;
  const int64_t wstride = int64_t(np_max / 4) * 64;
  const size_t ngroups = (size_t(count) + 15) / 16;
  const size_t nxc = ngroups * 16 * d, nv = ngroups * G * 16;
  double* ob = static_cast<double*>(sgp_scratch(
      ctx, 9, (nxc + 7 * nv + ngroups * 2 * d + 8 + (ngroups / 8 + 1) * (4 * G + 2 * d)) * 8));
  SGP_CHECK(ctx, ob, "device allocation failed: %s", ctx->err.c_str());
  double* Wp = static_cast<double*>(sgp_scratch(ctx, 10, ngroups * G * size_t(wstride) * 8));
  SGP_CHECK(ctx, Wp, "device allocation failed: %s", ctx->err.c_str());
  int32_t* dfl = static_cast<int32_t*>(sgp_scratch(ctx, 11, ngroups * 16 * G * 4 + 64));
  SGP_CHECK(ctx, dfl, "device allocation failed: %s", ctx->err.c_str());
  double* res = reinterpret_cast<double*>(reinterpret_cast<char*>(dfl) + ngroups * 16 * G * 4);
  res = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(res) + 7) & ~uintptr_t(7));
  double *dxc = ob, *dres = ob + nxc, *ddel = dres + nv, *dis2 = ddel + nv, *dtn2 = dis2 + nv;
  SGP_HIP(ctx, hipMemsetAsync(ob, 0, (nxc + nv) * 8, ctx->stream));
  SGP_HIP(ctx, hipMemsetAsync(dfl, 0, ngroups * 16 * G * 4, ctx->stream));
  SGP_TRY(launch_pass_stage(g, list, count, dxc, dres));
  SGP_TRY(stage_gpdev(g, host, G));
  ExpanderOps ops{};
  ops.xc = dxc;
  ops.resid = dres;
  ops.Wpack = Wp;
  ops.delta = ddel;
  ops.inv_s2 = dis2;
  ops.tn2 = dtn2;
  ops.wstride = wstride;
  ops.m = count;
  ops.Gs = G;
  ExpanderArgs ea{};
  for (int i = 0; i < SGP_MAX_GPS; ++i) {
    ea.fmin[i] = (i < G) ? fmin[i] : -INFINITY;
    ea.active[i] = (i < G) && (fmin[i] != -INFINITY);
    ops.active[i] = ea.active[i];
  }
  SGP_TRY(expander_operands_all(ctx, g->gpdev, host, G, d, ops, nullptr));
  ea.Wpack = Wp;
  ea.xc = dxc;
  ea.delta = ddel;
  ea.inv_s2 = dis2;
  ea.tn2 = dtn2;
  ea.stn = dtn2 + nv;
  ea.svc = dtn2 + 2 * nv;
  ea.agg = dtn2 + 3 * nv;          // (ngroups G 4 <= nv)
  ea.box = dtn2 + 4 * nv;
  ea.sagg = ea.box + ngroups * 2 * d + 8;
  ea.sbox = ea.sagg + (ngroups / 8 + 1) * 4 * G;
  ea.m = count;
  ea.beta = beta;
  ea.S = g->S;
  ea.mean = g->mean;
  ea.var = g->var;
  ea.flags = dfl;
  ea.wstride = wstride;
  ea.near_frac = 0.0;
  SweepPoints sp{g->pts, g->N, 1, g->N};
  SGP_TRY(launch_expander_many(ctx, g->gpdev, G, d, sp, ea));
  return pass_result_and_argmax(g, list, count, dfl, fmin, mode, scaling, res, out6);
}

// The same pass with Lipschitz certificates (gp_opt.py:558-576 instead of :577-606): selection
// as above, then the distance test of ALL listed candidates in one scan (k_lip_*).  out6 as
// sgp_grid_expander_pass.  One rank.
int sgp_grid_lipschitz_pass(sgp_grid* g, int G, const double* fmin, const double* lipschitz,
                            int mode, double cut_w, int64_t cut_idx, double key_lo,
                            double key_hi, int want, const double* scaling, double* out6) {
  sgp_ctx* ctx = g->ctx;
  SGP_HIP(ctx, hipSetDevice(ctx->device));
  SGP_CHECK(ctx, G == g->G, "grid was created for %d GPs, got %d", g->G, G);
  SGP_CHECK(ctx, want >= 1, "want = %d", want);
  SGP_CHECK(ctx, key_hi > key_lo, "empty key range %g .. %g", key_lo, key_hi);
  SGP_CHECK(ctx, g->N < (int64_t(1) << 31), "%lld rows", (long long)g->N);
  for (int i = 0; i < 6; ++i) out6[i] = 0.0;
  const size_t nl = (size_t(g->N) * 4 + 63) & ~size_t(63);      // list | histogram | sel | counts
  char* sb = static_cast<char*>(sgp_scratch(ctx, 8, nl + 16384 + 256 + (size_t(g->N) / 256 + 2) * 4));
  SGP_CHECK(ctx, sb, "device allocation failed: %s", ctx->err.c_str());
  int* list = reinterpret_cast<int*>(sb);
  unsigned* hist = reinterpret_cast<unsigned*>(sb + nl);
  char* sel = sb + nl + 16384;
  int* counts = reinterpret_cast<int*>(sb + nl + 16384 + 256);
  SGP_TRY(launch_pass_select(g, mode, cut_w, cut_idx, key_lo, key_hi, want, sel, list, hist, counts));
  struct { double thr; int count, est; } hs;
  SGP_TRY(sgp_d2h(ctx, &hs, sel, sizeof(hs)));
  const int count = hs.count;
  out6[0] = double(count);
  out6[4] = hs.thr;
  if (count == 0) {
    out6[4] = -INFINITY;
    return 0;
  }
  const int d = g->d;
  const size_t ngroups = (size_t(count) + 15) / 16;
  double* work = static_cast<double*>(
      sgp_scratch(ctx, 9, (size_t(count) * (d + G) + ngroups * (2 * d + 1) + 8) * 8));
  SGP_CHECK(ctx, work, "device allocation failed: %s", ctx->err.c_str());
  int32_t* dfl = static_cast<int32_t*>(sgp_scratch(ctx, 11, size_t(count) * G * 4 + 64));
  SGP_CHECK(ctx, dfl, "device allocation failed: %s", ctx->err.c_str());
  double* res = reinterpret_cast<double*>(reinterpret_cast<char*>(dfl) + size_t(count) * G * 4);
  res = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(res) + 7) & ~uintptr_t(7));
  SGP_TRY(launch_lipschitz_many(g, G, fmin, lipschitz, list, count, nullptr, nullptr, work, dfl));
  return pass_result_and_argmax(g, list, count, dfl, fmin, mode, scaling, res, out6);
}

// ---- the same pass on N ranks, in three calls with the ranks' agreement in between ----------
// (SafeOpt._visit_in_big_passes_nrank): every rank's histogram of the keys behind the cut
// (the ranks sum them and pick ONE threshold), every rank's candidates above it with what the
// other ranks need of them (the ranks gather them: the same list everywhere), and the test of
// ALL those candidates against this rank's unsafe rows (the ranks OR the flags).
int sgp_grid_pass_hist(sgp_grid* g, int mode, double cut_w, int64_t cut_idx, double key_lo,
                       double key_hi, uint32_t* hist) {
  sgp_ctx* ctx = g->ctx;
  SGP_HIP(ctx, hipSetDevice(ctx->device));
  SGP_CHECK(ctx, key_hi > key_lo, "empty key range %g .. %g", key_lo, key_hi);
  const size_t nl = (size_t(g->N) * 4 + 63) & ~size_t(63);      // list | histogram | sel | counts
  char* sb = static_cast<char*>(sgp_scratch(ctx, 8, nl + 16384 + 256 + (size_t(g->N) / 256 + 2) * 4));
  SGP_CHECK(ctx, sb, "device allocation failed: %s", ctx->err.c_str());
  unsigned* dh = reinterpret_cast<unsigned*>(sb + nl);
  SGP_TRY(launch_pass_hist(g, mode, cut_w, cut_idx, key_lo, key_hi, dh));
  return sgp_d2h(ctx, hist, dh, 4096 * sizeof(uint32_t));
}

int sgp_grid_pass_list(sgp_grid* g, int mode, double cut_w, int64_t cut_idx, double thr, int cap,
                       int* count, int64_t* gidx, double* key, double* x, double* resid) {
  sgp_ctx* ctx = g->ctx;
  SGP_HIP(ctx, hipSetDevice(ctx->device));
  SGP_CHECK(ctx, g->N < (int64_t(1) << 31), "%lld rows", (long long)g->N);
  const size_t nl = (size_t(g->N) * 4 + 63) & ~size_t(63);      // list | histogram | sel | counts
  char* sb = static_cast<char*>(sgp_scratch(ctx, 8, nl + 16384 + 256 + (size_t(g->N) / 256 + 2) * 4));
  SGP_CHECK(ctx, sb, "device allocation failed: %s", ctx->err.c_str());
  int* list = reinterpret_cast<int*>(sb);
  char* sel = sb + nl + 16384;
  struct { double thr; int count, est; } hs = {thr, 0, 0};
  SGP_TRY(sgp_h2d(ctx, sel, &hs, sizeof(hs)));
  int* counts = reinterpret_cast<int*>(sb + nl + 16384 + 256);
  const int upper = mode & 2;            // resid = u_i (Lipschitz certificates) instead of u_i - mu_i
  mode &= 1;
  SGP_TRY(launch_pass_list(g, mode, cut_w, cut_idx, sel, list, counts));
  SGP_TRY(sgp_d2h(ctx, &hs, sel, sizeof(hs)));
  *count = hs.count;
  if (hs.count == 0) return 0;
  SGP_CHECK(ctx, hs.count <= cap, "%d candidates above the threshold, room for %d", hs.count, cap);
  const int d = g->d, G = g->G;
  const size_t per = 2 + size_t(d) + size_t(G);
  double* ob = static_cast<double*>(sgp_scratch(ctx, 9, size_t(hs.count) * per * 8));
  SGP_CHECK(ctx, ob, "device allocation failed: %s", ctx->err.c_str());
  int64_t* dg = reinterpret_cast<int64_t*>(ob);
  double *dk = ob + hs.count, *dx = dk + hs.count, *dr = dx + size_t(hs.count) * d;
  SGP_TRY(launch_pass_gather(g, list, hs.count, mode | upper, dg, dk, dx, dr));
  std::vector<double> host(size_t(hs.count) * per);
  SGP_TRY(sgp_d2h(ctx, host.data(), ob, host.size() * 8));
  memcpy(gidx, host.data(), size_t(hs.count) * 8);
  memcpy(key, host.data() + hs.count, size_t(hs.count) * 8);
  memcpy(x, host.data() + 2 * size_t(hs.count), size_t(hs.count) * d * 8);
  memcpy(resid, host.data() + size_t(hs.count) * (2 + d), size_t(hs.count) * G * 8);
  return 0;
}

// xc [K][d], resid [K][G] (u_g - mu_g at the candidate): flags[c * G + i] != 0 when candidate c
// lifts one of THIS shard's unsafe rows above fmin_i.
int sgp_grid_pass_test(sgp_grid* g, sgp_gp* const* gps, int G, double beta, const double* fmin,
                       int K, const double* xc, const double* resid, int32_t* flags) {
  sgp_ctx* ctx = g->ctx;
  SGP_HIP(ctx, hipSetDevice(ctx->device));
  SGP_CHECK(ctx, G == g->G, "grid was created for %d GPs, got %d", g->G, G);
  if (K <= 0) return 0;
  GpDev host[SGP_MAX_GPS];
  SGP_TRY(collect_gps(ctx, gps, G, g->d, host));
  const int d = g->d;
  int np_max = 0;
  for (int i = 0; i < G; ++i) np_max = host[i].n_pad > np_max ? host[i].n_pad : np_max;
  const int64_t wstride = int64_t(np_max / 4) * 64;
  const size_t ngroups = (size_t(K) + 15) / 16;
  const size_t nxc = ngroups * 16 * d, nv = ngroups * G * 16;
  double* ob = static_cast<double*>(sgp_scratch(
      ctx, 9, (nxc + 7 * nv + ngroups * 2 * d + 8 + (ngroups / 8 + 1) * (4 * G + 2 * d)) * 8));
  SGP_CHECK(ctx, ob, "device allocation failed: %s", ctx->err.c_str());
  double* Wp = static_cast<double*>(sgp_scratch(ctx, 10, ngroups * G * size_t(wstride) * 8));
  SGP_CHECK(ctx, Wp, "device allocation failed: %s", ctx->err.c_str());
  int32_t* dfl = static_cast<int32_t*>(sgp_scratch(ctx, 11, ngroups * 16 * G * 4 + 64));
  SGP_CHECK(ctx, dfl, "device allocation failed: %s", ctx->err.c_str());
  double *dxc = ob, *dres = ob + nxc, *ddel = dres + nv, *dis2 = ddel + nv, *dtn2 = dis2 + nv;
  // operand block on the host: xc | resid[group][g][16]
  std::vector<double> st(nxc + nv, 0.0);
  memcpy(st.data(), xc, size_t(K) * d * 8);
  for (int c = 0; c < K; ++c)
    for (int i = 0; i < G; ++i)
      st[nxc + (size_t(c >> 4) * G + i) * 16 + (c & 15)] = resid[size_t(c) * G + i];
  SGP_TRY(sgp_h2d(ctx, ob, st.data(), st.size() * 8));
  SGP_HIP(ctx, hipMemsetAsync(dfl, 0, ngroups * 16 * G * 4, ctx->stream));
  SGP_TRY(stage_gpdev(g, host, G));
  ExpanderOps ops{};
  ops.xc = dxc;
  ops.resid = dres;
  ops.Wpack = Wp;
  ops.delta = ddel;
  ops.inv_s2 = dis2;
  ops.tn2 = dtn2;
  ops.wstride = wstride;
  ops.m = K;
  ops.Gs = G;
  ExpanderArgs ea{};
  for (int i = 0; i < SGP_MAX_GPS; ++i) {
    ea.fmin[i] = (i < G) ? fmin[i] : -INFINITY;
    ea.active[i] = (i < G) && (fmin[i] != -INFINITY);
    ops.active[i] = ea.active[i];
  }
  SGP_TRY(expander_operands_all(ctx, g->gpdev, host, G, d, ops, nullptr));
  ea.Wpack = Wp;
  ea.xc = dxc;
  ea.delta = ddel;
  ea.inv_s2 = dis2;
  ea.tn2 = dtn2;
  ea.stn = dtn2 + nv;
  ea.svc = dtn2 + 2 * nv;
  ea.agg = dtn2 + 3 * nv;          // (ngroups G 4 <= nv)
  ea.box = dtn2 + 4 * nv;
  ea.sagg = ea.box + ngroups * 2 * d + 8;
  ea.sbox = ea.sagg + (ngroups / 8 + 1) * 4 * G;
  ea.m = K;
  ea.beta = beta;
  ea.S = g->S;
  ea.mean = g->mean;
  ea.var = g->var;
  ea.flags = dfl;
  ea.wstride = wstride;
  ea.near_frac = 0.0;
  SweepPoints sp{g->pts, g->N, 1, g->N};
  SGP_TRY(launch_expander_many(ctx, g->gpdev, G, d, sp, ea));
  return sgp_d2h(ctx, flags, dfl, size_t(K) * G * 4);
}

// ... and with Lipschitz certificates: the distance test of ALL K gathered candidates (rows xc
// [K][d], upper bounds uc [K][G]: sgp_grid_pass_list with mode | 2) against this shard's unsafe
// rows; flags[c * G + i] != 0: some unsafe row of the shard is in reach of candidate c for every
// GP with a constraint (one flag per candidate, see k_lip_items) -- the ranks OR them.
int sgp_grid_pass_lipschitz_test(sgp_grid* g, int G, const double* fmin, const double* lipschitz,
                                 int K, const double* xc, const double* uc, int32_t* flags) {
  sgp_ctx* ctx = g->ctx;
  SGP_HIP(ctx, hipSetDevice(ctx->device));
  SGP_CHECK(ctx, G == g->G, "grid was created for %d GPs, got %d", g->G, G);
  if (K <= 0) return 0;
  const int d = g->d;
  const size_t ngroups = (size_t(K) + 15) / 16;
  double* work = static_cast<double*>(
      sgp_scratch(ctx, 9, (size_t(K) * (d + G) + ngroups * (2 * d + 1) + 8) * 8));
  SGP_CHECK(ctx, work, "device allocation failed: %s", ctx->err.c_str());
  int32_t* dfl = static_cast<int32_t*>(sgp_scratch(ctx, 11, size_t(K) * G * 4 + 64));
  SGP_CHECK(ctx, dfl, "device allocation failed: %s", ctx->err.c_str());
  SGP_TRY(launch_lipschitz_many(g, G, fmin, lipschitz, nullptr, K, xc, uc, work, dfl));
  return sgp_d2h(ctx, flags, dfl, size_t(K) * G * 4);
}

// Host copy of a front block ([0] max width | [1..2] counts (u64) | [3] w_top | [4]
// idx_top (i64) | [5] n_found, n_tied (int) | x[d] | mean[G] | q[2G]) -> the caller's
// arrays; out5[4] = -1 when the shard / grid has no candidate, out5[5] = number of
// candidates that share the first one's width.
static void unpack_front(const double* host, int d, int G, double* out5, double* x_top,
                         double* mean_top, double* q_top) {
  unsigned long long cnt[2];
  int64_t idx;
  int nfound, ntied;
  memcpy(cnt, &host[1], 16);
  memcpy(&idx, &host[4], 8);
  memcpy(&nfound, &host[5], 4);
  memcpy(&ntied, reinterpret_cast<const char*>(&host[5]) + 4, 4);
  out5[0] = host[0];
  out5[1] = double(cnt[0]);
  out5[2] = double(cnt[1]);
  out5[3] = host[3];
  out5[4] = (nfound > 0) ? double(idx) : -1.0;
  out5[5] = double(ntied);
  memcpy(x_top, &host[6], size_t(d) * 8);
  memcpy(mean_top, &host[6 + d], size_t(G) * 8);
  memcpy(q_top, &host[6 + d + G], size_t(2 * G) * 8);
}

// Single-rank fast path, front half of compute_sets (gp_opt.py:511-552) with
// ONE stream sync: M, max_var, candidate mask, counts and the first candidate
// in visiting order together with its rows.
int sgp_grid_sets_front(sgp_grid* g, double max_l, int have_max_var,
                        double max_var, const double* scaling,
                        const double* thr_beta, double* out5, double* x_top,
                        double* mean_top, double* q_top) {
  sgp_ctx* ctx = g->ctx;
  SGP_HIP(ctx, hipSetDevice(ctx->device));
  const int d = g->d, G = g->G;
  // result block: [0] max width | [1..2] counts (u64) | [3] w_top | [4] idx_top
  //               (i64) | [5] n_found (int) | x[d] | mean[G] | q[2G]
  const size_t nres = 6 + size_t(d) + 3 * size_t(G);
  double* res = static_cast<double*>(sgp_scratch(ctx, 1, (nres + 8) * 8));
  SGP_CHECK(ctx, res, "device allocation failed: %s", ctx->err.c_str());
  if (have_max_var) {
    // multi-rank: M is already set (sgp_grid_maximizers) and max_var is the
    // all-reduced value
    SGP_HIP(ctx, hipMemsetAsync(res, 0, 8, ctx->stream));
    SGP_TRY(launch_candidates(g, max_var, nullptr, scaling, thr_beta, 0,
                              reinterpret_cast<unsigned long long*>(res + 1)));
  } else {
    SGP_TRY(launch_maximizers(g, max_l));
    SGP_TRY(launch_reduce_max(ctx, g->partial, (g->N + 255) / 256, res));
    SGP_TRY(launch_candidates(g, 0.0, res, scaling, thr_beta, 0,
                              reinterpret_cast<unsigned long long*>(res + 1)));
  }
  SGP_TRY(launch_topk(g, 0, INFINITY, INT64_MAX, 1, res + 3,
                      reinterpret_cast<int64_t*>(res + 4),
                      reinterpret_cast<int*>(res + 5)));
  // (the fused single-rank pass counts the ties in k_front_final; here nothing
  // else writes that word)
  SGP_TRY(launch_count_ties(g, res + 3, reinterpret_cast<const int*>(res + 5),
                            reinterpret_cast<int*>(res + 5) + 1));
  SGP_TRY(launch_gather_top(g, reinterpret_cast<int64_t*>(res + 4), res + 6,
                            res + 6 + d, res + 6 + d + G));
  std::vector<double> host(nres);
  SGP_TRY(sgp_d2h(ctx, host.data(), res, nres * 8));
  unpack_front(host.data(), d, G, out5, x_top, mean_top, q_top);
  return 0;
}

// ---- the collectives of the N-rank step on DEVICE operands -------------------------
// Two transports behind one table: RCCL on the context's stream (sgp_comm_init: the
// collective is one more operation in the stream, no host round trip), or the caller's
// own collectives on host buffers (sgp_comm_init_host: stream sync + D2H, the callback,
// H2D behind it -- ranks without a GPU each / without xGMI between them, and the way the
// in-stream step is exercised with real foreign data on a one-GPU box).
static bool have_comm(const sgp_ctx* ctx) {
  return ctx->comm != nullptr || ctx->hostcomm.allgather != nullptr;
}
static int coll_allreduce_max_f64(sgp_ctx* ctx, double* dev, size_t n) {
  if (ctx->comm) {
    SGP_NCCL(ctx, g_rccl.AllReduce(dev, dev, n, ncclFloat64, ncclMax,
                                   static_cast<ncclComm_t>(ctx->comm), ctx->stream));
    return 0;
  }
  if (!ctx->hostcomm.allreduce_max_f64) return 0;
  std::vector<double> h(n);
  SGP_TRY(sgp_d2h(ctx, h.data(), dev, n * 8));
  SGP_CHECK(ctx, ctx->hostcomm.allreduce_max_f64(ctx->hostcomm.user, h.data(), int(n)) == 0,
            "host collective (all-reduce max, %zu f64) failed", n);
  return sgp_h2d(ctx, dev, h.data(), n * 8);
}

static int coll_allreduce_max_i32(sgp_ctx* ctx, int32_t* dev, size_t n) {
  if (ctx->comm) {
    SGP_NCCL(ctx, g_rccl.AllReduce(dev, dev, n, ncclInt32, ncclMax,
                                   static_cast<ncclComm_t>(ctx->comm), ctx->stream));
    return 0;
  }
  if (!ctx->hostcomm.allreduce_max_i32) return 0;
  std::vector<int32_t> h(n);
  SGP_TRY(sgp_d2h(ctx, h.data(), dev, n * 4));
  SGP_CHECK(ctx, ctx->hostcomm.allreduce_max_i32(ctx->hostcomm.user, h.data(), int(n)) == 0,
            "host collective (all-reduce max, %zu i32) failed", n);
  return sgp_h2d(ctx, dev, h.data(), n * 4);
}

// recv (device) = the nbytes of every rank, in rank order
static int coll_allgather(sgp_ctx* ctx, const void* send, void* recv, size_t nbytes) {
  if (ctx->comm) {
    SGP_NCCL(ctx, g_rccl.AllGather(send, recv, nbytes, ncclInt8,
                                   static_cast<ncclComm_t>(ctx->comm), ctx->stream));
    return 0;
  }
  if (!ctx->hostcomm.allgather) {
    SGP_HIP(ctx, hipMemcpyAsync(recv, send, nbytes, hipMemcpyDeviceToDevice, ctx->stream));
    return 0;
  }
  std::vector<char> hs(nbytes), hr(nbytes * size_t(ctx->world));
  SGP_TRY(sgp_d2h(ctx, hs.data(), send, nbytes));
  SGP_CHECK(ctx, ctx->hostcomm.allgather(ctx->hostcomm.user, hs.data(), hr.data(),
                                         int64_t(nbytes)) == 0,
            "host collective (all-gather, %zu bytes per rank) failed", nbytes);
  return sgp_h2d(ctx, recv, hr.data(), hr.size());
}

// Front half of compute_sets on this rank's shard with the cross-rank scalars kept
// on the device: max l0[S] (left in g->scal[0] by a confidence pass without
// read-back) and the maximiser width are all-reduced IN STREAM, the kernels read
// them from device memory; `res` (device) receives the front block of
// unpack_front.  Without a communicator (one rank) the all-reduces are skipped.
static int front_half_in_stream(sgp_grid* g, const double* scaling, const double* thr_beta,
                                double* res) {
  sgp_ctx* ctx = g->ctx;
  const int d = g->d, G = g->G;
  SGP_TRY(settle_max_l(g));
  const bool comm = have_comm(ctx);
  SGP_CHECK(ctx, comm || ctx->world <= 1,
            "rank %d of %d has no communicator in the grid's context: the "
            "in-stream collectives cannot run (sgp_comm_init on THIS context)",
            ctx->rank, ctx->world);
  if (comm) SGP_TRY(coll_allreduce_max_f64(ctx, g->scal, 1));
  SGP_TRY(launch_maximizers(g, 0.0, g->scal));
  SGP_TRY(launch_reduce_max(ctx, g->partial, (g->N + 255) / 256, res));
  if (comm) SGP_TRY(coll_allreduce_max_f64(ctx, res, 1));
  SGP_TRY(launch_candidates(g, 0.0, res, scaling, thr_beta, 0,
                            reinterpret_cast<unsigned long long*>(res + 1)));
  SGP_TRY(launch_topk(g, 0, INFINITY, INT64_MAX, 1, res + 3,
                      reinterpret_cast<int64_t*>(res + 4),
                      reinterpret_cast<int*>(res + 5)));
  // (the fused single-rank pass counts the ties in k_front_final; here nothing
  // else writes that word)
  SGP_TRY(launch_count_ties(g, res + 3, reinterpret_cast<const int*>(res + 5),
                            reinterpret_cast<int*>(res + 5) + 1));
  SGP_TRY(launch_gather_top(g, reinterpret_cast<int64_t*>(res + 4), res + 6,
                            res + 6 + d, res + 6 + d + G));
  return 0;
}

// N-rank front half with the cross-rank scalars kept on the device: max l0[S]
// (left in g->scal[0] by a confidence pass without read-back) and the maximiser
// width are all-reduced IN STREAM (RCCL on the context's stream), the kernels
// read them from device memory, and one read-back returns this rank's counts,
// its first candidate and the global max l0.  Without a communicator (one
// rank) the all-reduces are skipped.
int sgp_grid_sets_front_comm(sgp_grid* g, const double* scaling,
                             const double* thr_beta, double* out5, double* x_top,
                             double* mean_top, double* q_top, double* max_l_out) {
  sgp_ctx* ctx = g->ctx;
  SGP_HIP(ctx, hipSetDevice(ctx->device));
  const int d = g->d, G = g->G;
  const size_t nres = 7 + size_t(d) + 3 * size_t(G);   // front block + max_l
  double* res = static_cast<double*>(sgp_scratch(ctx, 1, (nres + 8) * 8));
  SGP_CHECK(ctx, res, "device allocation failed: %s", ctx->err.c_str());
  SGP_TRY(front_half_in_stream(g, scaling, thr_beta, res));
  SGP_HIP(ctx, hipMemcpyAsync(res + nres - 1, g->scal, 8,
                              hipMemcpyDeviceToDevice, ctx->stream));
  std::vector<double> host(nres);
  SGP_TRY(sgp_d2h(ctx, host.data(), res, nres * 8));
  unpack_front(host.data(), d, G, out5, x_top, mean_top, q_top);
  *max_l_out = host[nres - 1];
  return 0;
}

// Back half for ONE candidate (the common case: the first candidate is the
// expander): probe scan, conditional G mark and the M|G arg-max with one sync.
int sgp_grid_sets_back(sgp_grid* g, sgp_gp* const* gps, int G, double beta,
                       const double* fmin, const double* xc, const double* mu_c,
                       const double* u_c, double near_frac, int64_t gidx_c,
                       int mark, const double* scaling, int32_t* flags,
                       double* value, int64_t* gidx) {
  sgp_ctx* ctx = g->ctx;
  SGP_HIP(ctx, hipSetDevice(ctx->device));
  const int64_t li = gidx_c - g->goff;
  SGP_CHECK(ctx, !mark || (li >= 0 && li < g->N),
            "candidate %lld is not owned by this shard", (long long)gidx_c);
  int32_t* dfl = nullptr;
  SGP_TRY(enqueue_expander(g, gps, G, beta, fmin, 1, xc, mu_c, u_c, near_frac,
                           &dfl));
  if (mark) SGP_TRY(launch_mark_if(g, li, dfl, fmin));
  // results right behind the flags block: value (f64) | index (i64)
  char* res = reinterpret_cast<char*>(dfl) + size_t(SGP_TOPK) * G * 4;
  res += (8 - (reinterpret_cast<uintptr_t>(res) & 7)) & 7;
  SGP_TRY(launch_argmax(g, SGP_ARGMAX_MG_WIDTH, scaling,
                        reinterpret_cast<double*>(res),
                        reinterpret_cast<int64_t*>(res + 8)));
  const size_t span = size_t(res + 16 - reinterpret_cast<char*>(dfl));
  std::vector<char> host(span);
  SGP_TRY(sgp_d2h(ctx, host.data(), dfl, span));
  memcpy(flags, host.data(), size_t(G) * 4);
  memcpy(value, host.data() + (span - 16), 8);
  memcpy(gidx, host.data() + (span - 8), 8);
  return 0;
}

// Single-rank fast path, both halves with ONE stream sync: the front half
// leaves the first candidate on the device, the probe scan runs on it right
// away, G is marked if it is certified and the M|G arg-max follows.  The host
// decides afterwards which of the results apply (no candidate / nothing unsafe:
// the back results are void except the arg-max; candidate not certified: the
// exact scan and the general loop take over).
int sgp_grid_sets_fused(sgp_grid* g, sgp_gp* const* gps, int G, double beta,
                        const double* fmin, double max_l, const double* scaling,
                        const double* thr_beta, double near_frac, double* out5,
                        double* x_top, double* mean_top, double* q_top,
                        int32_t* flags, double* value, int64_t* gidx,
                        double* max_l_out) {
  sgp_ctx* ctx = g->ctx;
  SGP_HIP(ctx, hipSetDevice(ctx->device));
  const int d = g->d;
  SGP_CHECK(ctx, G == g->G, "grid was created for %d GPs, got %d", g->G, G);
  // result block: [0] max width | [1..2] counts (u64) | [3] w_top | [4] idx_top
  //   (i64) | [5] n_found (int) | x[d] | mean[G] | q[2G] | flags[G] (i32, padded
  //   to 8 B) | value | index (i64) | max_l (when it was resident)
  const size_t nfront = 6 + size_t(d) + 3 * size_t(G);
  const size_t nfl = (size_t(G) + 1) / 2;
  const size_t nres = nfront + nfl + 3;
  double* res = static_cast<double*>(sgp_scratch(ctx, 1, (nres + 8) * 8));
  SGP_CHECK(ctx, res, "device allocation failed: %s", ctx->err.c_str());
  // max_l = NaN: the confidence pass was issued without read-back; the value
  // is in g->scal[0] or still spread over the sweep's per-wave partials
  const bool resident = max_l != max_l;
  const bool pending = resident && g->l0_pending > 0;
  GpDev ghost[SGP_MAX_GPS];
  SGP_TRY(collect_gps(ctx, gps, G, d, ghost));
  ExpanderBufs eb;
  SGP_TRY(expander_bufs(g, ghost, G, &eb));
  // Seven launches and one stream synchronisation: maximisers -> candidates (+ the first
  // one per workgroup) -> [k_expkt: first candidate of the shard, staged as the operand of
  // the expander test, AND L^-1 k_c] -> k_expw1 -> pre-filter -> listed rows -> conditional
  // G mark + M|G arg-max per workgroup.  The reductions in between are folded into the
  // consumers; the result block and the arg-max partials land in mapped host memory, where
  // the last level of the arg-max is taken (no final launch, no read-back copy).
  const int nbp = argmax_marked_blocks(g->N);
  const size_t need = nres + 8 + 2 * size_t(nbp);
  const bool zero_copy = need * 8 <= (size_t(4) << 20) && !getenv("SGP_SETS_COPY");
  if (zero_copy && ctx->sets_cap < need) {
    if (ctx->sets_host) (void)hipHostFree(ctx->sets_host);
    ctx->sets_host = nullptr;
    ctx->sets_cap = 0;
    void* h = nullptr;
    const size_t cap = std::max<size_t>(need, 4096);
    SGP_HIP(ctx, hipHostMalloc(&h, cap * 8, hipHostMallocMapped | hipHostMallocCoherent));
    memset(h, 0, cap * 8);
    void* dv = nullptr;
    SGP_HIP(ctx, hipHostGetDevicePointer(&dv, h, 0));
    ctx->sets_host = static_cast<double*>(h);
    ctx->sets_dev = static_cast<double*>(dv);
    ctx->sets_cap = cap;
  }
  if (!zero_copy) {
    // (grids beyond ~2.5e8 rows, or SGP_SETS_COPY=1: the round-4 chain -- final launches
    // and one read-back copy)
    SGP_TRY(launch_sets_front_fused(
        g, max_l, pending ? g->partial : nullptr, g->l0_pending,
        (resident && !pending) ? g->scal : nullptr, scaling, thr_beta, res,
        res + nfront + nfl + 2, eb.xc, int((eb.bx + eb.bv) / 8), eb.flags,
        int(eb.bf / 4)));
    g->l0_pending = 0;
    int32_t* dfl = nullptr;
    SGP_TRY(enqueue_expander(g, gps, G, beta, fmin, 1, nullptr, nullptr, nullptr,
                             near_frac, &dfl, nullptr, true));
    SGP_TRY(launch_argmax_marked(
        g, scaling, fmin, dfl, reinterpret_cast<int64_t*>(res + 4),
        reinterpret_cast<int*>(res + 5), reinterpret_cast<int32_t*>(res + nfront),
        res + nfront + nfl, reinterpret_cast<int64_t*>(res + nfront + nfl + 1)));
    std::vector<double> host(nres);
    SGP_TRY(sgp_d2h(ctx, host.data(), res, nres * 8));
    unpack_front(host.data(), d, G, out5, x_top, mean_top, q_top);
    memcpy(flags, &host[nfront], size_t(G) * 4);
    *value = host[nfront + nfl];
    memcpy(gidx, &host[nfront + nfl + 1], 8);
    if (max_l_out) *max_l_out = resident ? host[nfront + nfl + 2] : max_l;
    return 0;
  }
  double* hres = ctx->sets_host;
  double* dres = ctx->sets_dev;
  double* hpart = hres + nres + 8 - (nres & 1);      // (16-byte aligned pairs)
  double* dpart = dres + (hpart - hres);
  FrontArgs fold{};
  SGP_TRY(launch_sets_front_fused(
      g, max_l, pending ? g->partial : nullptr, g->l0_pending,
      (resident && !pending) ? g->scal : nullptr, scaling, thr_beta, res,
      dres + nfront + nfl + 2, eb.xc, int((eb.bx + eb.bv) / 8), eb.flags,
      int(eb.bf / 4), &fold));
  fold.res_host = dres;
  g->l0_pending = 0;
  int32_t* dfl = nullptr;
  SGP_TRY(enqueue_expander(g, gps, G, beta, fmin, 1, nullptr, nullptr, nullptr,
                           near_frac, &dfl, nullptr, true, &fold));
  SGP_TRY(launch_argmax_marked(
      g, scaling, fmin, dfl, reinterpret_cast<int64_t*>(res + 4),
      reinterpret_cast<int*>(res + 5), reinterpret_cast<int32_t*>(dres + nfront),
      nullptr, nullptr, dpart));
  SGP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  unpack_front(hres, d, G, out5, x_top, mean_top, q_top);
  memcpy(flags, &hres[nfront], size_t(G) * 4);
  {   // np.argmax over the workgroups' results: largest value, first index among equals
    const int64_t* hidx = reinterpret_cast<const int64_t*>(hpart + nbp);
    double bv = -INFINITY;
    int64_t bi = -1;
    for (int e = 0; e < nbp; ++e) {
      const int64_t i = hidx[e];
      if (i < 0) continue;
      const double v = hpart[e];
      if (bi < 0 || v > bv || (v == bv && i < bi)) {
        bv = v;
        bi = i;
      }
    }
    *value = bv;
    *gidx = bi;
  }
  if (max_l_out) *max_l_out = resident ? hres[nfront + nfl + 2] : max_l;
  return 0;
}

// A whole SafeOpt.optimize() of a SMALL grid -- intervals, S, M, candidates, the exact
// expander test of the first candidate, the G mark and the arg-max -- in ONE launch of one
// workgroup and one read-back (step_small.hip).  Same result block as sgp_grid_sets_fused.
int sgp_grid_step_small(sgp_grid* g, sgp_gp* const* gps, int G, double beta,
                        const double* fmin, const double* scaling, const double* thr_beta,
                        double* out5, double* x_top, double* mean_top, double* q_top,
                        int32_t* flags, double* value, int64_t* gidx, double* max_l_out) {
  sgp_ctx* ctx = g->ctx;
  SGP_HIP(ctx, hipSetDevice(ctx->device));
  const int d = g->d;
  SGP_CHECK(ctx, G == g->G, "grid was created for %d GPs, got %d", g->G, G);
  GpDev host[SGP_MAX_GPS];
  SGP_TRY(collect_gps(ctx, gps, G, d, host));
  SGP_CHECK(ctx, step_small_eligible(ctx, host, G, g->N),
            "sgp_grid_step_small: %lld rows / a GP with more than 48 observations "
            "(sgp_grid_step_small_ok)", (long long)g->N);
  SGP_TRY(stage_gpdev(g, host, G));
  const size_t nfront = 6 + size_t(d) + 3 * size_t(G);
  const size_t nfl = (size_t(G) + 1) / 2;
  const size_t nres = nfront + nfl + 3;
  SGP_CHECK(ctx, nres < size_t(kStepResWords), "result block of %zu words", nres);
  if (!ctx->step_host) {
    void* h = nullptr;
    SGP_HIP(ctx, hipHostMalloc(&h, kStepResWords * 8, hipHostMallocMapped | hipHostMallocCoherent));
    memset(h, 0, kStepResWords * 8);
    void* dv = nullptr;
    SGP_HIP(ctx, hipHostGetDevicePointer(&dv, h, 0));
    ctx->step_host = static_cast<double*>(h);
    ctx->step_dev = static_cast<double*>(dv);
  }
  g->l0_pending = 0;
  const uint64_t seq = ++ctx->step_seq;
  SGP_TRY(launch_step_small(g, g->gpdev, host, G, beta, fmin, scaling, thr_beta, ctx->step_dev,
                            int(nfront), int(nfl), seq));
  // No read-back copy, no stream synchronisation: the kernel writes its result block into
  // host memory and a completion word behind it; spin on that word (a launch of 20-40 us),
  // fall back to the stream when it does not turn up.
  {
    volatile uint64_t* done = reinterpret_cast<volatile uint64_t*>(ctx->step_host) +
                              (kStepResWords - 1);
    const auto t0 = std::chrono::steady_clock::now();
    uint32_t spins = 0;
    while (*done != seq) {
      if ((++spins & 1023u) == 0 &&
          std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) {
        SGP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        SGP_CHECK(ctx, *done == seq, "sgp_grid_step_small: the kernel left no result");
        break;
      }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
  }
  double hostres[kStepResWords];
  memcpy(hostres, ctx->step_host, nres * 8);
  if (getenv("SGP_STEP_STAMPS")) {      // (-DSTEP_STAMPS builds of step_small.hip)
    const uint64_t* st = reinterpret_cast<const uint64_t*>(ctx->step_host) + 40;
    static int shown = 0;
    if (++shown % 500 == 0) {
      fprintf(stderr, "step stamps (ticks):");
      for (int i = 1; i < 8; ++i) fprintf(stderr, " %d:%lld", i, (long long)(st[i] - st[i - 1]));
      fprintf(stderr, "\n");
    }
  }
  unpack_front(hostres, d, G, out5, x_top, mean_top, q_top);
  memcpy(flags, &hostres[nfront], size_t(G) * 4);
  *value = hostres[nfront + nfl];
  memcpy(gidx, &hostres[nfront + nfl + 1], 8);
  *max_l_out = hostres[nfront + nfl + 2];
  return 0;
}

// Every candidate of a small grid at once (step_small.hip: k_cand_ops, k_cand_scan):
// flags[c * G + i] != 0 when candidate gidx[c] lifts an unsafe row above fmin_i after the
// rank-1 update by (x_c, u_i(x_c)) -- the test of gp_opt.py:579-606 for m candidates in
// two launches and one round trip.  Grids of at most 16384 rows, GPs with at most 48
// observations (sgp_grid_step_small_ok).
int sgp_grid_expanders_small(sgp_grid* g, sgp_gp* const* gps, int G, double beta,
                             const double* fmin, const int64_t* gidx, int m, int32_t* flags) {
  sgp_ctx* ctx = g->ctx;
  SGP_HIP(ctx, hipSetDevice(ctx->device));
  SGP_CHECK(ctx, G == g->G, "grid was created for %d GPs, got %d", g->G, G);
  if (m <= 0) return 0;
  GpDev host[SGP_MAX_GPS];
  SGP_TRY(collect_gps(ctx, gps, G, g->d, host));
  SGP_CHECK(ctx, step_small_eligible(ctx, host, G, g->N),
            "sgp_grid_expanders_small: %lld rows / a GP with more than 48 observations",
            (long long)g->N);
  for (int c = 0; c < m; ++c)
    SGP_CHECK(ctx, gidx[c] >= g->goff && gidx[c] < g->goff + g->N,
              "candidate %lld is not a row of this grid", (long long)gidx[c]);
  SGP_TRY(stage_gpdev(g, host, G));
  const size_t nops = cand_ops_doubles(m, G);
  char* buf = static_cast<char*>(sgp_scratch(ctx, 7, nops * 8 + size_t(m) * 8 + size_t(m) * G * 4));
  SGP_CHECK(ctx, buf, "device allocation failed: %s", ctx->err.c_str());
  double* ops = reinterpret_cast<double*>(buf);
  int64_t* cl = reinterpret_cast<int64_t*>(buf + nops * 8);
  int32_t* dfl = reinterpret_cast<int32_t*>(buf + nops * 8 + size_t(m) * 8);
  SGP_TRY(sgp_h2d(ctx, cl, gidx, size_t(m) * 8));
  SGP_HIP(ctx, hipMemsetAsync(dfl, 0, size_t(m) * G * 4, ctx->stream));
  SGP_TRY(launch_cand_all(g, g->gpdev, host, G, beta, fmin, cl, m, ops, dfl));
  return sgp_d2h(ctx, flags, dfl, size_t(m) * G * 4);
}

// ... of EVERY candidate of the grid, listed on the device in row order: global rows, widths and
// flags come back in ONE read-back (the candidate mask and the widths do not travel first).
// *count > cap: nothing else is valid (the caller takes the two-step path).
int sgp_grid_expanders_small_all(sgp_grid* g, sgp_gp* const* gps, int G, double beta,
                                 const double* fmin, int cap, int* count, int64_t* gidx,
                                 double* width, int32_t* flags) {
  sgp_ctx* ctx = g->ctx;
  SGP_HIP(ctx, hipSetDevice(ctx->device));
  SGP_CHECK(ctx, G == g->G, "grid was created for %d GPs, got %d", g->G, G);
  SGP_CHECK(ctx, cap >= 1, "cap = %d", cap);
  *count = 0;
  GpDev host[SGP_MAX_GPS];
  SGP_TRY(collect_gps(ctx, gps, G, g->d, host));
  SGP_CHECK(ctx, step_small_eligible(ctx, host, G, g->N),
            "sgp_grid_expanders_small_all: %lld rows / a GP with more than 48 observations",
            (long long)g->N);
  SGP_TRY(stage_gpdev(g, host, G));
  if (cap > g->N) cap = int(g->N);
  // the list: every candidate, row order (the selection of the big passes with no threshold)
  const size_t nl = (size_t(g->N) * 4 + 63) & ~size_t(63);
  char* sb = static_cast<char*>(sgp_scratch(ctx, 8, nl + 16384 + 256 + (size_t(g->N) / 256 + 2) * 4));
  SGP_CHECK(ctx, sb, "device allocation failed: %s", ctx->err.c_str());
  int* list = reinterpret_cast<int*>(sb);
  char* sel = sb + nl + 16384;
  int* counts = reinterpret_cast<int*>(sb + nl + 16384 + 256);
  struct { double thr; int count, est; } hs = {-INFINITY, 0, 0};
  SGP_TRY(sgp_h2d(ctx, sel, &hs, sizeof(hs)));
  SGP_TRY(launch_pass_list(g, 0, INFINITY, -1, sel, list, counts));
  const int* count_dev = reinterpret_cast<const int*>(sel + 8);
  // [count | rows | widths | flags] for the read-back, the operands behind them
  const size_t bh = 8, bc = size_t(cap) * 8, bw = size_t(cap) * 8,
               bf = (size_t(cap) * G * 4 + 7) & ~size_t(7);
  const size_t nops = cand_ops_doubles(cap, G);
  char* buf = static_cast<char*>(sgp_scratch(ctx, 7, bh + bc + bw + bf + nops * 8));
  SGP_CHECK(ctx, buf, "device allocation failed: %s", ctx->err.c_str());
  int64_t* hdr = reinterpret_cast<int64_t*>(buf);
  int64_t* cl = reinterpret_cast<int64_t*>(buf + bh);
  double* dw = reinterpret_cast<double*>(buf + bh + bc);
  int32_t* dfl = reinterpret_cast<int32_t*>(buf + bh + bc + bw);
  double* ops = reinterpret_cast<double*>(buf + bh + bc + bw + bf);
  SGP_TRY(launch_small_pack(g, list, count_dev, cap, hdr, cl, dw, dfl));
  SGP_TRY(launch_cand_all(g, g->gpdev, host, G, beta, fmin, cl, cap, ops, dfl, count_dev));
  std::vector<char> hb(bh + bc + bw + bf);
  SGP_TRY(sgp_d2h(ctx, hb.data(), buf, hb.size()));
  int64_t n64;
  memcpy(&n64, hb.data(), 8);
  *count = int(n64);
  if (n64 > cap) return 0;
  memcpy(gidx, hb.data() + bh, size_t(n64) * 8);
  memcpy(width, hb.data() + bh + bc, size_t(n64) * 8);
  memcpy(flags, hb.data() + bh + bc + bw, size_t(n64) * G * 4);
  return 0;
}

// 1 when sgp_grid_step_small serves this grid with these GPs (at most 16384 rows, every GP
// with at most 48 observations, sweep kernel not forced), else 0
int sgp_grid_step_small_ok(sgp_grid* g, sgp_gp* const* gps, int G) {
  if (!g || G != g->G || G < 1 || G > SGP_MAX_GPS) return 0;
  GpDev host[SGP_MAX_GPS];
  for (int i = 0; i < G; ++i) {
    if (!gps[i] || gps[i]->n <= 0 || gps[i]->ctx != g->ctx) return 0;
    host[i] = gps[i]->dev;
  }
  return step_small_eligible(g->ctx, host, G, g->N) ? 1 : 0;
}

// N-rank certified step with ONE stream sync (the N-rank counterpart of
// sgp_grid_sets_fused; gp_opt.py:511-557, 611-612, 635, 642-644): the front half of
// sgp_grid_sets_front_comm on this rank's shard, then IN STREAM
//   all-gather of every rank's front block  -> k_merge_front: the first candidate
//     of the whole grid in visiting order, total counts, ties over all shards,
//     staged as the operand of the expander test on every rank;
//   the probe scan of that candidate over this rank's unsafe rows;
//   all-reduce (max) of the G flags;
//   conditional G mark by the rank that owns the candidate + local M | G arg-max;
//   all-gather of the (value, index) pairs -> k_merge_argmax.
// Every rank reads back the same block.  Without a communicator (one rank) the
// collectives are skipped and the merges run over one block.
int sgp_grid_sets_fused_comm(sgp_grid* g, sgp_gp* const* gps, int G, double beta,
                             const double* fmin, const double* scaling,
                             const double* thr_beta, double near_frac, double* out5,
                             double* x_top, double* mean_top, double* q_top,
                             int32_t* flags, double* value, int64_t* gidx,
                             double* max_l_out) {
  sgp_ctx* ctx = g->ctx;
  SGP_HIP(ctx, hipSetDevice(ctx->device));
  const int d = g->d;
  SGP_CHECK(ctx, G == g->G, "grid was created for %d GPs, got %d", g->G, G);
  const bool comm = have_comm(ctx);
  SGP_CHECK(ctx, comm || ctx->world <= 1,
            "rank %d of %d has no communicator in the grid's context: the "
            "in-stream collectives cannot run (sgp_comm_init on THIS context)",
            ctx->rank, ctx->world);
  const int world = comm ? ctx->world : 1;
  // device block: merged result (layout of sgp_grid_sets_fused) | this rank's front
  // block | gathered front blocks | this rank's (value, index) | gathered pairs
  const size_t nfront = 6 + size_t(d) + 3 * size_t(G);
  const size_t nfl = (size_t(G) + 1) / 2;
  const size_t nres = nfront + nfl + 3;
  const size_t total = nres + 1 + nfront * (1 + size_t(world)) + 2 * (1 + size_t(world));
  double* res = static_cast<double*>(sgp_scratch(ctx, 1, (total + 8) * 8));
  SGP_CHECK(ctx, res, "device allocation failed: %s", ctx->err.c_str());
  double* mine = res + nres + 1;
  double* all = mine + nfront;
  double* pair = all + nfront * size_t(world);
  double* pairs = pair + 2;
  GpDev ghost[SGP_MAX_GPS];
  SGP_TRY(collect_gps(ctx, gps, G, d, ghost));
  ExpanderBufs eb;
  SGP_TRY(expander_bufs(g, ghost, G, &eb));

  // ---- front half on this shard
  SGP_TRY(front_half_in_stream(g, scaling, thr_beta, mine));

  // ---- first candidate of the whole grid
  const double* blocks = mine;
  if (comm) {
    SGP_TRY(coll_allgather(ctx, mine, all, nfront * 8));
    blocks = all;
  }
  SGP_TRY(launch_merge_front(g, blocks, world, int(nfront), res, eb.xc,
                             int((eb.bx + eb.bv) / 8), eb.flags, int(eb.bf / 4)));
  // ---- probe scan over this shard, flags over all shards
  int32_t* dfl = nullptr;
  SGP_TRY(enqueue_expander(g, gps, G, beta, fmin, 1, nullptr, nullptr, nullptr,
                           near_frac, &dfl, nullptr, true));
  if (comm) SGP_TRY(coll_allreduce_max_i32(ctx, dfl, size_t(G)));
  // ---- conditional G mark (owner of the candidate) + arg-max over the whole grid
  SGP_TRY(launch_argmax_marked(
      g, scaling, fmin, dfl, reinterpret_cast<int64_t*>(res + 4),
      reinterpret_cast<int*>(res + 5), reinterpret_cast<int32_t*>(res + nfront),
      pair, reinterpret_cast<int64_t*>(pair + 1)));
  const double* prs = pair;
  if (comm) {
    SGP_TRY(coll_allgather(ctx, pair, pairs, 16));
    prs = pairs;
  }
  SGP_TRY(launch_merge_argmax(ctx, prs, world, res + nfront + nfl,
                              reinterpret_cast<int64_t*>(res + nfront + nfl + 1)));
  SGP_HIP(ctx, hipMemcpyAsync(res + nfront + nfl + 2, g->scal, 8,
                              hipMemcpyDeviceToDevice, ctx->stream));

  std::vector<double> host(nres);
  SGP_TRY(sgp_d2h(ctx, host.data(), res, nres * 8));
  unpack_front(host.data(), d, G, out5, x_top, mean_top, q_top);
  memcpy(flags, &host[nfront], size_t(G) * 4);
  *value = host[nfront + nfl];
  memcpy(gidx, &host[nfront + nfl + 1], 8);
  *max_l_out = host[nfront + nfl + 2];
  return 0;
}

// Fitness of P <= kSmallPoints particles (row-major, device) through the
// small-point posterior path: mean / var per GP, then the shaping kernel.
static int fitness_small(sgp_ctx* ctx, const GpDev* gps_dev, const GpDev* gps_host,
                         int G, const double* pts_rowmajor, int64_t P,
                         const FitnessArgs& fa) {
  const int Geff = (fa.swarm_type == SGP_SWARM_GREEDY) ? 1 : G;
  double* mv = static_cast<double*>(
      sgp_scratch(ctx, 2, size_t(2) * SGP_MAX_GPS * kSmallPoints * sizeof(double)));
  SmallBufs sb;
  SGP_CHECK(ctx, mv && small_reserve(ctx, gps_host, Geff, int(P), &sb) == 0,
            "device allocation failed: %s", ctx->err.c_str());
  double* mean = mv;
  double* var = mv + size_t(SGP_MAX_GPS) * kSmallPoints;
  SGP_TRY(posterior_small_all(ctx, gps_dev, gps_host, Geff, pts_rowmajor, int(P), sb,
                              mean, var));
  return launch_fitness_small(ctx, G, P, mean, var, fa);
}

static bool small_path_pays_all(sgp_gp* const* gps, int G, int64_t P) {
  for (int g = 0; g < G; ++g)
    if (!small_path_pays(gps[g], P)) return false;
  return true;
}

// ---- swarm ----------------------------------------------------------------------
int sgp_swarm_fitness(sgp_ctx* ctx, sgp_gp* const* gps, int G, int swarm_type,
                      const double* particles, int64_t P, double beta,
                      const double* fmin, const double* scaling,
                      double best_lower_bound, double* values, uint8_t* safe) {
  SGP_HIP(ctx, hipSetDevice(ctx->device));
  SGP_CHECK(ctx, swarm_type >= SGP_SWARM_GREEDY && swarm_type <= SGP_SWARM_SAFE_SET,
            "Invalid swarm type %d", swarm_type);
  SGP_CHECK(ctx, G >= 1 && gps[0], "no GP");
  if (P <= 0) return 0;
  const int d = gps[0]->kern.d;
  GpDev host[SGP_MAX_GPS];
  SGP_TRY(collect_gps(ctx, gps, G, d, host));
  const size_t nd = size_t(P) * 8;
  double* stage = static_cast<double*>(sgp_scratch(ctx, 3, nd * d));
  char* work = static_cast<char*>(
      sgp_scratch(ctx, 4, nd * d + nd + size_t(P) + sizeof(GpDev) * SGP_MAX_GPS + 64));
  SGP_CHECK(ctx, stage && work, "device allocation failed: %s", ctx->err.c_str());
  double* pts = reinterpret_cast<double*>(work);
  double* dval = reinterpret_cast<double*>(work + nd * d);
  GpDev* gdev = reinterpret_cast<GpDev*>(work + nd * d + nd);
  uint8_t* dsafe = reinterpret_cast<uint8_t*>(work + nd * d + nd + sizeof(GpDev) * SGP_MAX_GPS);
  SGP_TRY(sgp_h2d(ctx, stage, particles, nd * d));
  SGP_TRY(launch_import_points(ctx, stage, P, d, d, 1, pts));
  SGP_TRY(sgp_h2d(ctx, gdev, host, sizeof(GpDev) * G));
  FitnessArgs fa{};
  fa.swarm_type = swarm_type;
  fa.beta = beta;
  fa.best_lower_bound = best_lower_bound;
  for (int i = 0; i < SGP_MAX_GPS; ++i) {
    fa.fmin[i] = (i < G) ? fmin[i] : -INFINITY;
    fa.scaling[i] = (i < G) ? scaling[i] : 1.0;
  }
  fa.values = dval;
  fa.safe = dsafe;
  if (small_path_pays_all(gps, G, P)) {
    SGP_TRY(fitness_small(ctx, gdev, host, G, stage, P, fa));
  } else {
    SweepPoints sp{pts, P, 1, P};
    SGP_TRY(launch_sweep_fitness(ctx, gdev, host, G, d, sp, fa));
  }
  SGP_TRY(sgp_d2h(ctx, values, dval, nd));
  SGP_TRY(sgp_d2h(ctx, safe, dsafe, size_t(P)));
  return 0;
}

// SwarmOptimization.init_swarm / run_swarm (swarm.py:61-146) with the state in
// HBM and the fitness fused in: one call = the whole run, one host round trip.
int sgp_swarm_run(sgp_ctx* ctx, sgp_gp* const* gps, int G, int swarm_type,
                  double beta, const double* fmin, const double* scaling,
                  double best_lower_bound, int64_t P, double* positions,
                  double* velocities, double* best_positions, double* best_values,
                  double* global_best, const double* velocity_scale,
                  const double* bounds, int init, int iters, double inertia0,
                  double step_size, const double* rand, uint64_t seed) {
  SGP_HIP(ctx, hipSetDevice(ctx->device));
  SGP_CHECK(ctx, swarm_type >= SGP_SWARM_GREEDY && swarm_type <= SGP_SWARM_SAFE_SET,
            "Invalid swarm type %d", swarm_type);
  SGP_CHECK(ctx, G >= 1 && gps[0], "no GP");
  SGP_CHECK(ctx, P >= 1 && iters >= 0, "bad swarm size %lld / iterations %d",
            (long long)P, iters);
  const int d = gps[0]->kern.d;
  GpDev host[SGP_MAX_GPS];
  SGP_TRY(collect_gps(ctx, gps, G, d, host));
  const size_t nd = size_t(P) * d * 8, nv = size_t(P) * 8;
  const size_t nrand = rand ? (size_t(init ? 1 : 0) + 2 * size_t(iters)) * nd : 0;
  // pos | vel | best | best_values | values | gbest | vscale | bounds | gpdev | safe
  const size_t small = size_t(d) * 8 * 4 + sizeof(GpDev) * SGP_MAX_GPS + 64;
  char* buf = static_cast<char*>(sgp_scratch(ctx, 4, 3 * nd + 2 * nv + small + size_t(P)));
  double* drand = rand ? static_cast<double*>(sgp_scratch(ctx, 3, nrand)) : nullptr;
  SGP_CHECK(ctx, buf && (!rand || drand), "device allocation failed: %s",
            ctx->err.c_str());
  double* dpos = reinterpret_cast<double*>(buf);
  double* dvel = reinterpret_cast<double*>(buf + nd);
  double* dbest = reinterpret_cast<double*>(buf + 2 * nd);
  double* dbv = reinterpret_cast<double*>(buf + 3 * nd);
  double* dval = reinterpret_cast<double*>(buf + 3 * nd + nv);
  double* dgb = reinterpret_cast<double*>(buf + 3 * nd + 2 * nv);
  double* dvs = dgb + d;
  double* dbd = dvs + d;                       // 2 d entries
  GpDev* gdev = reinterpret_cast<GpDev*>(dbd + 2 * d);
  uint8_t* dsafe = reinterpret_cast<uint8_t*>(gdev + SGP_MAX_GPS);
  SGP_TRY(sgp_h2d(ctx, dpos, positions, nd));
  if (!init) {
    SGP_TRY(sgp_h2d(ctx, dvel, velocities, nd));
    SGP_TRY(sgp_h2d(ctx, dbest, best_positions, nd));
    SGP_TRY(sgp_h2d(ctx, dbv, best_values, nv));
    SGP_TRY(sgp_h2d(ctx, dgb, global_best, size_t(d) * 8));
  }
  SGP_TRY(sgp_h2d(ctx, dvs, velocity_scale, size_t(d) * 8));
  if (bounds) SGP_TRY(sgp_h2d(ctx, dbd, bounds, size_t(d) * 16));
  if (rand) SGP_TRY(sgp_h2d(ctx, drand, rand, nrand));
  SGP_TRY(sgp_h2d(ctx, gdev, host, sizeof(GpDev) * G));
  FitnessArgs fa{};
  fa.swarm_type = swarm_type;
  fa.beta = beta;
  fa.best_lower_bound = best_lower_bound;
  for (int i = 0; i < SGP_MAX_GPS; ++i) {
    fa.fmin[i] = (i < G) ? fmin[i] : -INFINITY;
    fa.scaling[i] = (i < G) ? scaling[i] : 1.0;
  }
  fa.values = dval;
  fa.safe = dsafe;
  const SweepPoints sp{dpos, P, d, 1};          // row-major (P, d) in place
  const bool few = P <= kSmallSwarm && small_path_pays_all(gps, G, P);
  // a small swarm against GPs with few observations (SafeOptSwarm's defaults on the
  // reference's own examples: 20 particles, n <= 20): the posterior is one sweep launch
  // (sweep_tiny.hip up to 48 observations), everything else of the iteration the same ONE
  // workgroup as on the few-points path -- two launches per iteration instead of five
  const bool few_swept = P <= kSmallSwarm && !few;
  const double* r = drand;
  double inertia = inertia0;
  if (few || few_swept) {
    // small swarm: three launches per iteration -- k(X, particles), the block
    // products on the matrix cores, and ONE workgroup for everything else
    // (fitness, bests, and the move that opens the next iteration)
    const int Geff = (swarm_type == SGP_SWARM_GREEDY) ? 1 : G;
    SmallBufs sb{};
    ConfOut post{};
    if (few) {
      SGP_CHECK(ctx, small_reserve(ctx, host, Geff, int(P), &sb) == 0,
                "device allocation failed: %s", ctx->err.c_str());
    } else {
      const size_t np = size_t(Geff) * size_t(P);
      SGP_TRY(sgp_reserve(ctx, &ctx->pair_post, 2 * np * sizeof(double)));
      post.mean = static_cast<double*>(ctx->pair_post.p);
      post.var = post.mean + np;
      for (int i = 0; i < SGP_MAX_GPS; ++i) post.fmin[i] = -INFINITY;
    }
    PsoSmallArgs ps{};
    ps.pos = dpos;
    ps.vel = dvel;
    ps.best = dbest;
    ps.best_values = dbv;
    ps.gbest = dgb;
    ps.vscale = dvs;
    ps.bounds = bounds ? dbd : nullptr;
    ps.seed = seed;
    ps.P = int(P);
    ps.d = d;
    auto step = [&](int is_init, int it_next) -> int {   // it_next < 0: no move
      if (few)
        SGP_TRY(posterior_small_all(ctx, gdev, host, Geff, dpos, int(P), sb, nullptr,
                                    nullptr));
      else
        SGP_TRY(launch_sweep_conf(ctx, gdev, host, Geff, d, sp, post));
      ps.init = is_init;
      ps.move = it_next >= 0;
      ps.rand = r;
      ps.draw = uint32_t(it_next + 1);
      ps.inertia = inertia;
      SGP_TRY(launch_pso_small_step(ctx, gdev, G, sb, fa, ps, few ? nullptr : post.mean,
                                    few ? nullptr : post.var));
      if (ps.move) {
        if (r) r += 2 * size_t(P) * d;
        inertia += step_size;
      }
      return 0;
    };
    if (init) {
      SGP_TRY(launch_pso_init_vel(ctx, P, d, dvel, dvs, r, seed));
      if (r) r += size_t(P) * d;
      SGP_TRY(step(1, iters > 0 ? 0 : -1));
    } else if (iters > 0) {
      SGP_TRY(launch_pso_move(ctx, P, d, dpos, dvel, dbest, dgb, dvs,
                              bounds ? dbd : nullptr, inertia, r, seed, 1u));
      if (r) r += 2 * size_t(P) * d;
      inertia += step_size;
    }
    for (int it = 0; it < iters; ++it)
      SGP_TRY(step(0, it + 1 < iters ? it + 1 : -1));
  } else {
    // (up to kSmallPoints particles still take the few-points posterior)
    const bool few_points = small_path_pays_all(gps, G, P);
    auto fitness = [&]() -> int {
      return few_points ? fitness_small(ctx, gdev, host, G, dpos, P, fa)
                        : launch_sweep_fitness(ctx, gdev, host, G, d, sp, fa);
    };
    if (init) {
      SGP_TRY(launch_pso_init_vel(ctx, P, d, dvel, dvs, r, seed));
      if (r) r += size_t(P) * d;
      SGP_TRY(fitness());
      SGP_TRY(launch_pso_best(ctx, P, d, dval, dsafe, dpos, dbest, dbv, dgb, 1));
    }
    for (int it = 0; it < iters; ++it) {
      SGP_TRY(launch_pso_move(ctx, P, d, dpos, dvel, dbest, dgb, dvs,
                              bounds ? dbd : nullptr, inertia, r, seed,
                              uint32_t(it + 1)));
      if (r) r += 2 * size_t(P) * d;
      inertia += step_size;
      SGP_TRY(fitness());
      SGP_TRY(launch_pso_best(ctx, P, d, dval, dsafe, dpos, dbest, dbv, dgb, 0));
    }
  }
  SGP_TRY(sgp_d2h(ctx, positions, dpos, nd));
  SGP_TRY(sgp_d2h(ctx, velocities, dvel, nd));
  SGP_TRY(sgp_d2h(ctx, best_positions, dbest, nd));
  SGP_TRY(sgp_d2h(ctx, best_values, dbv, nv));
  return sgp_d2h(ctx, global_best, dgb, size_t(d) * 8);
}

// SafeOptSwarm safe-set growth, gp_opt.py:1089-1111 (kernels in swarm.hip).
int sgp_swarm_grow(sgp_ctx* ctx, sgp_gp* gp0, const double* S, int64_t m,
                   const double* B, int64_t n, double scale2, double thr,
                   uint8_t* accept) {
  SGP_HIP(ctx, hipSetDevice(ctx->device));
  SGP_CHECK(ctx, gp0 != nullptr, "no GP");
  SGP_CHECK(ctx, m >= 0 && n >= 0 && n <= INT32_MAX, "bad sizes m=%lld n=%lld",
            (long long)m, (long long)n);
  if (n == 0) return 0;
  const int d = gp0->kern.d;
  const int nchunks = swarm_grow_chunks(m);
  const size_t bs = size_t(m) * d * 8, bb = size_t(n) * d * 8;
  const size_t bp = size_t(n) * size_t(nchunks > 0 ? nchunks : 1) * 8;
  const size_t bl = size_t(n) * 4, ba = size_t(n);
  char* buf = static_cast<char*>(
      sgp_scratch(ctx, 4, bs + bb + bp + bl + ba + 64));
  SGP_CHECK(ctx, buf, "device allocation failed: %s", ctx->err.c_str());
  double* dS = reinterpret_cast<double*>(buf);
  double* dB = reinterpret_cast<double*>(buf + bs);
  double* part = reinterpret_cast<double*>(buf + bs + bb);
  int* list = reinterpret_cast<int*>(buf + bs + bb + bp);
  uint8_t* dacc = reinterpret_cast<uint8_t*>(buf + bs + bb + bp + bl);
  SGP_TRY(sgp_h2d(ctx, dS, S, bs));
  SGP_TRY(sgp_h2d(ctx, dB, B, bb));
  SGP_TRY(launch_swarm_grow(ctx, gp0->kern, dS, m, dB, int(n), scale2, thr, part,
                            list, dacc));
  return sgp_d2h(ctx, accept, dacc, ba);
}

// ---- timing ---------------------------------------------------------------------
int64_t sgp_ctx_alloc_count(sgp_ctx* ctx) { return ctx ? ctx->n_allocs : -1; }

int sgp_ctx_set_share(sgp_ctx* ctx, int on) {
  if (!ctx) return -1;
  const int old = ctx->share_factors;
  if (on == 0 || on == 1) ctx->share_factors = on;
  return old;
}

int sgp_ctx_last_sweep(sgp_ctx* ctx) { return ctx ? ctx->last_sweep : 0; }

int sgp_ctx_set_sweep(sgp_ctx* ctx, int which) {
  if (!ctx) return -1;
  const int old = ctx->sweep_choice;
  if (which >= 0 && which < 64) ctx->sweep_choice = which;
  return old;
}

int sgp_timer_start(sgp_ctx* ctx) {
  SGP_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  return 0;
}

int sgp_timer_stop(sgp_ctx* ctx, float* ms) {
  SGP_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  SGP_HIP(ctx, hipEventSynchronize(ctx->ev1));
  SGP_HIP(ctx, hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
  return 0;
}

int sgp_profile_enable(sgp_ctx* ctx, int on) {
  ctx->profiling = on != 0;
  ctx->prof_used = 0;
  ctx->prof_flops = 0.0;
  return 0;
}

int sgp_profile_read(sgp_ctx* ctx, double* total_ms, int64_t* launches,
                     double* flops) {
  SGP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  double tot = 0.0;
  for (size_t i = 0; i + 1 < ctx->prof_used; i += 2) {
    float ms = 0.f;
    SGP_HIP(ctx, hipEventElapsedTime(&ms, ctx->prof_events[i],
                                     ctx->prof_events[i + 1]));
    tot += ms;
  }
  *total_ms = tot;
  *launches = int64_t(ctx->prof_used / 2);
  *flops = ctx->prof_flops;
  return 0;
}

// ---- RCCL -----------------------------------------------------------------------
int sgp_comm_unique_id(void* id128) {
  SGP_TRY(load_rccl(nullptr));
  ncclUniqueId id;
  SGP_NCCL(nullptr, g_rccl.GetUniqueId(&id));
  memcpy(id128, &id, sizeof(id));
  return 0;
}

int sgp_comm_init(sgp_ctx* ctx, const void* id128, int rank, int world) {
  SGP_HIP(ctx, hipSetDevice(ctx->device));
  SGP_TRY(load_rccl(ctx));
  SGP_CHECK(ctx, world >= 1 && rank >= 0 && rank < world,
            "bad rank %d / world %d", rank, world);
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  ncclComm_t comm;
  SGP_NCCL(ctx, g_rccl.CommInitRank(&comm, world, id, rank));
  ctx->comm = comm;
  ctx->rank = rank;
  ctx->world = world;
  return 0;
}

int sgp_comm_init_host(sgp_ctx* ctx, int rank, int world,
                       sgp_host_allreduce_max_f64 allreduce_f64,
                       sgp_host_allreduce_max_i32 allreduce_i32,
                       sgp_host_allgather allgather, void* user) {
  SGP_CHECK(ctx, world >= 1 && rank >= 0 && rank < world,
            "bad rank %d / world %d", rank, world);
  SGP_CHECK(ctx, !ctx->comm, "the context already has an RCCL communicator");
  SGP_CHECK(ctx, allreduce_f64 && allreduce_i32 && allgather,
            "sgp_comm_init_host: all three collectives are required");
  ctx->hostcomm.allreduce_max_f64 = allreduce_f64;
  ctx->hostcomm.allreduce_max_i32 = allreduce_i32;
  ctx->hostcomm.allgather = allgather;
  ctx->hostcomm.user = user;
  ctx->rank = rank;
  ctx->world = world;
  return 0;
}

int sgp_comm_count(sgp_ctx* ctx, int* n) {
  *n = 1;
  if (ctx->hostcomm.allgather) {  // (the caller's transport: its word for it)
    *n = ctx->world;
    return 0;
  }
  if (!ctx->comm) return 0;       // no communicator: a single rank
  SGP_NCCL(ctx, g_rccl.CommCount(static_cast<ncclComm_t>(ctx->comm), n));
  return 0;
}

// Small host -> device copy through the pinned staging half (no stream sync:
// the collective and the read-back that follow are ordered behind it and the
// read-back's sync completes all three); large payloads take the synchronous
// path.
static int stage_h2d(sgp_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (bytes > ctx->pinned_cap / 4) return sgp_h2d(ctx, dst, src, bytes);
  char* slot = static_cast<char*>(ctx->pinned) + ctx->pinned_cap / 2;
  memcpy(slot, src, bytes);
  SGP_HIP(ctx, hipMemcpyAsync(dst, slot, bytes, hipMemcpyHostToDevice,
                              ctx->stream));
  return 0;
}

int sgp_comm_allreduce_max(sgp_ctx* ctx, double* buf, int n) {
  if ((ctx->world <= 1 && !ctx->comm) || n <= 0) return 0;
  if (!ctx->comm && ctx->hostcomm.allreduce_max_f64) {
    SGP_CHECK(ctx, ctx->hostcomm.allreduce_max_f64(ctx->hostcomm.user, buf, n) == 0,
              "host collective (all-reduce max) failed");
    return 0;
  }
  SGP_HIP(ctx, hipSetDevice(ctx->device));
  SGP_CHECK(ctx, ctx->comm, "sgp_comm_init was not called");
  double* d = static_cast<double*>(sgp_scratch(ctx, 6, size_t(n) * 8));
  SGP_CHECK(ctx, d, "device allocation failed: %s", ctx->err.c_str());
  SGP_TRY(stage_h2d(ctx, d, buf, size_t(n) * 8));
  SGP_NCCL(ctx, g_rccl.AllReduce(d, d, size_t(n), ncclFloat64, ncclMax,
                                 static_cast<ncclComm_t>(ctx->comm),
                                 ctx->stream));
  return sgp_d2h(ctx, buf, d, size_t(n) * 8);
}

int sgp_comm_allgather(sgp_ctx* ctx, const void* send, void* recv,
                       int64_t nbytes) {
  if (nbytes <= 0) return 0;
  if (ctx->world <= 1 && !ctx->comm) {
    memcpy(recv, send, size_t(nbytes));
    return 0;
  }
  if (!ctx->comm && ctx->hostcomm.allgather) {
    SGP_CHECK(ctx, ctx->hostcomm.allgather(ctx->hostcomm.user, send, recv, nbytes) == 0,
              "host collective (all-gather) failed");
    return 0;
  }
  SGP_HIP(ctx, hipSetDevice(ctx->device));
  SGP_CHECK(ctx, ctx->comm, "sgp_comm_init was not called");
  char* d = static_cast<char*>(
      sgp_scratch(ctx, 6, size_t(nbytes) * (size_t(ctx->world) + 1)));
  SGP_CHECK(ctx, d, "device allocation failed: %s", ctx->err.c_str());
  char* r = d + nbytes;
  SGP_TRY(stage_h2d(ctx, d, send, size_t(nbytes)));
  SGP_NCCL(ctx, g_rccl.AllGather(d, r, size_t(nbytes), ncclInt8,
                                 static_cast<ncclComm_t>(ctx->comm),
                                 ctx->stream));
  return sgp_d2h(ctx, recv, r, size_t(nbytes) * size_t(ctx->world));
}

int sgp_comm_barrier(sgp_ctx* ctx) {
  double z = 0.0;
  SGP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return sgp_comm_allreduce_max(ctx, &z, 1);
}

}  // extern "C"
