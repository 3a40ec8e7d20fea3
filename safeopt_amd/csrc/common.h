// Internal declarations shared by the translation units of libsafeopt_hip.so.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "safeopt_hip.h"

// ---- error plumbing ---------------------------------------------------------
void sgp_set_error(sgp_ctx* ctx, const char* fmt, ...);

#define SGP_HIP(ctx, call)                                                     \
  do {                                                                         \
    hipError_t e_ = (call);                                                    \
    if (e_ != hipSuccess) {                                                    \
      sgp_set_error((ctx), "%s:%d %s -> %s", __FILE__, __LINE__, #call,        \
                    hipGetErrorString(e_));                                    \
      return -1;                                                               \
    }                                                                          \
  } while (0)

#define SGP_CHECK(ctx, cond, ...)                                              \
  do {                                                                         \
    if (!(cond)) {                                                             \
      sgp_set_error((ctx), __VA_ARGS__);                                       \
      return -2;                                                               \
    }                                                                          \
  } while (0)

#define SGP_TRY(expr)                                                          \
  do {                                                                         \
    int r_ = (expr);                                                           \
    if (r_ != 0) return r_;                                                    \
  } while (0)

// ---- device-side descriptors ------------------------------------------------
struct KernDesc {
  int d;
  int n_parts;
  int kind[SGP_MAX_PARTS];
  double variance[SGP_MAX_PARTS];
  double inv_ls[SGP_MAX_PARTS][SGP_MAX_D];
  double kdiag;  // product of the variances = k(x, x)
  // Single-part kernels: inputs are stored pre-multiplied by scale0 =
  // inv_ls[0] * kern_unit(kind[0]), in units where the exponent of the
  // covariance is a plain square / norm (see kern_eval.h, KernFast).
  double scale0[SGP_MAX_D];
  // Products of parts in the sweep (KernFast::manyn_t): squared weights
  // wsq[p][k] = (inv_ls[p][k] * kern_unit(kind[p]))^2 on the squared RAW coordinate
  // differences (zero for a column the part does not use, and for p >= n_parts);
  // the exponents of the parts then ADD: one 2^(U/32) per covariance.
  double wsq[SGP_MAX_PARTS][SGP_MAX_D];
};

// Input scaling of the fast covariance path: with z = x * inv_ls * kern_unit,
//   RBF      exp(-r^2/2)   = 2^(-|dz|^2 / 32)        kern_unit = sqrt(16 / ln 2)
//   Matern   exp(-sqrt(nu2) r) = 2^(-|dz| / 32)      kern_unit = sqrt(3|5) * 32 / ln 2
inline double kern_unit(int kind) {
  return kind == SGP_RBF ? 4.804489635145799
                         : (kind == SGP_MATERN32 ? 79.962275540715
                                                 : 103.23085383134594);
}

// One GP as the sweep kernels see it.  All pointers are device pointers.
struct GpDev {
  const double* Apack;  // L^-1 in MFMA A-operand order: [nblk][n_pad/4][64],
                        // zero above the diagonal and in the padding rows
  const double* Xpad;   // training inputs, [n_pad][d], zero padded
  const double* Xs;     // = Xpad * kern.scale0 for single-part kernels,
                        //   else = Xpad (KernFast::operator() convention)
  const double* alpha;  // Ky^-1 y, [n_pad], zero padded
  const double* XA;     // per j-block of 16 training points: [16 d of Xs | 16 of
                        // alpha], contiguous -- ONE LDS-DMA source per stage of
                        // the paired sweep (sweep_pair.hip)
  int n;                // training points
  int n_pad;            // n rounded up to 16
  int nblk;             // n_pad / 16
  int share;            // >= 0: this GP has the same training inputs, kernel and noise
                        // as the GP in front of it in the launch (the multi-output
                        // case): same L^-1, so the paired sweep takes |L^-1 k|^2
                        // from that GP and only forms alpha . k (collect_gps)
  int narrow;           // 1: the last row block has <= 4 real rows and Apack holds
                        // them in the "narrow" form (k_pack): the sweep then needs
                        // one MFMA per k-step for that block instead of four
  int last_rows;        // real rows of the last row block (1..16): the 4-wave sweep
                        // takes a last block of <= 12 rows as 1..3 narrow 4-row groups
                        // (sweep.hip, narrow_groups)
  // record of the last one-row append (sgp_gp_append), consumed by the
  // rank-1 update of the resident posterior: w = Ky_old^-1 k(X_old, x*)
  // (zero padded to n_pad), upd[0] = (y* - mu(x*)) / s2, upd[1] = 1 / s2,
  // upd[2..2+d) = x*
  const double* upd_w;
  const double* upd;
  // dense L^-1 (row pitch ld) and k(x,x) + noise + 1e-8 + jitter: what the
  // operands of the rank-1 expander test are built from (factor.hip, k_expw)
  const double* Linv;
  int64_t ld;
  double prior;
  KernDesc kern;
};

// A candidate grid that is a TENSOR grid (linearly_spaced_combinations,
// safeopt/utilities.py:21-54: global row i has column k equal to
// axis_k[(i / stride_k) % count_k]), declared with sgp_grid_set_axes and verified on
// the device, together with per-axis factor tables of every GP whose kernel is a
// product of RBF parts: k(X_j, x) = prod_k E_k[idx_k(x)][j].  The sweep then reads d
// table entries per covariance instead of evaluating an exponential (sweep.hip, SEP).
struct SepLaunch {
  int naxes;                  // axes with more than one point (1..4), in order of their
                              // stride in the flat index; constant columns (contexts)
                              // are folded into the tables of axis 0
  uint32_t count[4];          // points of axis a
  int64_t goff;               // global index of the shard's first row
  // per GP and axis: [n_pad / 16][count][16] doubles, entry 4 k4 + q of a block of 16
  // = training point 16 jb + 4 q + k4 (the four values of a lane side by side)
  const double* tab[SGP_MAX_GPS][4];
};

// ---- host-side objects ------------------------------------------------------
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
};

struct sgp_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  // small pinned staging area for scalar H2D / D2H traffic
  void* pinned = nullptr;
  size_t pinned_cap = 0;
  // device scratch (grown on demand)
  DevBuf scratch[13];
  int64_t n_allocs = 0;       // hipMalloc calls so far (sgp_ctx_alloc_count)
  // timing
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool profiling = false;
  std::vector<hipEvent_t> prof_events;  // start/stop pairs of sweep launches
  size_t prof_used = 0;
  double prof_flops = 0.0;
  // stage table of the posterior sweep (sweep.hip: stage_table)
  DevBuf stage_tab;
  std::vector<uint64_t> stage_sig;
  int stage_count = 0;
  int stage_slots = 0;       // 2-KB positions of all stages of that table
  // ... and of the paired sweep (sweep_pair.hip: pair_stage_table)
  DevBuf pstage_tab;
  std::vector<uint64_t> pstage_sig;
  int pstage_count = 0;
  std::vector<int> pstage_chunk_start;   // first stage of every accumulator chunk, then
                                         // the stage count
  int pstage_chunk_off[SGP_MAX_GPS + 1] = {0};   // first chunk number of every GP
  DevBuf pair_split;                     // per-lane partial sums of split remainder tiles
  DevBuf pair_post;                      // [G][P] mean | var of a swarm (sweep_pair.hip)
  int sweep_partials = 0;     // partials of max l0[S] the last confidence sweep left
  int sweep_choice = 0;       // sgp_ctx_set_sweep: 0 auto, 1 4-wave, 2 paired, 3 auto (mid kernel asked for)
  int last_sweep = 0;         // kernel of the last posterior sweep (sgp_ctx_last_sweep)
  int share_factors = 1;      // sgp_ctx_set_share: GPs with identical (X, kernel, noise)
                              // share the variance contraction (paired sweep)
  // sgp_grid_step_small: its result block lives in host memory the device writes directly
  // (pinned, mapped, coherent), the last word is a completion counter the host spins on
  double* step_host = nullptr;
  double* step_dev = nullptr;      // the same memory as the device sees it
  // ... and of the one-rank chain of the large-grid path (sgp_grid_sets_fused): result
  // block + the per-workgroup results of the last arg-max pass
  double* sets_host = nullptr;
  double* sets_dev = nullptr;
  size_t sets_cap = 0;             // doubles
  uint64_t step_seq = 0;
  // RCCL
  void* comm = nullptr;  // ncclComm_t
  int rank = 0, world = 1;
  // ... or the caller's collectives on host buffers (sgp_comm_init_host): the library
  // stages the device operands of an N-rank step through the host around them
  struct HostComm {
    int (*allreduce_max_f64)(void*, double*, int) = nullptr;
    int (*allreduce_max_i32)(void*, int32_t*, int) = nullptr;
    int (*allgather)(void*, const void*, void*, int64_t) = nullptr;
    void* user = nullptr;
  } hostcomm;
  int num_cu = 256;
};

struct sgp_gp {
  sgp_ctx* ctx = nullptr;
  KernDesc kern;
  double noise_var = 0.0;
  double jitter = 0.0;
  int64_t n = 0;
  int n_pad = 0;  // multiple of 16 (sweep blocks)
  int n_f = 0;    // multiple of 32 (factorisation leaves)
  int ld = 0;     // leading dimension / row capacity of Linv, Kmat, work
  uint64_t serial = 0;           // unique per sgp_gp_create
  uint64_t data_version = 0;     // bumped by every fit / append / removal
  std::vector<double> xhost;     // host copy of the training rows (n x d): the EXACT
                                 // check behind a hash match of two GPs' inputs
  std::vector<uint64_t> xhash;   // xhash[i]: hash of the first i + 1 training rows
  uint64_t prov = 0;             // hash of the operations that led to this factor
                                 // (fit at n, appends, removals): two GPs with equal
                                 // inputs AND equal history have the same bits in L^-1
  bool upd_valid = false;  // dev.upd* describes the step to the current data
  DevBuf X, Y, Xpad, Xs, XA, alpha, Apack, Linv, Kmat, work, tvec, updw, upd;
  GpDev dev;      // filled by set_data
};

struct sgp_grid {
  sgp_ctx* ctx = nullptr;
  int64_t N = 0;
  int d = 0;
  int G = 0;
  int64_t goff = 0;
  double* pts = nullptr;     // SoA [d][N]
  double* Q = nullptr;       // [N][2G]
  double* mean = nullptr;    // [G][N]
  double* var = nullptr;     // [G][N]
  uint8_t* S = nullptr;      // [N]
  uint8_t* M = nullptr;
  uint8_t* Gm = nullptr;
  uint8_t* cand = nullptr;   // candidate mask s
  double* w = nullptr;       // [N] max_i(u_i - l_i) of candidates
  double* partial = nullptr; // block partials
  int64_t partial_cap = 0;
  GpDev* gpdev = nullptr;    // [SGP_MAX_GPS] device copy of descriptors
  GpDev gpdev_host[SGP_MAX_GPS];   // ... and what it holds (stage_gpdev)
  int gpdev_count = 0;
  double* scal = nullptr;    // [8] resident scalars: [0] = max l0 over S
  int l0_pending = 0;        // > 0: scal[0] is still spread over that many
                             // entries of `partial` (deferred confidence pass)
  // tensor-grid structure (sgp_grid_set_axes; SepLaunch)
  bool axes_valid = false;
  uint32_t ax_count[SGP_MAX_D] = {0};
  uint32_t ax_stride[SGP_MAX_D] = {0};
  int ax_off[SGP_MAX_D] = {0};          // first entry of column k's axis in ax_vals
  std::vector<double> ax_host;          // axis values, concatenated
  DevBuf ax_vals;                       // ... on the device
  uint64_t ax_version = 0;              // bumped when an axis value changes (context)
  DevBuf sep_tab[SGP_MAX_GPS];          // factor tables of the GP in slot g
  uint64_t sep_key[SGP_MAX_GPS][3] = {{0}};   // (gp serial, data version, axes version)
};

// ---- helpers (api.hip) ------------------------------------------------------
int sgp_reserve(sgp_ctx* ctx, DevBuf* b, size_t bytes);
void* sgp_scratch(sgp_ctx* ctx, int slot, size_t bytes);  // nullptr on failure
int sgp_poison(sgp_ctx* ctx, void* p, size_t bytes);       // SGP_POISON=1: fill a fresh allocation with 0xFF
int sgp_h2d(sgp_ctx* ctx, void* dst, const void* src, size_t bytes);
int sgp_d2h(sgp_ctx* ctx, void* dst, const void* src, size_t bytes);  // syncs

// ---- kernel launchers (implemented in the .hip files) -----------------------
// factor.hip
int launch_kernel_matrix(sgp_ctx* ctx, const KernDesc& kd, const double* X1,
                         int64_t n1, const double* X2, int64_t n2, double* out,
                         int64_t ld, int symmetric_diag, double diag_add,
                         int64_t n_valid);
int factor_gp(sgp_gp* gp, int* info);  // Kmat -> Linv, Apack, alpha
int append_gp(sgp_gp* gp, double y, int* info);  // row n already in gp->X
int pop_gp(sgp_gp* gp);
int publish_gp(sgp_gp* gp);  // Apack / Xpad / Xs / dev descriptor from Linv
struct ExpanderOps {      // all arrays on the device; [g] blocks as noted
  const double* xc;       // [m][d] candidates
  const double* resid;    // [G][16]  u_c - mu_c
  double* Wpack;          // [G][wstride]  A operands (cand x j) of Ky^-1 k_c
  double* delta;          // [G][16]
  double* inv_s2;         // [G][16]
  double* tn2;            // [G][16]
  int64_t wstride;
  int m;                  // candidates; more than 16: groups of 16, arrays [group][Gs][...]
  int Gs;                 // GP slots of the [group][Gs] layouts (the launch's G)
  int active[SGP_MAX_GPS];
};
struct FrontArgs;   // sets_front.h: the fold of the front half's last step into k_expkt
// operands of every active GP in three launches (all GPs per launch)
int expander_operands_all(sgp_ctx* ctx, const GpDev* gps_dev, const GpDev* gps_host,
                          int G, int d, const ExpanderOps& ops,
                          const FrontArgs* fold = nullptr);

// sweep.hip
struct SweepPoints {
  const double* base;
  int64_t N;
  int64_t stride_row;  // elements
  int64_t stride_col;
};
struct ConfOut {
  double* Q;        // [N][2G] or null
  double* mean;     // [G][N]
  double* var;      // [G][N]
  uint8_t* S;       // [N] or null
  double* partial;  // [nblocks] max l0 over safe rows of the block, or null
  double beta;
  double fmin[SGP_MAX_GPS];
};
struct FitnessArgs {
  int swarm_type;
  double beta;
  double fmin[SGP_MAX_GPS];
  double scaling[SGP_MAX_GPS];
  double best_lower_bound;
  double* values;
  uint8_t* safe;
};
int sweep_num_partials(const sgp_ctx* ctx, int64_t N);
// rows_sharded: the rows are a rank's shard of a grid -- the sweep kernel is then chosen
// by the GPs alone (the same on every rank), never by the number of rows
int launch_sweep_conf(sgp_ctx* ctx, const GpDev* gps_dev, const GpDev* gps_host,
                      int G, int d, SweepPoints pts, ConfOut out,
                      const SepLaunch* sep = nullptr, bool rows_sharded = false);
// per-axis factor tables of one GP (SepLaunch::tab): out[a] = table of axis a =
// column cols[a] (count[cols[a]] points); the columns with one point are folded into
// out[0]
int launch_sep_tables(sgp_ctx* ctx, const GpDev& gp, int d, const uint32_t* count,
                      const double* axis_vals, const int* axis_off, int naxes,
                      const int* cols, double* const* out);
size_t sep_table_doubles(const GpDev& gp, uint32_t count);
// j-blocks a factor table has at least: the resident-factor kernel walks the j-blocks of the
// LARGEST factor of its launch (up to kMidMaxNB, sweep_mid.hip; at least 4) for every GP -- the
// blocks beyond a GP's own meet zeros of its L^-1, but they have to be there, and finite
constexpr int kSepMinBlocks = 8;
// rows == the declared tensor grid?  *mismatch (device int) counts the differences
int launch_verify_axes(sgp_grid* g, int* mismatch_dev);
int launch_sweep_fitness(sgp_ctx* ctx, const GpDev* gps_dev,
                         const GpDev* gps_host, int G, int d, SweepPoints pts,
                         FitnessArgs fa);
// few-points posterior / small-swarm step: small_path.h
constexpr int kSmallPoints = 4096;   // few-points posterior path (factor.hip)
constexpr int kSmallSwarm = 64;     // ... with the whole PSO step in one workgroup (swarm.hip)
struct ExpanderArgs {
  const double* Wpack;   // [G][n_pad_max/4][64] MFMA A-operand (cand x j)
  const double* xc;      // [m][d]
  const double* delta;   // [G][16]  (u_c - mu_c) / s2
  const double* inv_s2;  // [G][16]
  const double* tn2;     // [G][16]  |L^-1 k(X, x_c)|^2
  const double* stn;     // k_expander_many: [group][G][16] |L^-1 k_c| and the posterior
  const double* svc;     // standard deviation at x_c (written by launch_expander_many)
  const double* agg;     // ... [group][G][4] extremes of a group, box [group][2][d] (k_pass_agg)
  const double* box;
  const double* sagg;    // the same per SUPERGROUP of 8 consecutive groups (the scan of the grid
  const double* sbox;    // tests those first): [super][G][4], [super][2][d]
  int m;
  double beta;
  double fmin[SGP_MAX_GPS];
  int active[SGP_MAX_GPS];
  const uint8_t* S;
  const double* mean;    // [G][N]
  const double* var;
  int32_t* flags;        // [16][G] device
  int64_t wstride;       // doubles between consecutive GPs in Wpack
  double near_frac;      // >0: only rows with k(x,x_c) >= near_frac * k(x,x)
  int* count;            // m == 1: number of rows that passed the pre-filter
  int* list;             //         (zeroed by the caller) / their local indices
  int* wcount;           // k_expander_many: the 16-row segments with a possible pair, per GP
  int* wlist;            // (count, [G][N / 16] segments, masks of their listed rows); count /
  unsigned* wmask;       // list: the rows that pass the pair test of some candidate, [G][N]
};
int launch_expander_check(sgp_ctx* ctx, const GpDev* gps_dev,
                          const GpDev* gps_host, int G, int d, SweepPoints pts,
                          ExpanderArgs ea);
// ea.m candidates in groups of 16, every per-candidate array [group][G][...] (flags
// [candidate][G]): sweep.hip, k_expander_many
int launch_expander_many(sgp_ctx* ctx, const GpDev* gps_dev, int G, int d, SweepPoints pts,
                         ExpanderArgs ea);
struct Rank1Args {
  double* Q;
  double* mean;
  double* var;
  uint8_t* S;
  double* partial;
  double beta;
  double fmin[SGP_MAX_GPS];
  int which[SGP_MAX_GPS];
};
int rank1_num_blocks(int64_t N);
int launch_rank1(sgp_ctx* ctx, const GpDev* gps_dev, int G, int d,
                 SweepPoints pts, Rank1Args ra);

// sweep_tiny.hip: do the GPs of a launch go through the VALU kernel (every one with at most
// 48 observations)?
bool tiny_sweep_wanted(const sgp_ctx* ctx, const GpDev* gh, int Geff, int64_t rows,
                       bool rows_sharded);

// sweep_mid.hip: 49 .. 128 observations, single-part kernels, d <= 4, all GPs resident in LDS
bool mid_sweep_wanted(const sgp_ctx* ctx, const GpDev* gh, int Geff, int d);
// ... 129 .. 256 observations with factor tables (a tensor grid, RBF parts): in passes of row blocks
bool mid_passes_wanted(const sgp_ctx* ctx, const GpDev* gh, int Geff, const SepLaunch* sep,
                       const ConfOut& conf);

// sweep_pair.hip: does the launch take the paired-wave kernel (a GP with more than 256 rows)?
bool pair_sweep_wanted(const sgp_ctx* ctx, const GpDev* gh, int Geff);

// step_small.hip: a whole SafeOpt.optimize() of a small grid in one launch
constexpr int64_t kStepSmallRows = 16384;
bool step_small_eligible(const sgp_ctx* ctx, const GpDev* gh, int G, int64_t N);
int launch_step_small(sgp_grid* g, const GpDev* gps_dev, const GpDev* gh, int G, double beta,
                      const double* fmin, const double* scaling, const double* thr_beta,
                      double* res, int nfront, int nfl, uint64_t seq);
// count_dev != nullptr: the list was formed on the device, m is its room and *count_dev its length
int launch_cand_all(sgp_grid* g, const GpDev* gps_dev, const GpDev* gh, int G, double beta,
                    const double* fmin, const int64_t* clist_dev, int m, double* ops,
                    int32_t* flags, const int* count_dev = nullptr);
int launch_small_pack(sgp_grid* g, const int* list_dev, const int* count_dev, int cap,
                      int64_t* hdr, int64_t* clist, double* wout, int32_t* flags);
size_t cand_ops_doubles(int m, int G);
constexpr int kStepResWords = 64;     // doubles of the result block (6 + d + 3 G + ... <= 50)

// sets.hip
int launch_reduce_max(sgp_ctx* ctx, const double* in, int64_t n, double* out);
int launch_safe_set(sgp_grid* g, const double* fmin);  // from Q -> S, partial
int launch_maximizers(sgp_grid* g, double max_l,
                      const double* max_l_dev = nullptr);  // device value wins
int launch_candidates(sgp_grid* g, double max_var, const double* max_width_dev,
                      const double* scaling, const double* thr_beta,
                      int full_sets, unsigned long long* counts_dev);
int launch_gather_top(sgp_grid* g, const int64_t* gidx_dev, double* x,
                      double* mean, double* Q);
int launch_stage_batch(sgp_grid* g, const int64_t* gidx_dev, const int* nfound_dev, int K,
                       double* xc, int n_xc_resid, int32_t* flags, int n_flag_words);
int launch_stage_top(sgp_grid* g, const double* x_top, const double* mean_top,
                     const double* q_top, double* xc, double* resid);
int launch_mark_top_if(sgp_grid* g, const int64_t* gidx_dev, const int* nfound_dev,
                       const int32_t* flags_dev, const double* fmin);
int launch_mark_if(sgp_grid* g, int64_t li, const int32_t* flags_dev,
                   const double* fmin);
// a pass of the expander loop over many candidates (sets.hip): selection by a key
// histogram (sel_dev: { double thr; int count; int est }), operand staging, hits
int launch_pass_select(sgp_grid* g, int mode, double cut_w, int64_t cut_idx, double lo,
                       double hi, int want, void* sel_dev, int* list_dev, unsigned* hist_dev,
                       int* counts_dev);
int launch_pass_hist(sgp_grid* g, int mode, double cut_w, int64_t cut_idx, double lo, double hi,
                     unsigned* hist_dev);
int launch_pass_list(sgp_grid* g, int mode, double cut_w, int64_t cut_idx, void* sel_dev,
                     int* list_dev, int* counts_dev);
int launch_pass_gather(sgp_grid* g, const int* list_dev, int count, int mode, int64_t* gidx,
                       double* key, double* x, double* resid);
int launch_pass_stage(sgp_grid* g, const int* list_dev, int count, double* xc, double* resid);
// the Lipschitz test of many candidates (sets.hip: k_lip_*): rows and upper bounds staged from
// list_dev (local rows) or copied from the host arrays; work: (count (d + G) + groups (2 d + 1))
// doubles on the device
int launch_lipschitz_many(sgp_grid* g, int G, const double* fmin, const double* lipschitz,
                          const int* list_dev, int count, const double* xc_in, const double* uc_in,
                          double* work, int32_t* flags_dev);
int launch_pass_result(sgp_grid* g, const int* list_dev, int count, const int32_t* flags_dev,
                       const double* fmin, int mode, double* res_dev);
int launch_topk(sgp_grid* g, int mode, double cut_w, int64_t cut_idx, int k,
                double* w_out_dev, int64_t* idx_out_dev, int* n_out_dev);
int launch_count_ties(sgp_grid* g, const double* w_top_dev, const int* n_found_dev,
                      int* ties_dev);
int launch_lipschitz(sgp_grid* g, int G, const double* fmin,
                     const double* lipschitz, int m, const double* xc_dev,
                     const double* uc_dev, int32_t* flags_dev);
int launch_argmax(sgp_grid* g, int mode, const double* scaling,
                  double* value_dev, int64_t* idx_dev);
int launch_sets_front_fused(sgp_grid* g, double max_l, const double* l0_part,
                            int n_l0, const double* max_l_dev,
                            const double* scaling, const double* thr_beta,
                            double* res, double* max_l_slot, double* xc,
                            int n_xc_resid, int32_t* flags, int n_flag_words,
                            FrontArgs* fold = nullptr);
int launch_front_final(sgp_ctx* ctx, const FrontArgs& fa);
int argmax_marked_blocks(int64_t N);
int launch_merge_front(sgp_grid* g, const double* all, int world, int nfront, double* res,
                       double* xc, int n_xc_resid, int32_t* flags, int n_flag_words);
int launch_merge_argmax(sgp_ctx* ctx, const double* all, int world, double* out_v,
                        int64_t* out_i);
int launch_argmax_marked(sgp_grid* g, const double* scaling, const double* fmin,
                         const int32_t* flags_dev, const int64_t* cand_gidx_dev,
                         const int* nfound_dev, int32_t* flags_out,
                         double* value_dev, int64_t* idx_dev, double* host_part = nullptr);
int launch_fill_cols(sgp_grid* g, const double* c, int nc);
int launch_gather_rows(sgp_grid* g, const int64_t* lidx_dev, int m, double* x,
                       double* mean, double* var, double* Q);
int launch_mark(sgp_grid* g, const int64_t* lidx_dev, int m, int value = 1);
int launch_swarm_grow(sgp_ctx* ctx, const KernDesc& kd, const double* S, int64_t m,
                      const double* B, int n, double scale2, double thr,
                      double* part, int* list, uint8_t* accept);
int swarm_grow_chunks(int64_t m);
int launch_pso_init_vel(sgp_ctx* ctx, int64_t P, int d, double* vel,
                        const double* vscale, const double* rand, uint64_t seed);
int launch_pso_move(sgp_ctx* ctx, int64_t P, int d, double* pos, double* vel,
                    const double* best, const double* gbest, const double* vscale,
                    const double* bounds, double inertia, const double* rand,
                    uint64_t seed, uint32_t draw);
int launch_pso_best(sgp_ctx* ctx, int64_t P, int d, const double* values,
                    const uint8_t* safe, const double* pos, double* best,
                    double* best_values, double* gbest, int init);
int launch_import_points(sgp_ctx* ctx, const double* src, int64_t N, int d,
                         int64_t stride_row, int64_t stride_col, double* dst);
