/*
 * safeopt_hip.h -- C ABI of libsafeopt_hip.so (MI355X / gfx950, HIP + RCCL).
 *
 * The drop-in boundary for SafeOpt's GP-posterior + safe-set hot path.  The
 * reference (befelix/SafeOpt, pure Python) has no FFI; its seam is the
 * duck-typed "GPy model" handle plus the NumPy sweep in safeopt/gp_opt.py.
 * Every entry point below names the reference call site(s) it replaces
 * (paths relative to /root/reference).  INTEGRATION.md shows the ctypes stub
 * a maintainer of the reference would add.
 *
 * Conventions
 *   - extern "C", plain pointers + sizes, no C++ / torch types.
 *   - every function returns 0 on success; <0 = HIP/RCCL/runtime error,
 *     >0 = numerical failure (e.g. Cholesky info).  sgp_last_error() gives
 *     the message.  No exception crosses the ABI.
 *   - all host buffers are borrowed for the duration of the call only;
 *     outputs go to caller-allocated buffers.  All real data is float64,
 *     masks are uint8 (NumPy bool), indices int64.
 *   - one sgp_ctx per (process, device); calls on one ctx are serialised by
 *     the caller (the reference is single-threaded too).
 */
#ifndef SAFEOPT_HIP_H
#define SAFEOPT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SGP_MAX_D 8        /* input dimension incl. context columns            */
#define SGP_MAX_PARTS 4    /* factors of a product kernel                      */
#define SGP_MAX_GPS 8      /* objective + constraints                          */
#define SGP_TOPK 16        /* expander candidates examined per pass            */

/* kernel kinds: GPy.kern.RBF / Matern32 / Matern52 (Stationary.K_of_r)        */
enum { SGP_RBF = 0, SGP_MATERN32 = 1, SGP_MATERN52 = 2 };
/* sgp_grid_download selectors                                                 */
enum { SGP_Q = 0, SGP_S = 1, SGP_M = 2, SGP_G = 3, SGP_MEAN = 4, SGP_VAR = 5,
       SGP_CAND = 6, SGP_WIDTH = 7 };
/* sgp_grid_argmax modes                                                       */
enum { SGP_ARGMAX_MG_WIDTH = 0, SGP_ARGMAX_UCB = 1, SGP_ARGMAX_LCB = 2 };
/* sgp_swarm_fitness swarm types (gp_opt.py:901-1013)                          */
enum { SGP_SWARM_GREEDY = 0, SGP_SWARM_MAXIMIZERS = 1, SGP_SWARM_EXPANDERS = 2,
       SGP_SWARM_SAFE_SET = 3 };

typedef struct sgp_ctx sgp_ctx;   /* device, stream, scratch, RCCL communicator */
typedef struct sgp_gp sgp_gp;     /* one GP: data, L^-1 (packed), alpha         */
typedef struct sgp_grid sgp_grid; /* resident candidate rows + Q/S/M/G          */

/* ---- context ---------------------------------------------------------------*/
int sgp_device_count(int* n);
int sgp_create(int device, sgp_ctx** out);
void sgp_destroy(sgp_ctx* ctx);
/* message of the last failing call on ctx (ctx may be NULL: process-global)   */
const char* sgp_last_error(sgp_ctx* ctx);
int sgp_sync(sgp_ctx* ctx);

/* ---- GP handle: replaces GPy.models.GPRegression as SafeOpt uses it --------
 * kernel = product over n_parts of  variance_p * k_kind_p(r_p),
 * r_p^2 = sum_k ((x_k - x'_k) * inv_ls[p*d + k])^2   (inv_ls = 0 on columns
 * outside the part's active_dims; = 1/lengthscale otherwise).               */
int sgp_gp_create(sgp_ctx* ctx, int d, int n_parts, const int* kinds,
                  const double* variances, const double* inv_ls,
                  double noise_var, sgp_gp** out);
void sgp_gp_destroy(sgp_gp* gp);
/* gp.set_XY (gp_opt.py:227, 267, 275): K = k(X,X), Ky = K + (noise+1e-8) I,
 * L = jitchol(Ky), L^-1, alpha = Ky^-1 Y -- all on the device.  X is (n,d)
 * row-major, Y is (n).  chol_info: 0 ok, k>0 = pivot k not positive even
 * after 5 jitter escalations (GPy jitchol); jitter_used: diagonal jitter.    */
int sgp_gp_set_data(sgp_gp* gp, const double* X, const double* Y, int64_t n,
                    int* chol_info, double* jitter_used);
/* gp.set_XY with ONE more / ONE fewer row -- what add_new_data_point
 * (gp_opt.py:230-255 -> :227) and remove_last_data_point (:269-278 -> :275) do
 * every iteration: bordered update of L^-1 and alpha, O(n^2) instead of the
 * O(n^3) re-factorisation.  append: info = 0 ok,
 * > 0 the bordered pivot is not positive (refit with sgp_gp_set_data, which
 * applies GPy's jitter), -1 no spare capacity (same remedy).                  */
int sgp_gp_append(sgp_gp* gp, const double* x, double y, int* info);
int sgp_gp_pop(sgp_gp* gp);
/* gp.predict_noiseless / gp._raw_predict (gp_opt.py:469, 591, 929, 973, 1117,
 * 1132; utilities.py:203, 282, 355).  Xnew element (r,c) at
 * Xnew[r*stride_row + c*stride_col] (strides in elements: C or F order).
 * var is clipped to [1e-15, inf) like GPy.                                   */
int sgp_gp_predict(sgp_gp* gp, const double* Xnew, int64_t N,
                   int64_t stride_row, int64_t stride_col, double* mean,
                   double* var);
/* test hook: dense L^-1 (n x n, row-major) and alpha (n)                     */
int sgp_gp_get_factor(sgp_gp* gp, double* Linv, double* alpha);
/* gp.kern.K(X, X2) (gp_opt.py:847, 1093; utilities.py:89, 135): out is
 * (n1,n2) row-major; X1 (n1,d), X2 (n2,d) row-major.                         */
int sgp_kern_K(sgp_ctx* ctx, int d, int n_parts, const int* kinds,
               const double* variances, const double* inv_ls, const double* X1,
               int64_t n1, const double* X2, int64_t n2, double* out);

/* ---- resident candidate grid: SafeOpt.inputs / Q / S / M / G ----------------
 * (gp_opt.py:358-390).  `base` element (r,c) at byte offset
 * r*stride_row_B + c*stride_col_B (the stock grid is F-ordered); rows are
 * this rank's shard, global index = global_offset + local row.               */
int sgp_grid_create(sgp_ctx* ctx, const double* base, int64_t N, int d,
                    int64_t stride_row_B, int64_t stride_col_B, int G,
                    int64_t global_offset, sgp_grid** out);
void sgp_grid_destroy(sgp_grid* grid);
/* SafeOpt.context setter (gp_opt.py:439-451): fill the last nc columns       */
int sgp_grid_set_context(sgp_grid* grid, const double* c, int nc);
/* Declares the rows a TENSOR grid, which is what SafeOpt's parameter_set is
 * (linearly_spaced_combinations, utilities.py:21-54, followed by constant context
 * columns, gp_opt.py:439-451): global row i has column k equal to
 * values_k[(i / stride[k]) % count[k]], values = the d axes concatenated (count[k]
 * doubles each; count 1 = a constant column).  The declaration is checked against the
 * resident rows bit for bit; *ok = 1 when it holds, 0 when it does not (nothing
 * changes then).  On a verified tensor grid the sweeps of GPs whose kernels are
 * products of RBF parts -- k(X_j, x) is then a product of one factor per axis -- read
 * per-axis factor tables instead of evaluating the covariance (same results within
 * rounding; sgp_ctx_set_sweep(... + 8) switches it off).  Replaces nothing in the
 * reference: gp.predict_noiseless(self.inputs) (gp_opt.py:469) evaluates every
 * covariance of the grid in full.                                               */
int sgp_grid_set_axes(sgp_grid* grid, int d, const int64_t* count,
                      const int64_t* stride, const double* values, int* ok);
/* update_confidence_intervals + compute_safe_set (gp_opt.py:453-481), fused:
 * for every GP mean/var over all rows, Q[:,2i] = mean - beta*std,
 * Q[:,2i+1] = mean + beta*std, S = all(Q[:, ::2] > fmin).
 * out2 = { max(l0[S]) or -inf, any(S) }  (local rows); out2 == NULL: no read-
 * back and no stream sync, the value stays on the device for
 * sgp_grid_sets_fused.                                                        */
int sgp_grid_confidence(sgp_grid* grid, sgp_gp* const* gps, int G, double beta,
                        const double* fmin, double* out2);
/* gp.predict_noiseless(self.inputs) for every GP (gp_opt.py:469, and the
 * re-predictions of the expander loop, :585-602): refreshes the resident
 * mean / var only -- Q, S, M, G are not touched.  Needed when the intervals
 * were assigned by hand (opt.Q = ...) or the data changed without an interval
 * update before compute_sets().                                              */
int sgp_grid_posterior(sgp_grid* grid, sgp_gp* const* gps, int G);
/* update_confidence_intervals after ONE sgp_gp_append on the GPs flagged in
 * which[]: closed-form rank-1 update of the resident mean/var,
 *   c(x) = k(x,x*) - k(X,x)^T Ky^-1 k(X,x*),  mean += c r / s2,
 *   var = max(var - c^2 / s2, 1e-15),
 * O(n) per row instead of the O(n^2) sweep, then Q and S for all GPs from the
 * resident mean/var with the given beta.  out2 as sgp_grid_confidence.        */
int sgp_grid_rank1_update(sgp_grid* grid, sgp_gp* const* gps, int G,
                          const int* which, double beta, const double* fmin,
                          double* out2);
/* replace Q by host values (N x 2G row-major) and recompute S (tests, and
 * users who edit opt.Q by hand).                                             */
int sgp_grid_upload_Q(sgp_grid* grid, const double* Q, const double* fmin,
                      double* out2);
/* gp_opt.py:511-513: M = S & (u0 >= max_l); out = max(u0[M]-l0[M]) or -inf   */
int sgp_grid_maximizers(sgp_grid* grid, double max_l, double* out);
/* gp_opt.py:527-540: candidate mask.  full_sets=0: s = S & ~M &
 * (max_i (u_i-l_i)/scaling_i > max_var) & any_i(u_i-l_i > thr_beta_i);
 * full_sets=1: s = S.  Also clears G.  counts = {#candidates, #unsafe}.      */
int sgp_grid_candidates(sgp_grid* grid, double max_var, const double* scaling,
                        const double* thr_beta, int full_sets, int64_t* counts);
/* gp_opt.py:542-552: next <=k candidates in visiting order after the cut
 * (cut_w, cut_idx), i.e. descending w = max_i(u_i - l_i), ties: higher global
 * index first.  mode 1 (full_sets): ascending global index after cut_idx.
 * Start with cut_w = +inf, cut_idx = INT64_MAX (mode 1: cut_idx = -1).       */
int sgp_grid_topk(sgp_grid* grid, int mode, double cut_w, int64_t cut_idx,
                  int k, double* w_out, int64_t* gidx_out, int* n_out);
/* rows of the resident arrays for m GLOBAL indices owned by this rank:
 * x (m,d), mean (m,G), var (m,G), Q (m,2G)                                   */
int sgp_grid_gather_rows(sgp_grid* grid, const int64_t* gidx, int m, double* x,
                         double* mean, double* var, double* Q);
/* expander test of gp_opt.py:579-606 for m<=SGP_TOPK candidates at once, as a
 * rank-1 posterior update instead of two re-factorisations: for every GP i
 * with finite fmin_i, flags[c*G+i] = any over UNSAFE local rows x of
 *   mu2 - beta*sqrt(var2) >= fmin_i,
 *   mu2 = mu_i(x) + c(x) (u_i(x_c) - mu_i(x_c)) / s2,
 *   var2 = max(var_i(x) - c(x)^2 / s2, 1e-15),
 *   c(x) = k(x,x_c) - k(X,x)^T Ky^-1 k(X,x_c),  s2 = var_i(x_c)+noise+1e-8.
 * xc (m,d), mu_c/u_c (m,G) come from sgp_grid_gather_rows on the owner.
 * near_frac > 0 restricts the scan to rows with k(x,x_c) >= near_frac*k(x,x)
 * (a cheap first probe: a hit there already proves the candidate an expander;
 * no hit proves nothing, repeat with near_frac = 0 for the exact answer).     */
int sgp_grid_expander_check(sgp_grid* grid, sgp_gp* const* gps, int G,
                            double beta, const double* fmin, int m,
                            const double* xc, const double* mu_c,
                            const double* u_c, double near_frac,
                            int32_t* flags);
/* Lipschitz variant, gp_opt.py:558-576: flags[c*G+i] = any over unsafe rows
 * of u_i(x_c) - L_i * ||x_c - x||_2 >= fmin_i                                */
int sgp_grid_lipschitz_check(sgp_grid* grid, int G, const double* fmin,
                             const double* lipschitz, int m, const double* xc,
                             const double* u_c, int32_t* flags);
/* N-rank variant of sgp_grid_sets_front after sgp_grid_confidence(out2 = NULL):
 * max(l0[S]) and the maximiser width are all-reduced in stream (RCCL on the
 * context's stream, device-resident operands), so the front half of
 * compute_sets costs one device round trip per rank instead of three.
 * out5[0] = GLOBAL max(u0[M]-l0[M]); *max_l_out = GLOBAL max(l0[S]) (-inf: no
 * safe point on any rank); the other outputs (out5 has SIX entries, as for
 * sgp_grid_sets_front) describe this rank's shard.                            */
int sgp_grid_sets_front_comm(sgp_grid* grid, const double* scaling,
                             const double* thr_beta, double* out5, double* x_top,
                             double* mean_top, double* q_top, double* max_l_out);
/* Fused passes (one stream sync each; results identical to the step-by-step
 * calls above):
 * front = gp_opt.py:511-552: M and max_var (have_max_var = 0; with
 *   have_max_var = 1 the caller already ran sgp_grid_maximizers and passes the
 *   all-reduced max_var), candidate mask, counts, this shard's first
 *   candidate in visiting order and its rows.
 *   out5 (SIX entries) = {max(u0[M]-l0[M]) (0 if given), #candidates, #unsafe,
 *           w_top, idx_top or -1, #candidates of this shard whose width equals
 *           w_top bit for bit (exact ties decide the reference's visiting
 *           order, gp_opt.py:542-552)}
 * back = expander test of ONE candidate on this shard's unsafe rows
 *   (gp_opt.py:579-606), G mark if mark != 0 and every active GP certified it
 *   (gp_opt.py:615; single rank only), then the M|G arg-max (:642-644).       */
int sgp_grid_sets_front(sgp_grid* grid, double max_l, int have_max_var,
                        double max_var, const double* scaling,
                        const double* thr_beta, double* out5, double* x_top,
                        double* mean_top, double* q_top);
int sgp_grid_sets_back(sgp_grid* grid, sgp_gp* const* gps, int G, double beta,
                       const double* fmin, const double* xc, const double* mu_c,
                       const double* u_c, double near_frac, int64_t gidx_c,
                       int mark, const double* scaling, int32_t* flags,
                       double* value, int64_t* gidx);
/* Both halves in one call and ONE stream sync (single GPU): the first candidate
 * stays on the device, the probe scan (near_frac) runs on it, G is marked if
 * every active GP certifies it, and the M|G arg-max follows.  Outputs as in
 * sgp_grid_sets_front + sgp_grid_sets_back, plus out5[5] (SIX entries here) =
 * number of candidates whose width equals the first one's bit for bit (> 1: the
 * reference's argsort()[::-1] decides among them, gp_opt.py:542-552; the host
 * settles it); flags are void when out5 reports no candidate or no unsafe row
 * (G is then left untouched).  max_l = NaN: use
 * the value a preceding sgp_grid_confidence / _rank1_update call with
 * out2 == NULL left on the device, and report it in *max_l_out (-inf = no safe
 * point) -- a whole SafeOpt.optimize() is then one device round trip.        */
int sgp_grid_sets_fused(sgp_grid* grid, sgp_gp* const* gps, int G, double beta,
                        const double* fmin, double max_l, const double* scaling,
                        const double* thr_beta, double near_frac, double* out5,
                        double* x_top, double* mean_top, double* q_top,
                        int32_t* flags, double* value, int64_t* gidx,
                        double* max_l_out);
/* One pass of the expander loop (safeopt/gp_opt.py:557-612) over the next k <= 16
 * candidates behind the cut (cut_w, cut_idx) in visiting order -- sgp_grid_topk,
 * sgp_grid_gather_rows and the exact sgp_grid_expander_check of those rows -- with one
 * stream synchronisation instead of three host round trips: w_out / gidx_out / n_out as
 * sgp_grid_topk, flags[c * G + i] != 0: candidate c lifts an unsafe row above fmin_i.
 * One rank (the shard is the grid).                                                 */
int sgp_grid_expander_batch(sgp_grid* grid, sgp_gp* const* gps, int G, double beta,
                            const double* fmin, int mode, double cut_w, int64_t cut_idx,
                            int k, double* w_out, int64_t* gidx_out, int* n_out,
                            int32_t* flags);

/* A pass of the expander loop over MANY candidates (safeopt/gp_opt.py:557-612 where the
 * loop goes far: no expander among the first candidates -- or none at all, the state of a
 * converged run -- and full_sets = True, :553-555, which visits every safe row): the next
 * ~want candidates behind the cut in visiting order (key descending -- the interval width,
 * or minus the row index with mode = 1 -- then index descending; strictly behind: key <
 * cut_w, or key == cut_w and row < cut_idx) are chosen by a histogram of the keys over
 * [key_lo, key_hi] (no sort) and ALL of them tested in ONE scan of the unsafe rows: the
 * rank-1 test of sgp_grid_expander_check per (row, candidate).  Two stream synchronisations
 * per pass whatever its size.
 *   out6 = { candidates tested, expanders among them,
 *            mode 0: key and global row of the FIRST expander in visiting order (nothing is
 *                    marked in G: the caller settles exact ties, gp_opt.py:542-552),
 *            key below which the candidates are still untested (-inf: none left),
 *            scaling != NULL and mode 0: the global row of sgp_grid_argmax(SGP_ARGMAX_MG_WIDTH)
 *            taken behind the test (gp_opt.py:631-641) -- the next query point when the pass
 *            ends the loop without an expander, for no further round trip; else -1 }
 *   mode 1 (full_sets): every expander of the pass is marked in G on the device.
 * One rank (the shard is the grid).                                                   */
int sgp_grid_expander_pass(sgp_grid* grid, sgp_gp* const* gps, int G, double beta,
                           const double* fmin, int mode, double cut_w, int64_t cut_idx,
                           double key_lo, double key_hi, int want, const double* scaling,
                           double* out6);
/* The same pass with Lipschitz certificates (safeopt/gp_opt.py:558-576 in place of :577-606):
 * candidate c is an expander when for every GP i with a constraint some unsafe row x has
 * u_i(x_c) - L_i |x_c - x|_2 >= fmin_i -- the comparison of sgp_grid_lipschitz_check per
 * (row, candidate), the pairs pruned by bounding boxes of 16 rows / 16 candidates.
 * Selection, modes and out6 as sgp_grid_expander_pass.  One rank.                       */
int sgp_grid_lipschitz_pass(sgp_grid* grid, int G, const double* fmin,
                            const double* lipschitz, int mode, double cut_w,
                            int64_t cut_idx, double key_lo, double key_hi, int want,
                            const double* scaling, double* out6);

/* The same pass on N ranks (safeopt/gp_opt.py:557-612 on a row-sharded grid), in three calls
 * with the ranks' agreement in between: (1) this shard's 4096-bin histogram of the keys of its
 * candidates behind the cut, over [key_lo, key_hi] -- the ranks sum the histograms and pick
 * ONE threshold thr; (2) this shard's candidates behind the cut with key >= thr (at most cap):
 * global rows, keys, the rows themselves [count][d] and u_i - mu_i [count][G] -- the ranks
 * gather them into one list; (3) the rank-1 test of ALL K listed candidates against this
 * shard's unsafe rows: flags[c * G + i] != 0 when candidate c lifts one of them above fmin_i
 * -- the ranks OR the flags; the first expander in visiting order is the hit with the
 * largest (key, row).  mode as in sgp_grid_expander_pass.                               */
int sgp_grid_pass_hist(sgp_grid* grid, int mode, double cut_w, int64_t cut_idx, double key_lo,
                       double key_hi, uint32_t* hist4096);
int sgp_grid_pass_list(sgp_grid* grid, int mode, double cut_w, int64_t cut_idx, double thr,
                       int cap, int* count, int64_t* gidx, double* key, double* x, double* resid);
int sgp_grid_pass_test(sgp_grid* grid, sgp_gp* const* gps, int G, double beta,
                       const double* fmin, int K, const double* xc, const double* resid,
                       int32_t* flags);
/* The test of (3) with Lipschitz certificates (safeopt/gp_opt.py:558-576): xc [K][d] and the
 * candidates' upper bounds uc [K][G] -- sgp_grid_pass_list with mode | 2 returns u_i in place of
 * u_i - mu_i -- against this shard's unsafe rows; flags as above (one value per candidate, set
 * for every GP: the comparison is monotone in the distance).                            */
int sgp_grid_pass_lipschitz_test(sgp_grid* grid, int G, const double* fmin,
                                 const double* lipschitz, int K, const double* xc,
                                 const double* uc, int32_t* flags);

/* SMALL grids -- the reference's own regime (safeopt/gp_opt.py:651-675 on the 1000-point
 * grid of examples/1d_example.ipynb with n <= 20 observations; BASELINE.json config 1):
 * one whole SafeOpt.optimize() = update_confidence_intervals (gp_opt.py:453-476) +
 * compute_sets (:483-615) + the arg-max of get_new_query_point (:635-649) in ONE launch
 * of one workgroup and one read-back, for grids of at most 16384 rows and GPs with at most
 * 48 observations.  The expander test of the first candidate is the exact scan over all
 * unsafe rows.  Outputs as sgp_grid_sets_fused; Q / S / M / G / mean / var / candidate
 * mask and widths stay resident as the large-grid path leaves them.
 * sgp_grid_step_small_ok: 1 when the grid and the GPs qualify.                        */
int sgp_grid_step_small(sgp_grid* grid, sgp_gp* const* gps, int G, double beta,
                        const double* fmin, const double* scaling,
                        const double* thr_beta, double* out5, double* x_top,
                        double* mean_top, double* q_top, int32_t* flags, double* value,
                        int64_t* gidx, double* max_l_out);
int sgp_grid_step_small_ok(sgp_grid* grid, sgp_gp* const* gps, int G);
/* ... and when its first candidate is no expander: the expander test of gp_opt.py:579-606
 * for EVERY candidate of the list (global row indices, any order) in two launches and one
 * round trip; flags[c * G + i] != 0: candidate c lifts an unsafe row above fmin_i.  The
 * caller walks the flags in the reference's visiting order (gp_opt.py:542-557).         */
int sgp_grid_expanders_small(sgp_grid* grid, sgp_gp* const* gps, int G, double beta,
                             const double* fmin, const int64_t* gidx, int m,
                             int32_t* flags);
/* The same for EVERY candidate of the grid (the mask of sgp_grid_candidates / the one-launch
 * step), listed on the device in row order: global rows, widths (max_i (u_i - l_i), the key of
 * the visiting order, safeopt/gp_opt.py:542-552) and flags of the first *count candidates in
 * one read-back -- the caller sorts and walks them.  *count > cap: only *count is valid.  */
int sgp_grid_expanders_small_all(sgp_grid* grid, sgp_gp* const* gps, int G, double beta,
                                 const double* fmin, int cap, int* count, int64_t* gidx,
                                 double* width, int32_t* flags);

/* The same on N ranks (row shards, sgp_comm_init on the grid's context) after a
 * confidence pass without read-back: max l0[S] and the maximiser width are
 * all-reduced in stream; every rank's first candidate (with its row, counts and
 * tie count) goes through ONE in-stream all-gather and is merged on the device
 * into the first candidate of the WHOLE grid in the visiting order of
 * gp_opt.py:542-552; every rank scans its own unsafe rows for it, the G flags are
 * all-reduced (max) in stream (gp_opt.py:611-612: any over all rows), the owner
 * of the candidate marks G if every active GP certified it, and the (value,
 * index) pairs of the local M|G arg-maxima are all-gathered and merged
 * (gp_opt.py:635, 642-644: first index among equals).  One stream sync; every
 * rank returns the same numbers: counts are totals over the grid, indices are
 * global.  Without a communicator (one rank) it equals sgp_grid_sets_fused with
 * max_l = NaN.                                                                  */
int sgp_grid_sets_fused_comm(sgp_grid* grid, sgp_gp* const* gps, int G, double beta,
                             const double* fmin, const double* scaling,
                             const double* thr_beta, double near_frac, double* out5,
                             double* x_top, double* mean_top, double* q_top,
                             int32_t* flags, double* value, int64_t* gidx,
                             double* max_l_out);
/* gp_opt.py:615: G[idx] = True for owned global indices                      */
int sgp_grid_mark_expanders(sgp_grid* grid, const int64_t* gidx, int m);
/* ... and G[gidx] = False: when exact ties in the visiting order
 * (gp_opt.py:542-552, argsort()[::-1]) make another candidate of the same
 * width the first expander.                                                    */
int sgp_grid_unmark_expanders(sgp_grid* grid, const int64_t* gidx, int m);
/* get_new_query_point / get_maximum arg-max (gp_opt.py:635, 642-644,
 * 708-710), first (lowest) global index wins among equal values.
 * value = -inf and gidx = -1 when the masked set is empty.                   */
int sgp_grid_argmax(sgp_grid* grid, int mode, const double* scaling,
                    double* value, int64_t* gidx);
/* copy a resident array to the host: Q (N,2G) f64 | S/M/G (N) u8 |
 * mean/var (G,N) f64                                                         */
/* S / M / G of the shard from the host: the reference's arrays are mutated in place
 * (safeopt/gp_opt.py:481, 505-506, 511, 615) and user code may write into them; the next
 * arg-max (get_new_query_point, gp_opt.py:635-649) reads what was uploaded.            */
int sgp_grid_upload_mask(sgp_grid* grid, int what, const uint8_t* mask);
int sgp_grid_download(sgp_grid* grid, int what, void* out);

/* ---- SafeOptSwarm._compute_particle_fitness (gp_opt.py:901-1013) -----------
 * particles (P,d) row-major; values (P) f64; safe (P) u8.                    */
int sgp_swarm_fitness(sgp_ctx* ctx, sgp_gp* const* gps, int G, int swarm_type,
                      const double* particles, int64_t P, double beta,
                      const double* fmin, const double* scaling,
                      double best_lower_bound, double* values, uint8_t* safe);

/* ---- SafeOptSwarm safe-set growth (gp_opt.py:1089-1111) ---------------------
 * After a maximizer / expander swarm run the reference appends, in order, every
 * best position B_j (n,d row-major) whose prior correlation
 * gp.kern.K(B_j, p) / scaling[0]^2 is <= thr (0.95) with every point p of the
 * safe set S (m,d row-major) and with every B_i accepted before it.
 * scale2 = scaling[0]^2; accept (n) u8 = 1 for the rows to append.            */
int sgp_swarm_grow(sgp_ctx* ctx, sgp_gp* gp0, const double* S, int64_t m,
                   const double* B, int64_t n, double scale2, double thr,
                   uint8_t* accept);

/* ---- SwarmOptimization on the device (swarm.py:61-146) -----------------------
 * init_swarm (init != 0: velocities = rand * velocity_scale, personal bests :=
 * positions / their fitness, global best := arg-max, unmasked as in the
 * reference) followed by `iters` iterations of run_swarm (inertia0, += step per
 * iteration; velocity clip 10 * velocity_scale; positions clipped to bounds
 * (d,2) when given), the fitness being the fused posterior kernel of
 * sgp_swarm_fitness.  State arrays are row-major (P,d) / (P) / (d), in and out
 * (with init only `positions` is read).
 * rand != NULL: the uniform numbers in the order the reference draws them --
 * P*d for init, then 2*P*d per iteration (np.random.rand(2*P, d): r1 rows, r2
 * rows) -- the run is then bit-identical to the host implementation.
 * rand == NULL: counter-based device generator (Philox4x32-10) keyed by seed. */
int sgp_swarm_run(sgp_ctx* ctx, sgp_gp* const* gps, int G, int swarm_type,
                  double beta, const double* fmin, const double* scaling,
                  double best_lower_bound, int64_t P, double* positions,
                  double* velocities, double* best_positions, double* best_values,
                  double* global_best, const double* velocity_scale,
                  const double* bounds, int init, int iters, double inertia0,
                  double step, const double* rand, uint64_t seed);

/* ---- multi-GPU: one process per GPU, RCCL over xGMI -------------------------
 * The only cross-rank traffic of the path is a handful of scalars per
 * iteration (max / any / arg-max / top-k merge).                             */
int sgp_comm_unique_id(void* id128);              /* rank 0: ncclGetUniqueId  */
int sgp_comm_init(sgp_ctx* ctx, const void* id128, int rank, int world);
int sgp_comm_count(sgp_ctx* ctx, int* n);           /* ncclCommCount: ranks that
                                                     * really joined (1 without a
                                                     * communicator)              */
/* The same role for a caller that brings its own transport (TCP, MPI, shared memory;
 * ranks that share a GPU or have no xGMI between them): three collectives on HOST
 * buffers -- in-place all-reduce(max) of n f64 / n i32, all-gather of nbytes per rank
 * into rank order -- returning 0 on success.  The N-rank entry points
 * (sgp_grid_sets_front_comm, sgp_grid_sets_fused_comm) then stage their device operands
 * through the host around these calls (stream sync, D2H, callback, H2D) at exactly the
 * points where they enqueue RCCL collectives otherwise; sgp_comm_allreduce_max /
 * _allgather / _barrier call them directly.  A context has one transport.           */
typedef int (*sgp_host_allreduce_max_f64)(void* user, double* buf, int n);
typedef int (*sgp_host_allreduce_max_i32)(void* user, int32_t* buf, int n);
typedef int (*sgp_host_allgather)(void* user, const void* send, void* recv, int64_t nbytes);
int sgp_comm_init_host(sgp_ctx* ctx, int rank, int world,
                       sgp_host_allreduce_max_f64 allreduce_f64,
                       sgp_host_allreduce_max_i32 allreduce_i32,
                       sgp_host_allgather allgather, void* user);
int sgp_comm_allreduce_max(sgp_ctx* ctx, double* buf, int n);   /* in place   */
int sgp_comm_allgather(sgp_ctx* ctx, const void* send, void* recv,
                       int64_t nbytes);           /* recv = world*nbytes      */
int sgp_comm_barrier(sgp_ctx* ctx);

/* ---- measurement -----------------------------------------------------------*/
/* hipEvent pair on the ctx stream around a region of calls                   */
int sgp_timer_start(sgp_ctx* ctx);
int sgp_timer_stop(sgp_ctx* ctx, float* ms);
/* per-launch hipEvent timing of the posterior sweep kernel (the dominant
 * kernel): enable, run, then read {total ms, launches, algorithmic flops}    */
int sgp_profile_enable(sgp_ctx* ctx, int on);
int sgp_profile_read(sgp_ctx* ctx, double* total_ms, int64_t* launches,
                     double* flops);
/* device allocations (hipMalloc) this context has made so far: buffers grow on
 * demand, and a growth is an allocation plus a stream sync -- a steady-state
 * loop (same n, same grid) must leave this number unchanged                  */
int64_t sgp_ctx_alloc_count(sgp_ctx* ctx);

/* Which kernel runs gp.predict_noiseless over a grid / a swarm
 * (safeopt/gp_opt.py:469, 929, 973).  0 = automatic (the paired-wave kernel once
 * a GP has more than 256 training rows, csrc/sweep_pair.hip; the VALU kernel up to
 * 48, csrc/sweep_tiny.hip; the resident-factor kernel for 49 .. 128 observations of
 * single-part kernels up to d = 4, csrc/sweep_mid.hip; the 4-wave kernel otherwise),
 * 1 = always the 4-wave kernel (csrc/sweep.hip), 2 = always the paired-wave kernel,
 * 3 = automatic without the VALU kernel and the one-launch step; + 4: the
 * paired-wave kernel does not cut remainder tiles into runs of chunks (same bits
 * either way, tests/test_gpu_posterior.py); + 8: no factor tables on tensor grids
 * (sgp_grid_set_axes); + 16: the 4-wave kernel streams small factors through its
 * double buffer instead of keeping them in LDS for the launch (same bits either
 * way); + 32: the paired-wave kernel runs one 16-point block of training points per
 * stage instead of merging the thin stages of a triangular chunk (the schedule until
 * round 5: another summation order, same results within rounding).  Same results
 * within rounding between the two kernels; the switch exists for A/B measurements and tests.  Returns
 * the previous setting.                                                        */
int sgp_ctx_set_sweep(sgp_ctx* ctx, int which);
/* Which kernel ran the LAST posterior sweep of this context (tests of the selection,
 * A/B scripts): 0 = none yet, 1 = the 4-wave kernel (n <= 256, csrc/sweep.hip),
 * 2 = the paired-wave kernel (csrc/sweep_pair.hip), 3 = the VALU kernel for GPs with
 * at most 48 observations (csrc/sweep_tiny.hip -- the regime of the reference's own
 * examples), 4 = the few-points path (csrc/factor.hip), 5 = the one-launch step of a
 * small grid (csrc/step_small.hip), 6 = the resident-factor kernel for 49 .. 128
 * observations (csrc/sweep_mid.hip).                                                */
int sgp_ctx_last_sweep(sgp_ctx* ctx);

/* Multi-output case: consecutive GPs of a launch with bit-identical training
 * inputs, kernel, noise and fitting history have the same L^-1, so the variance
 * contraction of gp.predict_noiseless (safeopt/gp_opt.py:466-476 loops over the
 * GPs independently) is computed once and only alpha . k per GP: up to two such
 * followers per leader take their alpha . k from the covariances the leader's
 * sweep evaluates anyway (both sweep kernels; single-part kernels, d <= 4 / 3),
 * further ones keep covariance-only stages in the paired-wave kernel.  Identical
 * results bit for bit (tests/test_gpu_posterior.py).  on = 1 (default) / 0; returns
 * the previous setting.                                                         */
int sgp_ctx_set_share(sgp_ctx* ctx, int on);

#ifdef __cplusplus
}
#endif
#endif /* SAFEOPT_HIP_H */
