/*
 * vilma_hip.h -- C-ABI of libvilma_hip.so: the MI355X (gfx950) implementation of the
 * `vilma fit` variational-inference hot path.
 *
 * The reference (jeffspence/vilma v0.0.16) is pure Python + numba and has NO FFI; the
 * boundary it exposes is the Python class API MultiPopVI / BlockDiagonalMatrix.  This header
 * is the device-side contract *below* that class API: each entry point names the reference
 * function(s) it replaces (paths relative to /root/reference/src/vilma/).  The Python host
 * (vilma_amd/) binds it with ctypes -- see INTEGRATION.md for the binding a reference
 * maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success, nonzero on failure; vilma_last_error() gives text.
 *   - all floating point is IEEE double; index arrays are int32/int64 as declared.
 *   - "host or device" pointers may be either (copies use hipMemcpyDefault).
 *   - `stream` is a hipStream_t passed as void*; NULL = the default stream.  Calls that take a
 *     stream are asynchronous on it; calls without one synchronise internally.  Work that depends
 *     only on the per-SNP pass of the latest evaluation (vilma_delta_sums of a trial state,
 *     vilma_mean_diff) may run on a context-owned side stream, concurrently with that
 *     evaluation's LD product; `stream` is made to wait for it before the call returns, so
 *     from the caller's side everything is still ordered on `stream` (VILMA_OVERLAP=0 in the
 *     environment disables the side stream).
 *   - one context drives ONE GPU (the device current at vilma_create).  Multi-GPU = one
 *     process and one context per GPU, each holding a shard of the SNPs; the host all-reduces
 *     the small `totals` vectors (RCCL via torch.distributed).
 *   - SNP-indexed arrays are in the caller's SNP order ("extract order"); LD blocks are given
 *     in LD order together with perm (LD position -> SNP index), exactly the reference's
 *     BlockDiagonalMatrix.perm (matrix_structures.py:246-257).
 */
#ifndef VILMA_HIP_H
#define VILMA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vilma_ctx vilma_ctx;

/* number of doubles in a `totals` vector for P cohorts:
 *   [0,P)    lin_p  = sum_i m_pi * adj_pi
 *   [P,2P)   var_p  = sum_i (ld_diag_pi / se_pi^2) * v_pi
 *   [2P,3P)  quad_p = sum_i (R_p z_p)_i * z_pi,  z = m / se
 *   3P       kl_mix = delta_kl + beta_kl  (numerics.py:132-146; summed because the
 *            log-determinant terms of the two cancel and are never computed on the device)
 *   3P+1     ip_comp  (numerics.py:98-115)
 * These are the per-shard sums from which fast_likelihood (numerics.py:31-46) and _beta_KL
 * (variational_inference.py:873-885) are assembled on the host after the all-reduce. */
#define VILMA_NTOTALS(P) (3 * (P) + 2)


const char *vilma_version(void);

/* Error text of the last failing call on `ctx` (or of vilma_create when ctx == NULL). */
const char *vilma_last_error(const vilma_ctx *ctx);

/* Create a context for a shard with P cohorts, N SNPs, M mixture components, A annotations
 * on the current HIP device.  Replaces the array allocations of VIScheme.__init__ /
 * MultiPopVI.__init__ (variational_inference.py:96-259, 599-630). */
int vilma_create(int P, int64_t N, int M, int A, vilma_ctx **out);
void vilma_destroy(vilma_ctx *ctx);

/* ---- static per-SNP data -------------------------------------------------------------- */

/* adj [P*N]   adj_marginal_effects           (variational_inference.py:226-243)
 * se  [P*N]   std_errs (ones when --scaled)  (variational_inference.py:205-214)
 * sld [P*N]   scaled_ld_diags = ld_diags / se^2   (variational_inference.py:215)
 * scalings [P*N]  (variational_inference.py:211-214)
 * annot [N]   annotation index per SNP       (variational_inference.py:217) */
int vilma_set_snp_data(vilma_ctx *ctx, const double *adj, const double *se, const double *sld,
                       const double *scalings, const int32_t *annot);

/* prec [M*P*P] mixture_prec, log_det [M]     (variational_inference.py:621-626) */
int vilma_set_mixture(vilma_ctx *ctx, const double *prec, const double *log_det);

/* error_scaling tau [P]; on device vi_sigma, nat_sigma, vi_sigma_log_det, vi_sigma_matches and
 * sigma_summary are functions of (prec, sld, tau) recomputed inside the kernels, so this call
 * is the whole of _set_vi_sigma (variational_inference.py:712-733). */
int vilma_set_tau(vilma_ctx *ctx, const double *tau);

/* hyper_delta [A*M]; defines nat_grad_vi_delta = fast_vi_delta_grad(hyper, log_det, annot)
 * (numerics.py:149-164, variational_inference.py:844-848) as an [A,M] table on device. */
int vilma_set_hyper(vilma_ctx *ctx, const double *hyper);

/* annotation_counts [A] = number of SNPs (of the WHOLE problem, not the shard) per annotation
 * (variational_inference.py:218); used by vilma_mstep. */
int vilma_set_annotation_counts(vilma_ctx *ctx, const double *counts);

/* ---- LD operator: BlockDiagonalMatrix (matrix_structures.py:237-447) -------------------- */

/* Start cohort `cohort`.  perm [N] int64: perm[t] = SNP index at LD position t; the first
 * n_ld positions are covered by blocks (in the order they are added), the remaining N-n_ld are
 * the `missing` SNPs (zero rows/columns).  total_elems = sum over blocks of the element count
 * each add call will store (see below) -- lets the library make one allocation. */
int vilma_ld_begin(vilma_ctx *ctx, int cohort, int n_blocks, int64_t n_ld, const int64_t *perm,
                   int64_t total_elems);

/* Element counts to use in total_elems (rows are padded to a multiple of 16 doubles = 128 B so
 * that every 1-KiB wave load covers whole cache lines; dense blocks keep the lower triangle; an
 * eigen-form block keeps U once -- column-major with the column length rounded up to even for the
 * fused product, row-major for blocks of more than 6 144 SNPs -- and s:
 * pad2(n) * pad16(r) + pad16(r)). */
int64_t vilma_ld_dense_elems(int n);
int64_t vilma_ld_lowrank_elems(int n, int r);

/* Add the next block as a dense SYMMETRIC n x n matrix R (row-major, host or device).  Only its
 * lower triangle is stored (by 128-column slabs) and read -- once -- per product.  For the
 * reference's blocks R must be the reconstruction U diag(s) U^T of the kept eigenpairs
 * (LowRankMatrix, matrix_structures.py:95-152), NOT the raw .npy matrix. */
int vilma_ld_add_dense(vilma_ctx *ctx, int cohort, int n, const double *R);

/* Add the next block in eigen form: U [n*r] row-major (host or device), s [r];
 * dot = U (s * (U^T x)) (LowRankMatrix.dot, matrix_structures.py:148-152).  Only U and s are
 * stored.  Blocks of up to 6 144 SNPs keep U column-major and a product reads it ONCE (the fused
 * kernel holds a slab of columns in registers between the two uses); taller blocks keep it
 * row-major and read it twice (column sums, then row sums).  The call repacks; the caller's
 * layout is always row-major. */
int vilma_ld_add_lowrank(vilma_ctx *ctx, int cohort, int n, int r, const double *U,
                         const double *s);

int vilma_ld_end(vilma_ctx *ctx, int cohort);

/* y = R_cohort . x for cohort in [0,P) or all cohorts when cohort < 0: BlockDiagonalMatrix.dot
 * (matrix_structures.py:389-408) including the perm gather, inv_perm scatter and zeros at
 * missing.  x, y: device pointers [P*N] in SNP order (row p = cohort p). */
int vilma_ld_matvec(vilma_ctx *ctx, void *stream, int cohort, const double *x, double *y);
/* The same product for TWO right-hand sides in one pass over the LD store (every element is loaded
 * once for both): what the product behind a two-step beta trial does; each result is bit-identical
 * to vilma_ld_matvec's.  BlockDiagonalMatrix.dot with a two-column right-hand side
 * (matrix_structures.py:389-408). */
int vilma_ld_matvec2(vilma_ctx *ctx, void *stream, int cohort, const double *x0, const double *x1,
                     double *y0, double *y1);

/* Algorithmic bytes one vilma_eval/vilma_trial_beta streams from the LD store (all cohorts):
 * 8 * sum_b n_b^2 (dense) or 8 * sum_b n_b r_b (eigen form, U counted once) -- SURVEY.md 8(d).
 * stored_bytes = bytes actually resident (dense symmetric blocks keep ~n^2/2 + 64 n elements;
 * the eigen form stores U and s, and reads U twice per product). */
int vilma_ld_bytes(const vilma_ctx *ctx, int64_t *algorithmic_bytes, int64_t *stored_bytes);

/* ---- variational state ---------------------------------------------------------------- */

/* vi_mu [M*P*N] (reference layout [M][P][N]).  vi_delta is never stored: it is the function
 * of (vi_mu, hyper, tau) given by _nat_to_not_vi_delta (variational_inference.py:632-641). */
int vilma_set_mu(vilma_ctx *ctx, const double *vi_mu);
int vilma_get_mu(vilma_ctx *ctx, double *vi_mu);
/* vi_delta [N*M] (reference layout) of the current state. */
int vilma_get_delta(vilma_ctx *ctx, double *vi_delta);
/* vi_sigma [M*P*P*N] (reference layout [M][P][P][N]) at `error_scaling` [P] (NULL: the context's): Sig_ki =
 * (mixture_prec_k + diag(scaled_ld_diags_i / tau))^-1, what _set_vi_sigma keeps as an array
 * (variational_inference.py:712-724; numerics.py:216-290) and `vilma fit` writes into its .npz.  The
 * library never stores it; this forms it on the device for the output (closed forms for one and two
 * cohorts in plain IEEE operations, Cholesky beyond). */
int vilma_get_vi_sigma(vilma_ctx *ctx, const double *error_scaling, double *vi_sigma);
/* posterior mean / marginal variance [P*N] of the current state, without scalings
 * (_posterior_mean, _posterior_marginal_variance, variational_inference.py:753-760). */
int vilma_get_moments(vilma_ctx *ctx, double *mean, double *var);

/* ---- objective evaluations (asynchronous on `stream`) ---------------------------------- */

/* Evaluate the ELBO pieces at the CURRENT vi_mu with the current hyper/tau: one fused per-SNP
 * pass (fast_posterior_mean, fast_pmv, fast_invert_nat_vi_delta, fast_delta_kl,
 * fast_inner_product_comp, fast_beta_kl: numerics.py:49-65, 98-146, 179-213) + one LD matvec
 * per cohort + the fast_likelihood sums.  Results become the *trial* moments; writes
 * VILMA_NTOTALS(P) doubles to totals_dev (device pointer).  Replaces elbo()/_log_likelihood/
 * _beta_KL (variational_inference.py:412-417, 452-470, 873-885). */
int vilma_eval(vilma_ctx *ctx, void *stream, double *totals_dev);

/* vilma_eval with the convergence statistics of vilma_mean_diff fused into its per-SNP pass: the
 * posterior means of the point being evaluated are compared with the snapshot and become the
 * snapshot.  For evaluations the caller accepts unconditionally (the one after the M-step,
 * variational_inference.py:849-858 followed by :374-382): saves a pass over [P*N] and a stream
 * hop per sweep.  The caller must vilma_accept(ctx, 0) this evaluation. */
int vilma_eval_diff(vilma_ctx *ctx, void *stream, double *totals_dev, double *out_sum3_dev,
                    double *out_max3_dev);

/* The same sums at (current vi_mu, a vi_delta SUPPLIED by the caller, current hyper/tau), for
 * callers of elbo(params) / real_posterior_mean(vi_mu, vi_delta, hyper_delta) that pass a vi_delta
 * which is not the fixed point of (vi_mu, hyper_delta, error_scaling) -- the reference evaluates
 * what it is handed (variational_inference.py:412-417, 740-760, 873-885; numerics.py:49-65,
 * 98-146).  delta_km_dev: device pointer, [M*N] component-major (vi_delta transposed).  Writes
 * VILMA_NTOTALS(P) + 1 doubles: the sums, then max_ik |vi_delta_ik - derived delta_ik|.  Needs an
 * accepted evaluation of the current state; the moments land in the trial buffers
 * (vilma_get_trial_moments) and that trial state cannot be accepted.  Off the sweep path. */
int vilma_eval_given_delta(vilma_ctx *ctx, void *stream, const double *delta_km_dev,
                           double *totals_dev);
int vilma_get_trial_moments(vilma_ctx *ctx, double *mean, double *var);

/* _initialize's per-SNP part (variational_inference.py:658-692) on the device: from fake_mu
 * [P*N] (host or device; the jittered ridge start, drawn by the host with the reference's RNG
 * call) compute the heuristic responsibilities, write vi_mu = Sigma_ki (avg Sigma_i)^-1 fake_mu
 * as the current state and the per-annotation responsibility sums [A*M] to sums_dev; the host
 * all-reduces them, forms hyper_delta (:667-674), calls vilma_set_hyper and vilma_eval. */
int vilma_init_state(vilma_ctx *ctx, void *stream, const double *fake_mu, double *sums_dev);

/* One line-search trial of _update_beta (variational_inference.py:777-787) from the current
 * state with step size `step` = 1/L: natural-gradient blend (sum_betas, numerics.py:11-15;
 * _nat_grad_beta, variational_inference.py:804-823), new_mu, new_vi_delta, and the objective
 * pieces of the candidate.  The candidate is held as the trial state. */
int vilma_trial_beta(vilma_ctx *ctx, void *stream, double step, double *totals_dev);

/* The same trial at TWO step sizes in one pass: step_a = 1/L, the step the line search tries now,
 * and step_b = 1/(L * line_search_rate), the one it would try next if step_a is rejected
 * (variational_inference.py:777-800).  vi_mu, the per-component matrices and -- the expensive
 * part -- the LD store are read once for both candidates (the LD kernel is HBM-bound; the second
 * right-hand side rides in the same loads), so a rejected first step no longer costs a second
 * pass over the LD store.  Each candidate's sums are bit-identical to vilma_trial_beta's at its
 * step.  Candidate A is held as the trial state exactly as after vilma_trial_beta(step_a);
 * candidate B is accepted with vilma_accept(ctx, 2).  Up to four cohorts (the per-SNP kernels of
 * five to eight evaluate one candidate per pass): an error beyond. */
int vilma_trial_beta2(vilma_ctx *ctx, void *stream, double step_a, double step_b,
                      double *totals_a_dev, double *totals_b_dev);

/* Make a trial state current.  take_mu = 1 after an accepted vilma_trial_beta / candidate A of
 * vilma_trial_beta2; 2 for candidate B; 0 after a vilma_eval (vi_mu unchanged, only the moments /
 * LD product move).  Buffers swap roles; nothing is copied. */
int vilma_accept(vilma_ctx *ctx, int take_mu);

/* sums [A*M] = sum_annotations(vi_delta) (numerics.py:118-129), the M-step statistic and the
 * payload of the cross-GPU all-reduce, of the state `which`: the current state, or the trial
 * state left by the last vilma_trial_beta / vilma_eval (so the statistic can be queued right
 * behind a line-search trial and fetched with its objective, saving a host round trip). */
#define VILMA_STATE_CURRENT 0
#define VILMA_STATE_TRIAL_BETA 1   /* trial of vilma_trial_beta (its own vi_mu) */
#define VILMA_STATE_TRIAL_EVAL 2   /* trial of vilma_eval (shares the current vi_mu) */
#define VILMA_STATE_TRIAL_BETA_B 3 /* candidate B of vilma_trial_beta2 */
int vilma_delta_sums(vilma_ctx *ctx, void *stream, double *sums_dev, int which);

/* The same statistic for the candidates of the LAST vilma_trial_beta(2) without reading their
 * vi_mu again: the trial's per-SNP pass keeps every softmax term of a tile of 64 SNPs on chip
 * until the tile's normaliser is known and leaves per-tile sums behind; this call only adds the
 * tiles up (fixed order).  sums_b_dev (may be NULL) = candidate B's.  Fails -- use
 * vilma_delta_sums -- when M is too large for the on-chip stash (LDS; M up to ~48 with two
 * candidates, ~100 with one) or VILMA_TILE_SUMS=0. */
int vilma_trial_sums(vilma_ctx *ctx, void *stream, double *sums_a_dev, double *sums_b_dev);
/* 0, 1 or 2: how many candidates of the last trial vilma_trial_sums can serve. */
int vilma_trial_sums_available(const vilma_ctx *ctx);

/* The M-step of _update_hyper_delta on the device, without a host round trip: from the
 * (all-reduced) sums_dev [A*M] of vilma_delta_sums compute
 * hyper = normalise(max(sums / (annotation_counts + 1e-100), 1e-100)) (variational_inference.py:
 * 837-842), install it exactly as vilma_set_hyper would, and write it to hyper_dev [A*M]. */
int vilma_mstep(vilma_ctx *ctx, void *stream, const double *sums_dev, double *hyper_dev);

/* Convergence statistics of real_posterior_mean (variational_inference.py:374-382, 292-314)
 * between the current state and the snapshot taken by the previous call (or by
 * vilma_snapshot_mean): out_sum3_dev = {#entries violating |new-old| <= 1e-6 + 1e-6|old|,
 * sum |new-old|, sum (new-old)^2} (additive over shards), out_max3_dev = {max |new|,
 * max |new-old|, max |(new-old)/(old+1e-100)|}.  Then the snapshot becomes the current mean. */
int vilma_mean_diff(vilma_ctx *ctx, void *stream, double *out_sum3_dev, double *out_max3_dev);
int vilma_snapshot_mean(vilma_ctx *ctx, void *stream);

/* Copy n doubles of a device result buffer to the host behind everything queued on `stream`
 * (pinned staging + stream synchronise): the one blocking point of a decision. */
int vilma_fetch(vilma_ctx *ctx, void *stream, const double *src_dev, double *dst_host, int64_t n);

/* ---- the sweep behind ONE call (vilma_sweep / vilma_elbo / vilma_posterior /
 *      vilma_set_state / vilma_get_state, communicator owned by the context) -------------------
 *
 * Everything above is the fine-grained surface (one evaluation, one trial, one decision).  The
 * entry points below run the reference's control flow INSIDE the library -- line search on L,
 * accept / reject, M-step, error-scaling update, running ELBO change -- so a host in any language
 * drives a fit with vilma_initialize (or vilma_set_state) + vilma_sweep in a loop.  The context
 * owns the small device-resident result vector all of them share; its layout, for P cohorts, A
 * annotations, M components (nt = VILMA_NTOTALS(P), am = A*M), all doubles:
 *   [0,3) convergence sums | [3,3+nt) totals of the current state | nt totals of trial candidate A |
 *   nt of candidate B | am responsibility sums of A | am of B | 3 convergence maxima | am hyper_delta
 * The part that is summed over ranks ([0, 3+3nt+2am)) is contiguous: one all-reduce per decision. */
int64_t vilma_results_size(const vilma_ctx *ctx);
double *vilma_results_dev(vilma_ctx *ctx);      /* device pointer, owned by the context */

/* Shard-independent constants of the objective: chi_stat [P], ld_ranks [P] of the WHOLE problem
 * (variational_inference.py:226-252) and whether error_scaling is learned (--learn-scaling,
 * :472-486).  Needed before vilma_set_state / vilma_initialize. */
int vilma_set_fit_constants(vilma_ctx *ctx, const double *chi_stat, const double *ld_ranks,
                            int scale_se);

/* Collective used for every cross-rank sum of the sweep (one all-reduce per decision).
 * Default: none (one rank).
 *   vilma_comm_init_rccl: ncclCommInitRank on this context's device; `id` = the 128 bytes of
 *     vilma_comm_unique_id(), produced on rank 0 and handed to every rank by the caller (any side
 *     channel: MPI, a file, torch.distributed's store).  The communicator is owned and destroyed by
 *     the context; all-reduces are queued on the sweep's stream (no host synchronisation).  RCCL is
 *     bound at run time (dlopen of the librccl already in the process, else the system one), so a
 *     single-GPU host needs no RCCL at all.
 *   vilma_comm_set_callback: the host supplies the all-reduce (rehearsals over gloo, MPI, ...):
 *     fn(user, stream, buf_dev, n, op) must reduce buf_dev[0..n) in place over all ranks, ordered
 *     after everything queued on `stream`, before it returns or in stream order; op 0 = sum, 1 = max. */
typedef int (*vilma_allreduce_fn)(void *user, void *stream, double *buf_dev, int64_t n, int op);
int vilma_comm_unique_id(char id[128]);
int vilma_comm_init_rccl(vilma_ctx *ctx, int world, int rank, const char id[128]);
int vilma_comm_set_callback(vilma_ctx *ctx, vilma_allreduce_fn fn, void *user, int world, int rank);
/* kind: 0 none, 1 RCCL, 2 callback */
int vilma_comm_info(const vilma_ctx *ctx, int *kind, int *world, int *rank);
/* In-place all-reduce of a device buffer through the context's collective (setup-time sums). */
int vilma_comm_allreduce(vilma_ctx *ctx, void *stream, double *buf_dev, int64_t n, int op);

/* _set_state (variational_inference.py:694-710) + the evaluation every caller follows it with:
 * make (vi_mu, hyper_delta, error_scaling) the current state and evaluate it; *objective = its
 * ELBO (elbo(), :412-417).  vi_mu [M*P*N] host or device, NULL = keep the vi_mu already on the
 * device (vilma_init_state); error_scaling NULL = keep.  vi_delta is implicit (the fixed point
 * _nat_to_not_vi_delta, :632-641). */
int vilma_set_state(vilma_ctx *ctx, void *stream, const double *vi_mu, const double *hyper_delta,
                    const double *error_scaling, double *objective);
/* Any pointer may be NULL.  vi_mu [M*P*N], vi_delta [N*M], hyper_delta [A*M], error_scaling [P]. */
int vilma_get_state(vilma_ctx *ctx, double *vi_mu, double *vi_delta, double *hyper_delta,
                    double *error_scaling);

/* _initialize (variational_inference.py:643-700) from the jittered ridge start fake_mu [P*N] (drawn
 * by the host with the reference's RNG call): per-SNP part on the device, responsibility sums
 * all-reduced, hyper_delta = normalise(sums + 1) clamped at 1e-100 (:667-674), then the first
 * evaluation.  *objective = ELBO of the starting point. */
int vilma_initialize(vilma_ctx *ctx, void *stream, const double *fake_mu, double *objective);

/* ELBO of the current state (elbo(), variational_inference.py:412-417); cached, no device work. */
int vilma_elbo(vilma_ctx *ctx, double *objective);

/* real_posterior_mean / real_posterior_variance of the current state (variational_inference.py:
 * 740-751): moments times scalings (squared for the variance).  [P*N] host pointers, may be NULL. */
int vilma_posterior(vilma_ctx *ctx, double *mean, double *var);

#define VILMA_MAX_COHORTS 8
#define VILMA_SWEEP_EVENTS 48
typedef struct {
    double elbo, running;            /* after the sweep (the call's *elbo / *running_delta) */
    double objective;                /* ELBO of the state the sweep ended in as evaluated on the
                                        device (vilma_elbo); `elbo` accumulates the sweep's changes
                                        on the caller's value like the reference does, so the two
                                        agree to rounding */
    double L[5];
    double diff_sum[3], diff_max[3]; /* convergence statistics of vilma_mean_diff for this sweep
                                        (filled with VILMA_SWEEP_DIFF) */
    double error_scaling[VILMA_MAX_COHORTS];
    int32_t n_evaluations;           /* candidate points whose objective was looked at */
    int32_t n_trials;                /* beta line-search trials among them */
    int32_t n_products;              /* passes over the LD store (a two-step trial is one) */
    int32_t ran_ahead;               /* this sweep's stage was decided on the device, ahead of the host */
    int32_t skipped_ahead;           /* stages queued ahead whose decision went the other way */
    /* the reference's per-update INFO lines (variational_inference.py:399, 424-449, 782):
     * kind 0 = paramset update {paramset, L}, 1 = objective pair {old, new}, 2 = error scaling */
    int32_t n_events;
    struct { int32_t kind, paramset; double a, b; } events[VILMA_SWEEP_EVENTS];
} vilma_sweep_stats;

#define VILMA_SWEEP_DIFF 1        /* fuse the convergence statistics into the sweep's last evaluation */
#define VILMA_SWEEP_LOOKAHEAD 2   /* the caller promises to call vilma_sweep again: the sweep loop --
                                     inner beta loop and its break rule, line-search retries, the
                                     M-step, with scale_se the error-scaling update and its
                                     re-evaluation -- is then decided by a kernel from a control
                                     block on the device, for every mixture size, and work beyond
                                     the state this call reports (at most ONE beta trial) may
                                     already be on `stream` when it returns.  Every decision is
                                     the one the same call without the flag takes (L to the bit,
                                     the same trials); values are its bits when the trials store
                                     their candidates (VILMA_STASH_LAZY=0 / VILMA_PIPE_LAZY=0) and
                                     equal to rounding otherwise -- up to four cohorts the queued
                                     sweeps keep the state as (stored vi_mu, a, c), mu_k = a mu_k
                                     + Sig_k c, store no vi_mu array and write the state out when
                                     somebody reads it (vilma_sweep_drain, any state read, the
                                     last sweep of a run).  The host replays every
                                     decision with the same source and returns an error if the
                                     two differ.  Ignored with VILMA_SWEEP_VERBOSE (per-update
                                     events need the host in the loop) and with VILMA_LOOKAHEAD=0
                                     in the environment.  A later call on a DIFFERENT stream first
                                     waits for the queued work and restores the reported state
                                     (as vilma_sweep_drain does) */
#define VILMA_SWEEP_VETO 4        /* the sweep queued ahead by this call must not proceed if no
                                     posterior mean moved in THIS sweep (optimize() stops then,
                                     variational_inference.py:374-382): checked on the device */
#define VILMA_SWEEP_VETO_NEXT 8   /* reserved */
#define VILMA_SWEEP_VERBOSE 16    /* record events; all-reduce the convergence maxima as well */

/* One outer iteration: _optimize_step (variational_inference.py:396-410) = _nat_grad_step
 * (:419-450): beta line search (_update_beta, :762-802) until the inner loop's break rule, the
 * M-step (_update_hyper_delta, :825-860), with scale_se the error-scaling update (:472-486), then
 * the running ELBO change.  L [5] in/out; *elbo in = ELBO before the sweep, out = after;
 * *running_delta in/out, NaN = None (first sweep).  line_search_rate = 2 in optimize() (:361).
 * Returns nonzero with the reference's messages on a line-search failure ("Encountered a numerical
 * error.", :790-799).  With VILMA_SWEEP_DIFF, vilma_snapshot_mean must have been called once. */
int vilma_sweep(vilma_ctx *ctx, void *stream, double L[5], double *elbo, double *running_delta,
                double line_search_rate, int flags, vilma_sweep_stats *stats);

/* The three updates vilma_sweep is made of, one at a time, for hosts (and tests) that drive the
 * reference's private steps themselves.  Each starts from the current state (vilma_set_state /
 * the previous update) and leaves the updated state current.
 *   vilma_update_beta: _update_beta (variational_inference.py:762-802): ONE damped natural-gradient
 *     step with backtracking on *L0 (multiplied by line_search_rate per rejected trial);
 *     *orig_obj / *new_obj = objective before / after (equal when the search gave up beyond L_MAX).
 *   vilma_update_hyper_delta: _update_hyper_delta (:825-860), the closed-form M-step + re-evaluation.
 *   vilma_update_error_scaling: _update_error_scaling (:472-486) followed by the re-evaluation
 *     _nat_grad_step does behind it (:442-447); error_scaling through vilma_get_state. */
int vilma_update_beta(vilma_ctx *ctx, void *stream, double *L0, double line_search_rate,
                      double *orig_obj, double *new_obj);
int vilma_update_hyper_delta(vilma_ctx *ctx, void *stream, double *orig_obj, double *new_obj);
int vilma_update_error_scaling(vilma_ctx *ctx, void *stream, double *orig_obj, double *new_obj);

/* Forget work queued ahead by VILMA_SWEEP_LOOKAHEAD (the caller breaks its promise): waits for
 * it, restores the state after the last sweep reported.  A no-op otherwise. */
int vilma_sweep_drain(vilma_ctx *ctx);
/* The form in which the sweeps queued ahead held the state behind the last decision the host has
 * looked at: 0 = a stored vi_mu array; 1 = (stored vi_mu, a, c), mu_k = a mu_k + Sig_k c; 2 = the same
 * with a == 0 -- what a fit started by vilma_initialize is (the reference's _initialize builds
 * vi_mu_k = vi_sigma_k temp_nat_mu, variational_inference.py:683-690) and stays: no per-SNP pass
 * of such sweeps reads a vi_mu array, their memory traffic is [P][N] vectors (bench.py prices them
 * accordingly). */
int vilma_prof_state_form(vilma_ctx *ctx, int *form);

/* ---- measurement ----------------------------------------------------------------------- */

/* vilma_prof_enable(ctx, k): k = 0 off; k >= 1 brackets every k-th LD product's streaming kernels
 * with HIP events on their stream (a pair costs a few microseconds of stream time, so small
 * shards sample).  vilma_prof_read synchronises the device and returns, per kernel kind, the accumulated kernel
 * milliseconds and number of bracketed launches since the last reset (arrays of VILMA_PROF_KINDS). */
#define VILMA_PROF_LD_SYM 0      /* the symmetric dense product's kernel (ld_tile_kernel; ld_sym_kernel with
                                  * VILMA_LD_TILE=0): dense blocks, lower triangle read once */
#define VILMA_PROF_LD_EIG 1      /* ld_eig_fused_kernel: one product of the eigen-form blocks (the fused
                                  * launches incl. the 512-thread one for 3 073 .. 6 144 SNPs, the
                                  * two-pass kernels of taller blocks, combine) */
#define VILMA_PROF_LD_SYM2 2     /* the same kernel with two right-hand sides (vilma_trial_beta2): one pass
                                  * over the store, two products */
#define VILMA_PROF_SNP_EVAL 3    /* snp_pass_kernel of a plain evaluation */
#define VILMA_PROF_SNP_TRIAL 4   /* ... of a one-step beta trial */
#define VILMA_PROF_SNP_TRIAL2 5  /* ... of a two-step beta trial */
#define VILMA_PROF_SUMS 6        /* delta_kernel + its column reduction: the responsibility sums of the
                                  * accepted candidate, for mixtures beyond the on-chip stash */
#define VILMA_PROF_SUMS_MAT 7    /* the same pass behind a lazy trial: it also re-derives and stores the
                                  * accepted candidate's vi_mu (read + write) */
#define VILMA_PROF_SNP_TRIAL_LAZY 8   /* one-step beta trial that stores no candidate (lazy) */
#define VILMA_PROF_SNP_TRIAL2_LAZY 9  /* two-step beta trial that stores no candidate (lazy) */
#define VILMA_PROF_KINDS 10
int vilma_prof_enable(vilma_ctx *ctx, int enable);
int vilma_prof_read(vilma_ctx *ctx, double *ms_total, int64_t *launches, int reset);

/* The yardstick for the LD kernels on THIS placement of the store: `passes` bare reads of every
 * cohort's LD store (a read-only streaming kernel, nothing computed), timed with HIP events on
 * `stream`; *ms_per_pass = average time of one pass over all stores, *bytes_per_pass = the bytes
 * it read (the stored bytes, rounded down to 32 KB per cohort).  Where an allocation lands in HBM
 * moves the streaming rate of a box by several percent from process to process
 * (profiles/r03r_placement_probe.txt); this is what the same bytes stream at with no arithmetic
 * at all.  Measurement only: nothing in a fit calls it. */
int vilma_prof_stream_store(vilma_ctx *ctx, void *stream, int passes, double *ms_per_pass,
                            int64_t *bytes_per_pass);

/* Diagnostics behind profiles/ld_levels_probe.py (why ld_sym_kernel streams at 0.85 - 0.97 of a
 * bare read of the same store, depending on the process).  Measurement only.
 *   vilma_prof_stream_pattern: the bare read of vilma_prof_stream_store with the store cut into
 *     chunks of chunk_kb (a multiple of 32) KB, each read front to back by one workgroup of a grid
 *     of `grid` workgroups; scattered != 0 visits the chunks in a scattered order instead of store
 *     order -- what an access pattern alone costs on this placement of the store; writes != 0 adds
 *     a thin stream of stores (1/128 of the bytes read, like ld_sym_kernel's partial sums) into the
 *     product's scratch: 1 = 8 doubles per wave per 8 KiB read, 2 = the same non-temporal, 3 =
 *     whole 128-B lines, 4 = half the bytes, 5 = 4 KiB per workgroup per 512 KB read, 6 = all of
 *     it at the end of the workgroup's life, 7 = as 5 / 8 = as 1 but addressed by the order in
 *     time of the chunks instead of their place in the store.
 *   vilma_prof_ld_order: the order of ld_sym_kernel's work items from now on: 0 longest chunk first
 *     (default), 1 the order the panels lie in the store, 2 store order dealt out so that each XCD's
 *     workgroups (b, b + 8, ... under round-robin dispatch) walk one contiguous eighth of it.
 *     Results are bit-identical in every order.  VILMA_LD_ORDER in the environment sets the default.
 *   vilma_prof_ld_trace: builds with -DLD_TRACE=1 only (else an error): every workgroup of
 *     ld_sym_kernel writes {start, end (100 MHz ticks), XCC id, bytes of its chunk, core-clock
 *     cycles from start to end} to row blockIdx.x of buf_dev [capacity_rows][5]; NULL switches
 *     it off. */
int vilma_prof_stream_pattern(vilma_ctx *ctx, void *stream, int passes, int chunk_kb,
                              int scattered, int grid, int writes, double *ms_per_pass,
                              int64_t *bytes_per_pass);
int vilma_prof_ld_order(vilma_ctx *ctx, int order);
/* Shape of the dense product's work items once the LD store is complete: rows per row strip and
 * 128-column slabs per column strip of ld_tile_kernel (chosen from the shard's size unless
 * VILMA_LD_TILE=rows,slabs says otherwise; 0, 0 with VILMA_LD_TILE=0: one workgroup per slab chunk,
 * ld_sym_kernel), and the number of work items of one launch over all cohorts. */
int vilma_prof_ld_tile(vilma_ctx *ctx, int *tile_rows, int *tile_slabs, int *n_items);
int vilma_prof_ld_trace(vilma_ctx *ctx, double *buf_dev, int64_t capacity_rows);

/* Debug poison.  With VILMA_DEBUG_POISON=1 in the environment when vilma_create runs, every beta trial
 * (host-decided or queued ahead) first fills what it is about to produce with NaN: the result slots
 * of candidates A and B (totals and responsibility sums) and the vi_mu buffers its candidates are
 * stored into.  A slot a kernel leaves unwritten, or one a decision reads although no candidate
 * produced it, then gives a non-finite objective at once (the line search ends in the reference's
 * "Encountered a numerical error.") instead of whatever the memory held.  Results of a correct
 * library are unchanged to the bit.  vilma_debug_result_slot copies the context's result slot of
 * candidate A (which = 0) or B (1) of the LAST trial, n <= 3 P + 2 doubles, to `out` (host). */
int vilma_debug_result_slot(vilma_ctx *ctx, int which, double *out, int n);

#ifdef __cplusplus
}
#endif
#endif /* VILMA_HIP_H */
