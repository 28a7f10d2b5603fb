// Internal: the context behind the C-ABI (include/vilma_hip.h), shared by capi.hip (LD store,
// state, evaluation entry points) and sweep.hip (the sweep state machine).  Not installed.
#pragma once
#include "../../include/vilma_hip.h"
#include "kernels.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

namespace vilma_detail {

struct BlockRec {
    int form;            // 0 dense, 1 eigen
    int n, r;
    int64_t off_a;       // element offset of R or U in the cohort store
    int64_t off_v;       // element offset of the eigenvalues s (eigen form)
    int32_t start;       // LD position of the block's first SNP
    int32_t t_off;       // offset in the cohort's t scratch (eigen form)
    int w;               // eigen form: rows per thread of the fused product (U stored column-major,
                         // stride pad2(n)), or 0 = row-major U [n x ld(r)] for the two-pass
                         // kernels (block too tall for the fused one)
};

struct CohortLd {
    bool begun = false, ended = false;
    int n_blocks = 0;
    int64_t n_ld = 0;
    double *store = nullptr;
    int64_t store_elems = 0, store_used = 0;
    int32_t next_start = 0;
    int64_t t_used = 0;
    std::vector<BlockRec> blocks;
    int64_t alg_bytes = 0;
    int64_t s_used = 0;                            // scratch entries of the LD product (set by ensure_ready)
    int64_t s_eig_used = 0;                        // ... of which for the eigen-form blocks (known at add time)
};

// device-resident work lists of one LD product (all cohorts, or one cohort)
// Eigen-form blocks are processed in groups of <= eigen_group_bytes of U (default: one group):
// first pass of the group (t' = s * U^T x), then the second pass on the SAME U (y = U t'), then
// the combine -- three launches per group.
struct EigenGroup { int a0, na, r0, nr, c0, nc; };
struct ItemSet {
    SymItem *sym = nullptr;
    SymCombItem *comb = nullptr;
    SymTile *tile = nullptr;         // tiled symmetric product (vilma_ctx::tile_rows > 0)
    TileCombItem *tcomb = nullptr;
    int n_tile = 0, n_tcomb = 0;
    LdItem *a = nullptr;
    RowItem *row = nullptr;
    RowCombItem *rcomb = nullptr;
    int n_sym = 0, n_comb = 0, n_a = 0, n_row = 0, n_rcomb = 0;
    std::vector<EigenGroup> groups;
    // fused eigen-form product: one list per block-height class (rows per thread 2, 4, 8, 12) and
    // the combine items of all fused blocks
    EigItem *eig[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};     // by block-height class
    int n_eig[5] = {0, 0, 0, 0, 0};
    EigItem *eig_all = nullptr;      // the same items in one list (largest first), for the
    int n_eig_all = 0;               // single-launch variant small shards use
    RowCombItem *fcomb = nullptr;
    int n_fcomb = 0;
};
inline int eig_class(int R) { return R == 2 ? 0 : R == 4 ? 1 : R == 8 ? 2 : R == 12 ? 3 : 4; }
inline int eig_class_rows(int k) { return k == 0 ? 2 : k == 1 ? 4 : k == 2 ? 8 : k == 3 ? 12 : 24; }
inline int pad2(int n) { return (n + 1) & ~1; }
// Columns of U one workgroup of the fused product takes ("slab"): the block's columns are cut into
// the fewest slabs of at most g_eig_slab_elems elements of U, equal to within a batch
// (VILMA_EIG_SLAB_ELEMS, read when a context is created, keeps the sweep reproducible).  Every slab
// re-reads the block's x, leaves a partial y of the block's height and ends in a store phase; a
// block of ONE slab writes y and its y.z partial itself (no scratch, no combine item).  History:
// 48 k elements (r02, guarded loads) -> 96 k with the single launch for all block heights (r03:
// C4 product 0.66 - 0.69 -> 0.59 - 0.63 ms, profiles/r03u_eigen_slab_size.txt) -> 384 k and no cap on
// the columns (r05: 0.603 - 0.643 -> 0.578 - 0.588 ms with slabs of <= 512 columns before the direct
// path, profiles/r05n_eigen_slabs.txt): as for the dense kernel, what the store phases cost falls
// with their number.
inline int g_eig_slab_elems = 393216;
inline int eig_n_slabs(int n, int r, int R) {
    // blocks of up to 512 rows (R = 2) are taken one slab per WAVE by the per-class launch: half the budget
    const int64_t budget = R == 2 ? g_eig_slab_elems / 2 : g_eig_slab_elems;
    return (int)std::max<int64_t>(1, ((int64_t)n * r + budget - 1) / budget);
}
inline int eig_slab_cols(int n, int r, int R) {
    const int C = eig_batch_cols(R), ns = eig_n_slabs(n, r, R);
    return std::max(C, ((r + ns - 1) / ns + C - 1) / C * C);
}

// number of doubles a dense block occupies: per 128-column slab J the panel of rows >= 128 J
inline int64_t sym_panel_elems(int n, int J);


// leading dimensions are multiples of 16 doubles (128 B): each 128-column slab of a row then
// starts on a cache-line boundary and no line is shared between two workgroups' slabs
inline int pad_ld(int n) { return (n + 15) & ~15; }
inline int slab_width(int n, int J) { return std::min(128, n - 128 * J); }
inline int64_t sym_panel_elems(int n, int J) {
    return (int64_t)(n - 128 * J) * pad_ld(slab_width(n, J));
}
inline int n_slabs(int n) { return (n + 127) / 128; }
// scratch of the symmetric product for one block: row sums S[slab][n] and the column-sum chunks
// C[slab][chunk][128] (laid out for ceil(n / chunk_rows) chunks per slab)
// (rows of S are pad2(n) apart: every partial-sum record then starts 16-byte aligned and goes out
// in 16-byte write-through stores)
inline int32_t sym_scratch_elems(int n, int chunk_rows) {
    return n_slabs(n) * (pad2(n) + ((n + chunk_rows - 1) / chunk_rows) * 128);
}
// scratch of the TILED symmetric product for one block (kernels.hip, ld_tile_kernel): S[slot][pad2(n)],
// entry j uses slots 0 .. j / cw + G - j / tr - 1 (cw = 128 tile_slabs, G = ceil(n / tr))
inline int tile_slots(int n, int tr, int cw) {
    const int G = (n + tr - 1) / tr;
    int most = 0;
    for (int j = 0; j < n; j += 128) most = std::max(most, j / cw + G - j / tr);
    return most;
}
inline int32_t tile_scratch_elems(int n, int tr, int cw) { return tile_slots(n, tr, cw) * pad2(n); }

}  // namespace vilma_detail
using namespace vilma_detail;

struct SweepState;      // sweep.hip: the sweep state machine's host-side bookkeeping

struct vilma_ctx {
    int P = 0, M = 0, A = 0, device = 0;
    SweepState *sw = nullptr;
    SweepCtl *ctl = nullptr;        // device: the control block of sweeps queued ahead (sweep.hip)
    int64_t N = 0;
    std::string err;

    double *adj = nullptr, *se = nullptr, *sld = nullptr, *scal = nullptr;
    int32_t *annot = nullptr, *invperm = nullptr;
    double *prec = nullptr, *log_det = nullptr, *lh = nullptr, *counts = nullptr;
    std::vector<double> log_det_host;
    double tau[VILMA_MAX_P];

    // Three buffers of each kind, in the roles current / candidate A / candidate B: a beta trial
    // may evaluate two step sizes at once (vilma_trial_beta2); plain evaluations and one-step
    // trials use the A role.  Accepting swaps roles, never copies.
    double *mu[3] = {nullptr, nullptr, nullptr};
    int mu_cur = 0, mu_ta = 1, mu_tb = 2;
    double *pool[3] = {nullptr, nullptr, nullptr};
    double *m[3] = {nullptr, nullptr, nullptr}, *v[3] = {nullptr, nullptr, nullptr},
           *lse[3] = {nullptr, nullptr, nullptr};
    int mom_cur = 0, mom_ta = 1, mom_tb = 2;
    bool have_b = false;            // candidate B holds the second step of the last trial
    int64_t pool_elems = 0;
    bool have_moments = false;
    bool trial_tainted = false;     // trial moments come from vilma_eval_given_delta

    // per-tile responsibility sums a stashing beta trial leaves behind ([candidate][tile][A*M] +
    // reduction scratch; nullptr when M is too large for the stash) and how many candidates of the
    // LAST trial have them (0: none -- the sums come from delta_kernel)
    double *sum_partials = nullptr;
    int tile_sums_ns = 0;
    // real_posterior_mean of the last completed sweep (convergence statistics).  Two buffers: the
    // host-decided path works in place on snap[snap_cur]; a queued sweep's evaluations read
    // snap[snap_cur] and write the other one, and the decision that ends the sweep flips snap_cur
    // (an error-scaling re-evaluation in the same sweep must compare with the same old means)
    double *snap[2] = {nullptr, nullptr};
    int snap_cur = 0;
    double *cvec[3] = {nullptr, nullptr, nullptr};  // [P][N] beside each set of moments: the vector c of a
                                    // state lazy trials reached (PhasePtrs, kernels.h)
    bool poison = false;            // VILMA_DEBUG_POISON=1: NaN into what a trial is about to write
    // vi_mu of role `current` equals Sig_k cvec[pure_c] for every component (what vilma_init_state leaves,
    // at the tau of pure_tau): a queued sweep may start from the lazy form a = 0; -1: not known to
    int pure_c = -1;
    double pure_tau[VILMA_MAX_P] = {};
    bool lazy_trial = false;        // the trials being queued store no vi_mu (set by sweep.hip)
    bool lazy_persist = false;      // ... and nothing else does: the evaluations being queued derive their state too
    bool lazy_stash = false;        // ... and the lazy trials being queued keep the on-chip stash (the mixture fits it)
    bool lazy_nobase = false;       // ... and their state has a == 0 for as long as they can run (armed base-free, no --learn-scaling)
    double *snp_partials = nullptr, *dot_partials = nullptr;
    double *delta_partials = nullptr, *diff_partials = nullptr;
    std::vector<int32_t> dot_start;     // first y.z partial slot of each cohort (+ end)

    std::vector<CohortLd> ld;
    ItemSet all;
    std::vector<ItemSet> solo;      // per cohort (vilma_ld_matvec with cohort >= 0)
    double *sym_scratch = nullptr;
    int dot_stride = 0;             // second right-hand side's y.z partials / scratch sit this far
    int64_t s_stride = 0;           // behind the first one's
    double *pinned = nullptr;       // host staging for vilma_fetch
    int64_t pinned_elems = 0;
    bool ready = false;
    // rows per work item of the symmetric product (multiple of 32); VILMA_LD_CHUNK_ROWS overrides
    int chunk_rows = 512;
    // tiled symmetric product: row strips of tile_rows rows (a multiple of 128 tile_slabs, at most 512),
    // column strips of tile_slabs slabs (at most 4); tile_rows = 0: one workgroup per slab chunk
    // (ld_sym_kernel).  VILMA_LD_TILE=rows,slabs overrides.
    int tile_rows = 512, tile_slabs = 4;
    bool tile_auto = true;          // pick tile_slabs from the shard's size when the work lists are built
    int n_cu = 256;                 // compute units of the device
    // order of the symmetric product's work items (capi.hip: sort_items); VILMA_LD_ORDER /
    // vilma_prof_ld_order override
    int ld_order = 0;
    // U bytes per eigen-form group.  Default: ONE group (first pass over every block, then the
    // second).  Running both passes group by group so that the second finds U in the 256 MB
    // Infinity Cache was measured and LOSES: 1.16 ms per product ungrouped, 1.50 / 1.76 / 2.32 ms
    // with 320 / 160 / 96 MB groups at C4 (profiles/r02g_ab_eigen.txt) -- ~10 us of launch ramp
    // and tail per extra kernel, no measurable gain from the cache.  VILMA_EIGEN_GROUP_MB keeps
    // the experiment reproducible.
    int64_t eigen_group_bytes = INT64_MAX;
    // eigen-form blocks go through the fused product (panel-major U, read once) unless
    // VILMA_EIG_FUSED=0 or the block is too tall for the LDS of a CU
    bool eig_fused = true;
    // blocks of up to 512 rows: one wave per slab of columns, no barrier (VILMA_EIG_WAVE=0: the
    // workgroup-wide kernel for them too)
    bool eig_wave = true;
    // fewer fused work items than this: one launch for all block heights (VILMA_EIG_MERGE_BELOW)
    int eig_merge_below = 8192;
    double *repack_tmp = nullptr;   // row-major staging of one block's U before the panel repack
    int64_t repack_elems = 0;

    // Work that depends only on the per-SNP pass of an evaluation (responsibility sums of the
    // trial state, convergence statistics) runs on `side`, concurrently with that evaluation's LD
    // product on the caller's stream: gated on ev_snp, joined back through ev_side.
    hipStream_t side = nullptr;
    hipEvent_t ev_snp = nullptr, ev_side = nullptr;
    bool overlap = true, snp_marked = false;

    int prof = 0;                   // 0 off, k >= 1: bracket every k-th LD launch
    int64_t prof_tick = 0;
    bool prof_now = false;
    struct Pending { hipEvent_t e0, e1; int kind; int64_t tag; };
    int64_t prof_tag = 0;           // stamped on the brackets recorded while it is set (sweep.hip drops
                                    // the brackets of queued launches that turned out empty)
    std::vector<Pending> pending;
    std::vector<hipEvent_t> event_pool;
    double prof_ms[VILMA_PROF_KINDS] = {};
    int64_t prof_launches[VILMA_PROF_KINDS] = {};
};

namespace vilma_detail {

extern std::string g_create_error;
inline int fail(vilma_ctx *c, const std::string &msg) {
    if (c) c->err = msg; else g_create_error = msg;
    return 1;
}

#define HIPCHK(c, call)                                                                   \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess)                                                             \
            return fail((c), std::string(#call) + ": " + hipGetErrorString(e_));          \
    } while (0)

template <typename T>
int dev_alloc(vilma_ctx *c, T **p, int64_t count, bool zero = true) {
    if (count <= 0) count = 1;
    HIPCHK(c, hipMalloc((void **)p, (size_t)count * sizeof(T)));
    if (zero) HIPCHK(c, hipMemset(*p, 0, (size_t)count * sizeof(T)));
    return 0;
}

inline void dev_free(void *p) { if (p) (void)hipFree(p); }

// sweep.hip
void sweep_destroy(vilma_ctx *c);
// capi.hip, for sweep.hip: one phase of a sweep queued ahead of the decision that assigns the
// buffers their roles (the kernels read them from c->ctl->phase[] when they start); prof_*: the
// HIP-event brackets recorded since a mark (dropped when the work turned out dead)
int queue_trial_phase(vilma_ctx *c, hipStream_t s, bool two, double *totals_a, double *totals_b,
                      double *sums_a, double *sums_b);
int queue_eval_phase(vilma_ctx *c, hipStream_t s, double *totals, double *dsum, double *dmax);
// responsibility sums of the state the queued EVAL phase starts from (the candidate the decision
// accepted), for mixtures too large for the trial pass's on-chip stash; and the M-step from them
int queue_sums_phase(vilma_ctx *c, hipStream_t s, double *sums_dev, bool writes_state = false);
int queue_mstep(vilma_ctx *c, hipStream_t s, const double *sums_dev, double *hyper_dev);
// a * mu[mu_from] + Sig cvec[c_buf] -> mu[mu_to] (the state lazy trials reached, written out by the host;
// lse[lse_buf]: its log-normaliser, for the responsibility sums that come with it)
int materialise_deferred(vilma_ctx *c, hipStream_t s, int mu_from, int mu_to, int c_buf, int lse_buf,
                         double a_def, const double *tau, double *sums_dev);
// HIP-event brackets recorded with vilma_ctx::prof_tag == tag / >= tag are forgotten (launches of
// a queued phase that did not happen exit at once: their microseconds are not kernel times)
void prof_drop_tag(vilma_ctx *c, int64_t tag);
void prof_drop_tags(vilma_ctx *c, int64_t from_tag);

}  // namespace vilma_detail
