// Internal launcher interface between capi.hip and kernels.hip (not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define VILMA_MAX_P 8

// One work item of the block-diagonal LD product: a 128-column slab of one row-major matrix
// (a dense symmetric block, U, or diag(s)U^T), all rows.  out[c] = sum_j a[j*ld + c] * x[j].
struct LdItem {
    const double *a;     // matrix base (row 0, col 0), 16-byte aligned, even ld
    int32_t rows;        // number of rows (the reduction dimension)
    int32_t ld;          // leading dimension in doubles (even)
    int32_t col0;        // first column of this slab (multiple of 128)
    int32_t ncols;       // total columns of the matrix
    int32_t x_off;       // offset of x[0] in the vector pool
    int32_t y_off;       // offset of out[0] (column 0) in the vector pool
    int32_t dot_off;     // offset in the pool of the vector to dot the output with, or -1
    int32_t dot_slot;    // where to store this item's partial of that dot product
    const double *scale; // out[c] is multiplied by scale[c] (the eigenvalues), or nullptr
};

// Second pass of an eigen-form block, y = U t' with t' = s * (U^T x): one work item = a chunk of
// rows of one 128-column slab of the SAME row-major U the first pass read (so it can still be in
// the Infinity Cache): partial row sums S[slab][i] = sum_{c in slab} U[i][c] t'[c].
struct RowItem {
    const double *a;     // U + r0 * ld + c0 (128-byte aligned)
    int32_t rows;        // rows in this chunk
    int32_t ld;          // leading dimension of U (multiple of 16 doubles)
    int32_t w;           // columns in this slab (<= 128)
    int32_t t_off;       // pool offset of t'[c0]
    int32_t s_off;       // scratch offset of S[slab][r0]
    int32_t pad;
};

// y[i] = sum_{J < ns} S[J][i] for 256 rows of one eigen-form block, with the y.z partial
struct RowCombItem {
    int32_t n, ns;            // block size, number of 128-column slabs of U
    int32_t s_base;           // scratch offset of S[0][0] (n entries per slab)
    int32_t y_off, dot_off, dot_slot;
    int32_t i0;               // first of the (up to) 256 rows this workgroup combines
    int32_t pad;
};

// Fused eigen-form product (both uses of U in one workgroup, U read from HBM once and held in
// registers in between).  U of such a block is stored COLUMN-major (column stride ldc = n rounded
// up to even, pad row zero).  One work item = a run of consecutive columns ("slab") of one block;
// the partial y of the slab goes to scratch S[slab][i] and ld_rowsum_combine_kernel adds the slabs
// in order.  No workgroup ever waits for another.
struct EigItem {
    const double *a;       // first column of this slab
    const double *scale;   // eigenvalue of the slab's first column
    int32_t n;             // rows of the block
    int32_t ncols;         // columns in this slab
    int32_t ldc;           // column stride in doubles (even)
    int32_t x_off;         // pool offset of x[0] of the block
    int32_t s_off;         // scratch offset of S[slab][0]
    int32_t direct;        // > 0: the slab is the whole block -- s_off is the POOL offset of y[0] and
                           // direct - 1 the slot of the block's y.z partial; nothing goes to scratch
};

// One work item of the SYMMETRIC dense product: the panel of one block below (and including)
// the diagonal tile of a 128-column slab, stored contiguously: rows j0..n-1, `ld` doubles each.
// A panel is cut into chunks of at most `chunk_rows` rows (a multiple of 32), one work item each,
// so no workgroup streams more than chunk_rows x 1 KiB: on a small shard (an 8-GPU rank holds
// ~1000 panels for 2048 workgroup slots) the longest panel would otherwise set the kernel time.
struct SymItem {
    const double *a;     // first row of this chunk (row j0 + r0 of the block, first column of the
                         // slab), 128-byte aligned
    int32_t rows;        // rows in this chunk
    int32_t ld;          // panel leading dimension (multiple of 16 doubles)
    int32_t w;           // columns in this slab (<= 128)
    int32_t j0;          // first row/column of the slab inside the block
    int32_t x_off;       // pool offset of x[0] of the block
    int32_t s_off;       // scratch offset of S[slab][0]: row sums, n entries per slab
    int32_t r0;          // first row of this chunk inside the panel (0: it holds the diagonal tile)
    int32_t c_off;       // scratch offset of C[slab][chunk][0]: this chunk's 128 column sums
};

// combine step of the symmetric product for one block:
//   y[j] = sum_{J < j/128} S[J][j] + sum_{chunks c of slab j/128} C[j/128][c][j%128]
struct SymCombItem {
    int32_t n;                // block size
    int32_t s_base;           // scratch offset of S[0][0] of the block
    int32_t y_off, dot_off, dot_slot;
    int32_t j0;               // first of the (up to) 256 columns this workgroup combines
    int32_t c_base;           // scratch offset of C[0][0][0] of the block
    int32_t nch_max;          // chunks per slab the C region is laid out for: ceil(n / chunk_rows)
    int32_t chunk_rows;
};

// One work item of the TILED symmetric product (ld_tile_kernel): rows [R0, R1) of the slabs
// J0 .. J0 + nJ - 1 of one block.  Row strips are `tile_rows` high, column strips `tile_slabs` slabs
// wide (tile_rows a multiple of the strip width); an item whose columns lie among its rows
// ("diagonal": col_off < 0) starts every slab at that slab's diagonal tile.
struct SymTile {
    const double *a;     // the block's store (panel 0, row 0), 128-byte aligned
    int32_t n;           // block size
    int32_t x_off;       // pool offset of x[0] of the block
    int32_t R0, R1;      // rows of the block this item takes
    int32_t J0, nJ;      // its slabs
    int32_t out0;        // first row it has a row sum for: max(R0, 128 J0)
    int32_t row_off;     // scratch offset of that row's partial (slot J0 / tile_slabs)
    int32_t col_off;     // scratch offset of column 128 J0's partial, or -1: merged into the rows' run
    int32_t direct;      // > 0: the item is the whole block -- row_off is the POOL offset of y[0] and
                         // direct - 1 the slot of the block's y.z partial; nothing goes to scratch
};
// combine step of the tiled product: y[j] = sum of slots 0 .. j / cw + G - j / tr - 1 of S[slot][pad2(n)]
struct TileCombItem {
    int32_t n;
    int32_t s_base;
    int32_t y_off, dot_off, dot_slot;
    int32_t j0;               // first of the (up to) 256 entries this workgroup combines
    int32_t cw, tr, G;        // column-strip width, row-strip height, number of row strips
};

struct TauArg { double v[VILMA_MAX_P]; };

// ---- device-resident sweep (sweep.hip): what changes from sweep to sweep lives on the device ----
// The buffers one phase of a queued sweep works on.  Kernels launched while a phase is set
// (set_launch_phase) take these instead of the pointers in their arguments, so the host can queue
// a sweep before it knows which candidate the previous line search accepted.
struct PhasePtrs {
    const double *mu_in;
    double *mu_out, *mu_out2;
    const double *pool_cur, *m_cur, *lse_ref;
    double *pool_out, *m_out, *v_out, *lse_out;
    double *pool_out2, *m_out2, *v_out2, *lse_out2;
    // real_posterior_mean of the last completed sweep, and where an evaluation that carries the
    // convergence statistics leaves the means it has just formed (the same buffer unless an
    // error-scaling re-evaluation may follow in the same sweep)
    const double *snap_in;
    double *snap_out;
    double step, step2;
    double tau[VILMA_MAX_P];            // error_scaling in force (the device may have updated it)
    // "lazy" trials (mixtures beyond the on-chip stash, see SnpKernelArgs::no_store) store no
    // candidate vi_mu.  Every state the beta loop reaches from a stored vi_mu ("base") is
    //     mu_k = a mu_k^base + Sig_k c,      Sig_k = (Prec_k + D)^-1,   a scalar, c [P] per SNP,
    // because a blend mu_k' = Sig_k (s g + (1 - s) Lam_k mu_k) = (1 - s) mu_k + s Sig_k g keeps that form:
    // a' = (1 - s) a, c' = (1 - s) c + s g.  A lazy trial reads the base, the coefficient a (a_def) and
    // the vector c of the state it starts from (c_cur; c_zero: that state IS the base) and leaves
    // its candidates' c' beside their moments (c_out, c_out2); an accepted update that does not end the
    // beta loop only moves (a, c) in the control block -- nothing of size [M][P][N] is read again or
    // written.  The pass that forms the responsibility sums when the loop ends derives the accepted
    // state from (mu_in = base, a_pend, c_pend) and writes it to mu_mat: ONE vi_mu array written per
    // SWEEP where round 4 wrote one per accepted update.
    double *mu_mat;                     // SUMS phase: where the accepted state's vi_mu goes, or nullptr
    const double *c_pend;               // ... its vector c [P][N]
    double a_pend;                      // ... its coefficient a
    const double *c_cur;                // TRIAL phase: c of the state the trial starts from
    double *c_out, *c_out2;             // ... where the candidates' c' go
    double a_def;                       // ... a of the state the trial starts from
    int32_t c_zero;                     // ... that state is the stored vi_mu itself (a = 1, c = 0)
};
#define VILMA_PHASE_EVAL 0      // an evaluation of the current vi_mu (after the M-step / a tau update)
#define VILMA_PHASE_TRIAL 1     // a beta trial from the current state
#define VILMA_PHASE_SUMS 2      // responsibility sums of the candidate a TRIAL decision accepted
                                // (mixtures beyond the stash), materialising it after a lazy trial
struct SweepCtl {
    int32_t alive;              // 0: every kernel queued under this block exits at once
    int32_t run_eval;           // the evaluation queued behind the last TRIAL decision runs (that
                                // decision ended the sweep's beta loop and did the M-step)
    int32_t run_eval2;          // the re-evaluation queued behind the last EVAL decision runs (that
                                // decision updated the error scaling)
    int32_t run_sums;           // the sums pass queued behind the last TRIAL decision runs (mixtures
                                // beyond the stash: that decision accepted a candidate)
    int32_t choice;             // last trial decision: 1 = candidate A accepted, 2 = B, 0 = neither
    int32_t stage;              // decisions taken since the block was armed
    int32_t running_none;       // running ELBO change not defined yet (first sweep)
    int32_t eval_pending;       // an evaluation has run whose sums no decision has looked at yet:
                                // 1 = the one after the M-step, 2 = the one after a tau update
    int32_t inner_it;           // beta updates accepted so far in the sweep in progress
    int32_t snap_cur;           // which of the two snapshot buffers holds the last completed sweep's means
    int32_t mu_role[3], mom_role[3];    // buffer indices in the roles current / candidate A / B
    double L_try;               // L[0] of the next (or just evaluated) trial's candidate A (B: L_try * rate)
    double L0;                  // L[0] after the last accepted line search
    double cur_obj;             // objective of the current (accepted) state
    double delta_sum;           // _nat_grad_step's delta_sum of the sweep in progress
    double running;             // running ELBO change as of the start of the sweep in progress
    double tau[VILMA_MAX_P];    // error_scaling in force
    double hrl[VILMA_MAX_P];    // 0.5 * ld_rank * log(tau) (det_log: the host gets the same bits)
    double a_def;               // lazy trials: the current state is a_def * (vi_mu of role 0) + Sig c, c in
    int32_t c_zero;             // the c buffer beside the current moments; c_zero: it is that vi_mu itself
    // PERSISTENT lazy state (SweepDecideParams::persist): the state is a_def * (vi_mu of buffer mu_base)
    // + Sig c for as long as the block lives -- no vi_mu is written by any sweep.  The three c buffers
    // then take the roles the vi_mu buffers would (mu_role: current / candidate A / candidate B, rotated
    // by every accept), so the c of the last reported state survives one trial queued beyond it exactly
    // as a stored vi_mu would; the evaluation behind the M-step derives mu_k from (base, a, c) as the
    // trials do.  Whoever needs the state as an array (a hand-back to the host, a drain, the end of the
    // run) writes it out once (sweep.hip: persist_writeback).
    int32_t mu_base;
    // --learn-scaling under a persistent lazy state: a tau update needs the state as an array (written
    // with the old tau).  tau_hot: the last sweep's EVAL decision moved tau -- the next sweep then ends
    // with the storing form of the sums pass (as without persistence: a tau that moves usually moves
    // again, and the state is in memory when it does); run_mat: the write-out pass queued behind the
    // last EVAL decision runs (tau moved under a state still in (a, c) form)
    int32_t tau_hot, run_mat;
    int32_t dbg_deferred;       // (tests) TRIAL decisions that found the state in that form so far
    PhasePtrs phase[3];
};
// base pointers of the three buffers of each kind (c: the vectors of lazy trials, one beside each set
// of moments), the two snapshot buffers
struct BufferBases { double *mu[3], *pool[3], *m[3], *v[3], *lse[3], *c[3], *snap[2]; };
// The pointers of a phase from the roles at the start of its stage (cur, ta, tb).  The
// evaluation reads the current vi_mu and writes the moments of role ta; the trial that follows
// treats those as current (the evaluation is accepted unconditionally), writes candidate A into
// the old current moments and role ta of vi_mu, candidate B into role tb.  (A trial that follows a
// trial -- the inner beta loop -- finds the roles arranged by the decision so that the same rule
// holds: the accepted candidate's moments sit in role ta.)
static __host__ __device__ inline void phase_ptrs(const BufferBases &b, const int32_t (&mu)[3],
                                                  const int32_t (&mom)[3], int phase, double step,
                                                  double step2, PhasePtrs &o, bool persist = false,
                                                  int mu_base = 0) {
    const int cur = phase == VILMA_PHASE_EVAL ? mom[0] : mom[1];      // moments treated as current
    const int ta = phase == VILMA_PHASE_EVAL ? mom[1] : mom[0];       // ... written as candidate A
    const int tb = mom[2];
    o.mu_in = b.mu[mu[0]]; o.mu_out = b.mu[mu[1]]; o.mu_out2 = b.mu[mu[2]];
    o.pool_cur = b.pool[cur]; o.m_cur = b.m[cur]; o.lse_ref = b.lse[cur];
    o.pool_out = b.pool[ta]; o.m_out = b.m[ta]; o.v_out = b.v[ta]; o.lse_out = b.lse[ta];
    o.pool_out2 = b.pool[tb]; o.m_out2 = b.m[tb]; o.v_out2 = b.v[tb]; o.lse_out2 = b.lse[tb];
    o.c_cur = b.c[cur]; o.c_out = b.c[ta]; o.c_out2 = b.c[tb];
    if (persist) {
        // (SweepCtl::mu_base) the stored vi_mu never moves; the c buffers go by the vi_mu roles
        o.mu_in = b.mu[mu_base]; o.mu_out = nullptr; o.mu_out2 = nullptr;
        o.c_cur = b.c[mu[0]]; o.c_out = b.c[mu[1]]; o.c_out2 = b.c[mu[2]];
    }
    o.step = step; o.step2 = step2;
}
// (snap_in / snap_out and tau of a PhasePtrs are set by set_phase_extras)
static __host__ __device__ inline void set_phase_extras(const BufferBases &b, int snap_cur,
                                                        bool two_snapshots, const double *tau,
                                                        double a_def, int32_t c_zero, PhasePtrs &o) {
    o.snap_in = b.snap[snap_cur];
    o.snap_out = b.snap[two_snapshots ? 1 - snap_cur : snap_cur];
    for (int p = 0; p < VILMA_MAX_P; ++p) o.tau[p] = tau[p];
    o.mu_mat = nullptr; o.c_pend = nullptr; o.a_pend = 1.0;
    o.a_def = a_def; o.c_zero = c_zero;
}
// launch attribute of the calling thread like set_launch_predicate: kernels launched while it is
// set work on *pp's buffers (read on the device when the kernel starts); nullptr = their arguments
void set_launch_phase(const PhasePtrs *pp);

struct SnpKernelArgs {
    int32_t N, M, A, P;
    const double *mu_in;      // [M][P][N]
    double *mu_out;           // [M][P][N] (trial) or nullptr
    const double *adj, *se, *sld;        // [P][N]
    const int32_t *annot;     // [N]
    const int32_t *invperm;   // [P][N] SNP -> LD position
    const double *prec;       // [M][P][P]
    const double *log_det;    // [M]
    const double *lh;         // [A][M] log hyper - 0.5 log_det
    const double *pool_cur;   // current vector pool: x_ld [P][N] | y_ld [P][N] | t scratch
    const double *m_cur;      // [P][N] posterior mean of the current state
    double *pool_out;         // trial pool (x_ld written)
    double *m_out, *v_out;    // [P][N]
    double *lse_out;          // [N]
    double *partials;         // [2P+2 | 6 | 2P+2][grid] (column-major: finalize reads columns):
                              // candidate sums, fused statistics, second candidate's sums
    const double *scal;       // [P][N] scalings                          } used when diff != 0
    const double *snapshot;   // [P][N] real_posterior_mean to compare with } (plain evaluations
    double *snapshot_out;     // [P][N] where the new means go (may be the same) } only)
    int32_t diff;             // fuse the convergence statistics into this evaluation
    double step;
    // second candidate of a two-step beta trial (launch_snp_pass with ns = 2)
    double step2;
    double *mu_out2, *pool_out2, *m_out2, *v_out2, *lse_out2;
    // shift of the softmax: the log-normaliser [N] of the current accepted state (nullptr: none yet)
    const double *lse_ref;
    // per-tile responsibility sums [candidate][tile][A*M] of the candidates (nullptr: not wanted)
    double *sum_partials;
    TauArg tau;
    // a LAZY beta trial (no_store != 0; launch_snp_pass without the stash only; queued sweeps only, so
    // its buffers come from PhasePtrs): the candidates' vi_mu are not stored -- see PhasePtrs --
    // and the pass that needs the accepted state derives it (DeltaArgs::mat)
    // no_store on a plain EVALUATION (a queued sweep with a persistent lazy state, SweepCtl::mu_base):
    // the state evaluated is a_def (stored vi_mu) + Sig c_cur (PhasePtrs), derived component by
    // component with the trials' expressions
    // (2: ... and the host knows the state has a == 0 for as long as the pass can run: the variant
    // without any vi_mu load, up to two cohorts)
    int32_t no_store;
    const int *pred;          // filled by the launcher (set_launch_predicate)
    const PhasePtrs *pp;      // filled by the launcher (set_launch_phase)
};

// launch attribute of the calling thread: kernels launched while it is set exit at once when
// *flag == 0 (sweeps queued ahead of the decision that may turn them dead); nullptr = unconditional
void set_launch_predicate(const int *flag);

// ns = 2: a beta trial at the two step sizes a.step / a.step2, second candidate into the *2 outputs
void launch_snp_pass(const SnpKernelArgs &a, bool blend, int ns, hipStream_t s);
// The vi_mu buffers' layout in HBM (kernels.hip, MU_TILED): elements to allocate per buffer (>= M*P*N),
// and the conversion between the reference's [M*P][N] array and that layout
int64_t mu_buffer_elems(int64_t N, int M, int P);
bool mu_is_tiled();
// debug poison (VILMA_DEBUG_POISON): NaN into p[0, n) -- which = 1 / 2: into the vi_mu buffer the current
// launch phase assigns to candidate A / B instead of p; honours the launch predicate
void launch_poison(double *p, int64_t n, int which, hipStream_t s);
void launch_mu_tile(double *buf, double *nat, int64_t N, int MP, int r0, int R, bool to_tiled, hipStream_t s);
int snp_pass_grid(int64_t N);       // workgroups of the thread-per-SNP kernels (256 SNPs each)
int snp_tile_grid(int64_t N);       // workgroups of launch_snp_pass (each loops over tiles of 64 SNPs) = rows of its partials
// can launch_snp_pass deliver the responsibility sums of ns candidates (LDS stash fits)?
bool snp_pass_can_stash(int M, int P, int ns);
// rows per candidate of SnpKernelArgs::sum_partials ([candidate][row][A*M]): one per workgroup with
// a single annotation, one per tile otherwise; doubles to allocate; and the reduction of the rows
// (candidate z's sums to out + z * out_zstride)
int snp_sum_rows(int64_t N, int A);
int64_t tile_sums_elems(int64_t N, int A, int M, int ns);
void launch_tile_sums(const double *rows_dev, int64_t N, int A, int M, int ns, double *out,
                      int64_t out_zstride, hipStream_t s);

// keep = true: default cache policy (the stream is read again by launch_ld_rowsum right after);
// false: non-temporal.  pool1 != nullptr: two right-hand sides per pass over U.
void launch_ld_colsum(const LdItem *items, int n_items, double *pool0, double *pool1, bool keep,
                      hipStream_t s);
void launch_ld_rowsum(const RowItem *items, int n_items, const double *pool0, const double *pool1,
                      double *scratch, int64_t s_stride, hipStream_t s);
void launch_ld_rowsum_combine(const RowCombItem *items, int n_items, double *pool0, double *pool1,
                              const double *scratch, int64_t s_stride, double *dot_partials,
                              int dot_stride, hipStream_t s);

// Fused eigen-form product on column-major U (see EigItem).  eig_rows_per_thread(n): 2, 4, 8, 12,
// 24 (= 12 rows per thread of a 512-thread workgroup: blocks of 3 073 .. 6 144 rows, launch_ld_eig_tall)
// or 0 = block too tall (two-pass kernels); columns are taken eig_batch_cols(R) at a time.
int eig_rows_per_thread(int n);
int eig_batch_cols(int R);
// One launch takes the items of ONE class R (rows per thread).
void launch_ld_eig_fused(const EigItem *items, int n_items, int R, const double *pool0,
                         const double *pool1, double *scratch, int64_t s_stride,
                         double *dot_partials, int dot_stride, hipStream_t s);
// the class of blocks of up to 512 rows (R = 2): one WAVE per slab of columns, no barrier
void launch_ld_eig_tall(const EigItem *items, int n_items, const double *pool0, const double *pool1,
                        double *scratch, int64_t s_stride, double *dot_partials, int dot_stride,
                        hipStream_t s);
void launch_ld_eig_wave(const EigItem *items, int n_items, const double *pool0, const double *pool1,
                        double *scratch, int64_t s_stride, double *dot_partials, int dot_stride,
                        hipStream_t s);
// the items of every class in one launch (small shards: one ramp and tail instead of four)
void launch_ld_eig_fused_all(const EigItem *items, int n_items, const double *pool0,
                             const double *pool1, double *scratch, int64_t s_stride,
                             double *dot_partials, int dot_stride, hipStream_t s);
// row-major [n x r] -> column-major [r][ldc], rows n .. ldc-1 zero (load time)
void launch_repack_columns(const double *src, int n, int r, int64_t ldc, double *dst,
                           hipStream_t s);

// pool1 != nullptr: two right-hand sides in one pass over the LD store; the second one's scratch
// sits s_stride doubles, its y.z partials dot_stride slots behind the first one's
void launch_ld_sym(const SymItem *items, int n_items, const double *pool0, const double *pool1,
                   double *scratch, int64_t s_stride, hipStream_t s);
// per-workgroup trace of ld_sym_kernel (builds with -DLD_TRACE=1 only; else returns 1): row b of
// buf = {s_memrealtime at start, at end (100 MHz ticks), XCC id, bytes of the chunk, core-clock
// cycles (s_memtime) between start and end}
int set_ld_trace(double *buf_dev, int64_t capacity_rows);
void launch_ld_tile(const SymTile *items, int n_items, int max_slabs, double *pool0, double *pool1,
                    double *scratch, int64_t s_stride, double *dot_partials, int dot_stride,
                    hipStream_t s);
void launch_ld_tile_combine(const TileCombItem *items, int n_items, double *pool0, double *pool1,
                            const double *scratch, int64_t s_stride, double *dot_partials,
                            int dot_stride, hipStream_t s);
void launch_ld_sym_combine(const SymCombItem *items, int n_items, double *pool0, double *pool1,
                           const double *scratch, int64_t s_stride, double *dot_partials,
                           int dot_stride, hipStream_t s);

// For each of ncand candidates (its per-SNP partial columns / y.z partials sit behind the first
// one's): totals[0..2P) and [3P..3P+2) from the per-SNP partials, totals[2P..3P) from the matvec
// dots.  With dsum/dmax non-null also the six fused convergence statistics (columns 2P+2..2P+7 of
// the first candidate's partials): three sums -> dsum[3], three maxima -> dmax[3].  With sum_rows
// non-null also the responsibility sums of each candidate from the per-tile rows of a stashing
// pass ([ncand][sum_nrows][AM]) -> sums_a / sums_b.  One launch.
void launch_finalize(const double *snp_partials, int snp_rows, int P, const double *dot_partials,
                     int dot_stride, const int32_t *dot_start /*[P+1] host*/, int ncand,
                     double *totals_a, double *totals_b, double *dsum, double *dmax,
                     const double *sum_rows, int sum_nrows, int AM, double *sums_a, double *sums_b,
                     hipStream_t s);

struct DeltaArgs {
    int32_t N, M, A, P;
    const PhasePtrs *pp;      // non-null: vi_mu, lse and tau of the state a queued sweep's EVAL
                              // phase starts from (PhasePtrs::mu_in, lse_ref, tau) instead of the
                              // three fields below
    const double *mu;         // [M][P][N]
    const double *sld;        // [P][N]
    const int32_t *annot;
    const double *prec, *log_det, *lh;
    const double *lse;        // [N]
    // mat != 0: `mu` is the stored vi_mu lazy trials started from; the state whose sums are wanted is
    // mu_k' = acoef mu_k + Sig_k cvec (PhasePtrs), which is also written to mu_mat
    // (with pp non-null: PhasePtrs::mu_mat / c_pend / a_pend)
    int32_t mat;
    double *mu_mat;           // [M][P][N]
    const double *cvec;       // [P][N]
    double acoef;
    double *out;              // mode 0: partial rows [grid*4][A*M]; mode 1: delta [M][N]
    TauArg tau;
    const int *pred;          // filled by the launcher (set_launch_predicate)
};
void launch_delta_sums(const DeltaArgs &a, double *sums_out /*[A*M]*/, hipStream_t s);
void launch_delta_write(const DeltaArgs &a, hipStream_t s);
// out [n][M] <- rows i0 .. i0 + n of the transpose of in [M][N] (vilma_get_delta)
void launch_transpose_km(const double *in, double *out, int64_t N, int M, int64_t i0, int n, hipStream_t s);
// vi_sigma of components [k0, k0 + nk) into out [nk][P][P][N] (vilma_get_vi_sigma)
void launch_vi_sigma(int P, const double *prec, const double *sld, const TauArg &tau, int64_t N, int k0, int nk,
                     double *out, hipStream_t s);
int delta_grid(int64_t N);
// rows ([A*M] doubles each) of the partials buffer launch_delta_sums needs: one per wave plus the
// scratch rows of its multi-pass column reduction
int64_t delta_partial_rows(int64_t N);

// _initialize's per-SNP part on the device: vi_mu [M][P][N] and the heuristic responsibility sums
struct InitArgs {
    int32_t N, M, A, P;
    const double *fake_mu;    // [P][N]
    const double *sld;        // [P][N]
    const int32_t *annot;
    const double *prec, *log_det;
    double *mu_out;           // [M][P][N]
    double *c_out;            // [P][N] or nullptr: the vector c with mu_out_k = Sig_k c (the reference's
                              // temp_nat_mu, variational_inference.py:683-690): the state _initialize
                              // builds IS of the lazy form a = 0 (PhasePtrs)
    double *partials;         // [init_partial_rows(N)][A*M]
    TauArg tau;
};
void launch_init_state(const InitArgs &a, double *sums_out /*[A*M]*/, hipStream_t s);
int64_t init_partial_rows(int64_t N);

// objective pieces with a caller-supplied vi_delta ([M][N], component-major); writes the trial
// moments / pool like launch_snp_pass and max |delta - derived delta| to *maxdev_out
void launch_snp_given_delta(const SnpKernelArgs &a, const double *delta_km, const double *lse_cur,
                            double *maxdev_partials /*[snp_pass_grid]*/, double *maxdev_out,
                            hipStream_t s);

void launch_gather_x(const double *x_snp, const int32_t *invperm, double *pool_x, int N, int P,
                     hipStream_t s);
void launch_scatter_y(const double *pool_y, const int32_t *invperm, double *y_snp, int N, int P,
                      hipStream_t s);

void launch_mstep(const double *sums, const double *counts, const double *log_det, int A, int M,
                  double *hyper, double *lh, hipStream_t s);

void launch_mean_diff(const double *m_cur, const double *scalings, double *snapshot, int64_t PN,
                      double *partials, double *out_sum3, double *out_max3, bool compare,
                      hipStream_t s);
int mean_diff_grid(int64_t PN);

// ---- device-resident sweep: the decision kernel (see kernels.hip) ----
#define VILMA_SNAP_EXTRA 56     // scalars of the control block behind the result vector in a
                                // snapshot; the last one is the serial number that completes it
// what a decision reports in its snapshot (offsets behind the result vector)
enum {
    SNAP_ALIVE = 0, SNAP_KIND, SNAP_OUTCOME, SNAP_STAGE, SNAP_CHOICE, SNAP_L_TRY, SNAP_L0,
    SNAP_CUR_OBJ, SNAP_DELTA_SUM, SNAP_RUNNING, SNAP_RUNNING_NONE, SNAP_INNER_IT, SNAP_ORIG,
    SNAP_FA, SNAP_FB, SNAP_EVAL_OBJ, SNAP_CONSUMED, SNAP_SWEEP_END, SNAP_SWEEP_CHANGE,
    SNAP_L_TRIED, SNAP_SNAP_CUR, SNAP_RUN_EVAL, SNAP_RUN_EVAL2, SNAP_EVAL_PENDING, SNAP_RUN_SUMS,
    SNAP_MU_ROLE = 26, SNAP_MOM_ROLE = 29, SNAP_TAU = 32, SNAP_HRL = 40, SNAP_A_DEF = 48, SNAP_C_ZERO = 49,
    SNAP_MU_BASE = 50, SNAP_TAU_HOT = 51, SNAP_RUN_MAT = 52,
    SNAP_SERIAL = VILMA_SNAP_EXTRA - 1
};
#define VILMA_DECIDE_TRIAL 0    // behind a beta trial (and the evaluation in front of it, if one ran)
#define VILMA_DECIDE_EVAL 1     // behind an evaluation: consume it; with scale_se decide the tau update
// outcomes
#define VILMA_OUT_NONE 0        // nothing to decide (dead block, or no evaluation had run)
#define VILMA_OUT_ACCEPT_MSTEP 1    // candidate accepted, beta loop over, M-step done: evaluation runs
#define VILMA_OUT_ACCEPT_CONTINUE 2 // candidate accepted, the inner beta loop goes on: next trial
#define VILMA_OUT_REJECTED 3        // every candidate rejected: next trial at larger L
#define VILMA_OUT_DEAD 4            // the host has to take this decision
#define VILMA_OUT_SWEEP_END 5       // (EVAL) evaluation consumed, sweep over
#define VILMA_OUT_TAU_UPDATED 6     // (EVAL) evaluation consumed, tau updated: re-evaluation runs
struct SweepDecideParams {
    int mode;                       // VILMA_DECIDE_TRIAL / VILMA_DECIDE_EVAL
    int P, A, M;
    int check_convergence, have_b, have_sums_b;
    int mstep_inside;               // the TRIAL decision does the M-step itself (the accepted
                                    // candidate's responsibility sums are in the result vector)
    int lazy;                       // the trials store no vi_mu: the sums pass behind an accepting
                                    // decision materialises the accepted candidate
    int persist;                    // ... or (lazy, no --learn-scaling, P <= 4) nothing does: the state
                                    // stays (stored vi_mu, a, c) from sweep to sweep (SweepCtl::mu_base)
    int scale_se;                   // an EVAL decision may update tau
    int two_snapshots;              // evaluations write their means to the other snapshot buffer
    int max_inner;                  // MAX_NUM_ITERS
    int debug_kill_deferred;        // tests (VILMA_DEBUG_KILL_DEFERRED=k): the k-th TRIAL decision that finds
                                    // the state carried as (a, c) hands the sweep back to the host undecided
    const double *chi, *ranks;      // host [P]
    double rel_tol, abs_tol, rate, l_max, em_tol;
    SweepCtl *ctl;
    double *results;
    int o_dsum, o_tot, o_ta, o_tb, o_sa, o_sb, o_hyper, n_results;
    double *lh;
    const double *counts, *log_det;
    double *snap;                   // host memory the device can write (hipHostMallocMapped)
    double serial;
    BufferBases bases;
};
void launch_sweep_decide(const SweepDecideParams &p, hipStream_t s);
