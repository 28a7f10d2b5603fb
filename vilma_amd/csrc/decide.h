// The decisions of a device-resident sweep as ONE piece of source for both sides: thread 0 of
// sweep_decide_kernel runs decide_core on the device's control block, and the host (sweep.hip)
// runs the very same function on its mirror of that block with the result vector the decision saw
// (from the snapshot) -- IEEE double arithmetic without fused multiply-add on both sides
// (detmath.h) -- and compares.  See the comment in front of sweep_decide_kernel (kernels.hip) for
// what is decided and which reference lines it follows.
#pragma once
#include "kernels.h"
#include "detmath.h"

struct SweepDecideArgs {
    int32_t mode, P, A, M;
    int32_t check_convergence;       // veto when dsum[0] == 0
    int32_t have_b, have_sums_b;     // candidate B evaluated / its responsibility sums available
    int32_t mstep_inside, lazy, persist, scale_se, two_snapshots, max_inner, debug_kill_deferred;
    double chi[VILMA_MAX_P], ranks[VILMA_MAX_P];
    double rel_tol, abs_tol, rate, l_max, em_tol;
    SweepCtl *ctl;
    const double *results;           // the context's result vector (include/vilma_hip.h layout)
    int32_t o_dsum, o_tot, o_ta, o_tb, o_sa, o_sb, o_hyper, n_results;
    double *hyper;                   // results + o_hyper
    double *lh;
    const double *counts, *log_det;
    double *snap;                    // [n_results + VILMA_SNAP_EXTRA], host memory mapped for the device
    double serial;                   // written behind the snapshot when it is complete
    BufferBases bases;
};

static inline SweepDecideArgs decide_args(const SweepDecideParams &p) {
    SweepDecideArgs a;
    a.mode = p.mode; a.P = p.P; a.A = p.A; a.M = p.M;
    a.check_convergence = p.check_convergence; a.have_b = p.have_b; a.have_sums_b = p.have_sums_b;
    a.mstep_inside = p.mstep_inside; a.lazy = p.lazy; a.persist = p.persist; a.scale_se = p.scale_se;
    a.two_snapshots = p.two_snapshots;
    a.max_inner = p.max_inner;
    a.debug_kill_deferred = p.debug_kill_deferred;
    for (int q = 0; q < VILMA_MAX_P; ++q) {
        a.chi[q] = q < p.P ? p.chi[q] : 0.0;
        a.ranks[q] = q < p.P ? p.ranks[q] : 1.0;
    }
    a.rel_tol = p.rel_tol; a.abs_tol = p.abs_tol; a.rate = p.rate; a.l_max = p.l_max; a.em_tol = p.em_tol;
    a.ctl = p.ctl; a.results = p.results;
    a.o_dsum = p.o_dsum; a.o_tot = p.o_tot; a.o_ta = p.o_ta; a.o_tb = p.o_tb; a.o_sa = p.o_sa;
    a.o_sb = p.o_sb; a.o_hyper = p.o_hyper; a.n_results = p.n_results;
    a.hyper = p.results + p.o_hyper; a.lh = p.lh; a.counts = p.counts; a.log_det = p.log_det;
    a.snap = p.snap; a.serial = p.serial; a.bases = p.bases;
    return a;
}

// what a decision found and did (beside the control block itself)
struct DecideReport {
    int outcome, consumed, sweep_end, mstep;
    double orig, fa, fb, eval_obj, sweep_change, L_tried;
    // the block as it stood when the sweep ended (before the same decision's line search moved on)
    double end_L0, end_running, end_tau[VILMA_MAX_P];
    int32_t end_mu_role[3], end_mom_role[3], end_snap_cur;
    double end_a_def;           // (persistent lazy state: the reported state is end_a_def (vi_mu of buffer
    int32_t end_c_zero;         // end_mu_base) + Sig c, c in the buffer of end_mu_role[0])
    int32_t end_mu_base;
};

static __host__ __device__ inline void decide_set_phases(const SweepDecideArgs &a, SweepCtl *ctl) {
#pragma clang fp contract(off)
    const double Lt = ctl->L_try;
    const double s1 = 1.0 / Lt;
    const double Lt2 = Lt * a.rate;
    const double s2 = 1.0 / Lt2;
    phase_ptrs(a.bases, ctl->mu_role, ctl->mom_role, VILMA_PHASE_EVAL, 0.0, 0.0, ctl->phase[0],
               a.persist != 0, ctl->mu_base);
    phase_ptrs(a.bases, ctl->mu_role, ctl->mom_role, VILMA_PHASE_TRIAL, s1, s2, ctl->phase[1],
               a.persist != 0, ctl->mu_base);
    set_phase_extras(a.bases, ctl->snap_cur, a.two_snapshots != 0, ctl->tau, ctl->a_def, ctl->c_zero, ctl->phase[0]);
    set_phase_extras(a.bases, ctl->snap_cur, a.two_snapshots != 0, ctl->tau, ctl->a_def, ctl->c_zero, ctl->phase[1]);
    // the sums pass works on the state the evaluation starts from (a lazy accept overrides this)
    ctl->phase[VILMA_PHASE_SUMS] = ctl->phase[VILMA_PHASE_EVAL];
}

static __host__ __device__ inline double decide_abs(double x) { return x < 0.0 ? -x : x; }

static __host__ __device__ inline void decide_core(const SweepDecideArgs &a, SweepCtl *ctl,
                                                   const double *results, DecideReport &rep) {
#pragma clang fp contract(off)
    const int P = a.P;
    rep.outcome = VILMA_OUT_NONE; rep.consumed = 0; rep.sweep_end = 0; rep.mstep = 0;
    rep.orig = 0.0; rep.fa = 0.0; rep.fb = 0.0; rep.eval_obj = 0.0; rep.sweep_change = 0.0;
    rep.L_tried = 0.0; rep.end_L0 = 0.0; rep.end_running = 0.0; rep.end_snap_cur = 0;
    rep.end_a_def = 1.0; rep.end_c_zero = 1; rep.end_mu_base = 0;
    for (int p = 0; p < VILMA_MAX_P; ++p) rep.end_tau[p] = 1.0;
    for (int q = 0; q < 3; ++q) { rep.end_mu_role[q] = 0; rep.end_mom_role[q] = 0; }
    if (!ctl->alive) return;
    bool dead = false;
    // ---- (1) an evaluation no decision has looked at yet
    if (ctl->eval_pending) {
        rep.consumed = ctl->eval_pending;
        const double *t = results + a.o_tot;
        rep.eval_obj = det_objective(P, a.chi, ctl->tau, ctl->hrl, t);
        ctl->delta_sum = ctl->delta_sum + (rep.eval_obj - ctl->cur_obj);
        ctl->cur_obj = rep.eval_obj;
        ctl->eval_pending = 0;
        bool end_now = true;
        if (a.mode == VILMA_DECIDE_EVAL && rep.consumed == 1 && a.scale_se &&
            ctl->delta_sum < a.em_tol) {
            // _update_error_scaling (variational_inference.py:472-486)
            double tau[VILMA_MAX_P];
            bool ok = true;
            for (int p = 0; p < P; ++p) {
                tau[p] = det_tau(a.chi[p], t[p], t[2 * P + p], t[P + p], a.ranks[p]);
                ok = ok && tau[p] > 0.0 && tau[p] < __builtin_huge_val();
            }
            if (!ok) {
                dead = true;            // the host reports the error
            } else {
                // A persistent lazy state a (stored vi_mu) + Sig_k c must stay put while Sig_k =
                // (Prec_k + D / tau)^-1 changes under it: the pass queued behind this decision (it runs
                // with the re-evaluation) writes it out with the OLD tau into the vi_mu buffer next to
                // the stored one, and that array is the stored state from here on
                const int32_t mat_from = ctl->mu_base, mat_to = (ctl->mu_base + 1) % 3;
                const int32_t mat_c = ctl->mu_role[0], mat_lse = ctl->mom_role[1];
                const double mat_a = ctl->a_def;
                double tau_old[VILMA_MAX_P];
                for (int p = 0; p < VILMA_MAX_P; ++p) tau_old[p] = ctl->tau[p];
                for (int p = 0; p < P; ++p) {
                    ctl->tau[p] = tau[p];
                    ctl->hrl[p] = det_hrl(a.ranks[p], tau[p]);
                }
                // the re-evaluation takes the evaluation's moments as its reference and writes the
                // other slot, which the trial behind it treats as current
                const int32_t m0 = ctl->mom_role[0];
                ctl->mom_role[0] = ctl->mom_role[1];
                ctl->mom_role[1] = m0;
                const bool write_out = a.persist && !ctl->c_zero;
                if (write_out) {
                    ctl->mu_base = mat_to;
                    ctl->a_def = 1.0;
                    ctl->c_zero = 1;
                }
                ctl->run_mat = write_out ? 1 : 0;
                ctl->tau_hot = 1;
                decide_set_phases(a, ctl);
                if (write_out) {
                    PhasePtrs &sp = ctl->phase[VILMA_PHASE_SUMS];
                    sp.mu_in = a.bases.mu[mat_from];
                    sp.mu_mat = a.bases.mu[mat_to];
                    sp.c_pend = a.bases.c[mat_c];
                    sp.a_pend = mat_a;
                    sp.lse_ref = a.bases.lse[mat_lse];
                    for (int p = 0; p < VILMA_MAX_P; ++p) sp.tau[p] = tau_old[p];
                }
                ctl->run_eval2 = 1;
                ctl->eval_pending = 2;
                rep.outcome = VILMA_OUT_TAU_UPDATED;
                end_now = false;
            }
        }
        if (end_now && !dead) {
            // the sweep is over: _optimize_step's running ELBO change (:403-409)
            rep.sweep_change = ctl->delta_sum;
            double r = ctl->running_none ? rep.sweep_change : ctl->running;
            r = r * 0.5;
            r = r + 0.5 * (rep.sweep_change > 0.0 ? rep.sweep_change : 0.0);
            ctl->running = r;
            ctl->running_none = 0;
            ctl->delta_sum = 0.0;
            ctl->inner_it = 0;
            rep.sweep_end = 1;
            if (a.two_snapshots) {
                ctl->snap_cur ^= 1;     // the means the sweep's last evaluation wrote
                decide_set_phases(a, ctl);
            }
            rep.end_L0 = ctl->L0; rep.end_running = r; rep.end_snap_cur = ctl->snap_cur;
            rep.end_a_def = ctl->a_def; rep.end_c_zero = ctl->c_zero; rep.end_mu_base = ctl->mu_base;
            for (int p = 0; p < VILMA_MAX_P; ++p) rep.end_tau[p] = ctl->tau[p];
            for (int q = 0; q < 3; ++q) {
                rep.end_mu_role[q] = ctl->mu_role[q];
                rep.end_mom_role[q] = ctl->mom_role[q];
            }
            if (a.mode == VILMA_DECIDE_EVAL) rep.outcome = VILMA_OUT_SWEEP_END;
            // optimize()'s "no posterior mean moved" stop: nothing queued behind may run
            if (a.check_convergence && results[a.o_dsum] == 0.0) dead = true;
        }
    }
    if (a.mode == VILMA_DECIDE_EVAL && rep.outcome != VILMA_OUT_TAU_UPDATED) {
        ctl->run_eval2 = 0;
        ctl->run_mat = 0;
        if (rep.consumed == 1) ctl->tau_hot = 0;        // this sweep's evaluation left tau alone
    }
    if (a.mode == VILMA_DECIDE_TRIAL && !dead && a.debug_kill_deferred > 0 && !ctl->c_zero) {
        ctl->dbg_deferred += 1;
        if (ctl->dbg_deferred == a.debug_kill_deferred) dead = true;
    }
    // ---- (2) the line search on the trial's candidates
    if (a.mode == VILMA_DECIDE_TRIAL && !dead) {
        ctl->run_eval2 = 0;
        ctl->run_mat = 0;
        ctl->run_sums = 0;
        int32_t lazy_uc = 0, lazy_ua = 0, lazy_macc = 0;
        int32_t store_to = -1, store_from = 0;          // (persistent lazy state, tau_hot: see below)
        double lazy_a = 1.0;
        bool lazy_materialise = false;
        rep.orig = ctl->cur_obj;
        rep.fa = det_objective(P, a.chi, ctl->tau, ctl->hrl, results + a.o_ta);
        rep.fb = a.have_b ? det_objective(P, a.chi, ctl->tau, ctl->hrl, results + a.o_tb) : 0.0;
        const double L = ctl->L_try;
        rep.L_tried = L;
        int choice = 0;
        double Lacc = L, fresh = rep.fa;
        const double bar = (rep.orig - a.rel_tol * decide_abs(rep.orig)) - a.abs_tol;
        if (rep.fa >= bar) {
            choice = 1;
        } else if (a.have_b && !(L > a.l_max)) {
            if (rep.fb >= bar) {
                choice = 2;
                Lacc = L * a.rate;
                fresh = rep.fb;
            }
        }
        if (choice == 0) {
            // every candidate rejected: on to the next larger L -- unless the search has to give
            // up beyond L_MAX, which is the host's business (:795-800)
            const double L_last = a.have_b ? L * a.rate : L;
            if (L > a.l_max || L_last > a.l_max) {
                dead = true;
            } else {
                ctl->L_try = L_last * a.rate;
                ctl->run_eval = 0;
                decide_set_phases(a, ctl);
                rep.outcome = VILMA_OUT_REJECTED;
            }
        } else if (Lacc > a.l_max) {
            dead = true;
        } else {
            const double dlt = fresh - rep.orig;
            const bool ends = ctl->running_none || decide_abs(dlt) <= 0.1 * ctl->running ||
                              Lacc == 1.0 || ctl->inner_it + 1 >= a.max_inner;
            if (ends && choice == 2 && a.mstep_inside && !a.have_sums_b) {
                dead = true;
            } else {
                ctl->delta_sum = ctl->delta_sum + dlt;
                ctl->inner_it += 1;
                ctl->cur_obj = fresh;
                ctl->L0 = Lacc;
                // vilma_accept's bookkeeping: the accepted candidate's vi_mu becomes current
                const int32_t mc = ctl->mom_role[0], ma = ctl->mom_role[1], mb = ctl->mom_role[2];
                const int32_t uc = ctl->mu_role[0], ua = ctl->mu_role[1], ub = ctl->mu_role[2];
                // (lazy trials stored nothing: the sums pass behind this decision writes the
                // accepted candidate, A or B alike, into role ua)
                const double step_acc = choice == 1 ? ctl->phase[VILMA_PHASE_TRIAL].step
                                                    : ctl->phase[VILMA_PHASE_TRIAL].step2;
                const int32_t macc = choice == 1 ? mc : mb;        // the accepted candidate's moments
                // lazy trials: the accepted state is a' (stored vi_mu) + Sig c', a' = (1 - step) a, its c'
                // beside its moments (PhasePtrs).  While the beta loop goes on that is all that moves;
                // when it ends the sums pass behind this decision writes the state out (role ua).
                const double a_acc = (a.lazy ? ctl->a_def : 1.0) * (1.0 - step_acc);
                if (a.persist) {
                    // the state stays (stored vi_mu, a, c) whether the loop ends or not; the c buffers
                    // change roles as stored candidates would (A's is in role 1, B's in role 2)
                    ctl->a_def = a_acc;
                    ctl->c_zero = 0;
                    if (choice == 1) { ctl->mu_role[0] = ua; ctl->mu_role[1] = uc; ctl->mu_role[2] = ub; }
                    else { ctl->mu_role[0] = ub; ctl->mu_role[1] = ua; ctl->mu_role[2] = uc; }
                    if (ends && a.scale_se && ctl->tau_hot && !a.mstep_inside) {
                        // tau moved in the last sweep and will probably move in this one: the sums pass
                        // behind this decision writes the state out (into the vi_mu buffer next to the
                        // stored one), so that the update finds an array
                        store_from = ctl->mu_base;
                        store_to = (ctl->mu_base + 1) % 3;
                        ctl->mu_base = store_to;
                        ctl->a_def = 1.0;
                        ctl->c_zero = 1;
                    }
                } else if (a.lazy && !ends) {
                    ctl->a_def = a_acc;
                    ctl->c_zero = 0;
                } else if (choice == 1 || a.lazy) {
                    ctl->mu_role[0] = ua; ctl->mu_role[1] = uc; ctl->mu_role[2] = ub;
                } else {
                    ctl->mu_role[0] = ub; ctl->mu_role[1] = ua; ctl->mu_role[2] = uc;
                }
                if (a.lazy && ends && !a.persist) {
                    ctl->a_def = 1.0;
                    ctl->c_zero = 1;
                }
                // the trial wrote candidate A's moments to slot mc and B's to mb (ma held the state
                // it started from).  An evaluation next: it takes the accepted candidate's as its
                // reference (role 0) and writes role 1.  A trial next (inner loop): it treats
                // role 1 as current.
                if (ends) {
                    if (choice == 1) { ctl->mom_role[0] = mc; ctl->mom_role[1] = ma; ctl->mom_role[2] = mb; }
                    else { ctl->mom_role[0] = mb; ctl->mom_role[1] = mc; ctl->mom_role[2] = ma; }
                    ctl->run_eval = 1;
                    ctl->eval_pending = 1;
                    rep.mstep = a.mstep_inside ? choice : 0;
                    rep.outcome = VILMA_OUT_ACCEPT_MSTEP;
                } else {
                    if (choice == 1) { ctl->mom_role[0] = ma; ctl->mom_role[1] = mc; ctl->mom_role[2] = mb; }
                    else { ctl->mom_role[0] = mc; ctl->mom_role[1] = mb; ctl->mom_role[2] = ma; }
                    ctl->run_eval = 0;
                    rep.outcome = VILMA_OUT_ACCEPT_CONTINUE;
                }
                ctl->run_sums = (!a.mstep_inside && ends) ? 1 : 0;
                lazy_uc = uc; lazy_ua = ua; lazy_macc = macc; lazy_a = a_acc;
                lazy_materialise = a.lazy != 0 && ends;
                double Lnext = Lacc / 1.25;
                Lnext = Lnext > 1.0 ? Lnext : 1.0;
                ctl->L_try = Lnext;
                decide_set_phases(a, ctl);
                if (lazy_materialise) {
                    // the sums pass derives the accepted state from the stored vi_mu the beta loop
                    // started from, (a', c'), and writes it where the roles above already say the
                    // current vi_mu is
                    PhasePtrs &sp = ctl->phase[VILMA_PHASE_SUMS];
                    sp.mu_in = a.bases.mu[lazy_uc];
                    sp.mu_mat = a.bases.mu[lazy_ua];
                    sp.c_pend = a.bases.c[lazy_macc];
                    sp.a_pend = lazy_a;
                    sp.lse_ref = a.bases.lse[lazy_macc];
                    if (a.persist) {
                        // ... and writes nothing: the pass only forms the sums of the accepted state
                        // (unless tau is on the move, see above)
                        sp.mu_in = a.bases.mu[store_to >= 0 ? store_from : ctl->mu_base];
                        sp.mu_mat = store_to >= 0 ? a.bases.mu[store_to] : nullptr;
                        sp.c_pend = a.bases.c[ctl->mu_role[0]];
                    }
                }
            }
        }
        ctl->choice = (rep.outcome == VILMA_OUT_ACCEPT_MSTEP ||
                       rep.outcome == VILMA_OUT_ACCEPT_CONTINUE) ? choice : 0;
    }
    if (dead) {
        ctl->alive = 0;
        ctl->run_eval = 0;
        ctl->run_eval2 = 0;
        ctl->run_mat = 0;
        ctl->run_sums = 0;
        rep.outcome = VILMA_OUT_DEAD;
        rep.mstep = 0;
    }
    ctl->stage += 1;
}

// the block's scalars after a decision, as they go behind the result vector in its snapshot
static __host__ __device__ inline void decide_snapshot_scalars(const SweepDecideArgs &a,
                                                               const SweepCtl *ctl,
                                                               const DecideReport &rep, double *x) {
    for (int q = 0; q < VILMA_SNAP_EXTRA; ++q) x[q] = 0.0;
    x[SNAP_ALIVE] = (double)ctl->alive; x[SNAP_KIND] = (double)a.mode; x[SNAP_OUTCOME] = (double)rep.outcome;
    x[SNAP_STAGE] = (double)ctl->stage; x[SNAP_CHOICE] = (double)ctl->choice;
    x[SNAP_L_TRY] = ctl->L_try; x[SNAP_L0] = ctl->L0; x[SNAP_CUR_OBJ] = ctl->cur_obj;
    x[SNAP_DELTA_SUM] = ctl->delta_sum; x[SNAP_RUNNING] = ctl->running;
    x[SNAP_RUNNING_NONE] = (double)ctl->running_none; x[SNAP_INNER_IT] = (double)ctl->inner_it;
    x[SNAP_ORIG] = rep.orig; x[SNAP_FA] = rep.fa; x[SNAP_FB] = rep.fb; x[SNAP_EVAL_OBJ] = rep.eval_obj;
    x[SNAP_CONSUMED] = (double)rep.consumed; x[SNAP_SWEEP_END] = (double)rep.sweep_end;
    x[SNAP_SWEEP_CHANGE] = rep.sweep_change; x[SNAP_L_TRIED] = rep.L_tried;
    x[SNAP_SNAP_CUR] = (double)ctl->snap_cur; x[SNAP_RUN_EVAL] = (double)ctl->run_eval;
    x[SNAP_RUN_EVAL2] = (double)ctl->run_eval2; x[SNAP_EVAL_PENDING] = (double)ctl->eval_pending;
    x[SNAP_RUN_SUMS] = (double)ctl->run_sums;
    for (int q = 0; q < 3; ++q) {
        x[SNAP_MU_ROLE + q] = (double)ctl->mu_role[q];
        x[SNAP_MOM_ROLE + q] = (double)ctl->mom_role[q];
    }
    for (int p = 0; p < VILMA_MAX_P; ++p) { x[SNAP_TAU + p] = ctl->tau[p]; x[SNAP_HRL + p] = ctl->hrl[p]; }
    x[SNAP_A_DEF] = ctl->a_def; x[SNAP_C_ZERO] = (double)ctl->c_zero;
    x[SNAP_MU_BASE] = (double)ctl->mu_base; x[SNAP_TAU_HOT] = (double)ctl->tau_hot;
    x[SNAP_RUN_MAT] = (double)ctl->run_mat;
}
