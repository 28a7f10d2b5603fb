// The sweep state machine behind the C-ABI: vilma_sweep & co. of include/vilma_hip.h.
//
// Control flow of the reference's outer iteration -- _optimize_step / _nat_grad_step /
// _update_beta / _update_hyper_delta / _update_error_scaling (reference
// variational_inference.py:396-450, 762-802, 825-860, 472-486) -- on top of the evaluation entry
// points of capi.hip.  Decisions are taken from the 3P+2 all-reduced sums per candidate point
// exactly as the reference takes them from its objectives (same tests, same order), so every
// rank walks the same accept / reject sequence and the L trajectory is the reference's.
#include "ctx.h"
#include "decide.h"

#include <deque>
#include <dlfcn.h>

namespace {

constexpr double L_MAX = 1e12;          // reference variational_inference.py:18-24
constexpr double REL_TOL = 1e-6;
constexpr double ABS_TOL = 1e-6;
constexpr double EM_TOL = 10.0;
constexpr double ELBO_MOMENTUM = 0.5;
constexpr int MAX_NUM_ITERS = 20;
constexpr double EPSILON = 1e-100;      // reference numerics.py:8

// ---------------------------------------------------------------------------------------------
// RCCL, bound at run time: the librccl already in the process (torch ships one) or the system's
// ---------------------------------------------------------------------------------------------
struct Id128 { char bytes[128]; };      // ncclUniqueId, passed by value
struct Rccl {
    void *handle = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, Id128, int) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string error;
};
constexpr int NCCL_FLOAT64 = 8;         // ncclDataType_t ncclFloat64 / ncclDouble
constexpr int NCCL_SUM = 0, NCCL_MAX = 2;   // ncclRedOp_t

Rccl &rccl() {
    static Rccl r;
    return r;
}

bool rccl_load() {
    Rccl &r = rccl();
    if (r.handle) return true;
    const char *defaults[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1",
                              "/opt/rocm/lib/librccl.so"};
    // VILMA_RCCL_LIB names the one library to bind instead (a site's own build; tests point it
    // at a file that does not exist to walk the "no RCCL on this rank" path)
    const char *only = std::getenv("VILMA_RCCL_LIB");
    std::vector<const char *> names;
    if (only && only[0]) names.push_back(only);
    else names.assign(defaults, defaults + 4);
    // the copy already mapped into the process first (two RCCL runtimes in one process would each
    // keep their own device state)
    void *h = nullptr;
    for (const char *n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
    if (!h)
        for (const char *n : names)
            if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) {
        const char *e = dlerror();          // one call: it returns the message and clears it
        r.error = std::string("cannot load librccl.so: ") + (e ? e : "not found");
        return false;
    }
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
    r.AllReduce = (decltype(r.AllReduce))dlsym(h, "ncclAllReduce");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.AllReduce || !r.CommDestroy || !r.GetErrorString) {
        r.error = "librccl.so lacks the ncclCommInitRank / ncclAllReduce entry points";
        r.GetUniqueId = nullptr; r.CommInitRank = nullptr; r.AllReduce = nullptr;
        r.CommDestroy = nullptr; r.GetErrorString = nullptr;
        (void)dlclose(h);
        return false;
    }
    r.handle = h;
    return true;
}

}  // namespace

// host-side bookkeeping of the sweep loop for one context
struct SweepState {
    int P = 0, M = 0, A = 0, nt = 0, am = 0;
    // result vector layout (doubles): see include/vilma_hip.h
    int o_dsum = 0, o_tot = 0, o_ta = 0, o_tb = 0, o_sa = 0, o_sb = 0, o_dmax = 0, o_hyper = 0;
    int size = 0, reduce_end = 0;
    double *results = nullptr;              // device
    std::vector<double> host;               // last download of the result vector

    std::vector<double> chi, ranks;
    bool have_consts = false, scale_se = false;

    // collective
    int comm_kind = 0, world = 1, rank = 0; // 0 none, 1 RCCL, 2 callback
    void *nccl_comm = nullptr;
    vilma_allreduce_fn cb = nullptr;
    void *cb_user = nullptr;

    // the current (accepted) state as the host knows it
    bool have_state = false;
    std::vector<double> hyper, totals;
    double objective = 0.0;
    int cur_sums = -1;                      // offset in the result vector of the all-reduced
                                            // responsibility sums of the current state, or -1
    // the last beta trial's candidates
    bool trial_sums = false;                // o_sa holds candidate A's all-reduced sums
    bool trial_sums_b = false;              // o_sb holds candidate B's
    int candidate = 1;                      // which candidate the objective just looked at belongs to
    bool alt_valid = false;                 // candidate B evaluated, not looked at yet
    double alt_step = 0.0, alt_obj = 0.0;
    std::vector<double> alt_totals;
    bool two_step = true;                   // VILMA_TWO_STEP / default P <= 2
    double L_rejected = -1.0;

    // a beta trial evaluated on the device but not looked at by the host's line search yet
    // (left by a queued sweep whose decision went to the host): candidate A's sums are in `host`
    bool pend_valid = false;
    double pend_step = 0.0;

    // ---- sweeps queued ahead (see "Sweeps queued ahead" below): groups of [trial, decision,
    // evaluation (, decision, re-evaluation)] run from the control block on the device; the host
    // reads one snapshot per decision and replays it on its mirror of the block
    bool armed = false;                     // the control block drives the stream
    hipStream_t pipe_stream = nullptr;      // ... this one
    bool pipe_diff = false;                 // queued evaluations carry the convergence statistics
    double pipe_rate = 2.0;
    static constexpr int RING = 8;          // snapshot buffers (at most 2 groups = 4 decisions are outstanding)
    double *land[RING] = {};                // host memory the decision kernels write
    int next_land = 0;
    double serial = 0.0;
    struct Queued {                         // a decision queued, its snapshot not looked at yet
        int land;                           // snapshot buffer
        double serial;                      // the serial number that completes it
        SweepDecideArgs args;               // what the kernel was launched with (the mirror replays it)
        int64_t tag;                        // serial number of its group (profiling brackets: 4 tag + phase)
        bool first_of_group;                // a TRIAL decision: one group of kernels sits in front of it
    };
    std::deque<Queued> outq;
    int groups_out = 0;                     // trial groups whose TRIAL decision is in outq
    int64_t group_serial = 0;
    bool fresh = false;                     // armed, no group queued yet
    SweepCtl mirror;                        // the control block as of the last decision looked at
    SweepCtl *ctl_host = nullptr;           // pinned staging of the control block
    // the state at the end of the last sweep reported from the device (what a caller who breaks
    // the LOOKAHEAD promise gets back): roles, tau, snapshot buffer
    struct Reported { int32_t mu_role[3], mom_role[3], snap_cur; double tau[VILMA_MAX_P]; bool valid = false;
                      double a_def = 1.0; int32_t c_zero = 1, mu_base = 0; } rep_end;   // (a_def, c_zero, mu_base: persistent lazy state)
    bool armed_pure = false; // the block was armed with a == 0 (pipeline_arm): without --learn-scaling a stays 0 until it is disarmed
    int last_form = 0;      // form of the state behind the last decision looked at: 0 a stored vi_mu, 1 (stored vi_mu, a, c),
                            // 2 the same with a == 0 (no pass reads vi_mu) -- vilma_prof_state_form
    // statistics of decisions looked at in one call that belong to the next sweep
    int carry_trials = 0, carry_evals = 0, carry_products = 0;
    // the host's line search resumes a sweep the device began (see Resume)
    struct Resume {
        bool active = false;
        int it0 = 0;                        // beta updates the device had accepted in this sweep
        double delta_sum = 0.0;             // ... and what they (and, after_mstep, the M-step) gained
        bool after_mstep = false;           // the M-step's evaluation has been looked at already
        double L_try = 0.0;                 // L[0] of the trial waiting in the pend cache
    } resume;

    // per-call
    int flags = 0;
    vilma_sweep_stats *stats = nullptr;
    bool have_diff = false;
    double lsr = 2.0;
};

void vilma_detail::sweep_destroy(vilma_ctx *c) {
    SweepState *s = c->sw;
    if (!s) return;
    if (s->nccl_comm && rccl().CommDestroy) (void)rccl().CommDestroy(s->nccl_comm);
    for (int b = 0; b < SweepState::RING; ++b)
        if (s->land[b]) (void)hipHostFree(s->land[b]);
    if (s->ctl_host) (void)hipHostFree(s->ctl_host);
    dev_free(c->ctl);
    c->ctl = nullptr;
    dev_free(s->results);
    delete s;
    c->sw = nullptr;
}

namespace {

SweepState *sweep_state(vilma_ctx *c) {
    if (c->sw) return c->sw;
    SweepState *s = new SweepState();
    s->P = c->P; s->M = c->M; s->A = c->A;
    s->nt = VILMA_NTOTALS(c->P);
    s->am = c->A * c->M;
    s->o_dsum = 0;
    s->o_tot = 3;
    s->o_ta = s->o_tot + s->nt;
    s->o_tb = s->o_ta + s->nt;
    s->o_sa = s->o_tb + s->nt;
    s->o_sb = s->o_sa + s->am;
    s->reduce_end = s->o_sb + s->am;
    s->o_dmax = s->reduce_end;
    s->o_hyper = s->o_dmax + 3;
    s->size = s->o_hyper + s->am;
    if (hipMalloc((void **)&s->results, (size_t)s->size * sizeof(double)) != hipSuccess ||
        hipMemset(s->results, 0, (size_t)s->size * sizeof(double)) != hipSuccess) {
        delete s;
        return nullptr;
    }
    s->host.assign(s->size, 0.0);
    s->totals.assign(s->nt, 0.0);
    s->alt_totals.assign(s->nt, 0.0);
    const char *two = std::getenv("VILMA_TWO_STEP");
    // Two steps per beta trial pay where the LD product dominates a trial (P <= 2); at P = 4 the
    // second candidate's per-SNP work outweighs the saved products (profiles/r02h_ab_twostep.txt)
    // (VILMA_TWO_STEP=1 is honoured up to four cohorts: beyond that the kernels evaluate one candidate)
    s->two_step = (two && (two[0] == '0' || two[0] == '1')) ? (two[0] == '1' && c->P <= 4) : c->P <= 2;
    c->sw = s;
    return s;
}

#define SW(c)                                                          \
    SweepState *s = sweep_state(c);                                    \
    if (!s) return fail((c), "cannot allocate the result vector")

int comm_allreduce(vilma_ctx *c, SweepState *s, hipStream_t st, double *buf, int64_t n, int op) {
    if (n <= 0) return 0;
    if (s->comm_kind == 1) {
        const int rc = rccl().AllReduce(buf, buf, (size_t)n, NCCL_FLOAT64, op ? NCCL_MAX : NCCL_SUM,
                                        s->nccl_comm, st);
        if (rc != 0) return fail(c, std::string("ncclAllReduce: ") + rccl().GetErrorString(rc));
    } else if (s->comm_kind == 2) {
        if (s->cb(s->cb_user, (void *)st, buf, n, op))
            return fail(c, "the all-reduce callback failed");
    }
    return 0;
}

// fast_likelihood (numerics.py:31-46) minus _beta_KL (variational_inference.py:873-885) from the
// all-reduced sums: the very function the device decisions use (detmath.h; no fused multiply-add,
// a logarithm that gives the same bits on both sides)
double objective_from(const vilma_ctx *c, const SweepState *s, const double *t) {
    double hrl[VILMA_MAX_P];
    for (int p = 0; p < s->P; ++p) hrl[p] = det_hrl(s->ranks[p], c->tau[p]);
    return det_objective(s->P, s->chi.data(), c->tau, hrl, t);
}

bool is_close(double a, double b) {     // numpy.isclose defaults (rtol 1e-5, atol 1e-8)
    return std::fabs(a - b) <= 1e-8 + 1e-5 * std::fabs(b);
}

void event(SweepState *s, int kind, int paramset, double a, double b) {
    vilma_sweep_stats *st = s->stats;
    if (!st || !(s->flags & VILMA_SWEEP_VERBOSE) || st->n_events >= VILMA_SWEEP_EVENTS) return;
    st->events[st->n_events].kind = kind;
    st->events[st->n_events].paramset = paramset;
    st->events[st->n_events].a = a;
    st->events[st->n_events].b = b;
    st->n_events += 1;
}

// all-reduce [lo, hi) of the result vector and download the whole vector
int reduce_and_fetch(vilma_ctx *c, SweepState *s, hipStream_t st, int lo, int hi, bool with_max) {
    if (s->comm_kind) {
        if (comm_allreduce(c, s, st, s->results + lo, hi - lo, 0)) return 1;
        if (with_max && comm_allreduce(c, s, st, s->results + s->o_dmax, 3, 1)) return 1;
    }
    return vilma_fetch(c, (void *)st, s->results, s->host.data(), s->size);
}

// Objective of the CURRENT vi_mu under the current hyper / tau; the evaluated point stays on the
// device as the trial state (vilma_accept(ctx, 0) makes it current).
int evaluate_current(vilma_ctx *c, SweepState *s, hipStream_t st, double *obj) {
    if (vilma_eval(c, (void *)st, s->results + s->o_tot)) return 1;
    if (reduce_and_fetch(c, s, st, s->o_tot, s->o_tot + s->nt, false)) return 1;
    s->trial_sums = s->trial_sums_b = false;
    s->alt_valid = false;
    if (s->stats) { s->stats->n_evaluations += 1; s->stats->n_products += 1; }
    *obj = objective_from(c, s, s->host.data() + s->o_tot);
    return 0;
}

int accept(vilma_ctx *c, SweepState *s, int take, double obj, const double *totals) {
    s->alt_valid = false;
    if (vilma_accept(c, take)) return 1;
    s->objective = obj;
    std::copy(totals, totals + s->nt, s->totals.begin());
    // responsibility sums fetched with the candidate now describe the current state
    s->cur_sums = (take == 1 && s->trial_sums) ? s->o_sa : (take == 2 && s->trial_sums_b) ? s->o_sb : -1;
    s->trial_sums = s->trial_sums_b = false;
    s->have_state = true;
    return 0;
}

// Objective of the candidate at `step`: the second candidate of the pair evaluated last if that is
// this step, else a new trial (with the step the search would try next riding along).
int trial(vilma_ctx *c, SweepState *s, hipStream_t st, double step, double next_step, double *obj,
          const double **totals) {
    if (s->alt_valid && s->alt_step == step) {
        s->alt_valid = false;
        s->candidate = 2;
        if (s->stats) { s->stats->n_evaluations += 1; s->stats->n_trials += 1; }
        *obj = s->alt_obj;
        *totals = s->alt_totals.data();
        return 0;
    }
    if (s->pend_valid && s->pend_step == step) {
        // the trial a queued sweep already ran for this very step (its decision came back to the
        // host): candidate A's sums are in s->host, B's in the alt cache
        s->pend_valid = false;
        s->candidate = 1;
        if (s->stats) { s->stats->n_evaluations += 1; s->stats->n_trials += 1; s->stats->n_products += 1; }
        *obj = objective_from(c, s, s->host.data() + s->o_ta);
        *totals = s->host.data() + s->o_ta;
        return 0;
    }
    s->pend_valid = false;
    s->alt_valid = false;
    s->cur_sums = -1;                   // the trial's sums overwrite the device copy
    const bool two = s->two_step;
    if (c->poison) launch_poison(s->results + s->o_ta, s->o_sb + s->am - s->o_ta, 0, st);
    if (two) {
        if (vilma_trial_beta2(c, (void *)st, step, next_step, s->results + s->o_ta, s->results + s->o_tb))
            return 1;
    } else if (vilma_trial_beta(c, (void *)st, step, s->results + s->o_ta)) {
        return 1;
    }
    // the M-step statistic of the candidates: from the per-tile sums the trial's own per-SNP pass
    // left behind (both candidates, no second pass over vi_mu) when it stashed them, else by
    // delta_kernel for candidate A
    const int avail = vilma_trial_sums_available(c);
    const bool sums_b = two && avail == 2;
    if (avail >= 1) {
        if (vilma_trial_sums(c, (void *)st, s->results + s->o_sa, sums_b ? s->results + s->o_sb : nullptr))
            return 1;
    } else if (vilma_delta_sums(c, (void *)st, s->results + s->o_sa, VILMA_STATE_TRIAL_BETA)) {
        return 1;
    }
    if (reduce_and_fetch(c, s, st, s->o_ta, sums_b ? s->o_sb + s->am : s->o_sa + s->am, false)) return 1;
    s->candidate = 1;
    s->trial_sums = true;
    s->trial_sums_b = sums_b;
    if (two) {
        s->alt_valid = true;
        s->alt_step = next_step;
        std::copy(s->host.begin() + s->o_tb, s->host.begin() + s->o_tb + s->nt, s->alt_totals.begin());
        s->alt_obj = objective_from(c, s, s->alt_totals.data());
    }
    if (s->stats) { s->stats->n_evaluations += 1; s->stats->n_trials += 1; s->stats->n_products += 1; }
    *obj = objective_from(c, s, s->host.data() + s->o_ta);
    *totals = s->host.data() + s->o_ta;
    return 0;
}

// One damped natural-gradient step with backtracking (variational_inference.py:762-802).
int update_beta(vilma_ctx *c, SweepState *s, hipStream_t st, double *L, double orig, double *new_obj) {
    for (;;) {
        double obj;
        const double *totals;
        if (trial(c, s, st, 1.0 / L[0], 1.0 / (L[0] * s->lsr), &obj, &totals)) return 1;
        event(s, 1, 0, orig, obj);
        const bool accepted = obj >= orig - REL_TOL * std::fabs(orig) - ABS_TOL;
        if (accepted) {
            if (L[0] > L_MAX && !is_close(orig, obj)) return fail(c, "Encountered a numerical error.");
            // `totals` may point into s->host, which accept() does not touch
            if (accept(c, s, s->candidate, obj, totals)) return 1;
            *new_obj = obj;
            return 0;
        }
        if (L[0] > L_MAX) {
            if (!is_close(orig, obj)) return fail(c, "Encountered a numerical error.");
            *new_obj = orig;
            return 0;
        }
        s->L_rejected = L[0];
        L[0] *= s->lsr;
    }
}

// Closed-form M-step for the mixture weights (variational_inference.py:825-860): responsibility
// sums (already all-reduced when they came with the accepted beta trial) -> hyper_delta and its
// table on the device -> re-evaluation.  Unconditional in the reference, so accepted at once.
int update_hyper(vilma_ctx *c, SweepState *s, hipStream_t st, bool with_diff, double *new_obj) {
    if (s->cur_sums < 0) {
        // no accepted beta step since the last evaluation (or a candidate without sums was taken):
        // the statistic of the current state is computed now
        if (vilma_delta_sums(c, (void *)st, s->results + s->o_sa, VILMA_STATE_CURRENT)) return 1;
        if (s->comm_kind && comm_allreduce(c, s, st, s->results + s->o_sa, s->am, 0)) return 1;
        s->cur_sums = s->o_sa;
    }
    if (vilma_mstep(c, (void *)st, s->results + s->cur_sums, s->results + s->o_hyper)) return 1;
    if (with_diff) {
        if (vilma_eval_diff(c, (void *)st, s->results + s->o_tot, s->results + s->o_dsum,
                            s->results + s->o_dmax)) return 1;
    } else if (vilma_eval(c, (void *)st, s->results + s->o_tot)) {
        return 1;
    }
    if (vilma_accept(c, 0)) return 1;
    const int lo = with_diff ? s->o_dsum : s->o_tot;
    if (reduce_and_fetch(c, s, st, lo, s->o_tot + s->nt, with_diff && (s->flags & VILMA_SWEEP_VERBOSE)))
        return 1;
    std::copy(s->host.begin() + s->o_tot, s->host.begin() + s->o_tot + s->nt, s->totals.begin());
    s->hyper.assign(s->host.begin() + s->o_hyper, s->host.begin() + s->o_hyper + s->am);
    if (with_diff && s->stats) {
        for (int q = 0; q < 3; ++q) {
            s->stats->diff_sum[q] = s->host[s->o_dsum + q];
            s->stats->diff_max[q] = s->host[s->o_dmax + q];
        }
        s->have_diff = true;
    }
    const double orig = s->objective;
    s->objective = objective_from(c, s, s->totals.data());
    s->cur_sums = -1;
    s->trial_sums = s->trial_sums_b = false;
    s->alt_valid = false;
    if (s->stats) { s->stats->n_evaluations += 1; s->stats->n_products += 1; }
    event(s, 1, 1, orig, s->objective);
    *new_obj = s->objective;
    return 0;
}

// EM update of the SE scaling (variational_inference.py:472-486, 735-738) from the sums of the
// current state; the sigma-dependent constants follow tau inside the kernels.
int update_error_scaling(vilma_ctx *c, SweepState *s, hipStream_t st, double *new_obj) {
    const int P = s->P;
    double tau[VILMA_MAX_P];
    const double *t = s->totals.data();
    for (int p = 0; p < P; ++p) tau[p] = det_tau(s->chi[p], t[p], t[2 * P + p], t[P + p], s->ranks[p]);
    if (vilma_set_tau(c, tau)) return 1;
    double obj;
    if (evaluate_current(c, s, st, &obj)) return 1;
    std::vector<double> tot(s->host.begin() + s->o_tot, s->host.begin() + s->o_tot + s->nt);
    if (accept(c, s, 0, obj, tot.data())) return 1;
    *new_obj = obj;
    return 0;
}

// variational_inference.py:419-450 with the redundant re-evaluations removed: the objective of the
// state a parameter-set update starts from is the one computed when that state was accepted.
int nat_grad_step(vilma_ctx *c, SweepState *s, hipStream_t st, double *L, double running,
                  double *delta_sum_out) {
    const double conv_tol = std::isnan(running) ? HUGE_VAL : 0.1 * running;
    double delta_sum = 0.0;
    // a sweep the device began and handed over (pipeline_takeover): the beta updates it had
    // accepted, what they gained, and -- unless the M-step is behind us too -- a trial at L[0]
    // already evaluated (the pend cache), whose L must not decay again
    const SweepState::Resume resume = s->resume;
    s->resume = SweepState::Resume();
    if (resume.active) delta_sum = resume.delta_sum;
    double orig = s->objective;
    double nw;
    if (!(resume.active && resume.after_mstep)) {
        // ---- paramset 0: variational family of beta
        const int it0 = resume.active ? resume.it0 : 0;
        for (int it = it0; it < MAX_NUM_ITERS; ++it) {
            if (!(resume.active && it == it0)) L[0] = std::max(1.0, L[0] / 1.25);
            event(s, 0, 0, L[0], 0.0);
            if (update_beta(c, s, st, L, orig, &nw)) return 1;
            delta_sum += nw - orig;
            // == np.isclose(new - orig, 0, atol=conv_tol, rtol=0) for finite objectives
            if (std::fabs(nw - orig) <= conv_tol || L[0] == 1.0 || L[0] > L_MAX) break;
            orig = nw;
        }
        // ---- paramset 1: mixture weights (L[1] stays 1: exactly one pass)
        L[1] = std::max(1.0, L[1] / 1.25);
        event(s, 0, 1, L[1], 0.0);
        // without --learn-scaling this is the sweep's last evaluation: the convergence statistics
        // ride in its per-SNP pass
        const bool last = !s->scale_se;
        orig = s->objective;
        if (update_hyper(c, s, st, (s->flags & VILMA_SWEEP_DIFF) && last, &nw)) return 1;
        delta_sum += nw - orig;
    }
    // ---- paramset 2: annotations -- nothing to do in this scheme (:862-866)
    L[2] = std::max(1.0, L[2] / 1.25);
    event(s, 0, 2, L[2], 0.0);
    if (s->scale_se && delta_sum < EM_TOL) {
        orig = s->objective;
        if (update_error_scaling(c, s, st, &nw)) return 1;
        delta_sum += nw - orig;
        event(s, 2, 0, orig, nw);
    }
    *delta_sum_out = delta_sum;
    return 0;
}


// ---------------------------------------------------------------------------------------------
// Sweeps queued ahead.  With VILMA_SWEEP_LOOKAHEAD the reference's loop (variational_inference.py:
// 419-450) is walked by the DEVICE: the host queues groups of launches
//     [beta trial (two candidates) -> all-reduce -> TRIAL decision -> evaluation
//      (-> all-reduce -> EVAL decision -> re-evaluation, with --learn-scaling)]
// and sweep_decide_kernel (kernels.hip, decide.h) decides from the control block what each group
// is: the next step of the line search after a rejection, the next update of the inner beta loop,
// or the first trial of the next sweep behind the M-step's evaluation.  Buffer roles, step sizes,
// L, the running ELBO change, delta_sum, tau all live in the control block (SweepCtl); kernels
// whose phase does not happen (the evaluation behind a trial that did not end the beta loop, the
// re-evaluation when tau stays) read a flag there and exit at once.  The host reads ONE snapshot
// per decision -- the decision kernel writes it straight into mapped host memory -- and replays
// the decision with the same source code (decide_core) on its mirror of the block; it raises if
// the two ever differ.  What the device will not decide (L beyond L_MAX, a non-positive tau, the
// convergence veto) turns the block dead and the host's own line search (above) resumes from
// where the device stood.
//
// Invariant: when vilma_sweep returns, at most ONE trial beyond the reported state has been queued,
// so the reported state's vi_mu is still in its buffer and a caller who breaks the promise
// (vilma_sweep_drain, any state read) gets exactly that state back.
// ---------------------------------------------------------------------------------------------
bool lookahead_enabled() {
    const char *e = std::getenv("VILMA_LOOKAHEAD");
    return !(e && e[0] == '0');
}

// can this sweep run from the control block?
bool pipeline_eligible(const vilma_ctx *c, const SweepState *s, int flags) {
    (void)c; (void)s;
    if (!(flags & VILMA_SWEEP_LOOKAHEAD) || (flags & VILMA_SWEEP_VERBOSE)) return false;
    return lookahead_enabled();
}

// the trial's own per-SNP pass leaves every candidate's responsibility sums (LDS stash): the TRIAL
// decision does the M-step itself.  Otherwise (mixtures beyond the stash) a pass over the accepted
// candidate's vi_mu behind the decision forms them, then an M-step kernel.
bool stash_sums(const vilma_ctx *c, const SweepState *s) {
    const char *e = std::getenv("VILMA_PIPE_STASH");        // =0: always the pass behind the decision (A/B)
    if (e && e[0] == '0') return false;
    return c->sum_partials != nullptr && snp_pass_can_stash(c->M, c->P, s->two_step ? 2 : 1);
}

// Without the stash the candidates' responsibility sums need a pass over the accepted candidate
// anyway; the trials then need not store their candidates' vi_mu at all ("lazy": that pass
// re-derives the accepted one and stores it) -- at M = 582 a trial writes 19.6 GB less.
// Round 5, late: mixtures that FIT the stash run lazy trials too (the lazy state persists: below) --
// the trial keeps its stash (the TRIAL decision still does the M-step itself) and stores two [P][N]
// vectors instead of two vi_mu arrays (C3: 1.35 GB per trial; the trial pass 0.37 -> 0.285 ms).  With
// --learn-scaling every tau update then needs the write-out pass (there is no sums pass that could
// store the state while tau keeps moving: decide.h, tau_hot) -- 0.29 ms at C3, which the sweeps in
// which tau moves earn back with their second trial (they run 3 - 16).
bool lazy_with_stash(const vilma_ctx *c, const SweepState *s) {
    const char *e = std::getenv("VILMA_STASH_LAZY");        // =0: trials of a mixture that fits the stash store (A/B)
    if (e && e[0] == '0') return false;
    const char *pe = std::getenv("VILMA_PIPE_PERSIST");
    if (pe && pe[0] == '0') return false;
    return stash_sums(c, s) && c->P <= 4;
}
bool lazy_trials(const vilma_ctx *c, const SweepState *s) {
    const char *e = std::getenv("VILMA_PIPE_LAZY");         // =0: the trials store both candidates (A/B)
    if (e && e[0] == '0') return false;
    return !stash_sums(c, s) || lazy_with_stash(c, s);
}
// ... and then nothing needs vi_mu as an array while the sweeps stay on the device: the state lives
// on as (stored vi_mu, a, c) from sweep to sweep (SweepCtl::mu_base), the sums pass stops storing
// (at M = 582 it wrote 9.8 GB per sweep) and the evaluation behind the M-step derives its state like
// the trials.  With --learn-scaling a tau update changes Sig_k under a state that must stay put: the
// EVAL decision that takes it has the state written out with the old tau (a pass queued behind it,
// beside the re-evaluation) and goes on from that array.  Not beyond four cohorts (the lazy
// evaluation is built for P <= 4).
bool lazy_persist(const vilma_ctx *c, const SweepState *s) {
    const char *e = std::getenv("VILMA_PIPE_PERSIST");      // =0: write vi_mu out at the end of every sweep (A/B)
    if (e && e[0] == '0') return false;
    return lazy_trials(c, s) && c->P <= 4;
}
// A persistent lazy state (k.c_zero == 0: a_def (vi_mu of buffer mu_base) + Sig c, c in buffer c_buf)
// written out for whoever needs vi_mu as an array; the host-side vi_mu roles then name the buffer
// it went to as current.  lse_buf: the state's log-normaliser (for the sums that come with the pass).
int persist_writeback(vilma_ctx *c, SweepState *s, hipStream_t st, int mu_base, int c_buf, int lse_buf,
                      double a_def, const double *tau) {
    const int to = (mu_base + 1) % 3;
    if (materialise_deferred(c, st, mu_base, to, c_buf, lse_buf, a_def, tau, s->results + s->o_sa)) return 1;
    c->pure_c = -1;
    if (a_def == 0.0) {
        // the array just written IS Sig_k c (the pass forms 0 * stored + Sig_k c with the passes' own
        // expressions): with c beside it the next arming starts from the base-free form again
        if (c_buf != to)
            HIPCHK(c, hipMemcpyAsync(c->cvec[to], c->cvec[c_buf], (size_t)c->P * c->N * sizeof(double),
                                     hipMemcpyDeviceToDevice, st));
        c->pure_c = to;
        for (int p = 0; p < VILMA_MAX_P; ++p) c->pure_tau[p] = p < c->P ? tau[p] : 1.0;
    }
    HIPCHK(c, hipStreamSynchronize(st));
    c->mu_cur = to; c->mu_ta = mu_base; c->mu_tb = (mu_base + 2) % 3;
    return 0;
}

int pipeline_buffers(vilma_ctx *c, SweepState *s) {
    if (c->ctl) return 0;
    if (dev_alloc(c, &c->ctl, 1)) return 1;
    HIPCHK(c, hipHostMalloc((void **)&s->ctl_host, sizeof(SweepCtl), hipHostMallocDefault));
    for (int b = 0; b < SweepState::RING; ++b) {
        HIPCHK(c, hipHostMalloc((void **)&s->land[b], (size_t)(s->size + VILMA_SNAP_EXTRA) * sizeof(double),
                                hipHostMallocMapped | hipHostMallocCoherent));
        std::memset(s->land[b], 0, (size_t)(s->size + VILMA_SNAP_EXTRA) * sizeof(double));
    }
    return 0;
}

BufferBases buffer_bases(const vilma_ctx *c) {
    BufferBases b;
    for (int q = 0; q < 3; ++q) {
        b.mu[q] = c->mu[q]; b.pool[q] = c->pool[q]; b.m[q] = c->m[q]; b.v[q] = c->v[q]; b.lse[q] = c->lse[q];
        b.c[q] = c->cvec[q];
    }
    b.snap[0] = c->snap[0]; b.snap[1] = c->snap[1];
    return b;
}

SweepDecideParams decide_params(vilma_ctx *c, SweepState *s, int mode, bool veto, bool allow_tau = true) {
    SweepDecideParams p;
    const bool two = s->two_step, stash = stash_sums(c, s);
    p.mode = mode;
    p.P = c->P; p.A = c->A; p.M = c->M;
    p.check_convergence = veto ? 1 : 0;
    p.have_b = two ? 1 : 0;
    p.have_sums_b = (two && stash) ? 1 : 0;
    p.mstep_inside = stash ? 1 : 0;
    p.lazy = lazy_trials(c, s) ? 1 : 0;
    p.persist = lazy_persist(c, s) ? 1 : 0;
    p.scale_se = (s->scale_se && allow_tau) ? 1 : 0;
    p.two_snapshots = 1;
    p.max_inner = MAX_NUM_ITERS;
    {
        const char *e = std::getenv("VILMA_DEBUG_KILL_DEFERRED");
        p.debug_kill_deferred = e ? std::atoi(e) : 0;
    }
    p.chi = s->chi.data(); p.ranks = s->ranks.data();
    p.rel_tol = REL_TOL; p.abs_tol = ABS_TOL; p.rate = s->pipe_rate; p.l_max = L_MAX; p.em_tol = EM_TOL;
    p.ctl = c->ctl; p.results = s->results;
    p.o_dsum = s->o_dsum; p.o_tot = s->o_tot; p.o_ta = s->o_ta; p.o_tb = s->o_tb;
    p.o_sa = s->o_sa; p.o_sb = s->o_sb; p.o_hyper = s->o_hyper; p.n_results = s->size;
    p.lh = c->lh; p.counts = c->counts; p.log_det = c->log_det;
    p.snap = nullptr; p.serial = 0.0; p.bases = buffer_bases(c);
    return p;
}

// one decision behind whatever has been queued; its snapshot goes to the next buffer of the ring
int queue_decision(vilma_ctx *c, SweepState *s, hipStream_t st, int mode, bool veto,
                   bool first_of_group, int64_t tag, bool allow_tau = true) {
    if ((int)s->outq.size() >= SweepState::RING)
        return fail(c, "internal: more decisions outstanding than snapshot buffers");
    SweepDecideParams p = decide_params(c, s, mode, veto, allow_tau);
    const int b = s->next_land;
    s->next_land = (s->next_land + 1) % SweepState::RING;
    void *dptr = nullptr;
    if (hipHostGetDevicePointer(&dptr, s->land[b], 0) != hipSuccess)
        return fail(c, "the snapshot buffer is not mapped for the device");
    s->serial += 1.0;
    p.snap = (double *)dptr;
    p.serial = s->serial;
    launch_sweep_decide(p, st);
    HIPCHK(c, hipGetLastError());
    SweepState::Queued q;
    q.land = b; q.serial = s->serial; q.args = decide_args(p); q.tag = tag;
    q.first_of_group = first_of_group;
    s->outq.push_back(q);
    return 0;
}

// One group: [trial, all-reduce, TRIAL decision, (sums pass, all-reduce, M-step,) evaluation
// (, all-reduce, EVAL decision, re-evaluation)], all behind the control block.  `fresh`: the first
// group behind an arming -- the evaluation in front of it was the host's, its sums are reduced.
int queue_group(vilma_ctx *c, SweepState *s, hipStream_t st, bool veto, bool fresh) {
    const bool two = s->two_step, stash = stash_sums(c, s);
    // profiling brackets of this group: 4 tag (trial), 4 tag + 1 (M-step, evaluation),
    // 4 tag + 2 (re-evaluation), 4 tag + 3 (sums pass)
    const int64_t tag = ++s->group_serial;
    int rc = 0;
    c->prof_tag = 4 * tag;
    c->lazy_trial = lazy_trials(c, s);
    c->lazy_persist = lazy_persist(c, s);
    c->lazy_stash = c->lazy_trial && stash;
    // (a tau update is the one thing that sets a back to 1 on the device behind the host's back)
    c->lazy_nobase = c->lazy_persist && s->armed_pure && !s->scale_se && c->P <= 2;
    if (const char *e = std::getenv("VILMA_NOBASE_KERNELS")) {    // =0: the run-time form everywhere (A/B)
        if (e[0] == '0') c->lazy_nobase = false;
    }
    set_launch_predicate(&c->ctl->alive);
    if (c->poison) launch_poison(s->results + s->o_ta, s->o_sb + s->am - s->o_ta, 0, st);
    rc = queue_trial_phase(c, st, two, s->results + s->o_ta, s->results + s->o_tb,
                           stash ? s->results + s->o_sa : nullptr,
                           (stash && two) ? s->results + s->o_sb : nullptr);
    set_launch_predicate(nullptr);
    // one all-reduce: the evaluation before this trial (sums and statistics), the trial's sums
    if (!rc && s->comm_kind) {
        const int lo = fresh ? s->o_ta : s->o_dsum;
        const int hi = stash ? (two ? s->o_sb + s->am : s->o_sa + s->am)
                             : (two ? s->o_tb + s->nt : s->o_ta + s->nt);
        rc = comm_allreduce(c, s, st, s->results + lo, hi - lo, 0);
    }
    if (!rc) rc = queue_decision(c, s, st, VILMA_DECIDE_TRIAL, veto, true, tag);
    c->prof_tag = 4 * tag + 1;
    if (!rc && !stash) {
        // (behind a lazy trial the pass runs on every accept: it is what stores the candidate)
        set_launch_predicate(&c->ctl->run_sums);
        c->prof_tag = 4 * tag + 3;
        rc = queue_sums_phase(c, st, s->results + s->o_sa);
        c->prof_tag = 4 * tag + 1;
        set_launch_predicate(nullptr);
        if (!rc && s->comm_kind) rc = comm_allreduce(c, s, st, s->results + s->o_sa, s->am, 0);
        set_launch_predicate(&c->ctl->run_eval);
        if (!rc) rc = queue_mstep(c, st, s->results + s->o_sa, s->results + s->o_hyper);
        set_launch_predicate(nullptr);
    }
    if (!rc) {
        set_launch_predicate(&c->ctl->run_eval);
        rc = queue_eval_phase(c, st, s->results + s->o_tot, s->pipe_diff ? s->results + s->o_dsum : nullptr,
                              s->pipe_diff ? s->results + s->o_dmax : nullptr);
        set_launch_predicate(nullptr);
    }
    if (!rc && s->scale_se) {
        if (s->comm_kind) rc = comm_allreduce(c, s, st, s->results + s->o_dsum, s->o_tot + s->nt - s->o_dsum, 0);
        if (!rc) rc = queue_decision(c, s, st, VILMA_DECIDE_EVAL, false, false, tag);
        c->prof_tag = 4 * tag + 2;
        if (!rc && c->lazy_persist) {
            // a tau update under a persistent lazy state: the state written out with the old tau (the
            // sums that come with the pass land in candidate B's slot, which lazy trials leave unused)
            set_launch_predicate(&c->ctl->run_mat);
            rc = queue_sums_phase(c, st, s->results + s->o_sb, /*writes_state=*/true);
            set_launch_predicate(nullptr);
        }
        if (!rc) {
            set_launch_predicate(&c->ctl->run_eval2);
            rc = queue_eval_phase(c, st, s->results + s->o_tot, s->pipe_diff ? s->results + s->o_dsum : nullptr,
                                  s->pipe_diff ? s->results + s->o_dmax : nullptr);
            set_launch_predicate(nullptr);
        }
    }
    c->prof_tag = 0;
    c->lazy_trial = false;
    c->lazy_persist = false;
    c->lazy_stash = false;
    c->lazy_nobase = false;
    if (!rc) s->groups_out += 1;
    s->pipe_stream = st;
    return rc;
}

// a decision that only looks at the evaluation that has just run (a sweep the caller did not
// promise to follow with another: nothing else would)
int queue_eval_decision(vilma_ctx *c, SweepState *s, hipStream_t st) {
    if (s->comm_kind &&
        comm_allreduce(c, s, st, s->results + s->o_dsum, s->o_tot + s->nt - s->o_dsum, 0)) return 1;
    // (no tau update here: with --learn-scaling the group's own EVAL decision has taken that one)
    return queue_decision(c, s, st, VILMA_DECIDE_EVAL, false, false, 0, /*allow_tau=*/false);
}

// Write the host's state into the control block; nothing is queued yet.
int pipeline_arm(vilma_ctx *c, SweepState *s, hipStream_t st, const double *L, double running,
                 double rate, bool diff) {
    if (pipeline_buffers(c, s)) return 1;
    s->pipe_rate = rate;
    s->pipe_diff = diff;
    s->pipe_stream = st;
    SweepCtl &k = *s->ctl_host;
    std::memset(&k, 0, sizeof(k));
    k.alive = 1;
    k.running_none = std::isnan(running) ? 1 : 0;
    k.running = std::isnan(running) ? 0.0 : running;
    k.snap_cur = c->snap_cur;
    // the trial phase treats the moments of role 1 as current (phase_ptrs): the host's current
    // moments take that place
    k.mu_role[0] = c->mu_cur; k.mu_role[1] = c->mu_ta; k.mu_role[2] = c->mu_tb;
    k.mom_role[0] = c->mom_ta; k.mom_role[1] = c->mom_cur; k.mom_role[2] = c->mom_tb;
    k.a_def = 1.0;          // the host's current vi_mu is stored as it is
    k.c_zero = 1;
    k.mu_base = c->mu_cur;  // (persistent lazy state: and stays there)
    {
        // ... unless it is known to be Sig_k c for a vector c at hand (what vilma_init_state leaves:
        // the reference's _initialize builds vi_mu = einsum(vi_sigma, temp_nat_mu)): the persistent
        // lazy state then starts with a = 0 and no pass of the sweeps reads a vi_mu array at all
        const char *e = std::getenv("VILMA_PURE_START");    // =0: start from the stored array (A/B)
        bool pure = !(e && e[0] == '0') && lazy_persist(c, s) && c->pure_c == c->mu_cur;
        for (int p = 0; p < c->P && pure; ++p) pure = c->pure_tau[p] == c->tau[p];
        if (pure) {
            k.a_def = 0.0;
            k.c_zero = 0;
        }
        s->armed_pure = pure;
        // (from here on the c buffers change roles on the device: what the host knows about them is
        // void until a write-out re-establishes it, persist_writeback)
        c->pure_c = -1;
    }
    k.L0 = L[0];
    k.L_try = std::max(1.0, L[0] / 1.25);
    k.cur_obj = s->objective;
    for (int p = 0; p < VILMA_MAX_P; ++p) {
        k.tau[p] = p < c->P ? c->tau[p] : 1.0;
        k.hrl[p] = p < c->P ? det_hrl(s->ranks[p], c->tau[p]) : 0.0;
    }
    const SweepDecideArgs a = decide_args(decide_params(c, s, VILMA_DECIDE_TRIAL, false));
    decide_set_phases(a, &k);
    HIPCHK(c, hipMemcpyAsync(c->ctl, &k, sizeof(k), hipMemcpyHostToDevice, st));
    s->mirror = k;
    s->outq.clear();
    s->groups_out = 0;
    s->armed = true;
    s->fresh = true;
    s->carry_trials = s->carry_evals = s->carry_products = 0;
    // what a drain before the first report has to give back: the state as it stands
    s->rep_end.valid = true;
    for (int q = 0; q < 3; ++q) { s->rep_end.mu_role[q] = k.mu_role[q]; s->rep_end.mom_role[q] = k.mom_role[q]; }
    s->rep_end.snap_cur = k.snap_cur;
    for (int p = 0; p < VILMA_MAX_P; ++p) s->rep_end.tau[p] = k.tau[p];
    s->rep_end.a_def = 1.0; s->rep_end.c_zero = 1; s->rep_end.mu_base = k.mu_base;
    return 0;
}

// Wait for the oldest outstanding decision, replay it on the mirror, check the device's account of
// it against the replay.  *r = the result vector as that decision saw it.
int consume_decision(vilma_ctx *c, SweepState *s, DecideReport *rep, const double **r,
                     SweepState::Queued *which) {
    if (s->outq.empty()) return fail(c, "internal: no decision outstanding");
    const SweepState::Queued q = s->outq.front();
    volatile double *stamp = s->land[q.land] + s->size + SNAP_SERIAL;
    for (uint64_t spins = 0; *stamp != q.serial; ++spins) {
        __builtin_ia32_pause();
        if ((spins & 0xfffff) == 0xfffff) {
            // a long wait: has the stream died (a failed launch would never write the stamp)?
            const hipError_t e = hipStreamQuery(s->pipe_stream);
            if (e != hipSuccess && e != hipErrorNotReady)
                return fail(c, std::string("the queued sweep failed: ") + hipGetErrorString(e));
            if (e == hipSuccess && *stamp != q.serial)
                return fail(c, "the queued sweep finished without writing its decision");
        }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    s->outq.pop_front();
    if (q.first_of_group) s->groups_out -= 1;
    const double *snap = s->land[q.land];
    decide_core(q.args, &s->mirror, snap, *rep);
    s->last_form = s->mirror.c_zero ? 0 : (s->mirror.a_def == 0.0 ? 2 : 1);
    double want[VILMA_SNAP_EXTRA];
    decide_snapshot_scalars(q.args, &s->mirror, *rep, want);
    const double *got = snap + s->size;
    for (int i = 0; i < SNAP_SERIAL; ++i) {
        const bool same = want[i] == got[i] || (std::isnan(want[i]) && std::isnan(got[i]));
        if (!same)
            return fail(c, "device and host decisions disagree (snapshot field " + std::to_string(i) +
                           ": device " + std::to_string(got[i]) + ", host " + std::to_string(want[i]) + ")");
    }
    *r = snap;
    if (which) *which = q;
    return 0;
}

// The block is dead (the decision just looked at went to the host): everything queued behind has
// exited.  Make the host-side roles and caches those of the device so that the host's own line
// search (nat_grad_step with s->resume) carries on from there.
int pipeline_takeover(vilma_ctx *c, SweepState *s, hipStream_t st, const DecideReport &rep,
                      const double *r, const SweepState::Queued &q, bool sweep_ended) {
    HIPCHK(c, hipStreamSynchronize(st));
    // brackets of launches that exited at once: the rest of this group and every later group
    if (q.tag) prof_drop_tags(c, 4 * q.tag + (q.args.mode == VILMA_DECIDE_TRIAL ? 1 : 2));
    s->outq.clear();
    s->groups_out = 0;
    const SweepCtl &k = s->mirror;
    c->mu_cur = k.mu_role[0]; c->mu_ta = k.mu_role[1]; c->mu_tb = k.mu_role[2];
    if (q.args.persist) {       // (the vi_mu roles name c buffers; the stored array is in mu_base)
        c->mu_cur = k.mu_base; c->mu_ta = (k.mu_base + 1) % 3; c->mu_tb = (k.mu_base + 2) % 3;
    }
    c->mom_cur = k.mom_role[1]; c->mom_ta = k.mom_role[0]; c->mom_tb = k.mom_role[2];
    c->snap_cur = k.snap_cur;
    for (int p = 0; p < c->P; ++p) c->tau[p] = k.tau[p];
    c->have_moments = true;
    c->snp_marked = false;
    c->trial_tainted = false;
    s->objective = k.cur_obj;
    s->cur_sums = -1;
    s->armed = false;
    s->resume = SweepState::Resume();
    if (!k.c_zero && q.args.persist) {
        // a persistent lazy state: (vi_mu of buffer mu_base, a, c of the current role) -- write it out
        // for the host's kernels
        if (persist_writeback(c, s, st, k.mu_base, k.mu_role[0], k.mom_role[1], k.a_def, k.tau)) return 1;
        s->mirror.a_def = 1.0;
        s->mirror.c_zero = 1;
    } else if (!k.c_zero) {
        // handed back in the middle of a beta loop of lazy trials: the current state exists as
        // (stored vi_mu, a, c) only -- write it out for the host's kernels
        if (materialise_deferred(c, st, k.mu_role[0], k.mu_role[1], k.mom_role[1], k.mom_role[1], k.a_def,
                                 k.tau, s->results + s->o_sa)) return 1;
        HIPCHK(c, hipStreamSynchronize(st));
        c->mu_cur = k.mu_role[1]; c->mu_ta = k.mu_role[0];
        c->pure_c = -1;
        s->mirror.a_def = 1.0;
        s->mirror.c_zero = 1;
    }
    const bool trial_pending = q.args.mode == VILMA_DECIDE_TRIAL;
    if (trial_pending && q.args.lazy) {
        // the trial's candidates exist as sums only (no vi_mu was stored): the host's line search
        // evaluates that trial again, with its own kernels
        c->have_b = false;
        c->tile_sums_ns = 0;
        if (s->comm_kind)
            HIPCHK(c, hipMemcpy(s->results + s->o_dsum, r + s->o_dsum,
                                (size_t)(s->reduce_end - s->o_dsum) * sizeof(double), hipMemcpyHostToDevice));
        s->pend_valid = s->alt_valid = false;
        s->trial_sums = s->trial_sums_b = false;
        if (!sweep_ended) {
            s->resume.active = true;
            s->resume.it0 = k.inner_it;
            s->resume.delta_sum = k.delta_sum;
            s->resume.L_try = k.L_try;
        }
    } else if (trial_pending) {
        // the trial's candidates are evaluated and nobody has used them: the host's line search
        // starts with them (trial()'s pend / alt caches)
        const bool stash = q.args.mstep_inside != 0;
        c->have_b = s->two_step;
        c->tile_sums_ns = stash ? (s->two_step ? 2 : 1) : 0;
        // the device copy of the summed part may have gone through the all-reduces of dead groups
        if (s->comm_kind)
            HIPCHK(c, hipMemcpy(s->results + s->o_dsum, r + s->o_dsum,
                                (size_t)(s->reduce_end - s->o_dsum) * sizeof(double), hipMemcpyHostToDevice));
        std::copy(r, r + s->size, s->host.begin());
        s->pend_valid = true;
        s->pend_step = 1.0 / k.L_try;
        s->trial_sums = stash;
        s->trial_sums_b = stash && s->two_step;
        s->alt_valid = s->two_step;
        if (s->two_step) {
            s->alt_step = 1.0 / (k.L_try * s->pipe_rate);
            std::copy(r + s->o_tb, r + s->o_tb + s->nt, s->alt_totals.begin());
            s->alt_obj = objective_from(c, s, s->alt_totals.data());
        }
        if (!sweep_ended) {
            s->resume.active = true;
            s->resume.it0 = k.inner_it;
            s->resume.delta_sum = k.delta_sum;
            s->resume.L_try = k.L_try;
        }
    } else {
        // an EVAL decision the device would not take (a tau it could not accept): the evaluation has
        // been looked at; the host goes on behind the M-step
        c->have_b = false;
        c->tile_sums_ns = 0;
        s->pend_valid = s->alt_valid = false;
        s->trial_sums = s->trial_sums_b = false;
        s->resume.active = true;
        s->resume.after_mstep = true;
        s->resume.delta_sum = k.delta_sum;
    }
    (void)rep;
    return 0;
}

// The caller does not go on with what is queued ahead: wait for it and put the state the last
// reported sweep ended in back (its vi_mu is still in its buffer -- see the invariant above; its
// moments are re-derived by one evaluation: equal to rounding, not to the bit).
int pipeline_rollback(vilma_ctx *c, SweepState *s, hipStream_t st) {
    if (!s->armed) return 0;
    HIPCHK(c, hipStreamSynchronize(st));
    // look at everything outstanding (keeps the mirror and the device's block in step; errors here
    // are real errors)
    int64_t first_tag = 0;
    while (!s->outq.empty()) {
        DecideReport rep;
        const double *r;
        SweepState::Queued q;
        if (consume_decision(c, s, &rep, &r, &q)) return 1;
        if (!first_tag) first_tag = q.tag;
    }
    // their launches' brackets describe work the caller never asked for
    if (first_tag) prof_drop_tags(c, 4 * first_tag);
    s->armed = false;
    s->groups_out = 0;
    s->pend_valid = s->alt_valid = false;
    s->trial_sums = s->trial_sums_b = false;
    s->resume = SweepState::Resume();
    s->carry_trials = s->carry_evals = s->carry_products = 0;
    if (!s->rep_end.valid) return fail(c, "internal: nothing to roll back to");
    // (a persistent lazy state: the vi_mu roles name c buffers, the stored array is in mu_base)
    const bool persisted = lazy_persist(c, s);
    const int cur = persisted ? s->rep_end.mu_base : s->rep_end.mu_role[0];
    c->mu_cur = cur; c->mu_ta = (cur + 1) % 3; c->mu_tb = (cur + 2) % 3;
    if (!s->rep_end.c_zero) {
        // the reported state is a persistent lazy one: its c is still in its buffer (role 0 of the
        // reported roles -- one trial beyond writes the other two), its vi_mu is written out now
        // (the log-normaliser handed over only shifts the sums that come with the pass: not used)
        if (persist_writeback(c, s, st, s->rep_end.mu_base, s->rep_end.mu_role[0], s->rep_end.mom_role[1],
                              s->rep_end.a_def, s->rep_end.tau)) return 1;
    }
    c->mom_cur = 0; c->mom_ta = 1; c->mom_tb = 2;
    c->snap_cur = s->rep_end.snap_cur;
    for (int p = 0; p < c->P; ++p) c->tau[p] = s->rep_end.tau[p];
    c->have_moments = false;
    c->have_b = false;
    c->tile_sums_ns = 0;
    c->snp_marked = false;
    // hyper_delta of the reported state (the host's copy): its table comes out to the same bits
    // the device's M-step gave it (det_lh)
    if (vilma_set_hyper(c, s->hyper.data())) return 1;
    vilma_sweep_stats *keep = s->stats;
    s->stats = nullptr;
    double obj;
    const int rc = evaluate_current(c, s, st, &obj);
    s->stats = keep;
    if (rc) return 1;
    std::vector<double> tot(s->host.begin() + s->o_tot, s->host.begin() + s->o_tot + s->nt);
    if (accept(c, s, 0, obj, tot.data())) return 1;
    s->cur_sums = -1;
    return 0;
}

// One sweep from the control block: see the comment above.  *done = false: the sweep has to be run
// (or finished) by the host's own line search (s->resume and the caches say where it stands).
int pipeline_sweep(vilma_ctx *c, SweepState *s, hipStream_t st, double *L, double *elbo,
                   double *running_delta, double rate, int flags, vilma_sweep_stats *out, bool *done) {
    *done = false;
    const bool promise = pipeline_eligible(c, s, flags);
    const bool diff = (flags & VILMA_SWEEP_DIFF) != 0;
    if (s->armed && (rate != s->pipe_rate || diff != s->pipe_diff || st != s->pipe_stream)) {
        // (a different stream: what is queued sits on the old one, and nothing orders the two)
        if (pipeline_rollback(c, s, s->pipe_stream)) return 1;
    }
    if (!s->armed) {
        if (!promise || s->pend_valid || s->resume.active) return 0;          // host path
        if (pipeline_arm(c, s, st, L, *running_delta, rate, diff)) return 1;
    }
    const bool veto = (flags & VILMA_SWEEP_VETO) != 0;
    out->n_trials = s->carry_trials; out->n_evaluations = s->carry_evals; out->n_products = s->carry_products;
    s->carry_trials = s->carry_evals = s->carry_products = 0;
    bool ended = false;
    double change = 0.0, running_after = 0.0, L0_after = L[0];
    for (;;) {
        // ---- keep the stream fed without ever queuing a second trial beyond the sweep's end
        if (s->mirror.eval_pending != 0) {
            // the beta loop of the sweep being reported is over; its evaluation runs (or has run)
            if (s->outq.empty()) {
                if (promise) { if (queue_group(c, s, st, veto, false)) return 1; }
                else if (queue_eval_decision(c, s, st)) return 1;
            }
        } else {
            const int target = promise ? 2 : 1;
            while (s->groups_out < target) {
                if (queue_group(c, s, st, veto, s->fresh)) return 1;
                s->fresh = false;
            }
        }
        // ---- the oldest outstanding decision
        DecideReport rep;
        const double *r;
        SweepState::Queued q;
        if (consume_decision(c, s, &rep, &r, &q)) return 1;
        if (rep.consumed) {
            // an evaluation of the sweep being reported has been looked at
            std::copy(r + s->o_tot, r + s->o_tot + s->nt, s->totals.begin());
            s->hyper.assign(r + s->o_hyper, r + s->o_hyper + s->am);
            s->objective = rep.eval_obj;
            out->n_evaluations += 1;
            out->n_products += 1;
            if (diff)
                for (int t = 0; t < 3; ++t) {
                    out->diff_sum[t] = r[s->o_dsum + t];
                    out->diff_max[t] = r[s->o_dmax + t];
                }
        }
        if (rep.outcome == VILMA_OUT_TAU_UPDATED)
            for (int p = 0; p < c->P; ++p) c->tau[p] = s->mirror.tau[p];
        if (rep.sweep_end) {
            ended = true;
            change = rep.sweep_change;
            running_after = rep.end_running;
            L0_after = rep.end_L0;
            for (int t = 0; t < 3; ++t) { s->rep_end.mu_role[t] = rep.end_mu_role[t]; s->rep_end.mom_role[t] = rep.end_mom_role[t]; }
            s->rep_end.snap_cur = rep.end_snap_cur;
            for (int p = 0; p < VILMA_MAX_P; ++p) s->rep_end.tau[p] = rep.end_tau[p];
            s->rep_end.a_def = rep.end_a_def; s->rep_end.c_zero = rep.end_c_zero; s->rep_end.mu_base = rep.end_mu_base;
            s->rep_end.valid = true;
        }
        if (q.args.mode == VILMA_DECIDE_TRIAL && rep.L_tried != 0.0) {
            // the line search looked at one or two candidates; behind a sweep's end they belong to
            // the next sweep (the next call reports them)
            const bool both = rep.outcome == VILMA_OUT_REJECTED ? q.args.have_b != 0
                              : (s->mirror.choice == 2 || (rep.outcome == VILMA_OUT_DEAD && q.args.have_b != 0 &&
                                                           !(rep.fa >= (rep.orig - REL_TOL * std::fabs(rep.orig)) - ABS_TOL)));
            if (rep.outcome != VILMA_OUT_DEAD) {
                const int n = both ? 2 : 1;
                if (ended) { s->carry_trials += n; s->carry_evals += n; s->carry_products += 1; }
                else { out->n_trials += n; out->n_evaluations += n; out->n_products += 1; }
            }
            // a trial that did not end the beta loop leaves its group's evaluation launches empty
            if (rep.outcome == VILMA_OUT_ACCEPT_CONTINUE || rep.outcome == VILMA_OUT_REJECTED) {
                prof_drop_tag(c, 4 * q.tag + 1);
                prof_drop_tag(c, 4 * q.tag + 2);
                // (so does the sums pass: the state lazy trials reach is only written out when the loop ends)
                prof_drop_tag(c, 4 * q.tag + 3);
            }
        } else if (q.args.mode == VILMA_DECIDE_EVAL && q.tag && rep.outcome != VILMA_OUT_TAU_UPDATED) {
            prof_drop_tag(c, 4 * q.tag + 2);        // the re-evaluation's launches exit at once
        }
        if (!s->mirror.alive) {
            out->skipped_ahead = 1;
            if (pipeline_takeover(c, s, st, rep, r, q, ended)) return 1;
            if (!ended) {
                // the host finishes this sweep: what the device had done of it is in s->resume
                if (s->resume.active && !s->resume.after_mstep) L[0] = s->resume.L_try;
                else L[0] = s->mirror.L0;
                return 0;
            }
            break;
        }
        if (ended) break;
    }
    // _optimize_step's bookkeeping (variational_inference.py:403-409), as the device did it
    L[0] = L0_after;
    L[1] = std::max(1.0, L[1] / 1.25);
    L[2] = std::max(1.0, L[2] / 1.25);
    *elbo = *elbo + change;
    *running_delta = running_after;
    out->ran_ahead = 1;
    s->have_diff = diff;
    s->cur_sums = -1;
    s->trial_sums = s->trial_sums_b = false;
    if (s->armed) s->alt_valid = false;
    if (s->armed && !promise && s->outq.empty() && s->mirror.eval_pending == 0 && s->groups_out == 0) {
        // nothing queued behind and no promise of another sweep: the host-side roles become the
        // device's (the sweep's last evaluation is accepted unconditionally)
        const SweepCtl &k = s->mirror;
        c->mu_cur = k.mu_role[0]; c->mu_ta = k.mu_role[1]; c->mu_tb = k.mu_role[2];
        if (lazy_persist(c, s)) {
            c->mu_cur = k.mu_base; c->mu_ta = (k.mu_base + 1) % 3; c->mu_tb = (k.mu_base + 2) % 3;
        }
        if (!k.c_zero) {
            // (a persistent lazy state: the run ends here, its vi_mu is written out)
            HIPCHK(c, hipStreamSynchronize(st));
            if (persist_writeback(c, s, st, k.mu_base, k.mu_role[0], k.mom_role[1], k.a_def, k.tau)) return 1;
            s->mirror.a_def = 1.0;
            s->mirror.c_zero = 1;
        }
        c->mom_cur = k.mom_role[1]; c->mom_ta = k.mom_role[0]; c->mom_tb = k.mom_role[2];
        c->snap_cur = k.snap_cur;
        for (int p = 0; p < c->P; ++p) c->tau[p] = k.tau[p];
        c->have_moments = true;
        c->have_b = false;
        c->tile_sums_ns = 0;
        c->snp_marked = false;
        s->armed = false;
    }
    *done = true;
    return 0;
}

}  // namespace

extern "C" {

int64_t vilma_results_size(const vilma_ctx *c) {
    if (!c) return 0;
    SweepState *s = sweep_state(const_cast<vilma_ctx *>(c));
    return s ? s->size : 0;
}

double *vilma_results_dev(vilma_ctx *c) {
    if (!c) return nullptr;
    SweepState *s = sweep_state(c);
    return s ? s->results : nullptr;
}

int vilma_set_fit_constants(vilma_ctx *c, const double *chi, const double *ranks, int scale_se) {
    if (!c) return 1;
    SW(c);
    s->chi.assign(chi, chi + c->P);
    s->ranks.assign(ranks, ranks + c->P);
    s->scale_se = scale_se != 0;
    s->have_consts = true;
    s->have_state = false;
    return 0;
}

int vilma_comm_unique_id(char id[128]) {
    // no context to carry the message: vilma_last_error(NULL) reports it
    if (!rccl_load()) return fail(nullptr, rccl().error);
    Id128 u;
    std::memset(&u, 0, sizeof(u));
    if (rccl().GetUniqueId(&u) != 0) return fail(nullptr, "ncclGetUniqueId failed");
    std::memcpy(id, u.bytes, 128);
    return 0;
}

int vilma_comm_init_rccl(vilma_ctx *c, int world, int rank, const char id[128]) {
    if (!c) return 1;
    SW(c);
    if (world < 1 || rank < 0 || rank >= world) return fail(c, "bad world size / rank");
    if (!rccl_load()) return fail(c, rccl().error);
    if (s->nccl_comm) { (void)rccl().CommDestroy(s->nccl_comm); s->nccl_comm = nullptr; }
    Id128 u;
    std::memcpy(u.bytes, id, 128);
    HIPCHK(c, hipSetDevice(c->device));
    const int rc = rccl().CommInitRank(&s->nccl_comm, world, u, rank);
    if (rc != 0) return fail(c, std::string("ncclCommInitRank: ") + rccl().GetErrorString(rc));
    s->comm_kind = 1; s->world = world; s->rank = rank;
    return 0;
}

int vilma_comm_set_callback(vilma_ctx *c, vilma_allreduce_fn fn, void *user, int world, int rank) {
    if (!c) return 1;
    SW(c);
    if (s->nccl_comm) { (void)rccl().CommDestroy(s->nccl_comm); s->nccl_comm = nullptr; }
    s->cb = fn; s->cb_user = user;
    s->comm_kind = fn ? 2 : 0;
    s->world = fn ? world : 1;
    s->rank = fn ? rank : 0;
    return 0;
}

int vilma_comm_info(const vilma_ctx *c, int *kind, int *world, int *rank) {
    if (!c) return 1;
    const SweepState *s = c->sw;
    if (kind) *kind = s ? s->comm_kind : 0;
    if (world) *world = s ? s->world : 1;
    if (rank) *rank = s ? s->rank : 0;
    return 0;
}

int vilma_comm_allreduce(vilma_ctx *c, void *stream, double *buf, int64_t n, int op) {
    if (!c) return 1;
    SW(c);
    return comm_allreduce(c, s, (hipStream_t)stream, buf, n, op);
}

int vilma_set_state(vilma_ctx *c, void *stream, const double *vi_mu, const double *hyper,
                    const double *tau, double *objective) {
    if (!c) return 1;
    SW(c);
    if (!s->have_consts) return fail(c, "vilma_set_fit_constants has not been called");
    if (!hyper) return fail(c, "hyper_delta is required");
    hipStream_t st = (hipStream_t)stream;
    if (vilma_sweep_drain(c)) return 1;
    if (tau && vilma_set_tau(c, tau)) return 1;
    if (vilma_set_hyper(c, hyper)) return 1;
    s->hyper.assign(hyper, hyper + s->am);
    if (vi_mu && vilma_set_mu(c, vi_mu)) return 1;
    s->stats = nullptr;
    double obj;
    if (evaluate_current(c, s, st, &obj)) return 1;
    std::vector<double> tot(s->host.begin() + s->o_tot, s->host.begin() + s->o_tot + s->nt);
    if (accept(c, s, 0, obj, tot.data())) return 1;
    s->cur_sums = -1;
    s->L_rejected = -1.0;
    if (objective) *objective = obj;
    return 0;
}

int vilma_get_state(vilma_ctx *c, double *vi_mu, double *vi_delta, double *hyper, double *tau) {
    if (!c) return 1;
    SW(c);
    // hyper_delta and error_scaling of the last reported sweep are known to the host; only the
    // per-SNP arrays need the device to stand at that state
    if ((vi_mu || vi_delta) && vilma_sweep_drain(c)) return 1;
    if (vi_mu && vilma_get_mu(c, vi_mu)) return 1;
    if (vi_delta && vilma_get_delta(c, vi_delta)) return 1;
    if (hyper) {
        if (s->hyper.empty()) return fail(c, "hyper_delta has not been set");
        std::copy(s->hyper.begin(), s->hyper.end(), hyper);
    }
    if (tau) std::copy(c->tau, c->tau + c->P, tau);
    return 0;
}

int vilma_initialize(vilma_ctx *c, void *stream, const double *fake_mu, double *objective) {
    if (!c) return 1;
    SW(c);
    if (!s->have_consts) return fail(c, "vilma_set_fit_constants has not been called");
    hipStream_t st = (hipStream_t)stream;
    if (vilma_sweep_drain(c)) return 1;
    if (vilma_init_state(c, stream, fake_mu, s->results + s->o_sa)) return 1;
    if (s->comm_kind && comm_allreduce(c, s, st, s->results + s->o_sa, s->am, 0)) return 1;
    std::vector<double> sums(s->am);
    if (vilma_fetch(c, stream, s->results + s->o_sa, sums.data(), s->am)) return 1;
    // hyper_delta of _initialize (variational_inference.py:667-674)
    std::vector<double> hyper(s->am);
    for (int a = 0; a < s->A; ++a) {
        double tot = 0.0;
        for (int k = 0; k < s->M; ++k) tot += sums[(size_t)a * s->M + k] + 1.0;
        for (int k = 0; k < s->M; ++k)
            hyper[(size_t)a * s->M + k] = std::max((sums[(size_t)a * s->M + k] + 1.0) / tot, EPSILON);
    }
    return vilma_set_state(c, stream, nullptr, hyper.data(), nullptr, objective);
}

int vilma_elbo(vilma_ctx *c, double *objective) {
    if (!c) return 1;
    SW(c);
    if (!s->have_state) return fail(c, "no state: call vilma_set_state or vilma_initialize");
    if (objective) *objective = s->objective;
    return 0;
}

int vilma_posterior(vilma_ctx *c, double *mean, double *var) {
    if (!c) return 1;
    if (vilma_sweep_drain(c)) return 1;
    if (vilma_get_moments(c, mean, var)) return 1;
    const size_t PN = (size_t)c->P * c->N;
    std::vector<double> scal(PN);
    HIPCHK(c, hipMemcpy(scal.data(), c->scal, PN * sizeof(double), hipMemcpyDeviceToHost));
    if (mean) for (size_t t = 0; t < PN; ++t) mean[t] *= scal[t];
    if (var) for (size_t t = 0; t < PN; ++t) var[t] *= scal[t] * scal[t];
    return 0;
}

int vilma_update_beta(vilma_ctx *c, void *stream, double *L0, double rate, double *orig_obj,
                      double *new_obj) {
    if (!c) return 1;
    SW(c);
    if (!s->have_state) return fail(c, "no state: call vilma_set_state or vilma_initialize");
    if (!L0 || !(rate > 1.0)) return fail(c, "vilma_update_beta: L0 and a line_search_rate > 1 are required");
    if (vilma_sweep_drain(c)) return 1;
    s->stats = nullptr;
    s->flags = 0;
    s->lsr = rate;
    double L[5] = {*L0, 1.0, 1.0, 1.0, 1.0};
    const double orig = s->objective;
    double nw;
    if (update_beta(c, s, (hipStream_t)stream, L, orig, &nw)) return 1;
    *L0 = L[0];
    if (orig_obj) *orig_obj = orig;
    if (new_obj) *new_obj = nw;
    return 0;
}

int vilma_update_hyper_delta(vilma_ctx *c, void *stream, double *orig_obj, double *new_obj) {
    if (!c) return 1;
    SW(c);
    if (!s->have_state) return fail(c, "no state: call vilma_set_state or vilma_initialize");
    if (vilma_sweep_drain(c)) return 1;
    s->stats = nullptr;
    s->flags = 0;
    const double orig = s->objective;
    double nw;
    if (update_hyper(c, s, (hipStream_t)stream, false, &nw)) return 1;
    if (orig_obj) *orig_obj = orig;
    if (new_obj) *new_obj = nw;
    return 0;
}

int vilma_update_error_scaling(vilma_ctx *c, void *stream, double *orig_obj, double *new_obj) {
    if (!c) return 1;
    SW(c);
    if (!s->have_state) return fail(c, "no state: call vilma_set_state or vilma_initialize");
    if (vilma_sweep_drain(c)) return 1;
    s->stats = nullptr;
    s->flags = 0;
    const double orig = s->objective;
    double nw;
    if (update_error_scaling(c, s, (hipStream_t)stream, &nw)) return 1;
    if (orig_obj) *orig_obj = orig;
    if (new_obj) *new_obj = nw;
    return 0;
}

int vilma_debug_result_slot(vilma_ctx *c, int which, double *out, int n) {
    if (!c || !out) return 1;
    SweepState *s = sweep_state(c);
    if (!s) return 1;
    if (which < 0 || which > 1 || n < 0 || n > s->nt) return fail(c, "vilma_debug_result_slot: bad slot or length");
    if (vilma_sweep_drain(c)) return 1;
    HIPCHK(c, hipDeviceSynchronize());
    HIPCHK(c, hipMemcpy(out, s->results + (which ? s->o_tb : s->o_ta), (size_t)n * sizeof(double), hipMemcpyDefault));
    return 0;
}

int vilma_prof_state_form(vilma_ctx *c, int *form) {
    if (!c || !form) return 1;
    SweepState *s = c->sw;
    *form = s ? s->last_form : 0;
    return 0;
}

int vilma_sweep_drain(vilma_ctx *c) {
    if (!c) return 1;
    SweepState *s = c->sw;
    if (!s || !s->armed) return 0;
    return pipeline_rollback(c, s, s->pipe_stream);
}

int vilma_sweep(vilma_ctx *c, void *stream, double L[5], double *elbo, double *running_delta,
                double line_search_rate, int flags, vilma_sweep_stats *stats) {
    if (!c) return 1;
    SW(c);
    if (!s->have_state) return fail(c, "no state: call vilma_set_state or vilma_initialize");
    if (!L || !elbo || !running_delta) return fail(c, "vilma_sweep: L, elbo and running_delta are required");
    if (!(line_search_rate > 1.0)) return fail(c, "line_search_rate must exceed 1");
    hipStream_t st = (hipStream_t)stream;
    vilma_sweep_stats local;
    vilma_sweep_stats *out = stats ? stats : &local;
    std::memset(out, 0, sizeof(*out));
    s->stats = out;
    s->flags = flags;
    s->lsr = line_search_rate;
    s->have_diff = false;
    event(s, 3, 0, *elbo, 0.0);
    {
        bool done = false;
        if (pipeline_sweep(c, s, st, L, elbo, running_delta, line_search_rate, flags, out, &done)) {
            s->stats = nullptr;
            return 1;
        }
        if (done) {
            s->stats = nullptr;
            out->elbo = *elbo;
            out->running = *running_delta;
            out->objective = s->objective;
            for (int q = 0; q < 5; ++q) out->L[q] = L[q];
            for (int p = 0; p < c->P; ++p) out->error_scaling[p] = c->tau[p];
            return 0;
        }
    }
    double change = 0.0;
    const int rc = nat_grad_step(c, s, st, L, *running_delta, &change);
    s->stats = nullptr;
    if (rc) return 1;
    // variational_inference.py:403-409
    double running = std::isnan(*running_delta) ? change : *running_delta;
    running *= ELBO_MOMENTUM;
    running += (1 - ELBO_MOMENTUM) * std::max(change, 0.0);
    *elbo = *elbo + change;
    *running_delta = running;
    if ((flags & VILMA_SWEEP_DIFF) && !s->have_diff) {
        // the sweep's last evaluation was not the M-step's (error-scaling update): a separate pass
        if (vilma_mean_diff(c, stream, s->results + s->o_dsum, s->results + s->o_dmax)) return 1;
        if (reduce_and_fetch(c, s, st, s->o_dsum, s->o_dsum + 3, (flags & VILMA_SWEEP_VERBOSE) != 0))
            return 1;
        for (int q = 0; q < 3; ++q) {
            out->diff_sum[q] = s->host[s->o_dsum + q];
            out->diff_max[q] = s->host[s->o_dmax + q];
        }
    }
    out->elbo = *elbo;
    out->running = running;
    out->objective = s->objective;
    for (int q = 0; q < 5; ++q) out->L[q] = L[q];
    for (int p = 0; p < c->P; ++p) out->error_scaling[p] = c->tau[p];
    return 0;
}

}  // extern "C"
