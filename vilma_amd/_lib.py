"""ctypes binding of libvilma_hip.so (include/vilma_hip.h).

There is no CPU fallback: if the library has not been built, or no GPU is present when a
context is created, this raises -- the product path never computes the VI sweep on the host.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# VILMA_HIP_LIB selects an alternative build of the same library (kernel A/B experiments)
LIB_PATH = os.environ.get('VILMA_HIP_LIB') or os.path.join(_HERE, 'libvilma_hip.so')

_lib = None


class VilmaHipError(RuntimeError):
    pass


def ntotals(P):
    return 3 * P + 2


STATE_CURRENT, STATE_TRIAL_BETA, STATE_TRIAL_EVAL, STATE_TRIAL_BETA_B = 0, 1, 2, 3

SWEEP_DIFF, SWEEP_LOOKAHEAD, SWEEP_VETO, SWEEP_VETO_NEXT, SWEEP_VERBOSE = 1, 2, 4, 8, 16
MAX_COHORTS, SWEEP_EVENTS = 8, 48


class SweepEvent(C.Structure):
    _fields_ = [('kind', C.c_int32), ('paramset', C.c_int32), ('a', C.c_double), ('b', C.c_double)]


class SweepStats(C.Structure):
    """vilma_sweep_stats of include/vilma_hip.h."""
    _fields_ = [('elbo', C.c_double), ('running', C.c_double), ('objective', C.c_double),
                ('L', C.c_double * 5),
                ('diff_sum', C.c_double * 3), ('diff_max', C.c_double * 3),
                ('error_scaling', C.c_double * MAX_COHORTS),
                ('n_evaluations', C.c_int32), ('n_trials', C.c_int32), ('n_products', C.c_int32),
                ('ran_ahead', C.c_int32), ('skipped_ahead', C.c_int32), ('n_events', C.c_int32),
                ('events', SweepEvent * SWEEP_EVENTS)]


# int fn(void *user, void *stream, double *buf_dev, int64_t n, int op)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int)

_c_double_p = C.POINTER(C.c_double)
_c_i32_p = C.POINTER(C.c_int32)
_c_i64_p = C.POINTER(C.c_int64)


def load():
    """Load the shared library (once) and declare every entry point of vilma_hip.h."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VilmaHipError(
            'libvilma_hip.so is not built (%s). Run `python -m vilma_amd.build` or '
            '`python -c "import __graft_entry__ as g; g.build()"`. There is no CPU fallback '
            'for the fit hot path.' % LIB_PATH)
    # torch ships its own libamdhip64; whichever HIP runtime is loaded FIRST in a process is the
    # one that can own the GPU (a second runtime initialised later sees no device).  The engine
    # shares device memory and streams with torch, so both must bind to the same runtime: load
    # torch's before dlopen resolves this library's libamdhip64 dependency by soname.
    try:
        import torch        # noqa: F401
    except ImportError:     # stand-alone use of the C-ABI: the system runtime
        pass
    lib = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    sigs = {
        'vilma_version': (C.c_char_p, []),
        'vilma_last_error': (C.c_char_p, [vp]),
        'vilma_create': (C.c_int, [C.c_int, C.c_int64, C.c_int, C.c_int, C.POINTER(vp)]),
        'vilma_destroy': (None, [vp]),
        'vilma_set_snp_data': (C.c_int, [vp, vp, vp, vp, vp, vp]),
        'vilma_set_mixture': (C.c_int, [vp, vp, vp]),
        'vilma_set_tau': (C.c_int, [vp, vp]),
        'vilma_set_hyper': (C.c_int, [vp, vp]),
        'vilma_set_annotation_counts': (C.c_int, [vp, vp]),
        'vilma_mstep': (C.c_int, [vp, vp, vp, vp]),
        'vilma_ld_begin': (C.c_int, [vp, C.c_int, C.c_int, C.c_int64, vp, C.c_int64]),
        'vilma_ld_dense_elems': (C.c_int64, [C.c_int]),
        'vilma_ld_lowrank_elems': (C.c_int64, [C.c_int, C.c_int]),
        'vilma_ld_add_dense': (C.c_int, [vp, C.c_int, C.c_int, vp]),
        'vilma_ld_add_lowrank': (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp]),
        'vilma_ld_end': (C.c_int, [vp, C.c_int]),
        'vilma_ld_matvec': (C.c_int, [vp, vp, C.c_int, vp, vp]),
        'vilma_ld_matvec2': (C.c_int, [vp, vp, C.c_int, vp, vp, vp, vp]),
        'vilma_ld_bytes': (C.c_int, [vp, _c_i64_p, _c_i64_p]),
        'vilma_set_mu': (C.c_int, [vp, vp]),
        'vilma_get_mu': (C.c_int, [vp, vp]),
        'vilma_get_delta': (C.c_int, [vp, vp]),
        'vilma_get_vi_sigma': (C.c_int, [vp, vp, vp]),
        'vilma_get_moments': (C.c_int, [vp, vp, vp]),
        'vilma_eval': (C.c_int, [vp, vp, vp]),
        'vilma_eval_diff': (C.c_int, [vp, vp, vp, vp, vp]),
        'vilma_eval_given_delta': (C.c_int, [vp, vp, vp, vp]),
        'vilma_get_trial_moments': (C.c_int, [vp, vp, vp]),
        'vilma_init_state': (C.c_int, [vp, vp, vp, vp]),
        'vilma_trial_beta': (C.c_int, [vp, vp, C.c_double, vp]),
        'vilma_trial_beta2': (C.c_int, [vp, vp, C.c_double, C.c_double, vp, vp]),
        'vilma_accept': (C.c_int, [vp, C.c_int]),
        'vilma_delta_sums': (C.c_int, [vp, vp, vp, C.c_int]),
        'vilma_trial_sums': (C.c_int, [vp, vp, vp, vp]),
        'vilma_trial_sums_available': (C.c_int, [vp]),
        'vilma_mean_diff': (C.c_int, [vp, vp, vp, vp]),
        'vilma_snapshot_mean': (C.c_int, [vp, vp]),
        'vilma_fetch': (C.c_int, [vp, vp, vp, vp, C.c_int64]),
        'vilma_results_size': (C.c_int64, [vp]),
        'vilma_results_dev': (vp, [vp]),
        'vilma_set_fit_constants': (C.c_int, [vp, vp, vp, C.c_int]),
        'vilma_comm_unique_id': (C.c_int, [vp]),
        'vilma_comm_init_rccl': (C.c_int, [vp, C.c_int, C.c_int, vp]),
        'vilma_comm_set_callback': (C.c_int, [vp, ALLREDUCE_FN, vp, C.c_int, C.c_int]),
        'vilma_comm_info': (C.c_int, [vp, vp, vp, vp]),
        'vilma_comm_allreduce': (C.c_int, [vp, vp, vp, C.c_int64, C.c_int]),
        'vilma_set_state': (C.c_int, [vp, vp, vp, vp, vp, vp]),
        'vilma_get_state': (C.c_int, [vp, vp, vp, vp, vp]),
        'vilma_initialize': (C.c_int, [vp, vp, vp, vp]),
        'vilma_elbo': (C.c_int, [vp, vp]),
        'vilma_posterior': (C.c_int, [vp, vp, vp]),
        'vilma_sweep': (C.c_int, [vp, vp, vp, vp, vp, C.c_double, C.c_int, vp]),
        'vilma_sweep_drain': (C.c_int, [vp]),
        'vilma_prof_state_form': (C.c_int, [vp, vp]),
        'vilma_update_beta': (C.c_int, [vp, vp, vp, C.c_double, vp, vp]),
        'vilma_update_hyper_delta': (C.c_int, [vp, vp, vp, vp]),
        'vilma_update_error_scaling': (C.c_int, [vp, vp, vp, vp]),
        'vilma_prof_enable': (C.c_int, [vp, C.c_int]),
        'vilma_prof_read': (C.c_int, [vp, _c_double_p, _c_i64_p, C.c_int]),
        'vilma_prof_stream_store': (C.c_int, [vp, vp, C.c_int, vp, vp]),
        'vilma_debug_result_slot': (C.c_int, [vp, C.c_int, vp, C.c_int]),
        'vilma_prof_stream_pattern': (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
        'vilma_prof_ld_order': (C.c_int, [vp, C.c_int]),
        'vilma_prof_ld_tile': (C.c_int, [vp, vp, vp, vp]),
        'vilma_prof_ld_trace': (C.c_int, [vp, vp, C.c_int64]),
        # include/vilma_numerics.h -- the Function API (vilma_amd/numerics.py)
        'vilma_num_last_error': (C.c_char_p, []),
        'vilma_num_sum_betas': (C.c_int, [vp, vp, C.c_double, C.c_int64, vp]),
        'vilma_num_divide': (C.c_int, [vp, vp, C.c_int64, vp]),
        'vilma_num_linked_ests': (C.c_int, [vp, vp, vp, vp, C.c_int64, vp]),
        'vilma_num_likelihood': (C.c_int, [vp] * 9 + [C.c_int, C.c_int64, vp]),
        'vilma_num_posterior_mean': (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int64, vp]),
        'vilma_num_pmv': (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int64, vp]),
        'vilma_num_nat_inner_product': (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int64,
                                                  C.c_double, vp]),
        'vilma_num_inner_product_comp': (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int64, vp]),
        'vilma_num_sum_annotations': (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int64, vp]),
        'vilma_num_delta_kl': (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int64, vp]),
        'vilma_num_beta_kl': (C.c_int, [vp, vp, C.c_int64, vp]),
        'vilma_num_vi_delta_grad': (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int64, vp]),
        'vilma_num_map_to_nat_cat': (C.c_int, [vp, C.c_int64, C.c_int, vp]),
        'vilma_num_invert_nat_cat': (C.c_int, [vp, C.c_int64, C.c_int, vp]),
        'vilma_num_invert_nat_vi_delta': (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int64,
                                                    vp]),
        'vilma_num_matrix_invert': (C.c_int, [vp, C.c_int64, C.c_int, C.c_int, vp]),
        'vilma_num_matrix_log_det': (C.c_int, [vp, C.c_int64, C.c_int, C.c_int, vp]),
        'vilma_num_vi_sigma_inv': (C.c_int, [vp, C.c_int, C.c_int, C.c_int64, vp]),
        'vilma_num_vi_sigma_log_det': (C.c_int, [vp, C.c_int, C.c_int, C.c_int64, vp]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)      # AttributeError if the header and the library diverge
        fn.restype = res
        fn.argtypes = args
    lib._vilma_symbols = sorted(sigs)
    _lib = lib
    return lib


def exported_symbols():
    return list(load()._vilma_symbols)
