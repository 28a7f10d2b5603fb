"""Function API, host side: argument checking happens before anything touches the GPU, so the
type rules of the reference's numba signatures (numerics.py:11-258: float64 / int64 arrays of a
fixed rank) can be tested here."""
import inspect
import os
import re

import numpy as np
import pytest

from vilma_amd import numerics as nm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

REFERENCE_NAMES = [
    'sum_betas', 'fast_divide', 'fast_linked_ests', 'fast_likelihood', 'fast_posterior_mean',
    'fast_pmv', 'fast_nat_inner_product_m2', 'fast_nat_inner_product', 'fast_inner_product_comp',
    'sum_annotations', 'fast_delta_kl', 'fast_beta_kl', 'fast_vi_delta_grad', 'map_to_nat_cat_2D',
    'invert_nat_cat_2D', 'fast_invert_nat_vi_delta', '_matrix_invert_4d_numba', 'matrix_invert',
    'vi_sigma_inv', '_matrix_log_det_4d_numba', 'matrix_log_det', 'vi_sigma_log_det']


def test_every_reference_function_is_present_with_its_argument_names():
    """Names and argument order of numerics.py, as the oracle restates them."""
    from oracle import numerics as onm
    for name in REFERENCE_NAMES:
        assert callable(getattr(nm, name)), name
        if hasattr(onm, name):
            assert list(inspect.signature(getattr(nm, name)).parameters) == \
                list(inspect.signature(getattr(onm, name)).parameters), name
    assert nm.EPSILON == 1e-100


def test_wrong_types_raise_before_any_device_call():
    f32 = np.zeros((2, 3), dtype=np.float32)
    with pytest.raises(TypeError):
        nm.fast_divide(f32, f32)
    with pytest.raises(TypeError):
        nm.fast_divide(np.zeros(3), np.zeros(3))                  # rank
    with pytest.raises(TypeError):
        nm.fast_divide([[1.0]], [[1.0]])                          # not an array
    with pytest.raises(TypeError):
        nm.sum_annotations(np.zeros((4, 2)), np.zeros(4, dtype=np.int32), 1)
    with pytest.raises(TypeError):
        nm.fast_posterior_mean(np.zeros((2, 1, 4)), np.zeros((4, 2), dtype=np.float32))
    with pytest.raises(TypeError):
        nm.matrix_invert(np.zeros((3, 2, 3)))                     # not square


def test_shape_mismatches_raise_value_error():
    with pytest.raises(ValueError):
        nm.fast_divide(np.zeros((2, 3)), np.zeros((2, 4)))
    with pytest.raises(ValueError):
        nm.fast_posterior_mean(np.zeros((2, 1, 4)), np.zeros((4, 3)))
    with pytest.raises(ValueError):
        nm.fast_inner_product_comp(np.zeros((3, 2, 10)), np.zeros((3, 2, 2, 10)),
                                   np.zeros((10, 3)))
    with pytest.raises(ValueError):
        nm.fast_invert_nat_vi_delta(np.zeros((3, 2, 10)), np.zeros((3, 2, 10)),
                                    np.zeros((10, 3)), np.zeros((10, 3)))
    with pytest.raises(NotImplementedError):
        nm.matrix_invert(np.zeros((2, 9, 9)))            # beyond 8 x 8: not instantiated
    with pytest.raises(ValueError):
        nm._matrix_invert_4d_numba(np.zeros((2, 2, 3, 3)))


def test_header_documents_every_entry_point_with_its_reference_lines():
    header = open(os.path.join(ROOT, 'include', 'vilma_numerics.h')).read()
    decls = re.findall(r'\bint (vilma_num_[a-z0-9_]+)\s*\(', header)
    assert len(decls) == 19
    assert header.count('numerics.py:') >= len(decls)
