"""`vilma sim` host logic (flags, RNG order, file formats) with the oracle-backed LD operator
standing in for the GPU one -- the same cases run on the HIP operator in tests/test_gpu_sim.py."""
import numpy as np
import pytest

import sim_cases
from oracle_engine import OracleEngine
from vilma_amd import matrix_structures as ms, sim


@pytest.fixture
def oracle_operator(monkeypatch):
    def own_engine(self):
        if self._engine is None:
            eng = OracleEngine(1, self.shape[0], 2, 1)
            eng.load_ld(0, self.device_blocks(), self.perm.astype(np.int64), int(self.starts[-1]))
            self._engine = eng
        return self._engine
    monkeypatch.setattr(ms.BlockDiagonalMatrix, '_own_engine', own_engine)


def test_rng_order_matches_reference():
    sim_cases.check_rng_order_against_reference()


def test_sim_components_frequencies():
    # reference tests/test.py:1935-1951
    annotations = np.zeros((20000, 2))
    annotations[:10000, 0] = 1
    annotations[10000:, 1] = 1
    weights = np.array([[0.5, 0.3, 0.2], [0.2, 0.3, 0.5]])
    np.random.seed(1)
    draws = sim.sim_components(annotations, weights)
    assert draws.shape == (20000, 3) and np.allclose(draws.sum(axis=1), 1)
    assert np.all(np.abs(draws[:10000].mean(axis=0) - weights[0]) < 0.025)
    assert np.all(np.abs(draws[10000:].mean(axis=0) - weights[1]) < 0.025)


def test_sim_true_effects_moments():
    # reference tests/test.py:1954-1984
    rng = np.random.default_rng(2)
    annotations = np.zeros((20000, 2))
    annotations[:10000, 0] = 1
    annotations[10000:, 1] = 1
    weights = np.eye(2)
    c1 = rng.random((3, 3)); c1 = c1 + c1.T + 5 * np.eye(3)
    c2 = 10 * rng.random((3, 3)); c2 = c2 + c2.T + 50 * np.eye(3)
    np.random.seed(2)
    eff = sim.sim_true_effects(annotations, weights, np.array([c1, c2]))
    assert eff.shape == (3, 20000)
    for part, cov in ((eff[:, :10000], c1), (eff[:, 10000:], c2)):
        assert np.all(np.abs(part.mean(axis=1) / np.sqrt(np.diag(cov))) < 5 / np.sqrt(10000))
        err = np.sqrt(np.outer(np.diag(cov), np.diag(cov)))
        assert np.all(np.abs(np.cov(part) - cov) < err * 5 / np.sqrt(10000))


def test_bad_weights_rejected():
    annotations = np.ones((5, 1))
    with pytest.raises(ValueError):
        sim.sim_components(annotations, np.array([[0.5, 0.4]]))
    with pytest.raises(ValueError):
        sim.sim_components(annotations, np.array([[1.5, -0.5]]))


def test_sim_gwas_matches_reference(oracle_operator):
    sim_cases.check_sim_gwas_against_reference()


def test_sim_gwas_moments(oracle_operator):
    sim_cases.check_sim_gwas_moments()


def test_cli_sim_golden(oracle_operator, tmp_path):
    sim_cases.check_cli_sim(tmp_path)


def test_sim_needs_the_gpu_operator():
    """No CPU fallback in the product: without the HIP library / a GPU the operator raises."""
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    bd = ms.BlockDiagonalMatrix([ms.LowRankMatrix(X=np.eye(3))])
    with pytest.raises(Exception):
        sim.sim_gwas(np.ones(3), np.ones(3), bd)
