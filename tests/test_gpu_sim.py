"""`vilma sim` on the HIP LD operator (vilma_ld_matvec): the reference's golden output and the
reference's own draws for pinned seeds."""
import pytest

import sim_cases

pytestmark = pytest.mark.gpu


def test_sim_gwas_matches_reference():
    sim_cases.check_sim_gwas_against_reference()


def test_sim_gwas_moments():
    sim_cases.check_sim_gwas_moments()


def test_cli_sim_golden(tmp_path):
    sim_cases.check_cli_sim(tmp_path)
