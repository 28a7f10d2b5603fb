"""`vilma fit` end to end on the reference's golden runs, host logic only: the CLI surface,
loaders, mixture grid, RNG order and output writers of the product with the oracle-backed test
engine underneath (the GPU engine runs the same cases in tests/test_gpu_fit.py).
Mirrors reference tests/test.py:2161-2197 (test_cli_fit) and example/example.sh."""
import os
import pickle

import numpy as np
import pandas as pd
import pytest

from helpers import GOLDEN
from oracle_engine import OracleEngine
from vilma_amd import frontend, vi_options

REF = os.path.join(GOLDEN, 'refdata')
EX = os.path.join(GOLDEN, 'example')


def run_fit(argv, engine_factory):
    args = frontend.build_parser().parse_args(['fit'] + argv)
    from helpers import engine_class
    with engine_class(engine_factory):
        vi_options.main(args)


def check_frames(truth, got):
    assert list(truth.columns) == list(got.columns)
    for col in truth.columns:
        if truth[col].dtype.kind == 'f':
            np.testing.assert_allclose(got[col], truth[col], rtol=1e-5, atol=1e-8, err_msg=col)
        else:
            assert (truth[col] == got[col]).all(), col


def fit_test_cli(tmp_path, engine_factory, manifest='ld_manifest.tsv'):
    out = str(tmp_path / 'vilma_run')
    run_fit(['--ld-schema', os.path.join(REF, manifest),
             '--sumstats', os.path.join(REF, 'good_sumstats_beta.tsv'),
             '--output', out, '-K', '80', '--ldthresh', '0.8', '--init-hg', '0.2',
             '--samplesizes', '10e3', '--names', 'test_cohort', '--learn-scaling',
             '--extract', os.path.join(REF, 'good_variants.tsv')], engine_factory)
    truth = np.load(os.path.join(REF, 'copy_vilma_run.npz'))
    got = np.load(out + '.npz')
    assert sorted(truth.files) == sorted(got.files)
    for key in truth.files:
        assert truth[key].shape == got[key].shape and got[key].dtype == np.float64
        np.testing.assert_allclose(got[key], truth[key], rtol=1e-5, atol=1e-8, err_msg=key)
    with open(os.path.join(REF, 'copy_vilma_run.covariance.pkl'), 'rb') as fh:
        tcov = pickle.load(fh)
    with open(out + '.covariance.pkl', 'rb') as fh:
        gcov = pickle.load(fh)
    assert np.allclose(tcov, gcov, rtol=1e-12, atol=0)
    check_frames(pd.read_csv(os.path.join(REF, 'copy_vilma_run.estimates.tsv'), sep='\t'),
                 pd.read_csv(out + '.estimates.tsv', sep='\t'))
    return out


def fit_example(tmp_path, engine_factory):
    out = str(tmp_path / 'example_vilma_run')
    common = ['--sumstats', os.path.join(EX, 'example_data', 'example_gwas_sumstats.txt'),
              '--ld-schema', os.path.join(EX, 'ld_mat', 'example_schema.schema'),
              '--seed', '42', '-K', '81', '--init-hg', '0.2', '--samplesizes', '300e3',
              '--names', 'ukbb', '--learn-scaling',
              '--extract', os.path.join(EX, 'keep_variants.txt')]
    run_fit(common + ['--output', out], engine_factory)
    check_frames(pd.read_csv(os.path.join(EX, 'copy_of_example_vilma_run.estimates.tsv'), sep='\t'),
                 pd.read_csv(out + '.estimates.tsv', sep='\t'))
    # example/checkpoint_example.sh: resume from the final model
    out2 = str(tmp_path / 'checkpoint_example_vilma_run')
    run_fit(common + ['--output', out2, '--load-checkpoint', out + '.npz',
                      out + '.covariance.pkl'], engine_factory)
    assert not os.path.exists(out2 + '.covariance.pkl')       # not rewritten on resume
    check_frames(pd.read_csv(os.path.join(EX, 'checkpoint_example_vilma_run.estimates.tsv'), sep='\t'),
                 pd.read_csv(out2 + '.estimates.tsv', sep='\t'))
    truth = np.load(os.path.join(EX, 'checkpoint_example_vilma_run.npz'))
    got = np.load(out2 + '.npz')
    for key in truth.files:
        np.testing.assert_allclose(got[key], truth[key], rtol=1e-5, atol=1e-8, err_msg=key)


def test_cli_fit_golden(tmp_path):
    fit_test_cli(tmp_path, OracleEngine)


def test_cli_fit_golden_svd_manifest(tmp_path):
    fit_test_cli(tmp_path, OracleEngine, manifest='ld_manifest_svd.tsv')


def test_cli_example_and_resume(tmp_path):
    fit_example(tmp_path, OracleEngine)


def test_cli_checkpoints_written(tmp_path):
    out = str(tmp_path / 'ck')
    run_fit(['--ld-schema', os.path.join(REF, 'ld_manifest.tsv'),
             '--sumstats', os.path.join(REF, 'good_sumstats_beta.tsv'), '--output', out,
             '-K', '10', '--num-its', '4', '--checkpoint-freq', '2',
             '--extract', os.path.join(REF, 'good_variants.tsv')], OracleEngine)
    for it in (0, 2):
        ck = np.load('%s-checkpoint.%d.npz' % (out, it))
        assert sorted(ck.files) == ['error_scaling', 'hyper_delta', 'scalings', 'vi_delta', 'vi_mu']
        assert ck['vi_mu'].shape == (12, 1, 13) and ck['vi_delta'].shape == (13, 12)


def test_cli_argument_errors(tmp_path):
    with pytest.raises(NotImplementedError):
        run_fit(['--ld-schema', 'a', '--sumstats', 'b', '--output', 'o', '--extract', 'e',
                 '--trait'], OracleEngine)
    with pytest.raises(ValueError):
        run_fit(['--ld-schema', 'a,b,c', '--sumstats', 'b', '--output', 'o', '--extract', 'e'],
                OracleEngine)
    with pytest.raises(SystemExit):
        frontend.main(['make_ld_schema'])
