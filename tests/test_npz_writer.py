"""vilma_amd.npz_writer.savez writes what numpy.savez writes (reference vi_options.py:263-265: the
fit's .npz): the same members with the same bytes, in a zip numpy.load and zipfile accept, with
correct CRCs -- here on arrays that cross the writer's piece size and on empty / scalar members."""
import zipfile
import zlib

import numpy as np
import pytest

from vilma_amd import npz_writer


@pytest.mark.parametrize('n1,n2', [(0, 0), (1, 0), (0, 5), (1, 1), (3, 70000), (65536, 1), (123457, 7654321)])
def test_crc32_combine_is_zlibs(n1, n2):
    rng = np.random.default_rng(n1 + 3 * n2)
    a, b = rng.bytes(n1), rng.bytes(n2)
    assert npz_writer.crc32_combine(zlib.crc32(a), zlib.crc32(b), len(b)) == zlib.crc32(a + b)


def test_savez_members_equal_numpys(tmp_path, monkeypatch):
    monkeypatch.setattr(npz_writer, '_PIECE', 1 << 16)       # many pieces, a ragged last one
    rng = np.random.default_rng(0)
    arrays = {
        'vi_mu': rng.normal(size=(7, 2, 1013)),
        'vi_delta': rng.random(size=(1013, 7)),
        'hyper_delta': rng.random(size=(1, 7)),
        'error_scaling': np.ones(2),
        'scalings': np.asfortranarray(rng.normal(size=(2, 1013))),     # not C-contiguous
        'vi_sigma': rng.normal(size=(7, 2, 2, 1013)),
        'empty': np.empty((0, 3)),
        'scalar': np.float64(2.5),
        'ints': np.arange(100000, dtype=np.int32),
    }
    ours = npz_writer.savez(str(tmp_path / 'ours'), threads=3, **arrays)
    assert ours.endswith('ours.npz')
    np.savez(str(tmp_path / 'theirs.npz'), **arrays)
    with zipfile.ZipFile(ours) as z:
        assert z.testzip() is None                            # every member's CRC checks out
        theirs = zipfile.ZipFile(str(tmp_path / 'theirs.npz'))
        assert z.namelist() == theirs.namelist()
        for name in z.namelist():
            assert z.read(name) == theirs.read(name), name    # header and data, byte for byte
            assert z.getinfo(name).compress_type == zipfile.ZIP_STORED
    back = np.load(ours)
    assert sorted(back.files) == sorted(arrays)
    for key, value in arrays.items():
        np.testing.assert_array_equal(back[key], value)
        assert back[key].dtype == np.asarray(value).dtype


def test_savez_refuses_object_arrays(tmp_path):
    with pytest.raises(ValueError):
        npz_writer.savez(str(tmp_path / 'x'), a=np.array([{'k': 1}], dtype=object))
