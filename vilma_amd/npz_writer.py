"""A `numpy.savez` that scales to the fit's outputs.

`vilma fit` ends with `np.savez(output, vi_mu, vi_delta, hyper_delta, error_scaling, scalings,
vi_sigma)` (reference vi_options.py:263-265).  With the default mixture grid (-K 12: 582 components)
and a million SNPs that is 34 GB, and numpy writes it through `zipfile` on one thread -- a CRC-32
over every byte, then a copy into the page cache: 23 s of a 100 s fit on the GPU box
(profiles/r05e_cli_fit_from_disk.txt).  `savez` here writes the SAME file format -- an
uncompressed zip (ZIP64) of `<key>.npy` members, readable by `numpy.load` and by any zip tool --
with the bytes of every array cut into pieces that a thread pool checksums (`zlib.crc32` releases
the GIL) and writes in place (`os.pwrite`), the piece CRCs being combined afterwards the way
zlib's `crc32_combine` does it.
"""
import io
import os
import struct
import time
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np

_PIECE = 64 << 20
_GF2_DIM = 32


def _gf2_times(mat, vec):
    total, i = 0, 0
    while vec:
        if vec & 1:
            total ^= mat[i]
        vec >>= 1
        i += 1
    return total


def _gf2_square(mat):
    return [_gf2_times(mat, mat[n]) for n in range(_GF2_DIM)]


_OPERATORS = {}


def _zeros_operator(length):
    """The GF(2) matrix that advances a CRC-32 register over `length` zero bytes."""
    if length in _OPERATORS:
        return _OPERATORS[length]
    odd = [0xedb88320] + [1 << n for n in range(_GF2_DIM - 1)]      # one zero BIT
    even = _gf2_square(odd)                                          # two bits
    odd = _gf2_square(even)                                          # four bits
    result = None                                                    # identity
    n = length
    while n:
        even = _gf2_square(odd)                                      # first pass: one zero BYTE
        if n & 1:
            result = even if result is None else [_gf2_times(even, col) for col in result]
        n >>= 1
        if not n:
            break
        odd = _gf2_square(even)
        if n & 1:
            result = odd if result is None else [_gf2_times(odd, col) for col in result]
        n >>= 1
    if result is None:
        result = [1 << n for n in range(_GF2_DIM)]
    _OPERATORS[length] = result
    return result


def crc32_combine(crc1, crc2, len2):
    """CRC-32 of A + B from crc32(A), crc32(B) and len(B) (zlib's crc32_combine)."""
    if len2 <= 0:
        return crc1
    return _gf2_times(_zeros_operator(len2), crc1) ^ crc2


def _npy_header(arr):
    buf = io.BytesIO()
    np.lib.format.write_array_header_1_0(buf, np.lib.format.header_data_from_array_1_0(arr))
    return buf.getvalue()


def _dos_time():
    t = time.localtime()
    year = max(t.tm_year, 1980)
    return (t.tm_hour << 11) | (t.tm_min << 5) | (t.tm_sec // 2), ((year - 1980) << 9) | (t.tm_mon << 5) | t.tm_mday


def savez(path, threads=None, **arrays):
    """np.savez(path, **arrays): same file name rule ('.npz' appended when missing), same members,
    same bytes inside the members; written by `threads` workers (default: the CPUs this process
    may use, at most 16)."""
    path = os.fspath(path)
    if not path.endswith('.npz'):
        path += '.npz'
    if threads is None:
        try:
            threads = len(os.sched_getaffinity(0))
        except AttributeError:
            threads = os.cpu_count() or 1
        threads = max(1, min(16, threads))
    dos_time, dos_date = _dos_time()
    members, offset = [], 0
    for key, value in arrays.items():
        arr = np.asanyarray(value)
        if arr.dtype.hasobject:
            raise ValueError('object arrays are not written by this savez (member %r)' % key)
        if not arr.flags.c_contiguous and not arr.flags.f_contiguous:
            arr = np.ascontiguousarray(arr)
        name = (key + '.npy').encode('utf-8')
        header = _npy_header(arr)           # (a Fortran-ordered array is stored as such, like numpy does)
        if not arr.flags.c_contiguous:
            arr = arr.T
        size = len(header) + arr.nbytes
        lfh_len = 30 + len(name) + 20
        members.append(dict(name=name, arr=arr, header=header, size=size, lfh=offset,
                            data=offset + lfh_len + len(header)))
        offset += lfh_len + size
    cd_offset = offset

    fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    try:
        tasks = []
        for m in members:
            raw = memoryview(m['arr'].reshape(-1).view(np.uint8)) if m['arr'].nbytes else memoryview(b'')
            m['pieces'] = []
            for lo in range(0, len(raw), _PIECE):
                tasks.append((m, raw[lo:lo + _PIECE], m['data'] + lo))

        def work(task):
            m, piece, where = task
            done = 0
            while done < len(piece):
                done += os.pwrite(fd, piece[done:], where + done)
            return m, where, zlib.crc32(piece), len(piece)

        with ThreadPoolExecutor(max_workers=threads) as pool:
            for m, where, crc, n in pool.map(work, tasks):
                m['pieces'].append((where, crc, n))
        for m in members:
            crc = zlib.crc32(m['header'])
            for _, piece_crc, n in sorted(m['pieces']):
                crc = crc32_combine(crc, piece_crc, n)
            m['crc'] = crc & 0xffffffff
            extra = struct.pack('<HHQQ', 1, 16, m['size'], m['size'])
            lfh = struct.pack('<IHHHHHIIIHH', 0x04034b50, 45, 0, 0, dos_time, dos_date, m['crc'],
                              0xffffffff, 0xffffffff, len(m['name']), len(extra)) + m['name'] + extra
            os.pwrite(fd, lfh + m['header'], m['lfh'])
        central = b''
        for m in members:
            extra = struct.pack('<HHQQQ', 1, 24, m['size'], m['size'], m['lfh'])
            central += struct.pack('<IHHHHHHIIIHHHHHII', 0x02014b50, 45, 45, 0, 0, dos_time, dos_date,
                                   m['crc'], 0xffffffff, 0xffffffff, len(m['name']), len(extra), 0, 0, 0,
                                   0o600 << 16, 0xffffffff) + m['name'] + extra
        n = len(members)
        end64 = struct.pack('<IQHHIIQQQQ', 0x06064b50, 44, 45, 45, 0, 0, n, n, len(central), cd_offset)
        locator = struct.pack('<IIQI', 0x07064b50, 0, cd_offset + len(central), 1)
        end = struct.pack('<IHHHHIIH', 0x06054b50, 0, 0, min(n, 0xffff), min(n, 0xffff),
                          min(len(central), 0xffffffff), 0xffffffff, 0)
        os.pwrite(fd, central + end64 + locator + end, cd_offset)
    finally:
        os.close(fd)
    return path
