"""Stand-in for numba (not installed here): decorators become no-ops so the
reference's jitted bodies run as the plain Python they are written in.
Used ONLY by tests/golden/make_golden.py in the build container."""


def njit(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]
    return lambda fn: fn


prange = range
