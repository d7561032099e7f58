"""Identity stand-in for numba: decorators return the function unchanged (oracle/shims/README.md)."""


def _identity_decorator(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]

    def wrap(fn):
        return fn

    return wrap


jit = njit = vectorize = guvectorize = _identity_decorator
prange = range
__version__ = "0.0-shim"
