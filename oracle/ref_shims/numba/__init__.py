"""Identity JIT decorators: the decorated numpy code runs as plain numpy (see README.md)."""


def jit(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]
    return lambda fn: fn


prange = range


class NumbaPerformanceWarning(Warning):
    pass
