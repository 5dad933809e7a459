"""ORACLE — TEST INFRASTRUCTURE ONLY.  The ONE reference configuration the fixtures are made under.

Import this module BEFORE numpy in every fixture generator.  numpy 1.26.4 ships Intel SVML kernels for
float32 / float64 power, cbrt, exp ... and dispatches to them on CPUs with AVX512_SKX; their results differ from
the C library's in the last bit of 20-40 % of the values and follow the CPU model of the host.  The reference
configuration switches that dispatch off (numpy's documented NPY_DISABLE_CPU_FEATURES), so that np.power /
np.cbrt on float32 arrays are glibc 2.35's powf / cbrtf — what any host without AVX-512 computes, and what
oracle/glibc_flt32.h and the HIP kernel restate bit for bit.  check() proves the switch took effect.
"""
import os
import sys

FEATURES = 'AVX512F AVX512CD AVX512_KNL AVX512_KNM AVX512_SKX AVX512_CLX AVX512_CNL AVX512_ICL AVX512_SPR'
assert 'numpy' not in sys.modules, 'import refconfig before numpy'
os.environ['NPY_DISABLE_CPU_FEATURES'] = FEATURES


def check():
    """np.power(float32, 2.4) and np.cbrt(float32) must BE libm's powf(x, 2.4f) / cbrtf(x) here."""
    import ctypes
    import numpy as np
    libm = ctypes.CDLL('libm.so.6')
    libm.powf.restype = ctypes.c_float
    libm.powf.argtypes = [ctypes.c_float, ctypes.c_float]
    libm.cbrtf.restype = ctypes.c_float
    libm.cbrtf.argtypes = [ctypes.c_float]
    x = np.random.RandomState(0).uniform(0.01, 300.0, 20000).astype(np.float32)
    p = np.power(x, 2.4)
    c = np.cbrt(x)
    assert p.dtype == np.float32 and c.dtype == np.float32
    assert all(libm.powf(float(v), 2.4) == float(r) for v, r in zip(x, p)), 'np.power is not libm powf'
    assert all(libm.cbrtf(float(v)) == float(r) for v, r in zip(x, c)), 'np.cbrt is not libm cbrtf'
    gnu = ctypes.CDLL('libc.so.6').gnu_get_libc_version
    gnu.restype = ctypes.c_char_p
    return 'numpy %s with NPY_DISABLE_CPU_FEATURES="%s" (float32 power / cbrt = glibc %s powf / cbrtf, checked)' \
        % (np.__version__, FEATURES, gnu().decode())
