"""csrc/fp80.h on the host against the CPU's own x87 arithmetic: every emulated operation must be bit-identical to
`long double`; the acosl replacement may differ from glibc's only by a final double rounding, rarely."""
import ctypes as C
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "hostcheck", "fp80_check.cpp")
OUT = os.path.join(ROOT, "tests", "hostcheck", "_build", "libfp80_check.so")


@pytest.fixture(scope="module")
def chk():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    inc = os.path.join(ROOT, "fre-nctools_amd", "csrc")
    srcs = [SRC, os.path.join(inc, "fp80.h"), os.path.join(inc, "atan_table.h")]
    if not os.path.exists(OUT) or any(os.path.getmtime(s) > os.path.getmtime(OUT) for s in srcs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-I", inc, SRC, "-o", OUT, "-lm"])
    L = C.CDLL(OUT)
    L.fp80_check_ops.argtypes = [C.c_long, C.c_int, C.POINTER(C.c_long)]
    L.fp80_check_ops.restype = C.c_long
    L.fp80_check_acosl.argtypes = [C.c_long, C.POINTER(C.c_long)]
    L.fp80_check_acosl.restype = C.c_long
    L.fp80_check_acosl_fast.argtypes = [C.c_long]
    L.fp80_check_acosl_fast.restype = C.c_long
    L.fp80_check_edges.argtypes = [C.c_long, C.POINTER(C.c_long)]
    L.fp80_check_edges.restype = C.c_long
    return L


def test_divide_and_sqrt_on_crafted_significands(chk):
    """extreme / exact / just-off-exact significands, both exponent parities, for the estimate-and-correct
    x80_div and x80_sqrt"""
    fails = (C.c_long * 2)()
    bad = chk.fp80_check_edges(2000000, fails)
    assert bad == 0, list(fails)      # [div, sqrt]


@pytest.mark.parametrize("spread", [4, 40, 600])
def test_extended_ops_bit_identical_to_x87(chk, spread):
    fails = (C.c_long * 8)()
    bad = chk.fp80_check_ops(400000, spread, fails)
    assert bad == 0, list(fails)      # [operands, add, sub, mul, div, sqrt, to_double, abs_lt]


def test_acosl_matches_glibc_up_to_rare_final_rounding(chk):
    n = 2000000
    ulp1 = C.c_long(0)
    bad = chk.fp80_check_acosl(n, C.byref(ulp1))
    assert bad == ulp1.value          # any difference is a last-place double rounding
    assert bad <= n * 2e-3, bad


def test_short_series_acosl_equals_the_long_series(chk):
    """fg_acosl takes a 7-term series with a double tail and keeps its result only when it is provably the nearest double;
    fg_acosl_long (13 double-double terms) is what it falls back to.  They must agree everywhere (random arguments, next to
    +-1, next to the table angles, results next to powers of two); dd_from_x80_pos must equal dd_from_x80."""
    assert chk.fp80_check_acosl_fast(6000000) == 0
