"""csrc/sincos_glibc.h on the host against the host libm: the device evaluates sin/cos of latitudes with this exact
operation sequence, so the legacy areas and centroid integrals carry the reference's bits.  Required: identical results
on millions of arguments over the supported range, including the branch boundaries and the table nodes."""
import ctypes as C
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "hostcheck", "sincos_check.cpp")
OUT = os.path.join(ROOT, "tests", "hostcheck", "_build", "libsincos_check.so")


@pytest.fixture(scope="module")
def chk():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    inc = os.path.join(ROOT, "fre-nctools_amd", "csrc")
    srcs = [SRC, os.path.join(inc, "sincos_glibc.h"), os.path.join(inc, "sincos_table.h")]
    if not os.path.exists(OUT) or any(os.path.getmtime(s) > os.path.getmtime(OUT) for s in srcs):
        subprocess.check_call(["g++", "-O1", "-fno-builtin", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-I", inc, SRC, "-o", OUT, "-lm"])
    L = C.CDLL(OUT)
    L.sincos_check.argtypes = [C.c_long, C.c_long, C.POINTER(C.c_long), C.POINTER(C.c_long), C.POINTER(C.c_double)]
    L.sincos_check.restype = C.c_long
    L.sincos_check_fused.argtypes = [C.c_long, C.c_long, C.POINTER(C.c_double)]
    L.sincos_check_fused.restype = C.c_long
    return L


def _host_has_fma():
    try:
        return " fma " in open("/proc/cpuinfo").read()
    except OSError:
        return True


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_sin_cos_bit_identical_to_host_libm(chk, seed):
    """fgs_sin / fgs_cos are libm's sin() / cos() as an FMA-capable x86-64 host runs them (the multiarch FMA build)."""
    if not _host_has_fma():
        pytest.skip("host CPU without FMA: libm runs its uncontracted sin/cos here")
    bs, bc, fb = C.c_long(0), C.c_long(0), C.c_double(0)
    bad = chk.sincos_check(6000000, seed, C.byref(bs), C.byref(bc), C.byref(fb))
    assert bad == 0, (bs.value, bc.value, fb.value)


@pytest.mark.parametrize("seed", [4, 5])
def test_sincos_bit_identical_to_host_libm(chk, seed):
    """fgs_sincos is libm's sincos() (uncontracted), the call gcc emits for sin(a) and cos(a) of one argument."""
    fb = C.c_double(0)
    bad = chk.sincos_check_fused(6000000, seed, C.byref(fb))
    assert bad == 0, fb.value
