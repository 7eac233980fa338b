import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_report_header(config):
    """Which libm the bit-identity assertions of the legacy path are judged against (tests/orc.py:host_has_fma): a host whose
    libm does not evaluate sin/cos the way csrc/sincos_glibc.h does relaxes them to 1e-10 -- make that visible in the log."""
    import platform
    try:
        fma = " fma " in open("/proc/cpuinfo").read()
    except OSError:
        fma = None
    libc = " ".join(platform.libc_ver())
    mode = "bit-identical areas / centroids asserted" if fma else "RELAXED to 1e-10 (host libm without the FMA sin/cos build)"
    return [f"fregrid-hip parity: host libc {libc}, cpu fma={fma}: {mode}"]


def load_package():
    """The package directory is called ``fre-nctools_amd`` (not an identifier): import it as fre_nctools_amd."""
    if "fre_nctools_amd" in sys.modules:
        return sys.modules["fre_nctools_amd"]
    pkg_dir = os.path.join(ROOT, "fre-nctools_amd")
    if not os.path.exists(os.path.join(pkg_dir, "libfregrid_hip.so")):
        # fresh checkout: build artefacts are git-ignored.  hipcc cross-compiles gfx950 without a GPU.
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(pkg_dir, "csrc"), "-j4"], stdout=subprocess.DEVNULL)
    spec = importlib.util.spec_from_file_location("fre_nctools_amd", os.path.join(pkg_dir, "__init__.py"),
                                                  submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["fre_nctools_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session")
def fg():
    return load_package()


@pytest.fixture(scope="session")
def gpu_ok(fg):
    n = fg.lib().fg_device_count()
    if n < 1:
        pytest.fail("GPU test selected but no HIP device is visible: " + fg._lib.last_error())
    return n
