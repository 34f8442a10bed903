import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import libs
    return libs.load_oracle()


@pytest.fixture(scope="session")
def ref():
    """The reference itself (oracle/_ref/libref_shim.so); built here when /root/reference is mounted."""
    import libs
    if not libs.have_ref():
        if os.path.isdir("/root/reference/src"):
            subprocess.check_call([os.path.join(ROOT, "build.sh"), "ref"])
        else:
            pytest.skip("oracle/_ref not built and /root/reference not present")
    return libs.load_ref()


def _emu_lib():
    from openfhe_amd import fhe_hip as fh
    so = os.path.join(ROOT, "tests", "emu", "libfhe_emu.so")
    srcs = [os.path.join(ROOT, "openfhe-development_amd", "csrc", f)
            for f in os.listdir(os.path.join(ROOT, "openfhe-development_amd", "csrc")) if f.endswith((".h", ".cpp"))]
    srcs += [os.path.join(ROOT, "tests", "emu", f) for f in ("emu_runtime.cpp", "emu_runtime.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call([os.path.join(ROOT, "build.sh"), "emu"])
    return fh.Lib(so)


def _hip_lib():
    from openfhe_amd import fhe_hip as fh
    lib = fh.Lib()  # raises if libfhe_hip.so is missing: the GPU tests must run the native HIP path
    assert "emulator" not in lib.version()
    if lib.device_count() < 1:
        raise fh.FheError("no HIP device visible: -m gpu tests need a real MI355X")
    return lib


@pytest.fixture(scope="session", params=["emu", pytest.param("hip", marks=pytest.mark.gpu)])
def backend(request):
    """`emu` = TEST-ONLY lane emulator build of the same kernels (CPU, checks index logic);
    `hip` = the product library on a real GPU (the parity tests proper)."""
    return _emu_lib() if request.param == "emu" else _hip_lib()


@pytest.fixture(scope="session")
def hip():
    return _hip_lib()
