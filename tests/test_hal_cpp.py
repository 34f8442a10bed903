"""The C++ host mirror (hal/dcrtpoly_hip.h) compiles against include/fhe_hip.h and passes a DCRTPoly-style smoke
test: on CPU linked against the TEST-ONLY emulator build, with -m gpu against the HIP library."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(libdir, libname, tmp_path):
    exe = str(tmp_path / "hal_smoke")
    subprocess.check_call(["g++", "-std=c++17", "-O1", os.path.join(ROOT, "tests", "hal_smoke.cpp"), "-o", exe,
                           f"-L{libdir}", f"-l{libname}", f"-Wl,-rpath,{libdir}", "-lpthread"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, f"hal_smoke failed with code {out.returncode}: {out.stdout} {out.stderr}"
    assert "hal_smoke OK" in out.stdout


def test_hal_cpp_on_emulator(backend, tmp_path):
    if "emulator" not in backend.version():
        pytest.skip("emulator variant")
    _run(os.path.join(ROOT, "tests", "emu"), "fhe_emu", tmp_path)


@pytest.mark.gpu
def test_hal_cpp_on_gpu(hip, tmp_path):
    _run(os.path.join(ROOT, "openfhe-development_amd", "csrc"), "fhe_hip", tmp_path)
