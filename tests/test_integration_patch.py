"""integration/with_hip.patch — the SOURCE-LEVEL form of the backend's pke / core hooks: applied to a copy of the
reference's sources, OpenFHE builds against the HIP backend of DCRTPoly with plain compiler flags (no objcopy on mangled names, no
-fno-inline-functions), every hooked function is bound to the backend's definition (checked by hal/Makefile on the linked library),
and the shim programs produce ciphertexts identical to the stock backend's with the hooked composites actually running
(must-run members: KeySwitchCore, the accumulate composite of EvalMult, the bootstrap transforms)."""
import os
import subprocess

import pytest

import test_hal_shim as shim

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INTEG = os.path.join(ROOT, "integration")
PATCHED = os.path.join(INTEG, "_build", "shim_ckks_hip_patched")


def backend_sources_mtime():
    """newest modification time of what the patched build is made of: the patch AND the backend's own sources / headers (round-5 advisor:
    comparing with the patch alone ran a stale backend after an edit of hip-runtime.cpp or dcrtpoly-hip.h)"""
    import glob
    hal = os.path.join(ROOT, "openfhe-development_amd", "hal")
    files = [os.path.join(INTEG, "with_hip.patch"), os.path.join(INTEG, "build_patched.sh"), os.path.join(ROOT, "include", "fhe_hip.h"),
             os.path.join(ROOT, "tests", "hal", "shim_ckks.cpp")]
    for pat in ("*.cpp", "*.h", "Makefile", "lattice/**/*.h", "math/**/*.h"):
        files += glob.glob(os.path.join(hal, pat), recursive=True)
    return max(os.path.getmtime(f) for f in files if os.path.exists(f))


def ensure_patched_build():
    shim.ensure_built()
    if os.path.exists(PATCHED) and os.path.getmtime(PATCHED) >= backend_sources_mtime():
        return
    if not os.path.isdir("/root/reference/src"):
        pytest.skip("integration/_build not present and /root/reference not mounted")
    subprocess.check_call([os.path.join(INTEG, "build_patched.sh")])


def test_patch_is_what_the_generator_writes(tmp_path):
    """the committed patch is the output of integration/make_patch.py on the mounted reference (skipped without it)"""
    if not os.path.isdir("/root/reference/src"):
        pytest.skip("/root/reference not mounted")
    r = subprocess.run(["python", os.path.join(INTEG, "make_patch.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_cmake_option_of_the_patch_configures(tmp_path):
    """the WITH_HIP block the patch adds to the reference's top-level CMakeLists.txt, run by cmake in a project of its own: refuses to
    configure without OPENFHE_HIP_DIR or with another math backend, and with them puts the backend's directory FIRST on the include
    path and defines WITH_HIP / FHE_HIP_PATCHED_PKE (the reference's own configure needs its un-vendored submodules, so the block is
    exercised outside it; the source lists of src/core and src/pke are checked textually)"""
    import shutil
    if shutil.which("cmake") is None:
        pytest.skip("cmake not installed")
    patch = open(os.path.join(INTEG, "with_hip.patch")).read()
    top = patch.split("+++ b/CMakeLists.txt")[1].split("+++ b/src/core/CMakeLists.txt")[0]
    added = "".join(l[1:] + "\n" for l in top.splitlines() if l.startswith("+") and not l.startswith("+++"))
    assert "option(WITH_HIP" in added and "include_directories(BEFORE ${OPENFHE_HIP_DIR})" in added
    assert "list(APPEND CORE_SRC_FILES ${OPENFHE_HIP_DIR}/hip-runtime.cpp)" in patch
    assert "list(APPEND PKE_SRC_FILES ${OPENFHE_HIP_DIR}/keyswitch-hybrid-hip.cpp)" in patch
    src = tmp_path / "proj"
    src.mkdir()
    (src / "CMakeLists.txt").write_text(
        "cmake_minimum_required(VERSION 3.5)\nproject(t NONE)\nif(NOT MATHBACKEND)\n set(MATHBACKEND 4)\nendif()\nset(NATIVE_SIZE 64)\n" + added +
        "get_directory_property(D COMPILE_DEFINITIONS)\nget_directory_property(I INCLUDE_DIRECTORIES)\n"
        "file(WRITE ${CMAKE_BINARY_DIR}/seen.txt \"${D}\\n${I}\\n\")\n")
    hal = os.path.join(ROOT, "openfhe-development_amd", "hal")

    def configure(*defs):
        b = tmp_path / ("b%d" % len(os.listdir(tmp_path)))
        r = subprocess.run(["cmake", "-S", str(src), "-B", str(b), *defs], capture_output=True, text=True)
        return r, b
    r, _ = configure("-DWITH_HIP=ON")
    assert r.returncode != 0 and "OPENFHE_HIP_DIR" in r.stderr
    r, _ = configure("-DWITH_HIP=ON", f"-DOPENFHE_HIP_DIR={hal}", "-DMATHBACKEND=2")
    assert r.returncode != 0 and "MATHBACKEND 4" in r.stderr
    r, b = configure("-DWITH_HIP=ON", f"-DOPENFHE_HIP_DIR={hal}")
    assert r.returncode == 0, r.stdout + r.stderr
    seen = (b / "seen.txt").read_text().splitlines()
    assert "WITH_HIP" in seen[0] and "FHE_HIP_PATCHED_PKE" in seen[0]
    assert seen[1].split(";")[0] == hal and os.path.exists(os.path.join(seen[1].split(";")[0], "lattice", "hal", "lat-backend.h"))
    r, b = configure()  # default OFF: nothing changes
    assert r.returncode == 0 and "WITH_HIP" not in (b / "seen.txt").read_text()


def patched_check(tmp_path, mode, logN, expect, must_run, extra=(), device_lib=None, threads=1):
    ensure_patched_build()
    so, sh = str(tmp_path / "stock.bin"), str(tmp_path / "patched.bin")
    out_stock = shim.run(shim.PROGS[0], so, mode, logN, extra=extra, threads=threads)
    out_hip = shim.run(PATCHED, sh, mode, logN, device_lib or shim.EMU, extra=extra, threads=threads)
    a, b = open(so, "rb").read(), open(sh, "rb").read()
    assert len(a) > 1000 and a == b, "the build from the patched sources differs from the default backend"
    shim.assert_ran_on_device(out_hip, must_run)
    import re
    cm = re.search(r"halcomposite calls (\d+) checksIdentical (\d+) checksDiffered (\d+)", out_hip)
    assert cm and int(cm.group(1)) > 0 and int(cm.group(3)) == 0, out_hip[-600:]
    for name, want in expect.items():
        got = shim.values(out_hip, name)
        assert len(got) == len(want) and all(abs(g - w) < 1e-3 for g, w in zip(got, want)), (name, got, want)


def test_patched_sources_leveled_ckks_on_emulator(tmp_path):
    patched_check(tmp_path, "leveled", 11, shim.LEVELED, shim.CKKS_MEMBERS)


def test_patched_sources_bootstrap_on_emulator(tmp_path):
    patched_check(tmp_path, "bootstrap", 10, shim.BOOT, shim.BOOT_MEMBERS)


# the same build on the MI355X (round 5: the build the benchmark's cc->EvalMult and bootstrap legs prefer)
@pytest.mark.gpu
def test_patched_sources_leveled_ckks_on_gpu(tmp_path):
    patched_check(tmp_path, "leveled", 14, shim.LEVELED, shim.CKKS_MEMBERS, device_lib=shim.HIP)


@pytest.mark.gpu
def test_patched_sources_bootstrap_on_gpu(tmp_path):
    patched_check(tmp_path, "bootstrap", 13, shim.BOOT, shim.BOOT_MEMBERS, device_lib=shim.HIP)


@pytest.mark.gpu
def test_patched_sources_lockstep_batch_on_gpu(tmp_path):
    """cc->EvalMult on 32 ciphertexts in lockstep groups of 12 (wide towers, windows of one allocation): every product's digest and the
    first and last product byte for byte equal to the stock backend's"""
    patched_check(tmp_path, "multbatch", 14, {"product 0": [0.5, 0.0, -3.0]}, (), extra=(8, 32, 1, 12), device_lib=shim.HIP)
