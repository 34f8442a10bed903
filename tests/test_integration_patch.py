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


def ensure_patched_build():
    shim.ensure_built()
    if os.path.exists(PATCHED) and os.path.getmtime(PATCHED) >= os.path.getmtime(os.path.join(INTEG, "with_hip.patch")):
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
