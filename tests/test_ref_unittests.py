"""The reference's OWN pke unit tests (src/pke/unittest: CKKS, BFV, BGV, bootstrapping, automorphisms, EvalMult, SHE, PRE, multiparty,
encodings ... — compiled where they lie, unmodified) on the HIP backend of lbcrypto::DCRTPoly ("lets the reference's
scheme tests run on the GPU path").  googletest is an empty submodule of the reference tree and is not installed here; the tests are
built on tests/hal/minigtest (a small stand-in for the part of googletest's interface they use) by tests/hal/Makefile.ut, once against
the stock libraries (oracle/_ref) and once against openfhe-development_amd/hal/_build.  Left out: tests that need a working
serialisation library (cereal is un-vendored) or read their case list from a .csv inside the reference tree.

CPU suite: the core library's lattice-layer tests (src/core/unittest: DCRTPoly itself, transforms, matrices, trapdoors ...) and a slice
of the pke tests on the lane emulator (the emulator runs workgroups sequentially: the full set would take hours).
GPU suite: all of them (1589 pke + 140 core tests) on the MI355X (about 90 s); the same binary built against the stock libraries passes all of them too."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = os.path.join(ROOT, "tests", "hal", "_build")
UT_HIP, UT_STOCK = os.path.join(B, "ut_hip"), os.path.join(B, "ut_stock")
UT_HIP_BIN, UT_STOCK_BIN = os.path.join(B, "ut_hip_binfhe"), os.path.join(B, "ut_stock_binfhe")
EMU = os.path.join(ROOT, "tests", "emu", "libfhe_emu.so")
HIP = os.path.join(ROOT, "openfhe-development_amd", "csrc", "libfhe_hip.so")


def ensure_built():
    if all(os.path.exists(p) for p in (UT_HIP, UT_STOCK, UT_HIP_BIN, UT_STOCK_BIN)):
        return
    if os.path.isdir("/root/reference/src"):
        subprocess.check_call([os.path.join(ROOT, "build.sh"), "hal"])
    else:
        pytest.skip("tests/hal/_build/ut_* not present and /root/reference not mounted")


def run(exe, flt, lib=None, threads=4, timeout=900):
    env = dict(os.environ, OMP_NUM_THREADS=str(threads))
    env.pop("FHE_HAL_ALLOW_HOST", None)
    if lib:
        env["FHE_HIP_LIB"] = lib
    r = subprocess.run([exe, f"--gtest_filter={flt}"], env=env, capture_output=True, text=True, timeout=timeout, cwd="/tmp")
    out = r.stdout + r.stderr
    m = re.search(r"(\d+) tests ran, (\d+) passed, (\d+) failed, (\d+) skipped", out)
    assert m, out[-2000:]
    ran, passed, failed, skipped = map(int, m.groups())
    h = re.search(r"hal: available (\d+) deviceOps (\d+) hostOps (\d+)", out)
    run.host_ops = int(h.group(3)) if h else 0  # fall-backs of members on rings the device takes (rings < 16 and host-produced words are counted apart)
    c = re.search(r"halcomposite calls (\d+) checksIdentical (\d+) checksDiffered (\d+)", out)
    run.composite = tuple(int(v) for v in c.groups()) if c else (0, 0, 0)
    # per member: (device operations, host-mirror executions, host reads); and why device plans / contexts were declined
    run.members = {m.group(1): tuple(int(m.group(i)) for i in (2, 3, 4)) for m in re.finditer(r"halmember (\S+) (\d+) (\d+) (\d+)", out)}
    run.declines = {m.group(1).strip(): int(m.group(2)) for m in re.finditer(r"haldecline (.*) x(\d+)", out)}
    return ran, passed, failed, (int(h.group(2)) if h else 0), out


SLICE = "*RELIN_TEST*:*EVAL_MULT_ERROR_HANDLING_0*:*UTGENERAL_ENCODING*"
# the core library's unit tests of the lattice layer (src/core/unittest): UnitTestDCRTElements exercises the DCRTPoly class itself.
# UTBinInt.GetInternalRepresentation depends on the limb width the dynamic big integer was configured with and fails on the stock
# build of this container in the same way: excluded everywhere.
CORE = ("UTDCRTPoly*:UTPoly*:UTNTT*:UTTransform*:UTLatticeParams*:UTMatrix*:UTTrapdoor*:UTField2n*:UTNbTheory*:UTmubintvec*:UTBinVect*:"
        "UTBinInt*-UTBinInt.GetInternalRepresentation")


def test_reference_unit_tests_slice_on_emulator():
    ensure_built()
    ran_s, passed_s, failed_s, _, out_s = run(UT_STOCK, SLICE)
    assert ran_s >= 20 and failed_s == 0, out_s[-1500:]
    ran, passed, failed, dev_ops, out = run(UT_HIP, SLICE, EMU)
    assert (ran, passed, failed) == (ran_s, passed_s, 0), out[-1500:]
    assert dev_ops > 500, "the HIP backend's device path did not run"


# src/binfhe/unittest (binaries of their own): NativePoly callers, must not notice the DCRTPoly backend the libraries were built with
def test_reference_binfhe_unit_tests_on_backend_libraries():
    ensure_built()
    flt = "-UnitTestFHEDeep*"  # (the *_VERY_LONG chains of gates: many minutes of CPU time on any backend)
    ran, passed, failed, _, out = run(UT_HIP_BIN, flt, EMU, threads=8)  # (the stock build of the same sources passes the same 73)
    assert ran >= 70 and (passed, failed) == (ran, 0), [l for l in out.split("\n") if "FAILED" in l][:10]


def test_reference_core_lattice_unit_tests_on_emulator():
    ensure_built()
    ran_s, passed_s, failed_s, _, out_s = run(UT_STOCK, CORE)
    assert ran_s >= 130 and failed_s == 0, out_s[-1500:]
    ran, passed, failed, dev_ops, out = run(UT_HIP, CORE, EMU)
    assert (ran, passed, failed) == (ran_s, passed_s, 0), [l for l in out.split("\n") if "FAILED" in l][:10]
    assert dev_ops > 500
    # per member, as on the GPU (HOST_ALLOW below): the core lattice suite needs NO host-mirror execution of any member on a ring the
    # device library takes (round 5: measured zero for every member; a member that appears here has lost its device path)
    on_mirror = {m: v[1] for m, v in run.members.items() if v[1] > 0}
    assert run.members and not on_mirror and run.host_ops == 0, (on_mirror, run.host_ops)
    assert not run.declines, run.declines


# Host-mirror executions per member over the reference's 1729 unit tests on the MI355X: the committed upper bounds (this round's record,
# profiles/r06_ref_unittests_trace.txt: the halmember lines).  The mirror is the reference's own class — what runs there proves nothing — so every member is
# bounded by name; a member that is not listed may not run on the mirror at all.
HOST_ALLOW = {
    # (KeySwitchBV — the BV key-switching technique of PRE, multiparty and the BV variants of the scheme tests — ran its digit decomposition
    # DCRTPoly::CRTDecompose 1772 times and, on the host-resident digits, EvalMult.KeySwitchAccumulate 749 times on the mirror until
    # round 6: fhe_crt_decompose cuts and lifts the digits on the device; no entry is left for either)
    # words produced or read on the host by pke itself: FHECKKSRNS::KeySwitchSparse fills limbs with SetElementAtIndex and adds them;
    # UnitTestMultipartyAborts reads limbs; PackedEncoding of a prime-cyclotomic plaintext transforms on the host
    "SetElementAtIndex": 128, "GetAllElements": 96, "operator+=": 64, "AssembleRows": 10, "SwitchFormat": 8,
    # (BFVrns_TestMultiplicativeDepthLimitation_{BEHZ,HPS,...} — ring dimension 32, multiplicative depths 32 ... 135, up to 129 distinct
    # moduli in one operation — ran 4 + 3 + 3 + 4 + 3 + 3 + 8 members on the mirror until round 5: a device context now holds 256 limbs,
    # the wide BEHZ plans 127 Q limbs, the ScaleAndRound plans 255 + 256: no entry, and no decline reason, is left for them)
    # key generation for an OLD key that carries more limbs than [P]_q has entries: the reference's TimesNoCheck leaves the trailing limbs
    # of its result unfilled (dcrtpoly-impl.h:594-601), a tower that has no device form, and the AssembleRows that follows takes the mirror
    # too (TimesNoCheck 50 + AssembleRows 62 inside the backend's KeySwitchGenInternal; every other key generation: zero)
    "KeySwitchGenInternal": 112,
}
DECLINE_REASONS = ()  # round 6: no device plan / context may be declined over the reference's unit tests


@pytest.mark.gpu
def test_reference_unit_tests_on_gpu():
    ensure_built()
    ran, passed, failed, dev_ops, out = run(UT_HIP, "-*SERIALIZE*:UTBinInt.GetInternalRepresentation", HIP, threads=8, timeout=600)
    failures = [l for l in out.split("\n") if "FAILED" in l][:20]
    assert failed == 0 and passed == ran, failures
    assert ran >= 1700, f"only {ran} tests ran"
    assert dev_ops > 1_000_000
    # the host mirror is the reference's own class: what ran there proves nothing.  Fall-backs of members (on rings the device library
    # takes) stay below 5 % of the device operations (round 3: 4063 against 1.76 M, profiles/r03_ref_unittests_trace.txt; round 2: 2.3 M
    # against 1.6 M), and no first-use check of a batched composite against the member-by-member path may differ
    over = {m: (v[1], HOST_ALLOW.get(m, 0)) for m, v in run.members.items() if v[1] > HOST_ALLOW.get(m, 0)}
    assert run.members and not over, f"host-mirror executions above the committed per-member bounds (got, bound): {over}"
    assert run.host_ops <= sum(HOST_ALLOW.values()), (run.host_ops, sum(HOST_ALLOW.values()))
    unknown = [r for r in run.declines if not (DECLINE_REASONS and r.startswith(DECLINE_REASONS))]
    assert not unknown, f"device plans / contexts declined for reasons that are not the documented domain limits: {unknown}"
    assert run.composite[0] > 1000 and run.composite[2] == 0, run.composite
