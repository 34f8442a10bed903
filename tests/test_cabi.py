"""The drop-in boundary: libfhe_hip.so loads (no GPU needed) and exports every symbol include/fhe_hip.h declares;
argument errors come back as status codes with the reference's messages; no compute without a device."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "fhe_hip.h")
SO = os.path.join(ROOT, "openfhe-development_amd", "csrc", "libfhe_hip.so")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fhe_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_boundary():
    syms = declared_symbols()
    for must in ("fhe_ctx_create", "fhe_ntt_fwd", "fhe_ntt_inv", "fhe_add", "fhe_mul", "fhe_automorph",
                 "fhe_approx_switch_basis", "fhe_switch_basis_exact", "fhe_keyswitch_hybrid", "fhe_ckks_eval_mult",
                 "fhe_approx_mod_down", "fhe_rescale"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    if not os.path.exists(SO):
        import subprocess
        subprocess.check_call([os.path.join(ROOT, "build.sh"), "hip"])
    lib = C.CDLL(SO)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, f"libfhe_hip.so does not export: {missing}"
    lib.fhe_version.restype = C.c_char_p
    assert b"gfx950" in lib.fhe_version()


def test_product_library_does_not_link_test_code():
    """the product must not depend on the oracle, the reference build or the emulator"""
    import subprocess
    out = subprocess.run(["ldd", SO], capture_output=True, text=True).stdout
    for bad in ("fhe_oracle", "ref_shim", "OPENFHE", "fhe_emu"):
        assert bad not in out
    syms = subprocess.run(["nm", "-D", "--defined-only", SO], capture_output=True, text=True).stdout
    assert "orc_" not in syms and "fhe_emu" not in syms


def test_no_device_means_loud_failure_not_fallback():
    """without a GPU (this container) context creation must fail with FHE_ERR_DEVICE — there is no CPU path"""
    lib = C.CDLL(SO)
    lib.fhe_device_count.restype = C.c_int
    if lib.fhe_device_count() > 0:
        pytest.skip("a HIP device is present")
    from openfhe_amd import fhe_hip as fh
    L = fh.Lib()
    q, psi = L.dcrt_chain(6, 2, 40)  # host-side helpers work without a device
    with pytest.raises(fh.FheError, match="no HIP device"):
        fh.Context(L, 6, q, psi)


def test_host_parameter_helpers_match_oracle(oracle):
    from openfhe_amd import fhe_hip as fh
    L = fh.Lib()
    for logN, n, bits in ((3, 3, 28), (12, 2, 60), (16, 3, 60)):
        q, psi = L.dcrt_chain(logN, n, bits)
        q2, psi2 = np.zeros(n, np.uint64), np.zeros(n, np.uint64)
        oracle.orc_dcrt_params(2 << logN, n, bits, q2, psi2)
        assert np.array_equal(q, q2) and np.array_equal(psi, psi2)
    q, _ = L.ckks_like_chain(12, 7, 60, 50)
    p, pp = L.select_p(12, q, 3)
    P, PP = np.zeros(64, np.uint64), np.zeros(64, np.uint64)
    n = oracle.orc_hybrid_select_p(4096, 7, q, 3, 60, P, PP)
    assert n == len(p) and np.array_equal(P[:n], p) and np.array_equal(PP[:n], pp)
    for m in (8, 1 << 13, 1 << 17):  # FindAutomorphismIndex2nComplex (the oracle's is pinned on the reference's)
        for index in (0, 1, -1, 2, 7, -13, 100, m - 1):
            assert L.find_automorphism_index(index, m) == oracle.orc_find_automorphism_index_2n_complex(index, m)
    with pytest.raises(fh.FheError, match="power of two"):
        L.find_automorphism_index(3, 24)
