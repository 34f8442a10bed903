"""The 8-residues-per-lane row pass (ntt_row8.h) against the oracle, every tile shape (1 / 2 / 4 / 8 waves per tile), on the lane
emulator (CPU: its C++ butterflies follow the generated plan and abort on a lazy-range violation) and on the GPU.

The library chooses the row kernel per pass shape (row8 for 9..11 stages, the 16-residue kernel for 12); FHE_NTT_ROW8=1 forces row8 for
12 stages too and FHE_NTT_T1=5 moves a stage of the 2^16 ring to the column pass.  Both are read once per process, so the forced shapes
run in a child process."""
import os
import subprocess
import sys

import numpy as np
import pytest

import libs
from openfhe_amd import fhe_hip as fh

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np, libs
from openfhe_amd import fhe_hip as fh
lib = fh.Lib({so!r})
o = libs.load_oracle()
rng = np.random.default_rng(61)
for logN, L, B in {shapes!r}:
    N, M = 1 << logN, 2 << logN
    q = [o.orc_last_prime(60, M)]
    for _ in range(L - 1):
        q.append(o.orc_previous_prime(q[-1], M))
    q = np.array(q, np.uint64)
    psi = np.array([o.orc_root_of_unity(M, int(v)) for v in q], np.uint64)
    ctx = fh.Context(lib, logN, q, psi)
    octx = o.orc_ctx_create(N, L, q, psi)
    x = libs.rand_tower(rng, q, N, B)
    x[0, :, 0] = 0
    x[0, :, 1] = q - np.uint64(1)
    want = x.copy()
    o.orc_ntt_fwd_tower(octx, want, None, L, B, 0)
    t = ctx.tower(x, fmt=fh.COEFFICIENT)
    t.SwitchFormat()
    assert np.array_equal(t.to_host(), want), f"forward mismatch logN={{logN}}"
    t.SwitchFormat()
    assert np.array_equal(t.to_host(), x), f"round trip mismatch logN={{logN}}"
    y = libs.rand_tower(rng, q, N, B)
    wanti = y.copy()
    o.orc_ntt_inv_tower(octx, wanti, None, L, B, 0)
    t2 = ctx.tower(y, fmt=fh.EVALUATION)
    t2.SwitchFormat()
    assert np.array_equal(t2.to_host(), wanti), f"inverse mismatch logN={{logN}}"
    ctx.close()
n = lib.launch_count("ntt_row8_kernel")
assert n >= {min_launches}, f"the row8 kernel ran {{n}} times"
print("ok", n)
"""


def run_child(so, shapes, env, min_launches):
    code = CHILD.format(root=ROOT, so=so, shapes=shapes, min_launches=min_launches)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_row8_default_shapes_run_row8(backend):
    """rings 2^13..2^15 take the row8 kernel by default (1, 2 and 4 waves per tile)"""
    before = backend.launch_count("ntt_row8_kernel")
    o = libs.load_oracle()
    rng = np.random.default_rng(62)
    for logN in (13, 14, 15):
        N, M = 1 << logN, 2 << logN
        q = np.array([o.orc_last_prime(60, M), o.orc_last_prime(45, M)], np.uint64)
        psi = np.array([o.orc_root_of_unity(M, int(v)) for v in q], np.uint64)
        ctx = fh.Context(backend, logN, q, psi)
        octx = o.orc_ctx_create(N, 2, q, psi)
        x = libs.rand_tower(rng, q, N, 1)
        want = x.copy()
        o.orc_ntt_fwd_tower(octx, want, None, 2, 1, 0)
        t = ctx.tower(x, fmt=fh.COEFFICIENT)
        t.SwitchFormat()
        assert np.array_equal(t.to_host(), want)
        t.SwitchFormat()
        assert np.array_equal(t.to_host(), x)
        o.orc_ctx_destroy(octx)
        ctx.close()
    assert backend.launch_count("ntt_row8_kernel") - before == 6


def test_row8_forced_shapes_on_emulator():
    """8 waves per tile (12 stages: 2^16 and 2^17) and the 5 + 11 split of 2^16, moduli of 60 and 36..45 bits (ladder reductions)"""
    so = os.path.join(ROOT, "tests", "emu", "libfhe_emu.so")
    run_child(so, [(16, 2, 1), (17, 1, 1)], {"FHE_NTT_ROW8": "1"}, 6)
    run_child(so, [(16, 1, 2)], {"FHE_NTT_ROW8": "1", "FHE_NTT_T1": "5"}, 3)


@pytest.mark.gpu
def test_row8_forced_shapes_on_gpu():
    so = os.path.join(ROOT, "openfhe-development_amd", "csrc", "libfhe_hip.so")
    run_child(so, [(16, 4, 3), (17, 2, 2)], {"FHE_NTT_ROW8": "1"}, 6)
    run_child(so, [(16, 3, 2)], {"FHE_NTT_ROW8": "1", "FHE_NTT_T1": "5"}, 3)
