#!/usr/bin/env python3
"""Generates tests/golden/ref_vectors_lt.npz by RUNNING THE REFERENCE ITSELF (oracle/_ref/libref_shim.so):
FHECKKSRNS::EvalLinearTransform (src/pke/lib/scheme/ckksrns/ckksrns-fhe.cpp:1832-1882; BSGS with double hoisting) on a
fresh sparsely packed CKKS ciphertext, with the reference's own rotation keys and EvalLinearTransformPrecompute diagonals.
Run from the repo root:  python tests/golden/make_golden_lt.py   (needs ./build.sh ref)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import libs  # noqa: E402
from test_oracle_vs_ref import ref_coeffs_to_slots_session, ref_linear_transform_session  # noqa: E402

r, o = libs.load_ref(), libs.load_oracle()
h, lt, ct, res, S = ref_linear_transform_session(r, o, 256, 3, 45, 55, 2, 8, 4)
out = {}
for k, v in S.items():
    if k == "keys":
        rots = sorted(v)
        out["rots"] = np.array(rots, np.int32)
        out["rotK"] = np.array([v[i][0] for i in rots], np.uint32)
        out["keyB"] = np.stack([v[i][1] for i in rots])
        out["keyA"] = np.stack([v[i][2] for i in rots])
    else:
        out[k] = np.array([v], np.uint64) if np.isscalar(v) else v
# what the result decrypts to (documentation of the fixture: the matrix-vector product of the reference's own inputs)
dec = np.zeros(2 * 8)
r.ref_ckks_decrypt_complex(h, res, dec, 8)
out["decrypted"] = dec
want = S["matrix"] @ S["values"]
assert np.abs(dec[0::2] - want).max() < 1e-4, (dec[0::2], want)  # the reference's result is the matrix-vector product
r.ref_ckks_lt_destroy(lt)
r.ref_ckks_destroy(h)
# FHECKKSRNS::EvalCoeffsToSlots (:1884-2040), fully packed N = 64, level budget 2: a main level (g = 8, b = 2) and a
# remainder level (g = 4, b = 2) with a rescale between them
h, S = ref_coeffs_to_slots_session(r, 64, 4, 2, 2)
for k in ("N", "q", "psiQ", "p", "psiP", "numPartQ", "sizeQl", "c", "out"):
    out["c2s_" + k] = np.array([S[k]], np.uint64) if np.isscalar(S[k]) else S[k]
rots = sorted(S["keys"])
out["c2s_rots"] = np.array(rots, np.int32)
out["c2s_rotK"] = np.array([S["keys"][i][0] for i in rots], np.uint32)
out["c2s_keyB"] = np.stack([S["keys"][i][1] for i in rots])
out["c2s_keyA"] = np.stack([S["keys"][i][2] for i in rots])
out["c2s_levels"] = np.array([len(S["levels"])], np.uint32)
for n, (s_, rot_in, rot_out, terms) in enumerate(S["levels"]):
    limbs = next(iter(S["diags"][n].values())).shape[0]
    d = np.zeros((len(rot_out), len(rot_in), limbs, S["N"]), np.uint64)
    present = np.zeros((len(rot_out), len(rot_in)), np.uint8)
    for (i, j), rows in S["diags"][n].items():
        d[i, j], present[i, j] = rows, 1
    out[f"c2s_l{n}_rot_in"], out[f"c2s_l{n}_rot_out"] = np.array(rot_in, np.int32), np.array(rot_out, np.int32)
    out[f"c2s_l{n}_diag"], out[f"c2s_l{n}_present"] = d, present
r.ref_ckks_destroy(h)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_vectors_lt.npz"), **out)
print("wrote tests/golden/ref_vectors_lt.npz with", len(out), "arrays")
