#!/usr/bin/env python3
"""Generates tests/golden/ref_vectors_bfv.npz by RUNNING THE REFERENCE ITSELF (oracle/_ref/libref_shim.so): two fresh
BFV ciphertexts and their LeveledSHEBFVRNS::EvalMult (BEHZ) product through the reference's scheme layer
(cc->EvalMultNoRelin, src/pke/lib/scheme/bfvrns/bfvrns-leveledshe.cpp:198-445), for config-5-shaped contexts at
small ring dimensions.  Run from the repo root:  python tests/golden/make_golden_bfv.py   (needs ./build.sh ref)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import libs  # noqa: E402
from test_oracle_vs_ref import ref_bfv_relin_session, ref_bfv_session  # noqa: E402

r = libs.load_ref()
out = {}
for ring, t, depth, sms in ((64, 65537, 2, 60), (1024, 786433, 3, 55)):
    h, N, q, pq, bsk, pb, A, B, D = ref_bfv_session(r, ring, t, depth, sms)
    k = f"bfv{ring}"
    out.update({k + "_t": np.array([t], np.uint64), k + "_q": q, k + "_psiQ": pq, k + "_bsk": bsk, k + "_psiBsk": pb,
                k + "_a": A, k + "_b": B, k + "_d": D})
    r.ref_bfv_destroy(h)
# cc->EvalMult (EvalMultNoRelin + HYBRID relinearisation) at N = 64: moduli of Q, Bsk and P, the key, inputs, result
h, S = ref_bfv_relin_session(r, 64, 65537, 2, 60, 2)
out.update({"bfvrelin64_" + k: (np.array([v], np.uint64) if np.isscalar(v) else v) for k, v in S.items()})
r.ref_bfv_destroy(h)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_vectors_bfv.npz"), **out)
print("wrote tests/golden/ref_vectors_bfv.npz with", len(out), "arrays")
