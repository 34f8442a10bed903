#!/usr/bin/env python3
"""Generates tests/golden/ref_vectors.npz by RUNNING THE REFERENCE ITSELF (oracle/_ref/libref_shim.so, the
reference's own sources compiled by oracle/Makefile) in the build container.  The vectors pin everything no
reference unit test pins bit-for-bit: ApproxSwitchCRTBasis, SwitchCRTBasis, HYBRID key switch /
EvalMult limb values, DropLastElementAndScale, plus parameter generation and NTT words.
Run from the repo root:  python tests/golden/make_golden.py     (needs /root/reference -> ./build.sh ref)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import libs  # noqa: E402

r = libs.load_ref()
out = {}
rng = np.random.default_rng(20260923)

# 1. parameter chains: ILDCRTParams(order, depth, bits)
for order, L, bits in ((16, 3, 28), (8192, 2, 60), (131072, 4, 60)):
    q = np.zeros(L, np.uint64)
    psi = np.zeros(L, np.uint64)
    r.ref_dcrt_params(order, L, bits, q, psi)
    out[f"params_{order}_{L}_{bits}_q"] = q
    out[f"params_{order}_{L}_{bits}_psi"] = psi

# 2. NTT words, N = 64 and 1024, 60-bit
for N in (64, 1024):
    q = np.zeros(2, np.uint64)
    psi = np.zeros(2, np.uint64)
    r.ref_dcrt_params(2 * N, 2, 60, q, psi)
    x = libs.rand_residues(rng, q[1], N)
    y = x.copy()
    r.ref_ntt(q[1], psi[1], N, y, 0)
    z = libs.rand_residues(rng, q[1], N)
    zi = z.copy()
    r.ref_ntt(q[1], psi[1], N, zi, 1)
    out[f"ntt{N}_q"] = q[1:2]
    out[f"ntt{N}_psi"] = psi[1:2]
    out[f"ntt{N}_in"] = x
    out[f"ntt{N}_fwd"] = y
    out[f"ntt{N}_evalin"] = z
    out[f"ntt{N}_inv"] = zi

# 3. CKKS EvalMult + HYBRID key switch + rescale, N = 64 (FIXEDMANUAL), ciphertexts at level 0 and level 1
h = r.ref_ckks_create(64, 4, 40, 50, 2, 0)
info = np.zeros(5, np.uint32)
r.ref_ckks_info(h, info)
N, sizeQ, sizeP, numPartQ, alpha = map(int, info)
q = np.zeros(sizeQ, np.uint64)
psiQ = q.copy()
p = np.zeros(sizeP, np.uint64)
psiP = p.copy()
r.ref_ckks_get_moduli(h, q, psiQ, p, psiP)
keyB = np.zeros((numPartQ, sizeQ + sizeP, N), np.uint64)
keyA = keyB.copy()
r.ref_ckks_get_relin_key(h, keyB, keyA)
out.update(ckks_info=info, ckks_q=q, ckks_psiQ=psiQ, ckks_p=p, ckks_psiP=psiP, ckks_keyB=keyB, ckks_keyA=keyA)
for lvl in (0, 1):
    c1 = r.ref_ckks_encrypt(h, 11 + lvl, lvl)
    c2 = r.ref_ckks_encrypt(h, 21 + lvl, lvl)
    ci = np.zeros(4, np.uint32)
    r.ref_ct_info(h, c1, ci)
    sizeQl = int(ci[1])

    def ex(ct, e, L=sizeQl):
        a = np.zeros((L, N), np.uint64)
        r.ref_ct_export(h, ct, e, a)
        return a
    cm = r.ref_ckks_eval_mult(h, c1, c2)
    rs = r.ref_ckks_rescale(h, cm)
    out[f"ckks_l{lvl}_a0"], out[f"ckks_l{lvl}_a1"] = ex(c1, 0), ex(c1, 1)
    out[f"ckks_l{lvl}_b0"], out[f"ckks_l{lvl}_b1"] = ex(c2, 0), ex(c2, 1)
    out[f"ckks_l{lvl}_c0"], out[f"ckks_l{lvl}_c1"] = ex(cm, 0), ex(cm, 1)
    out[f"ckks_l{lvl}_r0"], out[f"ckks_l{lvl}_r1"] = ex(rs, 0, sizeQl - 1), ex(rs, 1, sizeQl - 1)
r.ref_ckks_destroy(h)

# 4. basis conversions with explicit tables, N = 32, 3 -> 4 limbs of 60 bits
N, nS, nD = 32, 3, 4
q = np.zeros(nS + nD, np.uint64)
psi = q.copy()
r.ref_dcrt_params(2 * N, nS + nD, 60, q, psi)
src, dst = q[:nS].copy(), q[nS:].copy()
hatInv = np.zeros(nS, np.uint64)
hatMod = np.zeros((nS, nD), np.uint64)
for i in range(nS):
    hh = 1
    for k in range(nS):
        if k != i:
            hh = hh * int(src[k]) % int(src[i])
    hatInv[i] = pow(hh, -1, int(src[i]))
    for j in range(nD):
        v = 1
        for k in range(nS):
            if k != i:
                v = v * int(src[k]) % int(dst[j])
        hatMod[i, j] = v
Q = 1
for s in src:
    Q *= int(s)
alpha = np.array([[(a * Q) % int(pp) for pp in dst] for a in range(nS + 1)], np.uint64)
qinv = np.array([1.0 / float(int(s)) for s in src], np.float64)
x = libs.rand_tower(rng, src, N)
# adversarial coefficients for the double-precision overflow count: all residues near q-1 and near 0
x[:, 0] = src - np.uint64(1)
x[:, 1] = 0
x[:, 2] = src // np.uint64(2)
approx = np.zeros((nD, N), np.uint64)
exact = np.zeros((nD, N), np.uint64)
r.ref_approx_switch_crt_basis(N, nS, src, psi[:nS].copy(), x, hatInv, hatMod, nD, dst, psi[nS:].copy(), approx)
r.ref_switch_crt_basis(N, nS, src, psi[:nS].copy(), x, hatInv, np.ascontiguousarray(hatMod.T), alpha, nD, dst,
                       psi[nS:].copy(), qinv, exact)
out.update(conv_q=q, conv_psi=psi, conv_x=x, conv_approx=approx, conv_exact=exact)

np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_vectors.npz"), **out)
print("wrote tests/golden/ref_vectors.npz with", len(out), "arrays")
