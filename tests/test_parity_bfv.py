"""Parity of the BFV-side kernels (ScaleAndRound family, BEHZ trio) against the oracle; emulator on CPU,
the HIP library with -m gpu.  Bit-exact, including ScaleAndRound's double-precision rounding term."""
import json
import os

import numpy as np
import pytest

import libs
from openfhe_amd import fhe_hip as fh
from test_parity import params


def mu128(o, mods):
    out = np.zeros((len(mods), 2), np.uint64)
    for j, m in enumerate(mods):
        t = np.zeros(2, np.uint64)
        o.orc_barrett_mu128(int(m), t)
        out[j] = t
    return out


@pytest.mark.parametrize("logN,sizeI,sizeO,outputFirst,fscale,B",
                         [(6, 3, 2, 1, 1.0, 2), (10, 4, 3, 0, 1.0, 1), (12, 3, 2, 1, 2.0 ** 60, 1), (12, 7, 8, 0, 1.0, 2),
                          # round 6: more than 64 limbs on either side (the deep BFV sets of TestMultiplicativeDepthLimitation: 65 + 64)
                          (5, 65, 64, 0, 1.0, 1), (4, 70, 90, 1, 1.0, 1)])
def test_scale_and_round(backend, oracle, logN, sizeI, sizeO, outputFirst, fscale, B):
    o = oracle
    N, L = 1 << logN, sizeI + sizeO
    rng = np.random.default_rng(21)
    q, psi = params(o, logN, L)
    ctx = fh.Context(backend, logN, q, psi)
    off = 0 if outputFirst else sizeI
    out_idx = np.arange(off, off + sizeO, dtype=np.uint32)
    om = q[off:off + sizeO].copy()
    tab = np.stack([rng.integers(0, int(m), size=sizeI + 1, dtype=np.uint64) for m in om])
    frac = rng.random(sizeI) * fscale
    x = libs.rand_tower(rng, q, N, B)
    x[0, :, 0] = 0
    x[0, :, 1] = q - np.uint64(1)
    want = np.zeros((B, sizeO, N), np.uint64)
    for b in range(B):
        o.orc_scale_and_round(x[b], sizeI, sizeO, N, outputFirst, tab, frac, om, mu128(o, om), want[b])
    plan = fh.ScaleAndRoundPlan(ctx, sizeI, out_idx, tab, frac)
    assert np.array_equal(plan.run(ctx.tower(x, fmt=fh.COEFFICIENT), outputFirst).to_host(), want)
    plan.close()
    if not outputFirst:  # ApproxScaleAndRound: input basis first, output basis last, no fractional part
        wanta = np.zeros((B, sizeO, N), np.uint64)
        for b in range(B):
            o.orc_approx_scale_and_round(x[b], sizeI, sizeO, N, tab, om, mu128(o, om), wanta[b])
        plana = fh.ScaleAndRoundPlan(ctx, sizeI, out_idx, tab, None)
        assert np.array_equal(plana.run(ctx.tower(x, fmt=fh.COEFFICIENT), 0).to_host(), wanta)
        plana.close()
    ctx.close()


def test_times_q_over_t_and_mod_switch_round(backend, oracle):
    """fhe_times_q_over_t (DCRTPolyImpl::TimesQovert) and fhe_mod_switch_round (SetValuesModSwitch) vs the oracle"""
    import ctypes as C
    o = oracle
    rng = np.random.default_rng(48)
    for logN, L, B, t in [(4, 2, 2, 65537), (11, 3, 1, 2), (12, 4, 2, 786433), (13, 2, 1, 65537)]:
        N = 1 << logN
        q, psi = np.zeros(L, np.uint64), np.zeros(L, np.uint64)
        o.orc_dcrt_params(2 * N, L, 58, q, psi)
        ctx = fh.Context(backend, logN, q, psi)
        Q = 1
        for v in q:
            Q *= int(v)
        neg = (t - Q % t) % t
        tinv = np.array([pow(t, -1, int(v)) for v in q], np.uint64)
        x = rng.integers(0, t, (B, L, N), dtype=np.uint64)
        want = x.copy()
        for b in range(B):
            o.orc_times_q_over_t(want[b], L, N, q, t, neg, tinv)
        tw = ctx.tower(x, fmt=fh.COEFFICIENT)
        backend.check(backend.L.fhe_times_q_over_t(ctx.h, tw.ptr, tw.ptr, t, neg, tinv.ctypes.data_as(C.POINTER(C.c_uint64)), None, L, B, None))
        assert np.array_equal(tw.to_host(), want)
        y = rng.integers(0, int(q[0]), (1, 1, N), dtype=np.uint64)
        y[0, 0, :2] = (0, q[0] - np.uint64(1))
        wy = np.zeros(N, np.uint64)
        o.orc_set_values_mod_switch(y[0, 0], N, int(q[0]), int(q[L - 1]), wy)
        ty = ctx.tower(y, limb_idx=[0], fmt=fh.COEFFICIENT)
        out = ctx.empty(1, 1, [L - 1], fh.COEFFICIENT)
        backend.check(backend.L.fhe_mod_switch_round(ctx.h, ty.ptr, int(q[0]), int(q[L - 1]), out.ptr, N, None))
        assert np.array_equal(out.to_host()[0, 0], wy)
        ctx.close()


def test_scale_and_round_p_over_q(backend, oracle):
    o = oracle
    rng = np.random.default_rng(22)
    for logN, sizeQ, bits, B in [(5, 2, 40, 2), (12, 4, 60, 1)]:
        N = 1 << logN
        q, psi = params(o, logN, sizeQ + 1, bits)
        ctx = fh.Context(backend, logN, q, psi)
        x = libs.rand_tower(rng, q, N, B)
        pinv = np.array([pow(int(q[sizeQ]), -1, int(q[i])) for i in range(sizeQ)], np.uint64)
        want = np.zeros((B, sizeQ, N), np.uint64)
        for b in range(B):
            o.orc_scale_and_round_p_over_q(x[b], sizeQ, N, q[:sizeQ].copy(), int(q[sizeQ]), pinv, want[b])
        got = fh.scale_and_round_p_over_q(ctx, ctx.tower(x, fmt=fh.COEFFICIENT), np.arange(sizeQ + 1)).to_host()
        assert np.array_equal(got, want)
        ctx.close()


def behz_setup(lib, o, logN, numQ, t, bits=60):
    N = 1 << logN
    q, psiQ = params(o, logN, numQ, bits)
    hb = o.orc_behz_create(N, numQ, q, t)
    nb = o.orc_behz_num_bsk(hb)
    bsk, psiB = np.zeros(nb, np.uint64), np.zeros(nb, np.uint64)
    o.orc_behz_get_bsk(hb, bsk, psiB)
    b2, p2 = lib.behz_bsk(logN, q, t)  # product-side selection of the Bsk basis
    assert np.array_equal(b2, bsk) and np.array_equal(p2, psiB)
    ctx = fh.Context(lib, logN, np.concatenate([q, bsk]), np.concatenate([psiQ, psiB]))
    plan = fh.Behz(ctx, np.arange(numQ), np.arange(numQ, numQ + nb), t)
    return N, q, psiQ, bsk, psiB, hb, ctx, plan


# (numQ = 15: the last size of the register-resident kernels; 16, 20, 40, 63: the wide plans of round 5 — deep BFV parameter sets, VERDICT r4
# item 6 — with the y_i in a per-lane array; 63 Q + 64 Bsk limbs = 127 rows was the largest tower a context of 128 limbs held)
@pytest.mark.parametrize("logN,numQ,t,B", [(4, 2, 65537, 2), (10, 3, 65537, 2), (12, 6, 786433, 1), (5, 15, 65537, 2), (5, 16, 65537, 2),
                                           (6, 20, 786433, 2), (5, 40, 65537, 1), (5, 63, 65537, 1),
                                           # round 6: 64 Q + 65 Bsk limbs = 129 rows (the case the reference's test reaches), and 100 + 101
                                           (5, 64, 65537, 1), (4, 100, 65537, 1)])
def test_behz_trio(backend, oracle, logN, numQ, t, B):
    o = oracle
    rng = np.random.default_rng(23)
    N, q, psiQ, bsk, psiB, hb, ctx, plan = behz_setup(backend, o, logN, numQ, t)
    nb = len(bsk)
    octxQ = o.orc_ctx_create(N, numQ, q, psiQ)
    octxB = o.orc_ctx_create(N, nb, bsk, psiB)
    x = libs.rand_tower(rng, q, N, B)
    for eval_fmt in (False, True):
        xc = x.copy()
        if eval_fmt:  # oracle: to COEFFICIENT first (dcrtpoly-impl.h:1708-1712)
            o.orc_ntt_inv_tower(octxQ, xc, None, numQ, B, 1)
        want_b = np.zeros((B, nb, N), np.uint64)
        for b in range(B):
            o.orc_behz_q_to_bsk_montgomery(hb, xc[b], want_b[b])
        o.orc_ntt_fwd_tower(octxB, want_b, None, nb, B, 1)
        want_q = x.copy()
        if not eval_fmt:
            o.orc_ntt_fwd_tower(octxQ, want_q, None, numQ, B, 1)
        got = plan.FastBaseConvqToBskMontgomery(x, eval_fmt).to_host()
        assert np.array_equal(got[:, :numQ], want_q) and np.array_equal(got[:, numQ:], want_b), f"q->Bsk eval={eval_fmt}"
    allm = np.concatenate([q, bsk])
    y = libs.rand_tower(rng, allm, N, B)
    wy = y.copy()
    for b in range(B):
        o.orc_behz_fast_rns_floorq(hb, wy[b])
    ty = ctx.tower(y, limb_idx=np.arange(numQ + nb), fmt=fh.COEFFICIENT)
    assert np.array_equal(plan.FastRNSFloorq(ty).to_host(), wy), "FastRNSFloorq"
    z = libs.rand_tower(rng, allm, N, B)
    wz = np.zeros((B, numQ, N), np.uint64)
    for b in range(B):
        o.orc_behz_fast_base_conv_sk(hb, z[b], wz[b])
    tz = ctx.tower(z, limb_idx=np.arange(numQ + nb), fmt=fh.COEFFICIENT)
    assert np.array_equal(plan.FastBaseConvSK(tz).to_host(), wz), "FastBaseConvSK"
    plan.close()
    ctx.close()
    o.orc_behz_destroy(hb)


@pytest.mark.parametrize("logN,numQ,t,B", [(4, 2, 65537, 2), (10, 3, 65537, 3), (12, 4, 786433, 1), (6, 20, 65537, 2)])
def test_bfv_eval_mult_behz(backend, oracle, logN, numQ, t, B):
    """fhe_bfv_eval_mult_behz vs the oracle's composite (itself pinned to the reference's scheme-layer EvalMultNoRelin)"""
    o = oracle
    rng = np.random.default_rng(29)
    N, q, psiQ, bsk, psiB, hb, ctx, plan = behz_setup(backend, o, logN, numQ, t)
    call = o.orc_ctx_create(N, numQ + len(bsk), np.concatenate([q, bsk]), np.concatenate([psiQ, psiB]))
    X = [libs.rand_tower(rng, q, N, B) for _ in range(4)]
    want = np.zeros((3, B, numQ, N), np.uint64)
    for b in range(B):
        o.orc_bfv_eval_mult_behz(hb, call, X[0][b], X[1][b], X[2][b], X[3][b], want[0, b], want[1, b], want[2, b])
    T = [ctx.tower(x, limb_idx=np.arange(numQ)) for x in X]
    got = plan.EvalMultNoRelin(*T)
    for k in range(3):
        assert np.array_equal(got[k].to_host(), want[k]), f"product element {k}"
    octxQ = o.orc_ctx_create(N, numQ, q, psiQ)
    got = plan.EvalMultNoRelin(*T, out_eval=True)
    for k in range(3):
        w = want[k].copy()
        o.orc_ntt_fwd_tower(octxQ, w, None, numQ, B, 1)
        assert np.array_equal(got[k].to_host(), w), f"product element {k} (EVALUATION)"
    plan.close()
    ctx.close()
    o.orc_ctx_destroy(call)
    o.orc_ctx_destroy(octxQ)
    o.orc_behz_destroy(hb)


@pytest.mark.parametrize("ring", [64, 1024])
def test_bfv_eval_mult_behz_reference_vectors(backend, ring):
    """the product against ciphertexts produced by the reference's own scheme layer (tests/golden/ref_vectors_bfv.npz,
    generated by tests/golden/make_golden_bfv.py running cc->EvalMultNoRelin): no oracle in the loop"""
    import os
    V = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_vectors_bfv.npz"))
    k = f"bfv{ring}"
    q, psiQ, bsk, psiB, t = V[k + "_q"], V[k + "_psiQ"], V[k + "_bsk"], V[k + "_psiBsk"], int(V[k + "_t"][0])
    A, B, D = V[k + "_a"], V[k + "_b"], V[k + "_d"]
    numQ, nb, logN = len(q), len(bsk), int(ring).bit_length() - 1
    b2, p2 = backend.behz_bsk(logN, q, t)
    assert np.array_equal(b2, bsk) and np.array_equal(p2, psiB)
    ctx = fh.Context(backend, logN, np.concatenate([q, bsk]), np.concatenate([psiQ, psiB]))
    plan = fh.Behz(ctx, np.arange(numQ), np.arange(numQ, numQ + nb), t)
    T = [ctx.tower(x[None], limb_idx=np.arange(numQ)) for x in (A[0], A[1], B[0], B[1])]
    got = plan.EvalMultNoRelin(*T)
    for e in range(3):
        assert np.array_equal(got[e].to_host()[0], D[e]), f"product element {e}"
    plan.close()
    ctx.close()


@pytest.mark.parametrize("logN,nQ,nP,B,inEval,resEval,rev", [(4, 2, 3, 2, 1, 1, 0), (10, 3, 3, 2, 0, 1, 1), (12, 4, 5, 1, 1, 0, 0),
                                                           (12, 3, 4, 1, 0, 0, 1), (13, 2, 3, 1, 1, 1, 1)])
def test_expand_crt_basis(backend, oracle, logN, nQ, nP, B, inEval, resEval, rev):
    """fhe_expand_crt_basis (ExpandCRTBasis / ExpandCRTBasisReverseOrder, the BFV HPS basis extension) vs the oracle"""
    o = oracle
    rng = np.random.default_rng(37)
    N = 1 << logN
    allq, allpsi = params(o, logN, nQ + nP, 58)
    q, p = allq[:nQ], allq[nQ:]
    hatInv, hatPre, hatMod, alpha, qInv, mu = libs.crt_tables(q, p)
    hm_pq = np.ascontiguousarray(hatMod.T)
    ctx = fh.Context(backend, logN, allq, allpsi)
    octx = o.orc_ctx_create(N, nQ + nP, allq, allpsi)
    x = libs.rand_tower(rng, q, N, B)
    want = np.zeros((B, nQ + nP, N), np.uint64)
    for b in range(B):
        o.orc_expand_crt_basis(octx, nQ, nP, x[b], inEval, hatInv, hatPre, hm_pq, alpha, mu, qInv, resEval, rev, want[b])
    conv = fh.Conv(ctx, np.arange(nQ), np.arange(nQ, nQ + nP))
    tin = ctx.tower(x, limb_idx=np.arange(nQ), fmt=fh.EVALUATION if inEval else fh.COEFFICIENT)
    got = conv.ExpandCRTBasis(tin, fh.EVALUATION if resEval else fh.COEFFICIENT, reverse=bool(rev))
    assert np.array_equal(got.to_host(), want)
    conv.close()
    ctx.close()
    o.orc_ctx_destroy(octx)


@pytest.mark.parametrize("logN,nQ,nP,B", [(4, 2, 3, 2), (10, 3, 3, 2), (12, 4, 5, 1)])
def test_fast_expand_crt_basis_p_over_q(backend, oracle, logN, nQ, nP, B):
    """fhe_fast_expand_crt_basis_p_over_q with the HPSPOVERQ tables (custom-table plan + exact plan) vs the oracle"""
    o = oracle
    rng = np.random.default_rng(38)
    N = 1 << logN
    allq, allpsi = params(o, logN, nQ + nP, 58)
    q, pl = allq[:nQ], allq[nQ:]
    m, mpre, qinvp = libs.p_over_q_tables(q, pl)
    hatInv2, hatPre2, hatMod2, alpha2, pInv, muQ = libs.crt_tables(pl, q)
    _, _, _, _, _, muP = libs.crt_tables(q, pl)
    hm2_qp = np.ascontiguousarray(hatMod2.T)
    x = libs.rand_tower(rng, q, N, B)
    want = np.zeros((B, nQ + nP, N), np.uint64)
    for b in range(B):
        o.orc_fast_expand_crt_basis_p_over_q(x[b], nQ, N, q, m, mpre, qinvp, nP, pl, muP, hatInv2, hatPre2, hm2_qp, alpha2, nQ, q,
                                             muQ, pInv, want[b])
    ctx = fh.Context(backend, logN, allq, allpsi)
    to_pl = fh.Conv(ctx, np.arange(nQ), np.arange(nQ, nQ + nP), hat_inv=m, hat_mod=qinvp)
    to_ql = fh.Conv(ctx, np.arange(nQ, nQ + nP), np.arange(nQ))
    tin = ctx.tower(x, limb_idx=np.arange(nQ), fmt=fh.COEFFICIENT)
    got = to_pl.FastExpandCRTBasisPloverQ(to_ql, tin)
    assert np.array_equal(got.to_host(), want)
    to_pl.close(), to_ql.close()
    ctx.close()


@pytest.mark.parametrize("logN,sizeQ,bits,t,B", [(4, 2, 28, 65537, 2), (10, 3, 45, 1 << 20, 2), (12, 3, 60, 65537, 1), (12, 4, 60, 1 << 30, 1),
                                                  (10, 2, 50, 786433, 2), (12, 3, 59, (1 << 34) - 41, 1), (4, 2, 30, (1 << 34) - 41, 2)])
def test_scale_and_round_native(backend, oracle, logN, sizeQ, bits, t, B):
    """fhe_scale_and_round_native (decryption ScaleAndRound -> residues mod t) vs the oracle, every branch family"""
    o = oracle
    rng = np.random.default_rng(41)
    N = 1 << logN
    q, psi = params(o, logN, sizeQ, bits)
    a, b, fr, bf = libs.decrypt_tables(q, t)
    x = libs.rand_tower(rng, q, N, B)
    x[0, :, 0] = q - np.uint64(1)
    want = np.zeros((B, N), np.uint64)
    for bb in range(B):
        o.orc_scale_and_round_native(x[bb], sizeQ, N, q, t, a, b, fr, bf, want[bb])
    ctx = fh.Context(backend, logN, q, psi)
    got = fh.scale_and_round_native(ctx, ctx.tower(x, fmt=fh.COEFFICIENT), t, a, fr, b, bf)
    assert np.array_equal(got, want)
    ctx.close()


@pytest.mark.parametrize("logN,sizeQ,bits,t,B", [(4, 2, 45, 65537, 2), (12, 3, 60, 786433, 1)])
def test_scale_and_round_behz_decrypt(backend, oracle, logN, sizeQ, bits, t, B):
    o = oracle
    rng = np.random.default_rng(42)
    N = 1 << logN
    q, psi = params(o, logN, sizeQ, bits)
    tg, a, b = libs.behz_decrypt_tables(q, t)
    x = libs.rand_tower(rng, q, N, B)
    want = np.zeros((B, N), np.uint64)
    for bb in range(B):
        o.orc_scale_and_round_behz_decrypt(x[bb], sizeQ, N, q, tg, a, b, want[bb])
    ctx = fh.Context(backend, logN, q, psi)
    got = fh.scale_and_round_behz_decrypt(ctx, ctx.tower(x, fmt=fh.COEFFICIENT), tg, a, b)
    assert np.array_equal(got, want)
    ctx.close()


def test_bfv_eval_mult_with_relinearisation_reference_vectors(backend):
    """fhe_bfv_eval_mult_relin_behz against the reference's own cc->EvalMult on BFV/BEHZ ciphertexts with its own HYBRID
    relinearisation key (tests/golden/ref_vectors_bfv.npz): config 5 end to end, no oracle in the loop"""
    import os
    V = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_vectors_bfv.npz"))
    g = lambda k: V["bfvrelin64_" + k]
    N, t, numPartQ = int(g("N")[0]), int(g("t")[0]), int(g("numPartQ")[0])
    q, bsk, p = g("q"), g("bsk"), g("p")
    numQ, sizeP, nb, logN = len(q), len(p), len(bsk), N.bit_length() - 1
    allq = np.concatenate([q, p, bsk])
    allpsi = np.concatenate([g("psiQ"), g("psiP"), g("psiBsk")])
    ctx = fh.Context(backend, logN, allq, allpsi)
    ks = fh.KeySwitchPlan(ctx, numQ, sizeP, numPartQ)
    ks.upload_key(g("keyB"), g("keyA"))
    behz = fh.Behz(ctx, np.arange(numQ), np.arange(numQ + sizeP, numQ + sizeP + nb), t)
    A, B, Cw = g("a"), g("b"), g("c")
    T = [ctx.tower(x[None], limb_idx=np.arange(numQ)) for x in (A[0], A[1], B[0], B[1])]
    c0, c1 = behz.EvalMult(ks, *T)
    assert np.array_equal(c0.to_host()[0], Cw[0]) and np.array_equal(c1.to_host()[0], Cw[1])
    behz.close()
    ks.close()
    ctx.close()
