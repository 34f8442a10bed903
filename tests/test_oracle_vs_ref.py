"""The oracle against the REFERENCE ITSELF (oracle/_ref, compiled from /root/reference by oracle/Makefile) on
fresh random inputs.  Skipped where the reference build is not available (e.g. a box without /root/reference
and without a travelled oracle/_ref)."""
import numpy as np
import pytest

import libs


def test_scalar_ops(oracle, ref):
    rng = np.random.default_rng(31)
    for bits in (20, 40, 59, 60):
        q = oracle.orc_last_prime(bits, 1 << 10)
        assert q == ref.ref_last_prime(bits, 1 << 10)
        assert oracle.orc_compute_mu(q) == ref.ref_compute_mu(q)
        for _ in range(200):
            a, b = (int(v) for v in rng.integers(0, q, size=2, dtype=np.uint64))
            pre = oracle.orc_prep_mod_mul_const(b, q)
            assert pre == ref.ref_prep_mod_mul_const(b, q)
            assert oracle.orc_mod_mul_fast_const(a, b, q, pre) == ref.ref_mod_mul_fast_const(a, b, q) == a * b % q
            assert oracle.orc_mod_mul_fast(a, b, q, oracle.orc_compute_mu(q)) == ref.ref_mod_mul_fast(a, b, q)
            lo, hi = (int(v) for v in rng.integers(0, 1 << 63, size=2, dtype=np.uint64))
            mu = np.zeros(2, np.uint64)
            oracle.orc_barrett_mu128(q, mu)
            assert oracle.orc_barrett128(lo, hi, q, int(mu[0]), int(mu[1])) == ref.ref_barrett128(lo, hi, q) \
                == ((hi << 64) | lo) % q


def test_primes_roots_automaps(oracle, ref):
    for bits, m in ((28, 16), (50, 1 << 13), (60, 1 << 17)):
        assert oracle.orc_first_prime(bits, m) == ref.ref_first_prime(bits, m)
        q = oracle.orc_last_prime(bits, m)
        assert oracle.orc_previous_prime(q, m) == ref.ref_previous_prime(q, m)
        assert oracle.orc_next_prime(q - 2 * m if False else oracle.orc_previous_prime(q, m), m) == q
        assert oracle.orc_root_of_unity(m, q) == ref.ref_root_of_unity(m, q)
    for n, k in ((8, 3), (64, 5), (4096, 2 * 4096 - 1), (4096, 3125)):
        a = np.zeros(n, np.uint32)
        b = np.zeros(n, np.uint32)
        oracle.orc_precompute_auto_map(n, k, a)
        ref.ref_precompute_auto_map(n, k, b)
        assert np.array_equal(a, b)
    for i in (1, 7, -3, 100):
        assert oracle.orc_find_automorphism_index_2n_complex(i, 1 << 13) == ref.ref_find_automorphism_index_2n_complex(i, 1 << 13)


@pytest.mark.parametrize("logN,L", [(3, 2), (4, 1), (8, 2), (12, 2)])
def test_tower_switch_format_and_arith(oracle, ref, logN, L):
    o, r = oracle, ref
    N = 1 << logN
    rng = np.random.default_rng(32)
    q = np.zeros(L, np.uint64)
    psi = np.zeros(L, np.uint64)
    o.orc_dcrt_params(2 * N, L, 60 if logN > 4 else 28, q, psi)
    x = libs.rand_tower(rng, q, N, 2)
    y = libs.rand_tower(rng, q, N, 2)
    octx = o.orc_ctx_create(N, L, q, psi)
    h = r.ref_towers_create(N, L, q, psi, x, 2, 0)
    g = r.ref_towers_create(N, L, q, psi, y, 2, 1)
    r.ref_towers_switch_format(h)  # COEFF -> EVAL
    want = np.zeros_like(x)
    r.ref_towers_export(h, want)
    got = x.copy()
    o.orc_ntt_fwd_tower(octx, got, None, L, 2, 0)
    assert np.array_equal(got, want)
    r.ref_towers_mul_eq(h, g)
    r.ref_towers_export(h, want)
    prod = np.zeros_like(x)
    for b in range(2):
        for l in range(L):
            o.orc_vec_mul(prod[b, l], got[b, l], y[b, l], N, q[l])
    assert np.array_equal(prod, want)
    r.ref_towers_switch_format(h)  # EVAL -> COEFF
    r.ref_towers_export(h, want)
    o.orc_ntt_inv_tower(octx, prod, None, L, 2, 0)
    assert np.array_equal(prod, want)
    r.ref_towers_destroy(h)
    r.ref_towers_destroy(g)
    o.orc_ctx_destroy(octx)


def test_automorphism_and_switch_modulus(oracle, ref):
    o, r = oracle, ref
    rng = np.random.default_rng(33)
    N, L = 64, 2
    q = np.zeros(L, np.uint64)
    psi = np.zeros(L, np.uint64)
    o.orc_dcrt_params(2 * N, L, 50, q, psi)
    x = libs.rand_tower(rng, q, N)
    x[:, 5] = 0
    for k in (3, 5, 127):
        for evalfmt in (1, 0):
            want = np.zeros_like(x)
            r.ref_automorph(N, L, q, psi, x, want, k, evalfmt, 0)
            got = np.zeros_like(x)
            for l in range(L):
                if evalfmt:
                    o.orc_automorph_eval_k(got[l], x[l], N, k)
                else:
                    o.orc_automorph_coeff(got[l], x[l], N, k, q[l])
            assert np.array_equal(got, want)
            if evalfmt:
                r.ref_automorph(N, L, q, psi, x, want, k, 1, 1)  # precomputed-table overload
                pre = np.zeros(N, np.uint32)
                o.orc_precompute_auto_map(N, k, pre)
                for l in range(L):
                    o.orc_automorph_eval(got[l], x[l], N, pre)
                assert np.array_equal(got, want)
    for oldq, newq in ((int(q[0]), int(q[1])), (int(q[1]), int(q[0])), (int(q[0]), 65537), (65537, int(q[0]))):
        v = rng.integers(0, oldq, size=257, dtype=np.uint64)
        v[0], v[1], v[2] = 0, oldq - 1, oldq // 2
        a, b = v.copy(), v.copy()
        o.orc_switch_modulus(a, len(a), oldq, newq)
        r.ref_switch_modulus(b, len(b), oldq, newq)
        assert np.array_equal(a, b)


def test_ckks_eval_mult_keyswitch_against_live_reference(oracle, ref):
    """fresh context + keys from the reference's own KeyGen; the oracle must reproduce EvalMult limb values"""
    o, r = oracle, ref
    h = r.ref_ckks_create(1 << 10, 5, 45, 55, 3, 0)
    info = np.zeros(5, np.uint32)
    r.ref_ckks_info(h, info)
    N, sizeQ, sizeP, numPartQ, alpha = map(int, info)
    q, psiQ = np.zeros(sizeQ, np.uint64), np.zeros(sizeQ, np.uint64)
    p, psiP = np.zeros(sizeP, np.uint64), np.zeros(sizeP, np.uint64)
    r.ref_ckks_get_moduli(h, q, psiQ, p, psiP)
    hy = o.orc_hybrid_create(N, sizeQ, q, psiQ, sizeP, p, psiP, numPartQ)
    A, B, Cc = np.zeros(sizeQ, np.uint64), np.zeros(sizeP, np.uint64), np.zeros((sizeP, sizeQ), np.uint64)
    r.ref_ckks_get_tables(h, A, B, Cc)
    A2, B2, C2 = A.copy(), B.copy(), Cc.copy()
    o.orc_hybrid_get_PInvModq(hy, A2)
    o.orc_hybrid_get_PHatInvModp(hy, B2)
    o.orc_hybrid_get_PHatModq(hy, C2)
    assert np.array_equal(A, A2) and np.array_equal(B, B2) and np.array_equal(Cc, C2)
    keyB = np.zeros((numPartQ, sizeQ + sizeP, N), np.uint64)
    keyA = keyB.copy()
    r.ref_ckks_get_relin_key(h, keyB, keyA)
    c1, c2 = r.ref_ckks_encrypt(h, 5, 1), r.ref_ckks_encrypt(h, 6, 1)
    ci = np.zeros(4, np.uint32)
    r.ref_ct_info(h, c1, ci)
    sizeQl = int(ci[1])

    def ex(ct, e, L=sizeQl):
        a = np.zeros((L, N), np.uint64)
        r.ref_ct_export(h, ct, e, a)
        return a
    cm = r.ref_ckks_eval_mult(h, c1, c2)
    o0, o1 = np.zeros((sizeQl, N), np.uint64), np.zeros((sizeQl, N), np.uint64)
    o.orc_ckks_eval_mult_relin(hy, ex(c1, 0), ex(c1, 1), ex(c2, 0), ex(c2, 1), sizeQl, keyB, keyA, o0, o1)
    assert np.array_equal(o0, ex(cm, 0)) and np.array_equal(o1, ex(cm, 1))
    rs = r.ref_ckks_rescale(h, cm)
    octx = o.orc_ctx_create(N, sizeQ, q, psiQ)
    f = np.zeros((sizeQl - 1, N), np.uint64)
    o.orc_drop_last_element_and_scale(octx, o0, sizeQl, f)
    assert np.array_equal(f, ex(rs, 0, sizeQl - 1))
    # sanity: the reference decrypts its own product to (approximately) the product of the messages
    vals = np.zeros(4, np.float64)
    r.ref_ckks_decrypt(h, cm, vals, 4)
    assert np.all(np.isfinite(vals))
    o.orc_ctx_destroy(octx)
    o.orc_hybrid_destroy(hy)
    r.ref_ckks_destroy(h)


@pytest.mark.parametrize("ring,t,depth,sms", [(16, 65537, 2, 60), (64, 65537, 3, 55), (1024, 786433, 4, 60)])
def test_behz_against_live_reference(oracle, ref, ring, t, depth, sms):
    """the reference's own CryptoParametersBFVRNS tables + DCRTPoly BEHZ members vs the oracle's derived tables"""
    o, r = oracle, ref
    rng = np.random.default_rng(41)
    h = r.ref_bfv_create(ring, t, depth, sms, 0)  # MultiplicationTechnique BEHZ = 0
    info = np.zeros(3, np.uint32)
    r.ref_bfv_info(h, info)
    N, numQ, numBsk = map(int, info)
    q, pq = np.zeros(numQ, np.uint64), np.zeros(numQ, np.uint64)
    bsk, pb = np.zeros(numBsk, np.uint64), np.zeros(numBsk, np.uint64)
    r.ref_bfv_get_moduli(h, q, pq, bsk, pb)
    hb = o.orc_behz_create(N, numQ, q, t)
    b2, p2 = np.zeros(numBsk, np.uint64), np.zeros(numBsk, np.uint64)
    o.orc_behz_get_bsk(hb, b2, p2)
    assert np.array_equal(b2, bsk) and np.array_equal(p2, pb)
    x = libs.rand_tower(rng, q, N)
    want = np.zeros((numQ + numBsk, N), np.uint64)
    r.ref_bfv_behz_q_to_bsk(h, x, 0, want)
    got = np.zeros((numBsk, N), np.uint64)
    o.orc_behz_q_to_bsk_montgomery(hb, x, got)
    cb = o.orc_ctx_create(N, numBsk, bsk, pb)
    o.orc_ntt_fwd_tower(cb, got, None, numBsk, 1, 1)
    assert np.array_equal(got, want[numQ:])
    allm = np.concatenate([q, bsk])
    y = libs.rand_tower(rng, allm, N)
    y2 = y.copy()
    r.ref_bfv_fast_rns_floorq(h, y)
    o.orc_behz_fast_rns_floorq(hb, y2)
    assert np.array_equal(y, y2)
    z = libs.rand_tower(rng, allm, N)
    w, w2 = np.zeros((numQ, N), np.uint64), np.zeros((numQ, N), np.uint64)
    r.ref_bfv_fast_base_conv_sk(h, z, w)
    o.orc_behz_fast_base_conv_sk(hb, z, w2)
    assert np.array_equal(w, w2)
    o.orc_ctx_destroy(cb)
    o.orc_behz_destroy(hb)
    r.ref_bfv_destroy(h)


@pytest.mark.parametrize("logN,sizeQl,t,ev", [(4, 3, 65537, 1), (6, 4, 786433, 0), (10, 3, 2, 1)])
def test_mod_reduce_against_live_reference(oracle, ref, logN, sizeQl, t, ev):
    """DCRTPoly::ModReduce (BGV modulus switch) of the reference vs the oracle"""
    o, r = oracle, ref
    rng = np.random.default_rng(77)
    N = 1 << logN
    q, psi = np.zeros(sizeQl, np.uint64), np.zeros(sizeQl, np.uint64)
    o.orc_dcrt_params(2 * N, sizeQl, 55, q, psi)
    x = libs.rand_tower(rng, q, N)
    want = np.zeros((sizeQl - 1, N), np.uint64)
    r.ref_mod_reduce(N, sizeQl, q, psi, x, t, ev, want)
    octx = o.orc_ctx_create(N, sizeQl, q, psi)
    got = np.zeros((sizeQl - 1, N), np.uint64)
    o.orc_mod_reduce(octx, x, sizeQl, t, ev, got)
    assert np.array_equal(got, want)
    o.orc_ctx_destroy(octx)


@pytest.mark.parametrize("logN,nQ,nP,inEval,resEval,rev", [(4, 2, 3, 1, 1, 0), (5, 3, 3, 0, 1, 1), (6, 3, 4, 1, 0, 0), (4, 2, 2, 0, 0, 1)])
def test_expand_crt_basis_against_live_reference(oracle, ref, logN, nQ, nP, inEval, resEval, rev):
    """DCRTPoly::ExpandCRTBasis / ExpandCRTBasisReverseOrder of the reference vs the oracle composite"""
    o, r = oracle, ref
    rng = np.random.default_rng(91)
    N = 1 << logN
    allq, allpsi = np.zeros(nQ + nP, np.uint64), np.zeros(nQ + nP, np.uint64)
    o.orc_dcrt_params(2 * N, nQ + nP, 58, allq, allpsi)
    q, p, psiQ, psiP = allq[:nQ].copy(), allq[nQ:].copy(), allpsi[:nQ].copy(), allpsi[nQ:].copy()
    hatInv, hatPre, hatMod, alpha, qInv, mu = libs.crt_tables(q, p)
    hm_pq = np.ascontiguousarray(hatMod.T)
    x = libs.rand_tower(rng, q, N)
    want = np.zeros((nQ + nP, N), np.uint64)
    r.ref_expand_crt_basis(N, nQ, q, psiQ, x, inEval, hatInv, hm_pq, alpha, nP, p, psiP, qInv, resEval, rev, want)
    octx = o.orc_ctx_create(N, nQ + nP, allq, allpsi)
    got = np.zeros((nQ + nP, N), np.uint64)
    o.orc_expand_crt_basis(octx, nQ, nP, x, inEval, hatInv, hatPre, hm_pq, alpha, mu, qInv, resEval, rev, got)
    assert np.array_equal(got, want)
    o.orc_ctx_destroy(octx)


@pytest.mark.parametrize("logN,nQ,nP", [(4, 2, 3), (6, 3, 3), (5, 4, 5)])
def test_fast_expand_crt_basis_p_over_q_against_live_reference(oracle, ref, logN, nQ, nP):
    """DCRTPoly::FastExpandCRTBasisPloverQ with the BFV HPSPOVERQ tables vs the oracle composite"""
    o, r = oracle, ref
    rng = np.random.default_rng(92)
    N = 1 << logN
    allq, allpsi = np.zeros(nQ + nP, np.uint64), np.zeros(nQ + nP, np.uint64)
    o.orc_dcrt_params(2 * N, nQ + nP, 58, allq, allpsi)
    q, pl, psiQ, psiP = allq[:nQ].copy(), allq[nQ:].copy(), allpsi[:nQ].copy(), allpsi[nQ:].copy()
    m, mpre, qinvp = libs.p_over_q_tables(q, pl)
    hatInv2, hatPre2, hatMod2, alpha2, pInv, muQ = libs.crt_tables(pl, q)  # Pl -> Ql (= Q)
    _, _, _, _, _, muP = libs.crt_tables(q, pl)
    hm2_qp = np.ascontiguousarray(hatMod2.T)
    x = libs.rand_tower(rng, q, N)
    want = np.zeros((nQ + nP, N), np.uint64)
    r.ref_fast_expand_crt_basis_p_over_q(N, nQ, q, psiQ, x, m, qinvp, nP, pl, psiP, hatInv2, hm2_qp, alpha2, nQ, q, psiQ, pInv, want)
    got = np.zeros((nQ + nP, N), np.uint64)
    o.orc_fast_expand_crt_basis_p_over_q(x, nQ, N, q, m, mpre, qinvp, nP, pl, muP, hatInv2, hatPre2, hm2_qp, alpha2, nQ, q, muQ,
                                         pInv, got)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("logN,sizeQ,sizeQl,dnum,t", [(4, 3, 3, 1, 0), (5, 4, 3, 2, 65537), (6, 4, 4, 2, 2), (4, 2, 2, 1, 786433)])
def test_approx_mod_down_against_live_reference(oracle, ref, logN, sizeQ, sizeQl, dnum, t):
    """DCRTPoly::ApproxModDown of the reference (CKKS/BFV form t = 0 and BGV form t > 0) vs the oracle"""
    o, r = oracle, ref
    rng = np.random.default_rng(93)
    N = 1 << logN
    q, psiQ = np.zeros(sizeQ, np.uint64), np.zeros(sizeQ, np.uint64)
    o.orc_dcrt_params(2 * N, sizeQ, 50, q, psiQ)
    p, psiP = np.zeros(64, np.uint64), np.zeros(64, np.uint64)
    sizeP = o.orc_hybrid_select_p(N, sizeQ, q, dnum, 60, p, psiP)
    p, psiP = p[:sizeP].copy(), psiP[:sizeP].copy()
    hy = o.orc_hybrid_create(N, sizeQ, q, psiQ, sizeP, p, psiP, dnum)
    ql, psiQl = q[:sizeQl].copy(), psiQ[:sizeQl].copy()
    hatInv, _, hatMod, _, _, _ = libs.crt_tables(p, ql)  # PHatInvModp[j], PHatModq[j][i]
    P = 1
    for v in p:
        P *= int(v)
    pinv = np.array([pow(P % int(v), -1, int(v)) for v in ql], np.uint64)
    x = libs.rand_tower(rng, np.concatenate([ql, p]), N)
    want = np.zeros((sizeQl, N), np.uint64)
    r.ref_approx_mod_down(N, sizeQl, ql, psiQl, sizeP, p, psiP, x, pinv, hatInv, hatMod, t, want)
    got = np.zeros((sizeQl, N), np.uint64)
    if t:
        o.orc_hybrid_approx_mod_down_t(hy, x, sizeQl, t, got)
    else:
        o.orc_hybrid_approx_mod_down(hy, x, sizeQl, got)
    assert np.array_equal(got, want)
    o.orc_hybrid_destroy(hy)


@pytest.mark.parametrize("logN,sizeQ,bits,t", [(4, 2, 28, 65537), (4, 2, 28, 1 << 16), (5, 3, 45, 65537), (5, 3, 45, 1 << 20),
                                                (4, 3, 60, 65537), (4, 4, 60, 1 << 30), (4, 2, 50, 786433), (4, 3, 55, 1 << 3),
                                                (4, 2, 30, (1 << 34) - 41), (4, 3, 59, (1 << 34) - 41)])
def test_scale_and_round_native_against_live_reference(oracle, ref, logN, sizeQ, bits, t):
    """DCRTPoly::ScaleAndRound -> NativePoly (BFV decryption, all branch families: t power of two or not, split or not,
    with or without modular products) of the reference vs the oracle"""
    o, r = oracle, ref
    rng = np.random.default_rng(101)
    N = 1 << logN
    q, psi = np.zeros(sizeQ, np.uint64), np.zeros(sizeQ, np.uint64)
    o.orc_dcrt_params(2 * N, sizeQ, bits, q, psi)
    a, b, fr, bf = libs.decrypt_tables(q, t)
    x = libs.rand_tower(rng, q, N)
    x[:, 0] = q - np.uint64(1)
    x[:, 1] = 0
    want, got = np.zeros(N, np.uint64), np.zeros(N, np.uint64)
    r.ref_scale_and_round_native(N, sizeQ, q, psi, x, t, a, b, fr, bf, want)
    o.orc_scale_and_round_native(x, sizeQ, N, q, t, a, b, fr, bf, got)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("logN,sizeQ,bits,t", [(4, 2, 45, 65537), (5, 3, 60, 786433), (4, 4, 55, 2)])
def test_scale_and_round_behz_decrypt_against_live_reference(oracle, ref, logN, sizeQ, bits, t):
    o, r = oracle, ref
    rng = np.random.default_rng(102)
    N = 1 << logN
    q, psi = np.zeros(sizeQ, np.uint64), np.zeros(sizeQ, np.uint64)
    o.orc_dcrt_params(2 * N, sizeQ, bits, q, psi)
    tg, a, b = libs.behz_decrypt_tables(q, t)
    x = libs.rand_tower(rng, q, N)
    want, got = np.zeros(N, np.uint64), np.zeros(N, np.uint64)
    r.ref_scale_and_round_behz_decrypt(N, sizeQ, q, psi, x, t, tg, a, b, want)
    o.orc_scale_and_round_behz_decrypt(x, sizeQ, N, q, tg, a, b, got)
    assert np.array_equal(got, want)


def ref_bfv_session(r, ring, t, depth, sms):
    """reference BFV/BEHZ context with two fresh ciphertexts and their EvalMultNoRelin product, exported as arrays"""
    h = r.ref_bfv_create(ring, t, depth, sms, 0)
    info = np.zeros(3, np.uint32)
    r.ref_bfv_info(h, info)
    N, numQ, numBsk = map(int, info)
    q, pq = np.zeros(numQ, np.uint64), np.zeros(numQ, np.uint64)
    bsk, pb = np.zeros(numBsk, np.uint64), np.zeros(numBsk, np.uint64)
    r.ref_bfv_get_moduli(h, q, pq, bsk, pb)
    r.ref_bfv_keygen(h)
    a, b = r.ref_bfv_encrypt(h, 1), r.ref_bfv_encrypt(h, 2)
    c = r.ref_bfv_eval_mult_no_relin(h, a, b)

    def export(ct):
        ci = np.zeros(3, np.uint32)
        r.ref_bfv_ct_info(h, ct, ci)
        out = np.zeros((int(ci[0]), int(ci[1]), N), np.uint64)
        for e in range(int(ci[0])):
            r.ref_bfv_ct_export(h, ct, e, out[e])
        return out, int(ci[2])

    (A, fa), (B, fb), (D, fd) = export(a), export(b), export(c)
    assert (fa, fb, fd) == (0, 0, 1) and D.shape[0] == 3  # inputs EVALUATION, product COEFFICIENT with 3 elements
    return h, N, q, pq, bsk, pb, A, B, D


def ref_bfv_relin_session(r, ring, t, depth, sms, dnum):
    """reference BFV/BEHZ context with HYBRID relinearisation: moduli, key, two ciphertexts and cc->EvalMult of them"""
    h = r.ref_bfv_create_hybrid(ring, t, depth, sms, dnum)
    info = np.zeros(3, np.uint32)
    r.ref_bfv_info(h, info)
    N, numQ, numBsk = map(int, info)
    hi = np.zeros(3, np.uint32)
    r.ref_bfv_hybrid_info(h, hi)
    sizeP, numPartQ = int(hi[0]), int(hi[1])
    q, pq = np.zeros(numQ, np.uint64), np.zeros(numQ, np.uint64)
    bsk, pb = np.zeros(numBsk, np.uint64), np.zeros(numBsk, np.uint64)
    r.ref_bfv_get_moduli(h, q, pq, bsk, pb)
    p, pp = np.zeros(sizeP, np.uint64), np.zeros(sizeP, np.uint64)
    r.ref_bfv_get_p(h, p, pp)
    keyB = np.zeros((numPartQ, numQ + sizeP, N), np.uint64)
    keyA = keyB.copy()
    r.ref_bfv_get_relin_key(h, keyB, keyA)
    a, b = r.ref_bfv_encrypt(h, 1), r.ref_bfv_encrypt(h, 2)
    c = r.ref_bfv_eval_mult(h, a, b)

    def export(ct):
        ci = np.zeros(3, np.uint32)
        r.ref_bfv_ct_info(h, ct, ci)
        out = np.zeros((int(ci[0]), int(ci[1]), N), np.uint64)
        for e in range(int(ci[0])):
            r.ref_bfv_ct_export(h, ct, e, out[e])
        return out, int(ci[2])
    (A, fa), (B, fb), (Cc, fc) = export(a), export(b), export(c)
    assert (fa, fb, fc) == (0, 0, 0) and Cc.shape[0] == 2
    return h, dict(N=N, t=t, q=q, psiQ=pq, bsk=bsk, psiBsk=pb, p=p, psiP=pp, numPartQ=numPartQ, keyB=keyB, keyA=keyA, a=A, b=B, c=Cc)


def oracle_bfv_eval_mult_relin(o, S):
    """oracle composition: EvalMultNoRelin (BEHZ) -> NTT -> HYBRID key switch of the third element -> adds"""
    N, q, numQ = S["N"], S["q"], len(S["q"])
    hb = o.orc_behz_create(N, numQ, q, S["t"])
    call = o.orc_ctx_create(N, numQ + len(S["bsk"]), np.concatenate([q, S["bsk"]]), np.concatenate([S["psiQ"], S["psiBsk"]]))
    d = np.zeros((3, numQ, N), np.uint64)
    A, B = S["a"], S["b"]
    o.orc_bfv_eval_mult_behz(hb, call, np.ascontiguousarray(A[0]), np.ascontiguousarray(A[1]), np.ascontiguousarray(B[0]),
                             np.ascontiguousarray(B[1]), d[0], d[1], d[2])
    cq = o.orc_ctx_create(N, numQ, q, S["psiQ"])
    o.orc_ntt_fwd_tower(cq, d, None, numQ, 3, 1)
    hy = o.orc_hybrid_create(N, numQ, q, S["psiQ"], len(S["p"]), S["p"], S["psiP"], S["numPartQ"])
    k0, k1 = np.zeros((numQ, N), np.uint64), np.zeros((numQ, N), np.uint64)
    o.orc_hybrid_key_switch(hy, d[2], numQ, S["keyB"], S["keyA"], k0, k1)
    out = np.zeros((2, numQ, N), np.uint64)
    for i in range(numQ):
        o.orc_vec_add(out[0, i], d[0, i], k0[i], N, int(q[i]))
        o.orc_vec_add(out[1, i], d[1, i], k1[i], N, int(q[i]))
    o.orc_hybrid_destroy(hy), o.orc_ctx_destroy(cq), o.orc_ctx_destroy(call), o.orc_behz_destroy(hb)
    return out


@pytest.mark.parametrize("ring,t,depth,sms,dnum", [(64, 65537, 2, 60, 2), (1024, 786433, 3, 55, 3)])
def test_bfv_eval_mult_with_relinearisation_against_live_reference(oracle, ref, ring, t, depth, sms, dnum):
    """cc->EvalMult on BFV/BEHZ ciphertexts with a HYBRID relinearisation key (config 5 end to end) vs the oracle"""
    h, S = ref_bfv_relin_session(ref, ring, t, depth, sms, dnum)
    assert np.array_equal(oracle_bfv_eval_mult_relin(oracle, S), S["c"])
    ref.ref_bfv_destroy(h)


@pytest.mark.parametrize("ring,t,depth,sms", [(64, 65537, 2, 60), (1024, 786433, 3, 55)])
def test_bfv_eval_mult_behz_against_live_reference(oracle, ref, ring, t, depth, sms):
    """LeveledSHEBFVRNS::EvalMult (BEHZ) through the reference's scheme layer vs the oracle's composite"""
    o, r = oracle, ref
    h, N, q, pq, bsk, pb, A, B, D = ref_bfv_session(r, ring, t, depth, sms)
    numQ = len(q)
    hb = o.orc_behz_create(N, numQ, q, t)
    call = o.orc_ctx_create(N, numQ + len(bsk), np.concatenate([q, bsk]), np.concatenate([pq, pb]))
    got = np.zeros((3, numQ, N), np.uint64)
    o.orc_bfv_eval_mult_behz(hb, call, A[0], A[1], B[0], B[1], got[0], got[1], got[2])
    assert np.array_equal(got, D)
    o.orc_ctx_destroy(call)
    o.orc_behz_destroy(hb)
    r.ref_bfv_destroy(h)


@pytest.mark.parametrize("sizeI,sizeO,outputFirst,fscale", [(3, 2, 1, 1.0), (4, 3, 0, 1.0), (3, 2, 1, 2.0 ** 60), (2, 3, 0, 32.0)])
def test_scale_and_round_against_live_reference(oracle, ref, sizeI, sizeO, outputFirst, fscale):
    o, r = oracle, ref
    rng = np.random.default_rng(42)
    N, L = 64, sizeI + sizeO
    q, psi = np.zeros(L, np.uint64), np.zeros(L, np.uint64)
    o.orc_dcrt_params(2 * N, L, 60, q, psi)
    x = libs.rand_tower(rng, q, N)
    off = 0 if outputFirst else sizeI
    om = q[off:off + sizeO].copy()
    tab = np.stack([rng.integers(0, int(m), size=sizeI + 1, dtype=np.uint64) for m in om])
    frac = rng.random(sizeI) * fscale
    mu = np.zeros((sizeO, 2), np.uint64)
    for j in range(sizeO):
        t = np.zeros(2, np.uint64)
        o.orc_barrett_mu128(int(om[j]), t)
        mu[j] = t
    want, got = np.zeros((sizeO, N), np.uint64), np.zeros((sizeO, N), np.uint64)
    r.ref_scale_and_round(N, sizeI, sizeO, outputFirst, q, psi, x, tab, frac, want)
    o.orc_scale_and_round(x, sizeI, sizeO, N, outputFirst, tab, frac, om, mu, got)
    assert np.array_equal(want, got)
    if not outputFirst:
        r.ref_approx_scale_and_round(N, sizeI, sizeO, q, psi, x, tab, want)
        o.orc_approx_scale_and_round(x, sizeI, sizeO, N, tab, om, mu, got)
        assert np.array_equal(want, got)


def test_scale_and_round_p_over_q_against_live_reference(oracle, ref):
    o, r = oracle, ref
    rng = np.random.default_rng(43)
    N, sizeQ = 64, 3
    q, psi = np.zeros(sizeQ + 1, np.uint64), np.zeros(sizeQ + 1, np.uint64)
    o.orc_dcrt_params(2 * N, sizeQ + 1, 50, q, psi)
    x = libs.rand_tower(rng, q, N)
    pinv = np.array([pow(int(q[sizeQ]), -1, int(q[i])) for i in range(sizeQ)], np.uint64)
    want, got = np.zeros((sizeQ, N), np.uint64), np.zeros((sizeQ, N), np.uint64)
    r.ref_scale_and_round_p_over_q(N, sizeQ, q, psi, x, pinv, want)
    o.orc_scale_and_round_p_over_q(x, sizeQ, N, q[:sizeQ].copy(), int(q[sizeQ]), pinv, got)
    assert np.array_equal(want, got)


def test_times_q_over_t_and_set_values_mod_switch_against_live_reference(oracle, ref):
    """DCRTPolyImpl::TimesQovert (dcrtpoly-impl.h:868-885) and SetValuesModSwitch (:630-647) run by the reference vs the oracle"""
    o, r = oracle, ref
    rng = np.random.default_rng(47)
    N, L = 64, 3
    q, psi = np.zeros(L, np.uint64), np.zeros(L, np.uint64)
    o.orc_dcrt_params(2 * N, L, 55, q, psi)
    for t in (65537, 2, 786433):
        Q = 1
        for v in q:
            Q *= int(v)
        neg = (t - Q % t) % t
        tinv = np.array([pow(t, -1, int(v)) for v in q], np.uint64)
        x = rng.integers(0, t, (L, N), dtype=np.uint64)
        want, got = x.copy(), x.copy()
        r.ref_times_q_over_t(N, L, q, psi, want, t, neg, tinv)
        o.orc_times_q_over_t(got, L, N, q, t, neg, tinv)
        assert np.array_equal(want, got), t
    x = rng.integers(0, int(q[0]), N, dtype=np.uint64)
    x[:2] = (0, q[0] - np.uint64(1))
    for qTo, psiTo in ((int(q[1]), int(psi[1])), (int(q[2]), int(psi[2]))):
        want, got = np.zeros(N, np.uint64), np.zeros(N, np.uint64)
        r.ref_set_values_mod_switch(N, int(q[0]), int(psi[0]), x, qTo, psiTo, want)
        o.orc_set_values_mod_switch(x, N, int(q[0]), qTo, got)
        assert np.array_equal(want, got)


def test_rotations_against_live_reference(oracle, ref):
    """EvalRotate and hoisted EvalFastRotation of the reference (its own rotation keys) vs the oracle"""
    o, r = oracle, ref
    h = r.ref_ckks_create(1 << 10, 4, 45, 55, 3, 0)
    info = np.zeros(5, np.uint32)
    r.ref_ckks_info(h, info)
    N, sizeQ, sizeP, numPartQ, alpha = map(int, info)
    q, psiQ = np.zeros(sizeQ, np.uint64), np.zeros(sizeQ, np.uint64)
    p, psiP = np.zeros(sizeP, np.uint64), np.zeros(sizeP, np.uint64)
    r.ref_ckks_get_moduli(h, q, psiQ, p, psiP)
    hy = o.orc_hybrid_create(N, sizeQ, q, psiQ, sizeP, p, psiP, numPartQ)
    idx = np.array([1, -3, 7], np.int32)
    r.ref_ckks_rotate_keygen(h, idx, 3)
    ct = r.ref_ckks_encrypt(h, 5, 1)
    ci = np.zeros(4, np.uint32)
    r.ref_ct_info(h, ct, ci)
    sizeQl = int(ci[1])

    def ex(c, e):
        a = np.zeros((sizeQl, N), np.uint64)
        r.ref_ct_export(h, c, e, a)
        return a
    c0, c1 = ex(ct, 0), ex(ct, 1)
    for index in (1, -3, 7):
        keyB = np.zeros((numPartQ, sizeQ + sizeP, N), np.uint64)
        keyA = keyB.copy()
        k = r.ref_ckks_get_rot_key(h, index, keyB, keyA)
        assert k == o.orc_find_automorphism_index_2n_complex(index, 2 * N)
        o0, o1 = np.zeros_like(c0), np.zeros_like(c0)
        o.orc_eval_automorphism(hy, c0, c1, sizeQl, k, keyB, keyA, o0, o1)
        rr = r.ref_ckks_eval_rotate(h, ct, index)
        rf = r.ref_ckks_eval_fast_rotate(h, ct, index)
        assert np.array_equal(o0, ex(rr, 0)) and np.array_equal(o1, ex(rr, 1))
        assert np.array_equal(o0, ex(rf, 0)) and np.array_equal(o1, ex(rf, 1))
        # double hoisting: EvalFastRotationExt keeps the extended basis, KeySwitchDown ends it
        for add_first in (1, 0):
            re = r.ref_ckks_eval_fast_rotate_ext(h, ct, index, add_first)
            cie = np.zeros(4, np.uint32)
            r.ref_ct_info(h, re, cie)
            assert int(cie[1]) == sizeQl + sizeP
            w0, w1 = np.zeros((sizeQl + sizeP, N), np.uint64), np.zeros((sizeQl + sizeP, N), np.uint64)
            r.ref_ct_export(h, re, 0, w0), r.ref_ct_export(h, re, 1, w1)
            e0, e1 = np.zeros_like(w0), np.zeros_like(w0)
            o.orc_eval_fast_rotation_ext(hy, c0, c1, sizeQl, k, add_first, keyB, keyA, e0, e1)
            assert np.array_equal(e0, w0) and np.array_equal(e1, w1), "EvalFastRotationExt"
            rd = r.ref_ckks_key_switch_down(h, re)
            d0, d1 = np.zeros_like(c0), np.zeros_like(c0)
            o.orc_hybrid_approx_mod_down(hy, e0, sizeQl, d0)
            o.orc_hybrid_approx_mod_down(hy, e1, sizeQl, d1)
            assert np.array_equal(d0, ex(rd, 0)) and np.array_equal(d1, ex(rd, 1)), "KeySwitchDown"
    o.orc_hybrid_destroy(hy)
    r.ref_ckks_destroy(h)


def ref_linear_transform_session(r, o, ring, depth, sms, fms, dnum, slots, bStep, level=1, seed=11):
    """The reference's own EvalLinearTransform (BSGS with double hoisting) on its own keys and plaintext diagonals.
    Returns everything the oracle / the product need: moduli, ciphertext, diagonals, rotation keys, and the result."""
    h = r.ref_ckks_create(ring, depth, sms, fms, dnum, 0)
    info = np.zeros(5, np.uint32)
    r.ref_ckks_info(h, info)
    N, sizeQ, sizeP, numPartQ, alpha = map(int, info)
    q, psiQ = np.zeros(sizeQ, np.uint64), np.zeros(sizeQ, np.uint64)
    p, psiP = np.zeros(sizeP, np.uint64), np.zeros(sizeP, np.uint64)
    r.ref_ckks_get_moduli(h, q, psiQ, p, psiP)
    gStep = -(-slots // bStep)
    rots = sorted(set(range(1, bStep)) | {bStep * j for j in range(1, gStep)})
    idx = np.array(rots, np.int32)
    r.ref_ckks_rotate_keygen(h, idx, len(idx))
    rng = np.random.default_rng(seed)
    vals, M = np.zeros((slots, 2)), np.zeros((slots, slots, 2))
    vals[:, 0], M[:, :, 0] = rng.uniform(-1, 1, slots), rng.uniform(-1, 1, (slots, slots))  # real data: the default CKKS mode
    ct = r.ref_ckks_encrypt_slots(h, vals, level, slots)
    ci = np.zeros(4, np.uint32)
    r.ref_ct_info(h, ct, ci)
    sizeQl = int(ci[1])
    lt = r.ref_ckks_lt_create(h, slots, bStep, M, sizeQl - 1)  # L: plaintexts get L + 1 Q limbs (the ciphertext's) + P
    assert r.ref_ckks_lt_get_diag(lt, 0, None) == sizeQl + sizeP
    diag = np.zeros((slots, sizeQl + sizeP, N), np.uint64)
    for i in range(slots):
        r.ref_ckks_lt_get_diag(lt, i, diag[i].ctypes.data)

    def ex(c, e):
        a = np.zeros((sizeQl, N), np.uint64)
        r.ref_ct_export(h, c, e, a)
        return a
    keys = {}
    for index in rots:
        keyB = np.zeros((numPartQ, sizeQ + sizeP, N), np.uint64)
        keyA = keyB.copy()
        k = r.ref_ckks_get_rot_key(h, index, keyB, keyA)
        keys[index] = (k, keyB, keyA)
    res = r.ref_ckks_eval_linear_transform(h, lt, ct)
    S = dict(N=N, q=q, psiQ=psiQ, p=p, psiP=psiP, numPartQ=numPartQ, sizeQl=sizeQl, slots=slots, bStep=bStep, gStep=gStep,
             c=np.stack([ex(ct, 0), ex(ct, 1)]), diag=diag, values=vals[:, 0].copy(), matrix=M[:, :, 0].copy(), keys=keys, out=np.stack([ex(res, 0), ex(res, 1)]))
    return h, lt, ct, res, S


def bsgs_arguments(S):
    """orc_ckks_bsgs_transform / fhe_ckks_bsgs_transform arguments of EvalLinearTransform: inner rotations 0..bStep-1,
    outer rotations bStep*j, diagonal bStep*j+i (absent beyond `slots`)"""
    bStep, gStep, slots, keys = S["bStep"], S["gStep"], S["slots"], S["keys"]
    inK = np.array([0] + [keys[i][0] for i in range(1, bStep)], np.uint32)
    outK = np.array([0] + [keys[bStep * j][0] for j in range(1, gStep)], np.uint32)
    inB = [None] + [keys[i][1] for i in range(1, bStep)]
    inA = [None] + [keys[i][2] for i in range(1, bStep)]
    outB = [None] + [keys[bStep * j][1] for j in range(1, gStep)]
    outA = [None] + [keys[bStep * j][2] for j in range(1, gStep)]
    diag = [S["diag"][bStep * j + i] if bStep * j + i < slots else None for j in range(gStep) for i in range(bStep)]
    return inK, inB, inA, outK, outB, outA, diag


@pytest.mark.parametrize("ring,depth,dnum,slots,bStep", [(1 << 9, 3, 2, 16, 4), (1 << 10, 4, 3, 8, 4), (1 << 9, 3, 2, 16, 5)])
def test_linear_transform_against_live_reference(oracle, ref, ring, depth, dnum, slots, bStep):
    """FHECKKSRNS::EvalLinearTransform of the reference (BSGS + double hoisting, its own keys and encoded diagonals) vs the
    oracle's composition of EvalFastRotationExt / EvalMultExt / KeySwitchDown"""
    o, r = oracle, ref
    h, lt, ct, res, S = ref_linear_transform_session(r, o, ring, depth, 45, 55, dnum, slots, bStep)
    hy = o.orc_hybrid_create(S["N"], len(S["q"]), S["q"], S["psiQ"], len(S["p"]), S["p"], S["psiP"], S["numPartQ"])
    inK, inB, inA, outK, outB, outA, diag = bsgs_arguments(S)
    out = np.zeros_like(S["c"])
    o.orc_ckks_bsgs_transform(hy, S["c"][0], S["c"][1], S["sizeQl"], len(inK), inK, libs.ptr_array(inB), libs.ptr_array(inA),
                              len(outK), outK, libs.ptr_array(outB), libs.ptr_array(outA), libs.ptr_array(diag), out[0], out[1])
    assert np.array_equal(out, S["out"])
    o.orc_hybrid_destroy(hy)
    r.ref_ckks_lt_destroy(lt)
    r.ref_ckks_destroy(h)


def c2s_levels(P, slots, N):
    """Rotation plan of FHECKKSRNS::EvalCoeffsToSlots (ckksrns-fhe.cpp:1884-1925), level by level in evaluation order:
    [(s, inner rotation indices, outer rotation indices, {(i, j): index into A[s]} )]"""
    lvlb, layers, rem, numRot, b, g, numRotRem, bRem, gRem = P
    M4 = N // 2
    flagRem = 1 if rem else 0
    stop = 0 if rem else -1
    out = []
    offset = (numRot + 1) // 2 - 1
    for s in range(lvlb - 1, stop, -1):
        scale = 1 << ((s - flagRem) * layers + rem)
        rot_out = [(scale * g * i) % M4 for i in range(b)]
        rot_in = [(scale * (j - offset)) % slots for j in range(g)]
        terms = {(i, j): g * i + j for i in range(b) for j in range(g) if g * i + j != numRot}
        out.append((s, rot_in, rot_out, terms))
    if flagRem:
        offset = (numRotRem + 1) // 2 - 1
        rot_out = [(gRem * i) % M4 for i in range(bRem)]
        rot_in = [(j - offset) % slots for j in range(gRem)]
        terms = {(i, j): gRem * i + j for i in range(bRem) for j in range(gRem) if gRem * i + j != numRotRem}
        out.append((0, rot_in, rot_out, terms))
    return out


def s2c_levels(P, slots, N):
    """Rotation plan of FHECKKSRNS::EvalSlotsToCoeffs (ckksrns-fhe.cpp:2041-2080): levels ascending, the remainder last"""
    lvlb, layers, rem, numRot, b, g, numRotRem, bRem, gRem = P
    M4 = N // 2
    flagRem = 1 if rem else 0
    smax = lvlb - flagRem
    offset = (numRot + 1) // 2 - 1
    out = []
    for s in range(smax):
        scale = 1 << (s * layers)
        rot_in = [((j - offset) * scale) % M4 for j in range(g)]
        rot_out = [(g * i * scale) % M4 for i in range(b)]
        out.append((s, rot_in, rot_out, {(i, j): g * i + j for i in range(b) for j in range(g) if g * i + j != numRot}))
    if flagRem:
        scale = 1 << (smax * layers)
        offset = (numRotRem + 1) // 2 - 1
        rot_in = [((j - offset) * scale) % M4 for j in range(gRem)]
        rot_out = [(gRem * i * scale) % M4 for i in range(bRem)]
        out.append((smax, rot_in, rot_out,
                    {(i, j): gRem * i + j for i in range(bRem) for j in range(gRem) if gRem * i + j != numRotRem}))
    return out


def ref_coeffs_to_slots_session(r, ring, depth, dnum, budget, level=0, seed=21, decode=False):
    """The reference's own EvalCoeffsToSlots (or, decode=True, EvalSlotsToCoeffs), fully packed, with its own keys and
    plaintexts; everything exported."""
    h = r.ref_ckks_create(ring, depth, 45, 55, dnum, 0)
    info = np.zeros(5, np.uint32)
    r.ref_ckks_info(h, info)
    N, sizeQ, sizeP, numPartQ, alpha = map(int, info)
    slots = N // 2
    q, psiQ = np.zeros(sizeQ, np.uint64), np.zeros(sizeQ, np.uint64)
    p, psiP = np.zeros(sizeP, np.uint64), np.zeros(sizeP, np.uint64)
    r.ref_ckks_get_moduli(h, q, psiQ, p, psiP)
    rng = np.random.default_rng(seed)
    vals = np.zeros((slots, 2))
    vals[:, 0] = rng.uniform(-1, 1, slots)
    ct = r.ref_ckks_encrypt_slots(h, vals, level, slots)
    ci = np.zeros(4, np.uint32)
    r.ref_ct_info(h, ct, ci)
    sizeQl = int(ci[1])
    create = r.ref_ckks_s2c_create if decode else r.ref_ckks_c2s_create
    c2s = create(h, slots, budget, sizeQl - budget)  # L + lvlb limbs at the first level = the ciphertext's
    P = np.zeros(9, np.uint32)
    r.ref_ckks_c2s_params(c2s, P)
    levels = (s2c_levels if decode else c2s_levels)([int(v) for v in P], slots, N)
    rots = sorted({x for _, ri, ro, _ in levels for x in ri + ro if x})
    idx = np.array(rots, np.int32)
    r.ref_ckks_rotate_keygen(h, idx, len(idx))
    keys = {}
    for index in rots:
        keyB = np.zeros((numPartQ, sizeQ + sizeP, N), np.uint64)
        keyA = keyB.copy()
        keys[index] = (r.ref_ckks_get_rot_key(h, index, keyB, keyA), keyB, keyA)
    diags = []
    for n, (s, ri, ro, terms) in enumerate(levels):
        limbs = sizeQl - n + sizeP
        d = {}
        for (i, j), a in terms.items():
            rows = np.zeros((limbs, N), np.uint64)
            assert r.ref_ckks_c2s_get_diag(c2s, s, a, rows.ctypes.data) == limbs
            d[(i, j)] = rows
        diags.append(d)
    res = (r.ref_ckks_eval_slots_to_coeffs if decode else r.ref_ckks_eval_coeffs_to_slots)(h, c2s, ct)
    r.ref_ct_info(h, res, ci)
    outQl = int(ci[1])
    assert outQl == sizeQl - (len(levels) - 1)

    def ex(c, e, nl):
        a = np.zeros((nl, N), np.uint64)
        r.ref_ct_export(h, c, e, a)
        return a
    S = dict(N=N, q=q, psiQ=psiQ, p=p, psiP=psiP, numPartQ=numPartQ, sizeQl=sizeQl, levels=levels, keys=keys, diags=diags,
             c=np.stack([ex(ct, 0, sizeQl), ex(ct, 1, sizeQl)]), out=np.stack([ex(res, 0, outQl), ex(res, 1, outQl)]))
    r.ref_ckks_c2s_destroy(c2s)
    return h, S


def c2s_level_arguments(S, n):
    """orc_ckks_bsgs_transform / fhe_ckks_bsgs_transform arguments of level n of the session"""
    s, rot_in, rot_out, terms = S["levels"][n]
    keys = S["keys"]

    def side(rots):
        return (np.array([keys[x][0] if x else 0 for x in rots], np.uint32), [keys[x][1] if x else None for x in rots],
                [keys[x][2] if x else None for x in rots])
    inK, inB, inA = side(rot_in)
    outK, outB, outA = side(rot_out)
    diag = [S["diags"][n].get((i, j)) for i in range(len(rot_out)) for j in range(len(rot_in))]
    return inK, inB, inA, outK, outB, outA, diag


@pytest.mark.parametrize("ring,depth,dnum,budget,decode", [(64, 4, 2, 2, False), (128, 5, 3, 3, False), (32, 3, 2, 1, False),
                                                           (64, 4, 2, 2, True), (128, 5, 3, 3, True)])
def test_coeffs_to_slots_against_live_reference(oracle, ref, ring, depth, dnum, budget, decode):
    """FHECKKSRNS::EvalCoeffsToSlots / EvalSlotsToCoeffs (collapsed-FFT levels, each a BSGS transform with double hoisting,
    a rescale between them) run by the reference vs the oracle: one orc_ckks_bsgs_transform per level +
    DropLastElementAndScale"""
    o, r = oracle, ref
    h, S = ref_coeffs_to_slots_session(r, ring, depth, dnum, budget, decode=decode)
    N, q = S["N"], S["q"]
    hy = o.orc_hybrid_create(N, len(q), q, S["psiQ"], len(S["p"]), S["p"], S["psiP"], S["numPartQ"])
    octx = o.orc_ctx_create(N, len(q), q, S["psiQ"])
    c, sizeQl = S["c"], S["sizeQl"]
    for n in range(len(S["levels"])):
        if n:  # ModReduceInternalInPlace between the levels (ckksrns-fhe.cpp:1936-1937)
            nxt = np.zeros((2, sizeQl - 1, N), np.uint64)
            for e in range(2):
                o.orc_drop_last_element_and_scale(octx, np.ascontiguousarray(c[e]), sizeQl, nxt[e])
            c, sizeQl = nxt, sizeQl - 1
        inK, inB, inA, outK, outB, outA, diag = c2s_level_arguments(S, n)
        out = np.zeros_like(c)
        o.orc_ckks_bsgs_transform(hy, np.ascontiguousarray(c[0]), np.ascontiguousarray(c[1]), sizeQl, len(inK), inK,
                                  libs.ptr_array(inB), libs.ptr_array(inA), len(outK), outK, libs.ptr_array(outB),
                                  libs.ptr_array(outA), libs.ptr_array(diag), out[0], out[1])
        c = out
    assert np.array_equal(c, S["out"])
    o.orc_ctx_destroy(octx)
    o.orc_hybrid_destroy(hy)
    r.ref_ckks_destroy(h)


# ---- round 2: the members that had no entry point (ApproxModUp, ExpandCRTBasisQlHat, MultAccEqNoCheck, EvalSquareCore,
# ---- the ModRaise constructor) ----
@pytest.mark.parametrize("logN,nQ,nP,inEval", [(4, 2, 3, 1), (5, 3, 2, 0), (6, 4, 4, 1), (4, 1, 2, 0)])
def test_approx_mod_up_against_live_reference(oracle, ref, logN, nQ, nP, inEval):
    """DCRTPolyImpl::ApproxModUp (dcrtpoly-impl.h:935-963) of the reference vs the oracle composite"""
    o, r = oracle, ref
    rng = np.random.default_rng(191)
    N = 1 << logN
    allq, allpsi = np.zeros(nQ + nP, np.uint64), np.zeros(nQ + nP, np.uint64)
    o.orc_dcrt_params(2 * N, nQ + nP, 57, allq, allpsi)
    q, p, psiQ, psiP = allq[:nQ].copy(), allq[nQ:].copy(), allpsi[:nQ].copy(), allpsi[nQ:].copy()
    hatInv, hatPre, hatMod, _, _, mu = libs.crt_tables(q, p)
    x = libs.rand_tower(rng, q, N)
    want = np.zeros((nQ + nP, N), np.uint64)
    r.ref_approx_mod_up(N, nQ, q, psiQ, x, inEval, hatInv, hatMod, nP, p, psiP, want)
    octx = o.orc_ctx_create(N, nQ + nP, allq, allpsi)
    got = np.zeros((nQ + nP, N), np.uint64)
    o.orc_approx_mod_up(octx, nQ, nP, x, inEval, hatInv, hatPre, hatMod, mu, got)
    assert np.array_equal(got, want)
    o.orc_ctx_destroy(octx)


@pytest.mark.parametrize("logN,sizeQ,sizeQl,ev", [(4, 3, 2, 1), (5, 4, 4, 0), (6, 5, 1, 1)])
def test_expand_crt_basis_ql_hat_against_live_reference(oracle, ref, logN, sizeQ, sizeQl, ev):
    """DCRTPolyImpl::ExpandCRTBasisQlHat (dcrtpoly-impl.h:1167-1187)"""
    o, r = oracle, ref
    rng = np.random.default_rng(192)
    N = 1 << logN
    q, psi = np.zeros(sizeQ, np.uint64), np.zeros(sizeQ, np.uint64)
    o.orc_dcrt_params(2 * N, sizeQ, 55, q, psi)
    x = libs.rand_tower(rng, q[:sizeQl], N)
    h = libs.rand_tower(rng, q[:sizeQl], 1)[:, 0].copy()
    want, got = np.zeros((sizeQ, N), np.uint64), np.ones((sizeQ, N), np.uint64)
    r.ref_expand_crt_basis_ql_hat(N, sizeQ, q, psi, x, sizeQl, ev, h, want)
    o.orc_expand_crt_basis_ql_hat(x, sizeQl, N, q, h, sizeQ, got)
    assert np.array_equal(got, want)


def test_mult_acc_and_mod_raise_against_live_reference(oracle, ref):
    """PolyImpl::MultAccEqNoCheck (mubintvecnat.cpp:132-142) and the ModRaise constructor (dcrtpoly-impl.h:87-93)"""
    o, r = oracle, ref
    rng = np.random.default_rng(193)
    for logN, L in [(4, 3), (7, 2)]:
        N = 1 << logN
        q, psi = np.zeros(L, np.uint64), np.zeros(L, np.uint64)
        o.orc_dcrt_params(2 * N, L, 59, q, psi)
        acc, v = libs.rand_tower(rng, q, N), libs.rand_tower(rng, q, N)
        consts = np.array([int(rng.integers(0, 1 << 62)) for _ in range(L)], np.uint64)  # also constants >= q (ModEq first)
        want = acc.copy()
        r.ref_mult_acc(N, L, q, psi, want, v, consts)
        got = acc.copy()
        for i in range(L):
            o.orc_vec_mult_acc(got[i], v[i], consts[i], N, q[i])
        assert np.array_equal(got, want)
        # ModRaise: descending and ascending moduli relative to q_0
        for qs, ps in ((q, psi), (q[::-1].copy(), psi[::-1].copy())):
            x = libs.rand_tower(rng, qs[:1], N)[0]
            x[:3] = (0, qs[0] >> np.uint64(1), qs[0] - np.uint64(1))
            want, got = np.zeros((L, N), np.uint64), np.zeros((L, N), np.uint64)
            r.ref_mod_raise(N, L, qs, ps, x, want)
            o.orc_mod_raise(x, N, qs, L, got)
            assert np.array_equal(got, want)


def test_eval_square_core_against_live_reference(oracle, ref):
    """LeveledSHEBase::EvalSquareCore through cc->EvalSquare without relinearisation (FIXEDMANUAL: no level adjustment)"""
    o, r = oracle, ref
    h = r.ref_ckks_create(1 << 8, 2, 50, 60, 2, 0)  # scalTech 0 = FIXEDMANUAL
    info = np.zeros(5, np.uint32)
    r.ref_ckks_info(h, info)
    N, sizeQ, sizeP = int(info[0]), int(info[1]), int(info[2])
    q, psiQ, p, psiP = (np.zeros(n, np.uint64) for n in (sizeQ, sizeQ, sizeP, sizeP))
    r.ref_ckks_get_moduli(h, q, psiQ, p, psiP)
    ct = r.ref_ckks_encrypt(h, 5, 0)
    sq = r.ref_ckks_eval_square_no_relin(h, ct)
    ci = np.zeros(4, np.uint32)
    r.ref_ct_info(h, sq, ci)
    assert int(ci[0]) == 3
    L = int(ci[1])
    a = [np.zeros((L, N), np.uint64) for _ in range(2)]
    for e in range(2):
        r.ref_ct_export(h, ct, e, a[e])
    want = [np.zeros((L, N), np.uint64) for _ in range(3)]
    for e in range(3):
        r.ref_ct_export(h, sq, e, want[e])
    got = [np.zeros((L, N), np.uint64) for _ in range(3)]
    o.orc_eval_square_core(a[0], a[1], L, N, q, got[0], got[1], got[2])
    for e in range(3):
        assert np.array_equal(got[e], want[e]), f"EvalSquareCore element {e}"
    r.ref_ckks_destroy(h)


def test_plus_minus_constants_against_live_reference(oracle, ref):
    """orc_vec_add_const / orc_vec_sub_const vs DCRTPolyImpl::Plus / Minus(vector<Integer>) (dcrtpoly-impl.h:520-548) of the
    compiled reference, both formats"""
    o, r = oracle, ref
    rng = np.random.default_rng(4242)
    for logN, L in [(4, 2), (10, 4)]:
        N = 1 << logN
        q = np.zeros(L, np.uint64)
        psi = np.zeros(L, np.uint64)
        o.orc_dcrt_params(2 * N, L, 59, q, psi)
        a = np.stack([rng.integers(0, int(m), N, dtype=np.uint64) for m in q])
        a[:, 0] = q - np.uint64(1)
        consts = np.array([int(rng.integers(0, 1 << 63)) for _ in range(L)], np.uint64)
        consts[0] = 0
        for fmt in (0, 1):
            for minus in (0, 1):
                want = np.empty_like(a)
                r.ref_plus_minus_const(N, L, q, psi, a, consts, fmt, minus, want)
                got = np.empty_like(a)
                for i in range(L):
                    if minus:
                        o.orc_vec_sub_const(got[i], a[i], consts[i], N, q[i])
                    else:
                        o.orc_vec_add_const(got[i], a[i], consts[i], N, q[i], fmt)
                assert np.array_equal(got, want), (logN, fmt, minus)


@pytest.mark.parametrize("logN,L,bits,baseBits", [(4, 3, 60, 0), (5, 4, 60, 4), (6, 2, 50, 7), (5, 3, 60, 1), (4, 3, 36, 16), (5, 2, 60, 30),
                                                  (5, 5, 45, 3)])
def test_crt_decompose_against_live_reference(oracle, ref, logN, L, bits, baseBits):
    """orc_crt_decompose == DCRTPolyImpl::CRTDecompose (dcrtpoly-impl.h:230-285; KeySwitchBV's digit decomposition) from both formats"""
    o = oracle
    N = 1 << logN
    rng = np.random.default_rng(900 + baseBits)
    q, psi = np.zeros(L, np.uint64), np.zeros(L, np.uint64)
    o.orc_dcrt_params(2 * N, L, bits, q, psi)
    x = libs.rand_tower(rng, q, N, 1)[0]
    x[:, 0] = 0
    x[:, 1] = q - np.uint64(1)
    x[:, 2] = q >> np.uint64(1)
    x[:, 3] = (q >> np.uint64(1)) + np.uint64(1)
    octx = o.orc_ctx_create(N, L, q, psi)
    towers = ref.ref_crt_decompose(N, L, q, psi, x, 0, baseBits, None)
    assert towers == o.orc_crt_decompose(octx, x.ctypes.data, L, baseBits, None)
    want = np.zeros((towers, L, N), np.uint64)
    got = np.zeros_like(want)
    ref.ref_crt_decompose(N, L, q, psi, x, 0, baseBits, want.ctypes.data)
    o.orc_crt_decompose(octx, x.ctypes.data, L, baseBits, got.ctypes.data)
    assert np.array_equal(got, want)
    # the reference takes an EVALUATION tower too (its coefficient copy is the inverse transform): the same towers
    xe = x.copy()
    o.orc_ntt_fwd_tower(octx, xe, None, L, 1, 0)
    want_e = np.zeros_like(want)
    ref.ref_crt_decompose(N, L, q, psi, xe, 1, baseBits, want_e.ctypes.data)
    assert np.array_equal(want_e, want)
    o.orc_ctx_destroy(octx)
