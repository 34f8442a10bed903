"""Parity AT THE BASELINE SHAPES (BASELINE.json configs[1..4]) on the GPU: the product library against the oracle on the same
seeded inputs, word for word, at the very sizes bench.py times.

    config 2   fwd / inv NTT and Hadamard product, N = 2^16, 30 x 60-bit limbs (ILDCRTParams chain), sampled towers
    config 3   KeySwitchCore, EvalMult (+ rescale), N = 2^16, l = 21, k = 7, dnum = 3 (alpha = 7)
    config 5   BFV EvalMult (BEHZ) without and with HYBRID relinearisation, N = 2^15, 7 Q limbs + 8 Bsk limbs
    config 4's inner loop: the BSGS linear transform at N = 2^17, l = 21, dnum = 3, 8 x 8 diagonals

These are `-m gpu` tests only: the lane emulator would need hours at these sizes (its small-ring cases cover the same
entry points in tests/test_parity*.py).  The oracle runs each case in seconds on the host cores.
"""
import numpy as np
import pytest

import libs
from openfhe_amd import fhe_hip as fh
from test_parity import ckks_like_params, params
from test_parity_lt import run_oracle

pytestmark = pytest.mark.gpu


def test_config2_ntt_and_hadamard_at_full_ring(hip, oracle):
    """N = 2^16, L = 30 (the bench's modulus chain): forward words, inverse words and the Hadamard product of a small batch"""
    o = oracle
    rng = np.random.default_rng(202)
    logN, L, B = 16, 30, 3
    N = 1 << logN
    q, psi = params(o, logN, L)
    q2, psi2 = hip.dcrt_chain(logN, L, 60)  # the product-side helper bench.py uses must give the same chain
    assert np.array_equal(q, q2) and np.array_equal(psi, psi2)
    ctx = fh.Context(hip, logN, q, psi)
    octx = o.orc_ctx_create(N, L, q, psi)
    x, y = libs.rand_tower(rng, q, N, B), libs.rand_tower(rng, q, N, B)
    want = x.copy()
    o.orc_ntt_fwd_tower(octx, want, None, L, B, 0)
    t = ctx.tower(x, fmt=fh.COEFFICIENT).SwitchFormat()
    assert np.array_equal(t.to_host(), want), "forward NTT words differ from the oracle at N=2^16, L=30"
    winv = y.copy()
    o.orc_ntt_inv_tower(octx, winv, None, L, B, 0)
    assert np.array_equal(ctx.tower(y, fmt=fh.EVALUATION).SwitchFormat().to_host(), winv), "inverse NTT words differ"
    wm = np.empty_like(x)
    for b in range(B):
        for l in range(L):
            o.orc_vec_mul(wm[b, l], want[b, l], y[b, l], N, q[l])
    assert np.array_equal(t.Times(ctx.tower(y)).to_host(), wm), "Hadamard product differs"
    o.orc_ctx_destroy(octx)
    ctx.close()


def test_config3_keyswitch_evalmult_rescale_at_full_shape(hip, oracle):
    """N = 2^16, l = 21 (60-bit first modulus, 59-bit scaling moduli), HYBRID with dnum = 3 => alpha = 7, k = 7"""
    o = oracle
    rng = np.random.default_rng(203)
    logN, sizeQ, dnum, B = 16, 21, 3, 2
    N = 1 << logN
    q, psiQ, p, psiP = ckks_like_params(o, logN, sizeQ, dnum, first_bits=60, scale_bits=59, aux_bits=60)
    assert len(p) == 7
    q2, _ = hip.ckks_like_chain(logN, sizeQ, 60, 59)  # what bench.py's EvalMult leg builds
    p2, _ = hip.select_p(logN, q2, dnum, 60)
    assert np.array_equal(q, q2) and np.array_equal(p, p2)
    sizeP = len(p)
    hy = o.orc_hybrid_create(N, sizeQ, q, psiQ, sizeP, p, psiP, dnum)
    assert o.orc_hybrid_alpha(hy) == 7
    allq = np.concatenate([q, p])
    ctx = fh.Context(hip, logN, allq, np.concatenate([psiQ, psiP]))
    plan = fh.KeySwitchPlan(ctx, sizeQ, sizeP, dnum)
    keyB, keyA = libs.rand_tower(rng, allq, N, dnum), libs.rand_tower(rng, allq, N, dnum)
    plan.upload_key(keyB, keyA)
    a0, a1, b0, b1 = (libs.rand_tower(rng, q, N, B) for _ in range(4))
    w0, w1 = np.empty_like(a0), np.empty_like(a0)
    for bb in range(B):
        o.orc_hybrid_key_switch(hy, a0[bb], sizeQ, keyB, keyA, w0[bb], w1[bb])
    g0, g1 = plan.KeySwitchCore(ctx.tower(a0))
    assert np.array_equal(g0.to_host(), w0) and np.array_equal(g1.to_host(), w1), "KeySwitchCore at config 3's shape"
    g0.free(), g1.free()
    c0, c1 = np.empty_like(a0), np.empty_like(a0)
    for bb in range(B):
        o.orc_ckks_eval_mult_relin(hy, a0[bb], a1[bb], b0[bb], b1[bb], sizeQ, keyB, keyA, c0[bb], c1[bb])
    r0, r1 = plan.EvalMult(ctx.tower(a0), ctx.tower(a1), ctx.tower(b0), ctx.tower(b1))
    assert np.array_equal(r0.to_host(), c0) and np.array_equal(r1.to_host(), c1), "EvalMult at config 3's shape"
    # rescale of the product (DropLastElementAndScale), both elements
    octx = o.orc_ctx_create(N, sizeQ, q, psiQ)
    for got, src in ((fh.rescale(ctx, r0), c0), (fh.rescale(ctx, r1), c1)):
        want = np.empty((B, sizeQ - 1, N), np.uint64)
        for bb in range(B):
            o.orc_drop_last_element_and_scale(octx, src[bb], sizeQ, want[bb])
        assert np.array_equal(got.to_host(), want), "rescale at config 3's shape"
    # one lower level as well (a different digit split: 3 digits of 7, 7, 5 limbs -> level 19)
    sizeQl = 19
    x = libs.rand_tower(rng, q[:sizeQl], N, 1)
    w0, w1 = np.empty_like(x), np.empty_like(x)
    o.orc_hybrid_key_switch(hy, x[0], sizeQl, keyB, keyA, w0[0], w1[0])
    g0, g1 = plan.KeySwitchCore(ctx.tower(x))
    assert np.array_equal(g0.to_host(), w0) and np.array_equal(g1.to_host(), w1), "KeySwitchCore at level 19"
    o.orc_ctx_destroy(octx)
    plan.close()
    ctx.close()
    o.orc_hybrid_destroy(hy)


def test_config5_bfv_eval_mult_at_full_shape(hip, oracle):
    """N = 2^15, Q = 7 x 60-bit limbs, Bsk = 8 limbs, t = 65537: EvalMultNoRelin (BEHZ) and cc->EvalMult (HYBRID, dnum = 3)"""
    o = oracle
    rng = np.random.default_rng(205)
    logN, numQ, t, dnum, B = 15, 7, 65537, 3, 2
    N = 1 << logN
    M = 2 * N
    q, psiQ = hip.ckks_like_chain(logN, numQ, 60, 60)  # the bench leg's moduli: 7 distinct 60-bit primes, descending
    for i, v in enumerate(q):
        assert o.orc_is_prime(int(v)) and (int(v) - 1) % M == 0 and o.orc_root_of_unity(M, int(v)) == int(psiQ[i])
    hb = o.orc_behz_create(N, numQ, q, t)
    nb = o.orc_behz_num_bsk(hb)
    assert nb == 8
    bsk, psiB = np.zeros(nb, np.uint64), np.zeros(nb, np.uint64)
    o.orc_behz_get_bsk(hb, bsk, psiB)
    b2, _ = hip.behz_bsk(logN, q, t)
    assert np.array_equal(b2, bsk)
    p, psiP = np.zeros(64, np.uint64), np.zeros(64, np.uint64)
    sizeP = o.orc_hybrid_select_p(N, numQ, q, dnum, 60, p, psiP)
    p, psiP = p[:sizeP].copy(), psiP[:sizeP].copy()
    allq = np.concatenate([q, p, bsk])
    ctx = fh.Context(hip, logN, allq, np.concatenate([psiQ, psiP, psiB]))
    behz = fh.Behz(ctx, np.arange(numQ), np.arange(numQ + sizeP, numQ + sizeP + nb), t)
    X = [libs.rand_tower(rng, q, N, B) for _ in range(4)]
    call = o.orc_ctx_create(N, numQ + nb, np.concatenate([q, bsk]), np.concatenate([psiQ, psiB]))
    want = np.zeros((3, B, numQ, N), np.uint64)
    for b in range(B):
        o.orc_bfv_eval_mult_behz(hb, call, X[0][b], X[1][b], X[2][b], X[3][b], want[0, b], want[1, b], want[2, b])
    T = [ctx.tower(x, limb_idx=np.arange(numQ)) for x in X]
    got = behz.EvalMultNoRelin(*T)
    for k in range(3):
        assert np.array_equal(got[k].to_host(), want[k]), f"BFV EvalMultNoRelin element {k} at config 5's shape"
    # with relinearisation: SetFormat(EVALUATION), KeySwitchCore on the third element, adds (base-leveledshe.cpp:201-214)
    hy = o.orc_hybrid_create(N, numQ, q, psiQ, sizeP, p, psiP, dnum)
    qp = np.concatenate([q, p])
    keyB, keyA = libs.rand_tower(rng, qp, N, dnum), libs.rand_tower(rng, qp, N, dnum)
    ks = fh.KeySwitchPlan(ctx, numQ, sizeP, dnum)
    ks.upload_key(keyB, keyA)
    octxQ = o.orc_ctx_create(N, numQ, q, psiQ)
    ev = want.copy()
    for k in range(3):
        o.orc_ntt_fwd_tower(octxQ, ev[k], None, numQ, B, 0)
    c0w, c1w = np.empty((B, numQ, N), np.uint64), np.empty((B, numQ, N), np.uint64)
    for b in range(B):
        k0, k1 = np.empty((numQ, N), np.uint64), np.empty((numQ, N), np.uint64)
        o.orc_hybrid_key_switch(hy, ev[2, b], numQ, keyB, keyA, k0, k1)
        for l in range(numQ):
            o.orc_vec_add(c0w[b, l], ev[0, b, l], k0[l], N, q[l])
            o.orc_vec_add(c1w[b, l], ev[1, b, l], k1[l], N, q[l])
    c0, c1 = behz.EvalMult(ks, *T)
    assert np.array_equal(c0.to_host(), c0w) and np.array_equal(c1.to_host(), c1w), "BFV EvalMult with relinearisation"
    for h in (call, octxQ):
        o.orc_ctx_destroy(h)
    o.orc_hybrid_destroy(hy)
    o.orc_behz_destroy(hb)
    ks.close()
    behz.close()
    ctx.close()


def test_bsgs_transform_at_the_bootstrapping_ring(hip, oracle):
    """fhe_ckks_bsgs_transform at the bench leg's shape: N = 2^17, l = 21, dnum = 3, 64 diagonals = 8 baby x 8 giant steps
    (7 + 7 rotation keys).  Keys and diagonals reuse a few host images (the arithmetic does not care), as bench.py does."""
    o = oracle
    rng = np.random.default_rng(204)
    logN, sizeQ, dnum, bStep, gStep = 17, 21, 3, 8, 8
    N = 1 << logN
    q, psiQ, p, psiP = ckks_like_params(o, logN, sizeQ, dnum, first_bits=60, scale_bits=59, aux_bits=60)
    sizeP = len(p)
    hy = o.orc_hybrid_create(N, sizeQ, q, psiQ, sizeP, p, psiP, dnum)
    allq = np.concatenate([q, p])
    ctx = fh.Context(hip, logN, allq, np.concatenate([psiQ, psiP]))
    plan = fh.KeySwitchPlan(ctx, sizeQ, sizeP, dnum)
    kimg = [libs.rand_tower(rng, allq, N, dnum) for _ in range(3)]
    dimg = [libs.rand_tower(rng, allq, N) for _ in range(3)]
    ddev = [ctx.upload(d) for d in dimg]
    handles = []

    def rot(index, n):
        k = o.orc_find_automorphism_index_2n_complex(index, 2 * N)
        kb, ka = kimg[n % 3], kimg[(n + 1) % 3]
        handles.append(plan.make_key(kb, ka))
        return (k, kb, ka), (k, handles[-1])
    ins = [(None, None)] + [rot(i, i) for i in range(1, bStep)]
    outs = [(None, None)] + [rot(bStep * j, j + 1) for j in range(1, gStep)]
    sel = [[(3 * i + j) % 3 for j in range(bStep)] for i in range(gStep)]
    diag = [[dimg[sel[i][j]] for j in range(bStep)] for i in range(gStep)]
    dptr = [[ddev[sel[i][j]] for j in range(bStep)] for i in range(gStep)]
    c0, c1 = libs.rand_tower(rng, q, N, 1), libs.rand_tower(rng, q, N, 1)
    want = run_oracle(o, hy, c0, c1, sizeQ, [r[0] for r in ins], [r[0] for r in outs], diag)
    g0, g1 = plan.BsgsTransform(ctx.tower(c0), ctx.tower(c1), [r[1] for r in ins], [r[1] for r in outs], dptr)
    assert np.array_equal(g0.to_host(), want[0]) and np.array_equal(g1.to_host(), want[1]), "BSGS transform at N=2^17, 8x8"
    for hnd in handles:
        hip.L.fhe_ks_key_destroy(hnd)
    plan.close()
    ctx.close()
    o.orc_hybrid_destroy(hy)
