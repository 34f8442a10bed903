"""Parity of the HIP kernels (through the C ABI) against the oracle on identical seeded inputs.

Every test runs twice: on the `emu` backend (TEST-ONLY lane emulator, CPU, small sizes) and, with
`-m gpu`, on the real library on an MI355X.  Bit-exact equality is required everywhere (integer path);
the only floating-point step (SwitchCRTBasis overflow count) is also required to be bit-exact because it
replicates the reference's operation order.
"""
import ctypes as C
import os

import numpy as np
import pytest

import libs
from openfhe_amd import fhe_hip as fh


def is_emu(lib):
    return "emulator" in lib.version()


def params(o, logN, L, bits=60):
    q = np.zeros(L, np.uint64)
    psi = np.zeros(L, np.uint64)
    o.orc_dcrt_params(2 << logN, L, bits, q, psi)
    return q, psi


def ntt_sizes(lib):
    small = [(4, 2, 3), (5, 1, 1), (7, 3, 2), (10, 2, 3), (11, 1, 5), (12, 2, 2), (13, 2, 1), (14, 1, 2)]
    if is_emu(lib):
        return small + [(15, 1, 1), (16, 1, 1), (17, 1, 1)]
    return small + [(15, 3, 2), (16, 4, 3), (17, 2, 2)]


def test_ntt_forward_inverse(backend, oracle):
    o = oracle
    rng = np.random.default_rng(11)
    for logN, L, B in ntt_sizes(backend):
        q, psi = params(o, logN, L)
        N = 1 << logN
        ctx = fh.Context(backend, logN, q, psi)
        octx = o.orc_ctx_create(N, L, q, psi)
        x = libs.rand_tower(rng, q, N, B)
        # edge values: 0, q-1
        x[0, :, 0] = 0
        x[0, :, 1] = q - np.uint64(1)
        want = x.copy()
        o.orc_ntt_fwd_tower(octx, want, None, L, B, 0)
        t = ctx.tower(x, fmt=fh.COEFFICIENT)
        t.SwitchFormat()
        got = t.to_host()
        assert np.array_equal(got, want), f"forward NTT mismatch logN={logN}"
        t.SwitchFormat()
        assert np.array_equal(t.to_host(), x), f"round trip mismatch logN={logN}"
        # inverse from an independent EVALUATION input
        y = libs.rand_tower(rng, q, N, B)
        wanti = y.copy()
        o.orc_ntt_inv_tower(octx, wanti, None, L, B, 0)
        t2 = ctx.tower(y, fmt=fh.EVALUATION)
        t2.SwitchFormat()
        assert np.array_equal(t2.to_host(), wanti), f"inverse NTT mismatch logN={logN}"
        o.orc_ctx_destroy(octx)
        ctx.close()


def test_ntt_limb_selection_and_oop(backend, oracle):
    """towers at a lower level / arbitrary limb subsets share one context through limbIdx"""
    o = oracle
    rng = np.random.default_rng(12)
    logN, L = (12, 5) if is_emu(backend) else (13, 5)
    N = 1 << logN
    q, psi = params(o, logN, L)
    ctx = fh.Context(backend, logN, q, psi)
    octx = o.orc_ctx_create(N, L, q, psi)
    sel = np.array([4, 1, 3], np.uint32)
    x = libs.rand_tower(rng, q[sel], N, 2)
    want = x.copy()
    o.orc_ntt_fwd_tower(octx, want, sel.ctypes.data, 3, 2, 0)
    t = ctx.tower(x, limb_idx=sel, fmt=fh.COEFFICIENT)
    out = ctx.empty(2, 3, sel)
    backend.check(backend.L.fhe_ntt_fwd_oop(ctx.h, t.ptr, out.ptr, sel.ctypes.data_as(fh.u32p), 3, 2, None))
    assert np.array_equal(out.to_host(), want)
    assert np.array_equal(t.to_host(), x), "out-of-place transform must not touch its input"
    o.orc_ctx_destroy(octx)
    ctx.close()


def test_config1_cpu_reference_case(backend, oracle):
    """BASELINE config 1: N=2^12, 2x60-bit limbs, fwd+inv round trip bit-exact, forward words vs oracle"""
    o = oracle
    q, psi = params(o, 12, 2)
    assert list(q) == [1152921504606830593, 1152921504606748673]
    rng = np.random.default_rng(1)
    x = libs.rand_tower(rng, q, 4096, 1)
    ctx = fh.Context(backend, 12, q, psi)
    octx = o.orc_ctx_create(4096, 2, q, psi)
    want = x.copy()
    o.orc_ntt_fwd_tower(octx, want, None, 2, 1, 0)
    t = ctx.tower(x, fmt=fh.COEFFICIENT).SwitchFormat()
    assert np.array_equal(t.to_host(), want)
    assert np.array_equal(t.SwitchFormat().to_host(), x)
    o.orc_ctx_destroy(octx)
    ctx.close()


def test_elementwise(backend, oracle):
    o = oracle
    rng = np.random.default_rng(13)
    for logN, L, B in [(6, 3, 2), (12, 2, 2), (13, 3, 1)]:
        N = 1 << logN
        q, psi = params(o, logN, L)
        ctx = fh.Context(backend, logN, q, psi)
        a = libs.rand_tower(rng, q, N, B)
        b = libs.rand_tower(rng, q, N, B)
        a[0, :, 0] = 0
        b[0, :, 0] = q - np.uint64(1)
        a[0, :, 1] = q - np.uint64(1)
        b[0, :, 1] = q - np.uint64(1)
        ta, tb = ctx.tower(a), ctx.tower(b)
        for name, fn in (("Plus", o.orc_vec_add), ("Minus", o.orc_vec_sub), ("Times", o.orc_vec_mul)):
            want = np.empty_like(a)
            for bb in range(B):
                for l in range(L):
                    fn(want[bb, l], a[bb, l], b[bb, l], N, q[l])
            got = getattr(ta, name)(tb).to_host()
            assert np.array_equal(got, want), f"{name} mismatch logN={logN}"
        consts = rng.integers(1, 1 << 59, size=L, dtype=np.uint64) % q
        want = np.empty_like(a)
        for bb in range(B):
            for l in range(L):
                o.orc_vec_mul_const(want[bb, l], a[bb, l], consts[l], N, q[l])
        assert np.array_equal(ta.Times(consts).to_host(), want)
        want = np.empty_like(a)
        for bb in range(B):
            for l in range(L):
                o.orc_vec_neg(want[bb, l], a[bb, l], N, q[l])
        assert np.array_equal(ta.Negate().to_host(), want)
        # fhe_mul_add: acc += a * b (the key-switch inner product's accumulation, keyswitch-hybrid.cpp:419-430)
        acc = libs.rand_tower(rng, q, N, B)
        want = np.empty_like(a)
        for bb in range(B):
            for l in range(L):
                o.orc_vec_mul(want[bb, l], a[bb, l], b[bb, l], N, q[l])
                o.orc_vec_add(want[bb, l], want[bb, l], acc[bb, l], N, q[l])
        tacc = ctx.tower(acc)
        backend.check(backend.L.fhe_mul_add(ctx.h, tacc.ptr, ta.ptr, tb.ptr, None, L, B, None))
        assert np.array_equal(tacc.to_host(), want)
        ctx.close()


def test_inner_product_over_separate_towers(backend, oracle):
    """fhe_inner_product = the sum of EvalFastKeySwitchCoreExt (keyswitch-hybrid.cpp:419-430) with every digit and every key
    element in its own allocation; key rows through keyRow (the reference's idx(i)), one or two outputs, batch; more than 8 terms run
    as several launches whose exact sums add up (the reference's loop over digits has no bound)"""
    o = oracle
    rng = np.random.default_rng(5150)
    vp = C.c_void_p
    for logN, rows, keyRows, nTerms, B, two in [(4, 2, 3, 1, 1, True), (11, 5, 8, 3, 2, True), (12, 4, 4, 8, 1, False),
                                                (13, 3, 6, 2, 1, True), (6, 3, 4, 9, 2, True), (5, 2, 2, 19, 1, False)]:
        N = 1 << logN
        q, psi = params(o, logN, keyRows)
        ctx = fh.Context(backend, logN, q, psi)
        # output row i lives on context limb limb[i] and multiplies key row keyRow[i] (= the same limb of the key tower)
        skip = keyRows - rows
        keyRow = np.array([i if i < rows - 1 else i + skip for i in range(rows)], np.uint32)
        limb = keyRow.copy()
        xs = [np.stack([np.stack([rng.integers(0, int(q[l]), N, dtype=np.uint64) for l in limb]) for _ in range(B)]) for _ in range(nTerms)]
        xs[0][0, 0, :3] = q[limb[0]] - np.uint64(1)
        k0 = [libs.rand_tower(rng, q, N) for _ in range(nTerms)]
        k1 = [libs.rand_tower(rng, q, N) for _ in range(nTerms)]
        k0[0][limb[0], :3] = q[limb[0]] - np.uint64(1)
        want0 = np.empty((B, rows, N), np.uint64)
        want1 = np.empty((B, rows, N), np.uint64)
        for bb in range(B):
            for i in range(rows):
                for keys, want in ((k0, want0), (k1, want1)):
                    xp = (vp * nTerms)(*[x[bb, i].ctypes.data for x in xs])
                    kp = (vp * nTerms)(*[k[keyRow[i]].ctypes.data for k in keys])
                    o.orc_vec_inner_product(want[bb, i], xp, kp, nTerms, N, q[limb[i]])
        tx = [ctx.tower(x, limb_idx=limb) for x in xs]
        tk0 = [ctx.tower(k[None]) for k in k0]
        tk1 = [ctx.tower(k[None]) for k in k1]
        out0, out1 = tx[0].like(), tx[0].like()
        px = (vp * nTerms)(*[t.ptr for t in tx])
        p0 = (vp * nTerms)(*[t.ptr for t in tk0])
        p1 = (vp * nTerms)(*[t.ptr for t in tk1])
        u32p = C.POINTER(C.c_uint32)
        backend.check(backend.L.fhe_inner_product(ctx.h, nTerms, px, p0, p1 if two else None, keyRow.ctypes.data_as(u32p),
                                                  limb.ctypes.data_as(u32p), rows, B, out0.ptr, out1.ptr if two else None, None))
        assert np.array_equal(out0.to_host(), want0), f"inner product (b half) logN={logN}"
        if two:
            assert np.array_equal(out1.to_host(), want1), f"inner product (a half) logN={logN}"
        ctx.close()


def test_plus_minus_constants(backend, oracle):
    """fhe_add_const / fhe_sub_const = DCRTPolyImpl::Plus / Minus(vector<Integer>) (dcrtpoly-impl.h:520-548): every word in
    EVALUATION, coefficient 0 only for Plus in COEFFICIENT; constants >= q are reduced first; in place and out of place"""
    o = oracle
    rng = np.random.default_rng(977)
    u64p = C.POINTER(C.c_uint64)
    for logN, L, B in [(4, 2, 1), (11, 5, 2), (13, 3, 1)]:
        N = 1 << logN
        q, psi = params(o, logN, L)
        ctx = fh.Context(backend, logN, q, psi)
        a = libs.rand_tower(rng, q, N, B)
        a[0, :, 0] = q - np.uint64(1)
        consts = np.array([int(rng.integers(0, 1 << 62)) for _ in range(L)], np.uint64)
        consts[0] = 0
        if L > 2:
            consts[2] = q[2]  # reduces to zero
        for mode in ("plus", "plus0", "minus"):
            want = np.empty_like(a)
            for bb in range(B):
                for l in range(L):
                    if mode == "minus":
                        o.orc_vec_sub_const(want[bb, l], a[bb, l], consts[l], N, q[l])
                    else:
                        o.orc_vec_add_const(want[bb, l], a[bb, l], consts[l], N, q[l], 1 if mode == "plus0" else 0)
            ta = ctx.tower(a)
            out = ta.like()
            cp = consts.ctypes.data_as(u64p)
            if mode == "minus":
                backend.check(backend.L.fhe_sub_const(ctx.h, out.ptr, ta.ptr, cp, None, L, B, None))
                backend.check(backend.L.fhe_sub_const(ctx.h, ta.ptr, ta.ptr, cp, None, L, B, None))
            else:
                backend.check(backend.L.fhe_add_const(ctx.h, out.ptr, ta.ptr, cp, None, L, B, int(mode == "plus0"), None))
                backend.check(backend.L.fhe_add_const(ctx.h, ta.ptr, ta.ptr, cp, None, L, B, int(mode == "plus0"), None))
            assert np.array_equal(out.to_host(), want), f"{mode} logN={logN}"
            assert np.array_equal(ta.to_host(), want), f"{mode} in place logN={logN}"
        ctx.close()


def test_mult_acc_square_ql_hat_async_consts(backend, oracle):
    """fhe_mult_acc (MultAccEqNoCheck), fhe_tensor_square (EvalSquareCore), fhe_expand_crt_basis_ql_hat (ExpandCRTBasisQlHat);
    the host constants of fhe_mul_const / fhe_mult_acc travel in the kernel arguments: the host array may change right after
    the call returns (no staging buffer, no synchronisation inside the call)"""
    o = oracle
    rng = np.random.default_rng(131)
    for logN, L, B in [(5, 3, 2), (12, 4, 2), (13, 2, 1)]:
        N = 1 << logN
        q, psi = params(o, logN, L)
        ctx = fh.Context(backend, logN, q, psi)
        a, v = libs.rand_tower(rng, q, N, B), libs.rand_tower(rng, q, N, B)
        consts = np.array([int(rng.integers(0, 1 << 62)) for _ in range(L)], np.uint64)  # some >= q: reduced first
        want = a.copy()
        for bb in range(B):
            for l in range(L):
                o.orc_vec_mult_acc(want[bb, l], v[bb, l], consts[l], N, q[l])
        ta, tv = ctx.tower(a), ctx.tower(v)
        c2 = consts.copy()
        ta.MultAccEqNoCheck(tv, c2)
        c2[:] = 0  # the call has returned: its constants are already in the launch
        assert np.array_equal(ta.to_host(), want), f"MultAccEqNoCheck logN={logN}"
        # many constant multiplications in flight on one stream (the former 64-slot ring would have wrapped)
        outs, wants = [], []
        tv0 = ctx.tower(v)
        for it in range(80):
            cs = np.array([int(rng.integers(1, int(m))) for m in q], np.uint64)
            if it % 16 == 0:
                w = np.empty_like(v)
                for bb in range(B):
                    for l in range(L):
                        o.orc_vec_mul_const(w[bb, l], v[bb, l], cs[l], N, q[l])
                wants.append(w)
                outs.append(tv0.Times(cs))
            else:
                tv0.Times(cs).free()
        for t, w in zip(outs, wants):
            assert np.array_equal(t.to_host(), w)
        # EvalSquareCore
        d = [ta.like() for _ in range(3)]
        backend.check(backend.L.fhe_tensor_square(ctx.h, tv.ptr, tv0.ptr, d[0].ptr, d[1].ptr, d[2].ptr, None, L, B, None))
        w = [np.empty_like(v) for _ in range(3)]
        for bb in range(B):
            o.orc_eval_square_core(v[bb], v[bb], L, N, q, w[0][bb], w[1][bb], w[2][bb])
        for e in range(3):
            assert np.array_equal(d[e].to_host(), w[e]), f"EvalSquareCore element {e}"
        # ExpandCRTBasisQlHat: the first sizeQl limbs scaled, the rest zero
        for sizeQl in sorted({1, L - 1, L} - {0}):
            x = libs.rand_tower(rng, q[:sizeQl], N, B)
            h = np.array([int(rng.integers(1, int(m))) for m in q[:sizeQl]], np.uint64)
            tx, out = ctx.tower(x), ctx.empty(B, L)
            backend.check(backend.L.fhe_expand_crt_basis_ql_hat(ctx.h, tx.ptr, sizeQl, h.ctypes.data_as(fh.u64p), None, L, B,
                                                                out.ptr, None))
            w = np.ones((B, L, N), np.uint64)
            for bb in range(B):
                o.orc_expand_crt_basis_ql_hat(x[bb], sizeQl, N, q, h, L, w[bb])
            assert np.array_equal(out.to_host(), w), f"ExpandCRTBasisQlHat sizeQl={sizeQl}"
        ctx.close()


def test_switch_modulus_and_mod_raise(backend, oracle):
    """fhe_switch_modulus (NativeVectorT::SwitchModulus, a9) directly: one source limb of a tower lifted, centred, into
    every row of the output — in both directions (target modulus above and below the source) — and the ModRaise
    constructor DCRTPolyImpl(const PolyType&, params) (dcrtpoly-impl.h:87-93) as its srcLimbs = 1 case"""
    o = oracle
    rng = np.random.default_rng(132)
    for logN, L, B in [(4, 4, 2), (12, 3, 2), (13, 5, 1)]:
        N = 1 << logN
        q, psi = ckks_like_params(o, logN, L, 2, first_bits=60, scale_bits=45)[:2]  # q_0 (60 bits) above the others
        ctx = fh.Context(backend, logN, q, psi)
        for srcPos in (0, L - 1):  # 0: every target below the source; L-1: every other target above it
            x = libs.rand_tower(rng, q, N, B)
            x[0, srcPos, :4] = (0, 1, q[srcPos] >> np.uint64(1), q[srcPos] - np.uint64(1))
            x[0, srcPos, 4] = (q[srcPos] >> np.uint64(1)) + np.uint64(1)
            tx, out = ctx.tower(x), ctx.empty(B, L)
            backend.check(backend.L.fhe_switch_modulus(ctx.h, out.ptr, None, L, tx.ptr, L, srcPos, srcPos, B, None))
            want = np.empty((B, L, N), np.uint64)
            for bb in range(B):
                for l in range(L):
                    want[bb, l] = x[bb, srcPos]
                    o.orc_switch_modulus(want[bb, l], N, q[srcPos], q[l])
            assert np.array_equal(out.to_host(), want), f"SwitchModulus logN={logN} srcPos={srcPos}"
        # ModRaise: a single polynomial modulo q_0 into a full tower
        x0 = libs.rand_tower(rng, q[:1], N, B)
        tx, out = ctx.tower(x0), ctx.empty(B, L)
        backend.check(backend.L.fhe_switch_modulus(ctx.h, out.ptr, None, L, tx.ptr, 1, 0, 0, B, None))
        want = np.empty((B, L, N), np.uint64)
        for bb in range(B):
            o.orc_mod_raise(x0[bb, 0], N, q, L, want[bb])
        assert np.array_equal(out.to_host(), want), "ModRaise"
        ctx.close()


@pytest.mark.parametrize("logN,nQ,nP,B", [(5, 2, 3, 2), (12, 3, 5, 1), (12, 7, 7, 1), (13, 4, 4, 2), (10, 12, 3, 1)])
def test_approx_mod_up(backend, oracle, logN, nQ, nP, B):
    """fhe_mod_up = DCRTPolyImpl::ApproxModUp (dcrtpoly-impl.h:935-963), from both formats"""
    o = oracle
    rng = np.random.default_rng(133)
    N = 1 << logN
    q, psi = params(o, logN, nQ + nP)
    ctx = fh.Context(backend, logN, q, psi)
    octx = o.orc_ctx_create(N, nQ + nP, q, psi)
    src, dst = q[:nQ], q[nQ:]
    hatInv, hatPre, hatMod, _, _, mu = libs.crt_tables(src, dst)
    conv = fh.Conv(ctx, np.arange(nQ), np.arange(nQ, nQ + nP))
    for fmt, inEval in ((fh.EVALUATION, 1), (fh.COEFFICIENT, 0)):
        x = libs.rand_tower(rng, src, N, B)
        want = np.empty((B, nQ + nP, N), np.uint64)
        for bb in range(B):
            o.orc_approx_mod_up(octx, nQ, nP, x[bb], inEval, hatInv, hatPre, hatMod, mu, want[bb])
        got = conv.ApproxModUp(ctx.tower(x, limb_idx=np.arange(nQ), fmt=fmt))
        assert np.array_equal(got.to_host(), want), f"ApproxModUp inEval={inEval}"
    conv.close()
    o.orc_ctx_destroy(octx)
    ctx.close()


def test_automorphism(backend, oracle):
    o = oracle
    rng = np.random.default_rng(14)
    for logN, L, B in [(5, 2, 2), (12, 2, 1), (13, 2, 2)]:
        N = 1 << logN
        q, psi = params(o, logN, L)
        ctx = fh.Context(backend, logN, q, psi)
        x = libs.rand_tower(rng, q, N, B)
        x[0, :, 3] = 0  # COEFF branch stores q - 0 = q unreduced, as the reference does
        for k in (3, 5, 2 * N - 1, o.orc_find_automorphism_index_2n_complex(7, 2 * N)):
            pre = np.zeros(N, np.uint32)
            o.orc_precompute_auto_map(N, k, pre)
            want = np.empty_like(x)
            wantc = np.empty_like(x)
            for bb in range(B):
                for l in range(L):
                    o.orc_automorph_eval(want[bb, l], x[bb, l], N, pre)
                    o.orc_automorph_coeff(wantc[bb, l], x[bb, l], N, k, q[l])
            assert np.array_equal(ctx.tower(x, fmt=fh.EVALUATION).AutomorphismTransform(k).to_host(), want)
            assert np.array_equal(ctx.tower(x, fmt=fh.COEFFICIENT).AutomorphismTransform(k).to_host(), wantc)
        ctx.close()


def test_automorphism_even_index_rejected(backend, oracle):
    q, psi = params(oracle, 5, 1)
    ctx = fh.Context(backend, 5, q, psi)
    t = ctx.tower(np.zeros((1, 1, 32), np.uint64))
    with pytest.raises(fh.FheError, match="Automorphism index not odd"):  # poly-impl.h:337-338
        t.AutomorphismTransform(4)
    ctx.close()


def test_context_argument_checks(backend, oracle):
    q, psi = params(oracle, 6, 2)
    with pytest.raises(fh.FheError):
        fh.Context(backend, 6, q, psi + np.uint64(1))  # not a primitive root
    with pytest.raises(fh.FheError):
        fh.Context(backend, 3, q, psi)  # logN out of range
    with pytest.raises(fh.FheError):
        fh.Context(backend, 6, np.array([97, 193], np.uint64) * np.uint64(1 << 58), psi)


@pytest.mark.parametrize("logN,L,B", [(5, 129, 2), (4, 256, 1), (12, 130, 1)])
def test_towers_of_more_than_128_rows(backend, oracle, logN, L, B):
    """One tower / one context with up to 256 limbs (round 6; the reference's UTBFVRNS TestMultiplicativeDepthLimitation reaches 129
    distinct moduli in one operation, bfvrns-cryptoparameters.cpp:673-712): the limb map of a launch is one byte per row for 256 rows,
    the by-value constant vector of Times(vector<Integer>) goes in windows of 128 rows (elemwise_kernels.h kConstVecLimbs)."""
    o = oracle
    rng = np.random.default_rng(1290 + L)
    N = 1 << logN
    q, psi = params(o, logN, L)
    assert len(set(int(v) for v in q)) == L
    ctx = fh.Context(backend, logN, q, psi)
    octx = o.orc_ctx_create(N, L, q, psi)
    x = libs.rand_tower(rng, q, N, B)
    x[0, :, 0] = 0
    x[0, :, 1] = q - np.uint64(1)
    want = x.copy()
    o.orc_ntt_fwd_tower(octx, want, None, L, B, 0)
    t = ctx.tower(x, fmt=fh.COEFFICIENT)
    t.SwitchFormat()
    assert np.array_equal(t.to_host(), want), "forward NTT"
    t.SwitchFormat()
    assert np.array_equal(t.to_host(), x), "round trip"
    # a selection of rows in another order than the context's (the map is what is being widened): the last 140 limbs, reversed
    nSel = min(L, 140)
    sel = np.arange(L - 1, L - 1 - nSel, -1).astype(np.uint32)
    ys = libs.rand_tower(rng, q[sel], N, B)
    wants = ys.copy()
    o.orc_ntt_fwd_tower(octx, wants, sel.ctypes.data, nSel, B, 0)
    ts = ctx.tower(ys, limb_idx=sel, fmt=fh.COEFFICIENT)
    ts.SwitchFormat()
    assert np.array_equal(ts.to_host(), wants), "forward NTT over a selection"
    # element-wise members: tower (.) tower, and the per-limb constants in windows
    a, b = libs.rand_tower(rng, q, N, B), libs.rand_tower(rng, q, N, B)
    ta, tb = ctx.tower(a), ctx.tower(b)
    for name, fn in (("Plus", o.orc_vec_add), ("Times", o.orc_vec_mul)):
        w = np.empty_like(a)
        for bb in range(B):
            for l in range(L):
                fn(w[bb, l], a[bb, l], b[bb, l], N, q[l])
        assert np.array_equal(getattr(ta, name)(tb).to_host(), w), name
    consts = rng.integers(1, 1 << 59, size=L, dtype=np.uint64) % q
    w = np.empty_like(a)
    for bb in range(B):
        for l in range(L):
            o.orc_vec_mul_const(w[bb, l], a[bb, l], consts[l], N, q[l])
    assert np.array_equal(ta.Times(consts).to_host(), w), "Times(vector<Integer>)"
    o.orc_ctx_destroy(octx)
    # basis conversion into more than 128 target limbs (ExpandCRTBasis of a deep BFV parameter set: Q -> Q u Bsk)
    nS = L // 2
    nD = L - nS
    src_idx, dst_idx = np.arange(nS, dtype=np.uint32), np.arange(nS, L, dtype=np.uint32)
    src, dst = q[:nS], q[nS:]
    hatInv, hatPre, hatMod, mu = conv_tables(o, src, dst)
    xs = libs.rand_tower(rng, src, N, B)
    conv = fh.Conv(ctx, src_idx, dst_idx)
    wc = np.empty((B, nD, N), np.uint64)
    for bb in range(B):
        o.orc_approx_switch_crt_basis(xs[bb], nS, N, src, hatInv, hatPre, hatMod, nD, dst, mu, wc[bb])
    assert np.array_equal(conv.run(ctx.tower(xs, limb_idx=src_idx, fmt=fh.COEFFICIENT)).to_host(), wc), "ApproxSwitchCRTBasis"
    conv.close()
    ctx.close()


def test_context_of_more_than_256_limbs_is_refused(backend, oracle):
    q, psi = params(oracle, 4, 257)
    with pytest.raises(Exception, match="256"):
        fh.Context(backend, 4, q, psi)


@pytest.mark.parametrize("logN,L,bits,baseBits,ev", [(4, 3, 60, 0, 0), (5, 4, 60, 4, 1), (12, 2, 50, 7, 0), (13, 3, 60, 0, 1), (5, 3, 60, 1, 0),
                                                     (6, 3, 36, 16, 1), (12, 2, 60, 30, 0), (5, 5, 45, 3, 1)])
def test_crt_decompose(backend, oracle, logN, L, bits, baseBits, ev):
    """fhe_crt_decompose = DCRTPolyImpl::CRTDecompose (dcrtpoly-impl.h:230-285; the digit decomposition of KeySwitchBV) from both formats;
    the oracle function is pinned on the reference's member (tests/test_oracle_vs_ref.py)"""
    o = oracle
    N = 1 << logN
    rng = np.random.default_rng(930 + baseBits)
    q, psi = params(o, logN, L, bits)
    ctx = fh.Context(backend, logN, q, psi)
    octx = o.orc_ctx_create(N, L, q, psi)
    x = libs.rand_tower(rng, q, N, 1)
    x[0, :, 0] = 0
    x[0, :, 1] = q - np.uint64(1)
    x[0, :, 2] = q >> np.uint64(1)
    x[0, :, 3] = (q >> np.uint64(1)) + np.uint64(1)
    towers = o.orc_crt_decompose(octx, x.ctypes.data, L, baseBits, None)
    want = np.zeros((towers, L, N), np.uint64)
    o.orc_crt_decompose(octx, x.ctypes.data, L, baseBits, want.ctypes.data)
    xin = x.copy()
    if ev:
        o.orc_ntt_fwd_tower(octx, xin, None, L, 1, 0)
    t = ctx.tower(xin, fmt=fh.EVALUATION if ev else fh.COEFFICIENT)
    got = t.CRTDecompose(baseBits)
    assert got is not None and got.batch == towers
    assert np.array_equal(got.to_host(), want)
    assert np.array_equal(t.to_host(), xin), "the member is const"
    o.orc_ctx_destroy(octx)
    ctx.close()


def test_crt_decompose_declines_windows_beyond_the_word(backend, oracle):
    """ceil(msb(q) / baseBits) * baseBits > 64 (e.g. 60-bit moduli cut into digits of 25 bits): the reference shifts a 64-bit word by
    64 and more there (undefined); the library reports 0 towers and the caller keeps its host path"""
    q, psi = params(oracle, 5, 2, 60)
    ctx = fh.Context(backend, 5, q, psi)
    assert backend.L.fhe_crt_decompose_towers(ctx.h, None, 2, 25) == 0
    assert backend.L.fhe_crt_decompose_towers(ctx.h, None, 2, 32) == 0
    assert backend.L.fhe_crt_decompose_towers(ctx.h, None, 2, 20) == 6
    ctx.close()


def conv_tables(o, src, dst):
    """host tables exactly as the oracle's hybrid code derives them: hatInv[i], hatMod[i][j], mu128[j]"""
    nS, nD = len(src), len(dst)
    hatInv = np.zeros(nS, np.uint64)
    hatPre = np.zeros(nS, np.uint64)
    hatMod = np.zeros((nS, nD), np.uint64)
    mu = np.zeros((nD, 2), np.uint64)
    for i in range(nS):
        h = 1
        for k in range(nS):
            if k != i:
                h = (h * int(src[k])) % int(src[i])
        hatInv[i] = pow(h, -1, int(src[i]))
        hatPre[i] = (int(hatInv[i]) << 64) // int(src[i])
        for j in range(nD):
            v = 1
            for k in range(nS):
                if k != i:
                    v = (v * int(src[k])) % int(dst[j])
            hatMod[i, j] = v
    for j in range(nD):
        m = (1 << 128) // int(dst[j])
        mu[j] = (m & ((1 << 64) - 1), m >> 64)
    return hatInv, hatPre, hatMod, mu


def test_approx_and_exact_switch_crt_basis(backend, oracle):
    o = oracle
    rng = np.random.default_rng(15)
    # (more than 32 source limbs: chunked plans, one launch per 32 — the reference's loop dcrtpoly-impl.h:895-915 has no bound)
    for logN, nS, nD, B in [(5, 2, 3, 2), (12, 3, 9, 1), (12, 7, 21, 1), (13, 4, 5, 2), (10, 12, 3, 1), (10, 20, 2, 1), (6, 33, 3, 2), (8, 40, 5, 1),
                            (5, 70, 2, 1)]:
        N = 1 << logN
        q, psi = params(o, logN, nS + nD)
        ctx = fh.Context(backend, logN, q, psi)
        src_idx = np.arange(nS, dtype=np.uint32)
        dst_idx = np.arange(nS, nS + nD, dtype=np.uint32)
        src, dst = q[:nS], q[nS:]
        hatInv, hatPre, hatMod, mu = conv_tables(o, src, dst)
        x = libs.rand_tower(rng, src, N, B)
        x[0, :, 0] = 0
        x[0, :, 1] = src - np.uint64(1)
        conv = fh.Conv(ctx, src_idx, dst_idx)
        tin = ctx.tower(x, limb_idx=src_idx, fmt=fh.COEFFICIENT)
        want = np.empty((B, nD, N), np.uint64)
        for bb in range(B):
            o.orc_approx_switch_crt_basis(x[bb], nS, N, src, hatInv, hatPre, hatMod, nD, dst, mu, want[bb])
        assert np.array_equal(conv.run(tin).to_host(), want), "ApproxSwitchCRTBasis mismatch"
        # exact variant: alphaQModp[a][j] = a*Q mod p_j, qInv = 1/q_i (double); QHatModp indexed [j][i]
        Q = 1
        for s in src:
            Q *= int(s)
        alpha = np.array([[(a * Q) % int(p) for p in dst] for a in range(nS + 1)], np.uint64)
        qinv = np.array([1.0 / float(int(s)) for s in src], np.float64)
        wante = np.empty((B, nD, N), np.uint64)
        hm_pq = np.ascontiguousarray(hatMod.T)
        for bb in range(B):
            o.orc_switch_crt_basis(x[bb], nS, N, src, hatInv, hatPre, hm_pq, alpha, nD, dst, mu, qinv, wante[bb])
        assert np.array_equal(conv.run(tin, exact=True).to_host(), wante), "SwitchCRTBasis mismatch"
        conv.close()
        ctx.close()


def ckks_like_params(o, logN, sizeQ, dnum, first_bits=60, scale_bits=50, aux_bits=60):
    """A CKKS-shaped tower without the reference's context: first modulus `first_bits`, the rest
    `scale_bits`, auxiliary P chosen the way PrecomputeCRTTables does (oracle restatement)."""
    M = 2 << logN
    q = [o.orc_last_prime(first_bits, M)]
    cur = o.orc_last_prime(scale_bits, M)
    for _ in range(sizeQ - 1):
        q.append(cur)
        cur = o.orc_previous_prime(cur, M)
    q = np.array(q, np.uint64)
    psiQ = np.array([o.orc_root_of_unity(M, int(v)) for v in q], np.uint64)
    p = np.zeros(64, np.uint64)
    psiP = np.zeros(64, np.uint64)
    sizeP = o.orc_hybrid_select_p(1 << logN, sizeQ, q, dnum, aux_bits, p, psiP)
    return q, psiQ, p[:sizeP].copy(), psiP[:sizeP].copy()


@pytest.mark.parametrize("logN,sizeQ,dnum,sizeQl,B", [(10, 4, 2, 4, 3), (8, 5, 2, 3, 2), (12, 6, 3, 6, 2), (12, 7, 2, 5, 1), (12, 5, 3, 2, 1), (13, 4, 2, 4, 1),
                                                      (13, 4, 2, 4, 4), (14, 5, 3, 4, 2),  # two-pass rings with whole groups of 8 (tower, tile) pairs (the shapes the round-6 fused-ModUp experiment took, commit efa3a5d)
                                                      (16, 2, 2, 2, 1), (17, 2, 2, 2, 1),  # 12-stage row passes: BASELINE configs[2] / [3] rings
                                                      # shapes past the kernels' per-launch bounds: a digit and a P basis of more than 32
                                                      # limbs (dnum = 1 on a 40-limb chain: chunked conversions), more than 8 digits (chunked
                                                      # inner products)
                                                      (10, 40, 1, 40, 1), (9, 36, 1, 34, 2), (9, 20, 10, 20, 2), (8, 24, 12, 19, 1)])
def test_hybrid_keyswitch_and_eval_mult(backend, oracle, logN, sizeQ, dnum, sizeQl, B):
    o = oracle
    if is_emu(backend) and logN > 12 and not (logN == 13 and B == 4) and not os.environ.get("FHE_TEST_BIG_EMU"):
        pytest.skip("emulator: keep the CPU suite short (FHE_TEST_BIG_EMU=1 runs these too)")
    rng = np.random.default_rng(16)
    N = 1 << logN
    q, psiQ, p, psiP = ckks_like_params(o, logN, sizeQ, dnum)
    sizeP = len(p)
    hy = o.orc_hybrid_create(N, sizeQ, q, psiQ, sizeP, p, psiP, dnum)
    allq = np.concatenate([q, p])
    ctx = fh.Context(backend, logN, allq, np.concatenate([psiQ, psiP]))
    plan = fh.KeySwitchPlan(ctx, sizeQ, sizeP, dnum)
    keyB = libs.rand_tower(rng, allq, N, dnum)
    keyA = libs.rand_tower(rng, allq, N, dnum)
    plan.upload_key(keyB, keyA)
    ql = q[:sizeQl]
    a0, a1, b0, b1 = (libs.rand_tower(rng, ql, N, B) for _ in range(4))
    # key switch alone
    w0 = np.empty_like(a0)
    w1 = np.empty_like(a0)
    for bb in range(B):
        o.orc_hybrid_key_switch(hy, a0[bb], sizeQl, keyB, keyA, w0[bb], w1[bb])
    g0, g1 = plan.KeySwitchCore(ctx.tower(a0))
    assert np.array_equal(g0.to_host(), w0) and np.array_equal(g1.to_host(), w1), "KeySwitchCore mismatch"
    # ... and accumulated into two towers (`cv[0] += ab[0]; cv[1] += ab[1]`, base-leveledshe.cpp:210-211)
    acc0, acc1 = ctx.tower(a1), ctx.tower(b0)
    plan.KeySwitchCoreAcc(ctx.tower(a0), acc0, acc1)
    wa0, wa1 = np.empty_like(a0), np.empty_like(a0)
    for bb in range(B):
        for l in range(sizeQl):
            o.orc_vec_add(wa0[bb, l], a1[bb, l], w0[bb, l], N, ql[l])
            o.orc_vec_add(wa1[bb, l], b0[bb, l], w1[bb, l], N, ql[l])
    assert np.array_equal(acc0.to_host(), wa0) and np.array_equal(acc1.to_host(), wa1), "KeySwitchCore (accumulating) mismatch"
    # EvalMult = tensor + key switch + add
    c0 = np.empty_like(a0)
    c1 = np.empty_like(a0)
    for bb in range(B):
        o.orc_ckks_eval_mult_relin(hy, a0[bb], a1[bb], b0[bb], b1[bb], sizeQl, keyB, keyA, c0[bb], c1[bb])
    r0, r1 = plan.EvalMult(ctx.tower(a0), ctx.tower(a1), ctx.tower(b0), ctx.tower(b1))
    assert np.array_equal(r0.to_host(), c0) and np.array_equal(r1.to_host(), c1), "EvalMult mismatch"
    # ApproxModDown alone
    x = libs.rand_tower(rng, np.concatenate([ql, p]), N, B)
    wd = np.empty((B, sizeQl, N), np.uint64)
    for bb in range(B):
        o.orc_hybrid_approx_mod_down(hy, x[bb], sizeQl, wd[bb])
    assert np.array_equal(plan.ApproxModDown(ctx.tower(x), sizeQl).to_host(), wd), "ApproxModDown mismatch"
    # ... and its BGV form (t^-1 mod p_j before, t mod q_i after the conversion)
    for t in (65537, 2):
        for bb in range(B):
            o.orc_hybrid_approx_mod_down_t(hy, x[bb], sizeQl, t, wd[bb])
        assert np.array_equal(plan.ApproxModDown(ctx.tower(x), sizeQl, t=t).to_host(), wd), f"ApproxModDown (BGV, t={t}) mismatch"
    plan.close()
    ctx.close()
    o.orc_hybrid_destroy(hy)


def test_rescale(backend, oracle):
    o = oracle
    rng = np.random.default_rng(17)
    # (N = 4096: the single static pass with the fused store; N >= 8192: the 4-launch form, column passes of 4 and of 5 stages)
    for logN, sizeQl, B in [(6, 3, 2), (9, 5, 3), (12, 4, 2), (13, 3, 1), (14, 9, 2), (17, 3, 1)]:
        N = 1 << logN
        q, psiQ, _, _ = ckks_like_params(o, logN, sizeQl + 1, 2)
        ctx = fh.Context(backend, logN, q, psiQ)
        octx = o.orc_ctx_create(N, len(q), q, psiQ)
        x = libs.rand_tower(rng, q[:sizeQl], N, B)
        want = np.empty((B, sizeQl - 1, N), np.uint64)
        for bb in range(B):
            o.orc_drop_last_element_and_scale(octx, x[bb], sizeQl, want[bb])
        got = fh.rescale(ctx, ctx.tower(x)).to_host()
        assert np.array_equal(got, want), f"DropLastElementAndScale mismatch logN={logN}"
        o.orc_ctx_destroy(octx)
        ctx.close()


@pytest.mark.parametrize("logN", [8, 12, 13])
def test_rescale_limbs_with_the_callers_tables(backend, oracle, logN):
    """fhe_rescale_limbs: a tower over scattered limbs of the context, first with the reference's tables (the fused form on rings of
    static passes), then with a perturbed scale table (A != -B: the member-by-member form computes exactly what the tables say)"""
    o = oracle
    rng = np.random.default_rng(23)
    N = 1 << logN
    q, psiQ, _, _ = ckks_like_params(o, logN, 7, 2)
    ctx = fh.Context(backend, logN, q, psiQ)
    limbs = [5, 0, 3, 6, 2]  # the tower's limbs in its own order: q_6 ... dropped limb is context limb 2
    sizeQl, B = len(limbs), 2
    qs = [int(q[i]) for i in limbs]
    sub = o.orc_ctx_create(N, sizeQl, np.array(qs, np.uint64), np.array([int(psiQ[i]) for i in limbs], np.uint64))
    x = libs.rand_tower(rng, np.array(qs, np.uint64), N, B)
    want = np.empty((B, sizeQl - 1, N), np.uint64)
    for bb in range(B):
        o.orc_drop_last_element_and_scale(sub, x[bb], sizeQl, want[bb])
    inv = [pow(qs[-1] % qi, -1, qi) for qi in qs[:-1]]
    neg = [(qi - v) % qi for v, qi in zip(inv, qs[:-1])]
    xt = ctx.tower(x, limbs)
    assert np.array_equal(fh.rescale_limbs(ctx, xt, neg, inv).to_host(), want)
    # perturbed table: out_i = x_i * inv_i + NTT(SwitchModulus(last) * a_i); with a_i = 2 * neg_i this is want + NTT(sm * neg) =
    # 2 * want - x_i * inv_i
    a2 = [(2 * v) % qi for v, qi in zip(neg, qs[:-1])]
    got = fh.rescale_limbs(ctx, xt, a2, inv).to_host()
    for i, qi in enumerate(qs[:-1]):
        w = (2 * want[:, i].astype(object) - x[:, i].astype(object) * inv[i]) % qi
        assert np.array_equal(got[:, i].astype(object), w), f"limb {i}"
    o.orc_ctx_destroy(sub)
    ctx.close()


@pytest.mark.parametrize("logN", [8, 12, 13, 17])
def test_pair_entries_two_separately_allocated_towers_in_one_launch(backend, oracle, logN):
    """fhe_add_pair / fhe_sub_pair / fhe_mul_const_pair / fhe_rescale_limbs_pair: the two elements of a ciphertext, each a buffer of
    its own (allocated in both address orders), equal what the single-tower entries give element by element"""
    if logN == 17 and "emulator" in backend.version():
        pytest.skip("the 5-stage column pass with separately allocated towers is covered on the GPU; N = 2^13 covers the emulator")
    o = oracle
    rng = np.random.default_rng(29)
    N, L = 1 << logN, 4
    q, psiQ, _, _ = ckks_like_params(o, logN, L, 2)
    ctx = fh.Context(backend, logN, q, psiQ)
    qs = [int(v) for v in q[:L]]
    mk = lambda: libs.rand_tower(rng, q[:L], N, 1)
    ha0, ha1, hb0, hb1 = mk(), mk(), mk(), mk()
    # (device addresses: a1 before a0 but b0 before b1, so the distances between the elements differ in sign across operands)
    a1, a0, b0, b1 = ctx.tower(ha1), ctx.tower(ha0), ctx.tower(hb0), ctx.tower(hb1)
    o0, o1 = fh.elem_pair(ctx, "add", a0, a1, b0, b1)
    for got, x, y in ((o0, ha0, hb0), (o1, ha1, hb1)):
        assert all(np.array_equal(got.to_host()[0, i], (x[0, i] + y[0, i]) % np.uint64(qs[i])) for i in range(L))
    o0, o1 = fh.elem_pair(ctx, "sub", a0, a1, b0, b1)
    for got, x, y in ((o0, ha0, hb0), (o1, ha1, hb1)):
        assert all(np.array_equal(got.to_host()[0, i], (x[0, i] + (np.uint64(qs[i]) - y[0, i])) % np.uint64(qs[i])) for i in range(L))
    k = [int(rng.integers(1, qi)) for qi in qs]
    o0, o1 = fh.elem_pair(ctx, "mul_const", a0, a1, consts=k)
    for got, x in ((o0, ha0), (o1, ha1)):
        assert all(np.array_equal(got.to_host()[0, i].astype(object), (x[0, i].astype(object) * k[i]) % qs[i]) for i in range(L))
    # rescale of both elements against the single-tower entry (itself checked against the oracle above)
    inv = [pow(qs[-1] % qi, -1, qi) for qi in qs[:-1]]
    neg = [(qi - v) % qi for v, qi in zip(inv, qs[:-1])]
    r0, r1 = fh.rescale_limbs_pair(ctx, a0, a1, neg, inv)
    assert np.array_equal(r0.to_host(), fh.rescale_limbs(ctx, a0, neg, inv).to_host())
    assert np.array_equal(r1.to_host(), fh.rescale_limbs(ctx, a1, neg, inv).to_host())
    octx = o.orc_ctx_create(N, L, q[:L], psiQ[:L])
    want = np.empty((L - 1, N), np.uint64)
    o.orc_drop_last_element_and_scale(octx, ha1[0], L, want)
    assert np.array_equal(r1.to_host()[0], want)
    o.orc_ctx_destroy(octx)
    # in place: a += b on both elements
    fh.elem_pair(ctx, "add", a0, a1, b0, b1, in_place=True)
    assert all(np.array_equal(a1.to_host()[0, i], (ha1[0, i] + hb1[0, i]) % np.uint64(qs[i])) for i in range(L))
    ctx.close()


@pytest.mark.parametrize("logN,L,B,terms", [(6, 3, 2, 1), (10, 4, 1, 5), (12, 3, 2, 16), (12, 2, 1, 37)])
def test_lincomb_weighted_sum_in_one_launch(backend, oracle, logN, L, B, terms):
    """fhe_lincomb (pke's internalEvalLinearWSumMutable, ckksrns-advancedshe.cpp:97-136: EvalMultInPlace(ct_i, c_i) + EvalAddInPlaceNoCheck
    per term): sum_i c_i (.) x_i per limb against the oracle's per-limb products and sums (exact residues), 1 ... 37 terms (> 16: several
    launches, the later ones accumulating), the accumulate form, and the limb-subset form"""
    o = oracle
    rng = np.random.default_rng(41)
    N = 1 << logN
    q, psi = params(o, logN, L)
    ctx = fh.Context(backend, logN, q, psi)
    qs = [int(v) for v in q]
    xs = [libs.rand_tower(rng, q, N, B) for _ in range(terms)]
    ks = [[int(rng.integers(0, qi)) if (i + r) % 7 else qi - 1 for r, qi in enumerate(qs)] for i in range(terms)]
    want = np.zeros((B, L, N), dtype=object)
    for i in range(terms):
        for r in range(L):
            want[:, r] = (want[:, r] + xs[i][:, r].astype(object) * ks[i][r]) % qs[r]
    # (the oracle's modular product agrees with the python integers on a sample: same arithmetic as every other parity test)
    probe = np.empty(N, np.uint64)
    o.orc_vec_mul(probe, xs[0][0, 0], np.full(N, ks[0][0], np.uint64), N, q[0])
    assert np.array_equal(probe.astype(object), xs[0][0, 0].astype(object) * ks[0][0] % qs[0])
    tw = [ctx.tower(x) for x in xs]
    got = fh.lincomb(ctx, tw, ks)
    assert np.array_equal(got.to_host().astype(object), want)
    # accumulate: acc + sum
    acc_h = libs.rand_tower(rng, q, N, B)
    acc = fh.lincomb(ctx, tw, ks, accumulate_into=ctx.tower(acc_h))
    want2 = np.empty_like(want)
    for r in range(L):
        want2[:, r] = (want[:, r] + acc_h[:, r].astype(object)) % qs[r]
    assert np.array_equal(acc.to_host().astype(object), want2)
    # in place on the LAST term (with 37 terms it sits in the third launch's chunk: fhe_lincomb sums the terms that alias `out` first)
    import ctypes as C
    ptrs = (C.c_void_p * terms)(*[t.ptr.value for t in tw])
    k = np.ascontiguousarray(ks, dtype=np.uint64)
    ctx.lib.check(ctx.lib.L.fhe_lincomb(ctx.h, tw[-1].ptr, ptrs, k.ctypes.data_as(C.POINTER(C.c_uint64)), terms, tw[-1]._li(), L, B, 0, None))
    ctx.sync(None)
    assert np.array_equal(tw[-1].to_host().astype(object), want)
    ctx.close()


@pytest.mark.parametrize("logN,sizeQl,t,B,ev", [(4, 3, 65537, 2, 1), (10, 4, 786433, 2, 1), (12, 3, 65537, 1, 0), (13, 3, 2, 1, 1)])
def test_mod_reduce(backend, oracle, logN, sizeQl, t, B, ev):
    """fhe_mod_reduce (DCRTPoly::ModReduce, BGV modulus switching) vs the oracle"""
    o = oracle
    rng = np.random.default_rng(31)
    N = 1 << logN
    q, psi = params(o, logN, sizeQl)
    ctx = fh.Context(backend, logN, q, psi)
    octx = o.orc_ctx_create(N, sizeQl, q, psi)
    x = libs.rand_tower(rng, q, N, B)
    want = np.zeros((B, sizeQl - 1, N), np.uint64)
    for b in range(B):
        o.orc_mod_reduce(octx, x[b], sizeQl, t, ev, want[b])
    got = fh.mod_reduce(ctx, ctx.tower(x, fmt=fh.EVALUATION if ev else fh.COEFFICIENT), t)
    assert np.array_equal(got.to_host(), want)
    o.orc_ctx_destroy(octx)
    ctx.close()


@pytest.mark.parametrize("logN,sizeQ,dnum,sizeQl,B", [(10, 4, 2, 3, 2), (12, 6, 3, 6, 1)])
def test_hoisted_rotations(backend, oracle, logN, sizeQ, dnum, sizeQl, B):
    """EvalAutomorphism and EvalFastRotation (one ModUp, several rotation keys) against the oracle"""
    o = oracle
    rng = np.random.default_rng(18)
    N = 1 << logN
    q, psiQ, p, psiP = ckks_like_params(o, logN, sizeQ, dnum)
    sizeP = len(p)
    hy = o.orc_hybrid_create(N, sizeQ, q, psiQ, sizeP, p, psiP, dnum)
    allq = np.concatenate([q, p])
    ctx = fh.Context(backend, logN, allq, np.concatenate([psiQ, psiP]))
    plan = fh.KeySwitchPlan(ctx, sizeQ, sizeP, dnum)
    ql = q[:sizeQl]
    c0, c1 = libs.rand_tower(rng, ql, N, B), libs.rand_tower(rng, ql, N, B)
    t0, t1 = ctx.tower(c0), ctx.tower(c1)
    ks = [o.orc_find_automorphism_index_2n_complex(i, 2 * N) for i in (1, -2, 5)] + [2 * N - 1]  # + conjugation
    keys = [(libs.rand_tower(rng, allq, N, dnum), libs.rand_tower(rng, allq, N, dnum)) for _ in ks]
    handles = [plan.make_key(kb, ka) for kb, ka in keys]
    want = []
    for k, (kb, ka) in zip(ks, keys):
        w0, w1 = np.empty_like(c0), np.empty_like(c0)
        for b in range(B):
            o.orc_eval_automorphism(hy, c0[b], c1[b], sizeQl, k, kb, ka, w0[b], w1[b])
        want.append((w0, w1))
    g0, g1 = plan.EvalAutomorphism(handles[0], t0, t1, ks[0])
    assert np.array_equal(g0.to_host(), want[0][0]) and np.array_equal(g1.to_host(), want[0][1])
    plan.EvalFastRotationPrecompute(t1)  # hoisted: digits computed once ...
    for k, hnd, (w0, w1) in zip(ks, handles, want):  # ... reused for every rotation key
        g0, g1 = plan.EvalFastRotation(hnd, t0, t1, k)
        assert np.array_equal(g0.to_host(), w0) and np.array_equal(g1.to_host(), w1), f"rotation k={k}"
    # double hoisting: rotations that stay in the extended basis, summed there, one KeySwitchDown at the end
    acc0 = acc1 = None
    wacc0 = np.zeros((B, sizeQl + sizeP, N), np.uint64)
    wacc1 = np.zeros_like(wacc0)
    extq = np.concatenate([ql, p])
    for j, (k, hnd, (kb, ka)) in enumerate(zip(ks, handles, keys)):
        add_first = j % 2 == 0
        e0, e1 = plan.EvalFastRotationExt(hnd, t0, t1, k, add_first)
        w0, w1 = np.empty_like(wacc0), np.empty_like(wacc0)
        for b in range(B):
            o.orc_eval_fast_rotation_ext(hy, c0[b], c1[b], sizeQl, k, 1 if add_first else 0, kb, ka, w0[b], w1[b])
        assert np.array_equal(e0.to_host(), w0) and np.array_equal(e1.to_host(), w1), f"EvalFastRotationExt k={k}"
        for i, m in enumerate(extq):  # EvalAddExt: limb-wise modular add in the extended basis
            wacc0[:, i] = (wacc0[:, i] + w0[:, i]) % m
            wacc1[:, i] = (wacc1[:, i] + w1[:, i]) % m
        if acc0 is None:
            acc0, acc1 = e0, e1
        else:
            idx = plan.ext_limbs(sizeQl)
            for acc, e in ((acc0, e0), (acc1, e1)):
                backend.check(backend.L.fhe_add(ctx.h, acc.ptr, acc.ptr, e.ptr, idx.ctypes.data_as(fh.u32p), len(idx), B, None))
    assert np.array_equal(acc0.to_host(), wacc0) and np.array_equal(acc1.to_host(), wacc1), "EvalAddExt"
    d0, d1 = plan.KeySwitchDown(acc0, acc1, sizeQl)
    wd0, wd1 = np.empty((B, sizeQl, N), np.uint64), np.empty((B, sizeQl, N), np.uint64)
    for b in range(B):
        o.orc_hybrid_approx_mod_down(hy, wacc0[b], sizeQl, wd0[b])
        o.orc_hybrid_approx_mod_down(hy, wacc1[b], sizeQl, wd1[b])
    assert np.array_equal(d0.to_host(), wd0) and np.array_equal(d1.to_host(), wd1), "KeySwitchDown"
    # KeySwitchExt: c * [P]_{q_i} on the Q_l limbs, zeros on the P limbs
    ext = plan.KeySwitchExt(t0).to_host()
    P = 1
    for v in p:
        P *= int(v)
    for i, m in enumerate(ql):
        assert np.array_equal(ext[:, i], (c0[:, i].astype(object) * (P % int(m)) % int(m)).astype(np.uint64)), "KeySwitchExt"
    assert not ext[:, sizeQl:].any()
    for hnd in handles:
        backend.L.fhe_ks_key_destroy(hnd)
    plan.close()
    ctx.close()
    o.orc_hybrid_destroy(hy)


@pytest.mark.gpu
def test_graph_capture_replays_eval_mult(oracle):
    """a captured EvalMult + key switch (fhe_graph_begin/end) replays bit-exactly, also on fresh inputs"""
    lib = fh.Lib()
    if lib.device_count() < 1:
        pytest.skip("needs a HIP device")
    o = oracle
    rng = np.random.default_rng(55)
    logN, sizeQ, dnum, B = 12, 5, 2, 2
    N = 1 << logN
    q, psiQ, p, psiP = ckks_like_params(o, logN, sizeQ, dnum)
    sizeP = len(p)
    hy = o.orc_hybrid_create(N, sizeQ, q, psiQ, sizeP, p, psiP, dnum)
    allq = np.concatenate([q, p])
    ctx = fh.Context(lib, logN, allq, np.concatenate([psiQ, psiP]))
    plan = fh.KeySwitchPlan(ctx, sizeQ, sizeP, dnum)
    keyB, keyA = libs.rand_tower(rng, allq, N, dnum), libs.rand_tower(rng, allq, N, dnum)
    plan.upload_key(keyB, keyA)
    ops = [libs.rand_tower(rng, q, N, B) for _ in range(4)]
    T = [ctx.tower(x) for x in ops]
    c0, c1 = T[0].like(), T[0].like()
    ws, wsb = plan.workspace(sizeQ, B)
    st = C.c_void_p()
    lib.check(lib.L.fhe_stream_create(ctx.h, C.byref(st)))

    def call():
        lib.check(lib.L.fhe_ckks_eval_mult(plan.h, plan.key, T[0].ptr, T[1].ptr, T[2].ptr, T[3].ptr, sizeQ, B, c0.ptr, c1.ptr,
                                           ws, wsb, st))
    call()  # builds the level's tables
    lib.check(lib.L.fhe_stream_sync(ctx.h, st))
    g = C.c_void_p()
    lib.check(lib.L.fhe_graph_begin(ctx.h, st))
    call()
    lib.check(lib.L.fhe_graph_end(ctx.h, st, C.byref(g)))
    for trial in range(2):
        if trial:  # new operands in the same buffers: the graph must pick them up
            for t, x in zip(T, ops):
                x[:] = libs.rand_tower(rng, q, N, B)
                lib.check(lib.L.fhe_memcpy_h2d(ctx.h, t.ptr, x.ctypes.data_as(C.c_void_p), x.nbytes, None))
            ctx.sync()
        lib.check(lib.L.fhe_graph_launch(ctx.h, g, st))
        lib.check(lib.L.fhe_stream_sync(ctx.h, st))
        for b in range(B):
            w0, w1 = np.empty((sizeQ, N), np.uint64), np.empty((sizeQ, N), np.uint64)
            o.orc_ckks_eval_mult_relin(hy, ops[0][b], ops[1][b], ops[2][b], ops[3][b], sizeQ, keyB, keyA, w0, w1)
            assert np.array_equal(c0.to_host()[b], w0) and np.array_equal(c1.to_host()[b], w1), f"graph replay {trial}"
    lib.L.fhe_graph_destroy(g)
    lib.check(lib.L.fhe_stream_destroy(ctx.h, st))
    plan.close()
    ctx.close()
    o.orc_hybrid_destroy(hy)


@pytest.mark.parametrize("variant", ["1"])
def test_conversion_kernel_variants_on_emulator(backend, variant):
    """the non-default column-sum variant of the basis-conversion / BEHZ kernels (FHE_CONV_SUM8 is read once per process:
    1 = carry-counted 64-bit columns; the default 2 = 30-bit split without carries runs in every other test) against the
    oracle, through the regular conversion / key-switch / BFV parity tests in a child process"""
    import subprocess
    import sys
    if not is_emu(backend):
        pytest.skip("emulator variant")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FHE_CONV_SUM8=variant)
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_parity.py"), "-q", "-x", "-m", "not gpu",
                          "-k", "test_approx_and_exact_switch_crt_basis or (test_hybrid_keyswitch_and_eval_mult and 12-6)"],
                         env=env, capture_output=True, text=True, timeout=1200, cwd=root)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    if variant == "1":  # the same knob switches the BEHZ dot products of the BFV multiplication
        out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_parity_bfv.py"), "-q", "-x", "-m",
                              "not gpu", "-k", "behz or eval_mult"], env=env, capture_output=True, text=True, timeout=1200, cwd=root)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]


@pytest.mark.gpu
def test_conversion_kernel_variants_on_gpu():
    """the same forced variant (carry-counted 64-bit columns, FHE_CONV_SUM8=1: 16- and 32-source-limb instances, the 1-row kernel) on
    the MI355X: until round 6 the GPU ran these instances only where a shape happened to select them"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FHE_CONV_SUM8="1")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_parity.py"), "-q", "-x", "-m", "gpu",
                          "-k", "hip and (test_approx_and_exact_switch_crt_basis or test_approx_mod_up or (test_hybrid_keyswitch_and_eval_mult and 12-6))"],
                         env=env, capture_output=True, text=True, timeout=1200, cwd=root)
    assert out.returncode == 0 and " passed" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_parity_bfv.py"), "-q", "-x", "-m", "gpu",
                          "-k", "hip and (behz or eval_mult)"], env=env, capture_output=True, text=True, timeout=1200, cwd=root)
    assert out.returncode == 0 and " passed" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_whole_tower_checksums(backend, oracle):
    """fhe_checksum: {sum, position-weighted sum} mod 2^64 of every limb-row of a resident batch in one read (the all-towers parity check of bench.py
    and of the full-shape tests)"""
    rng = np.random.default_rng(23)
    for logN, L, B in [(4, 3, 2), (9, 2, 3), (12, 2, 2), (13, 3, 1)]:
        q, psi = params(oracle, logN, L)
        ctx = fh.Context(backend, logN, q, psi)
        x = libs.rand_tower(rng, q, 1 << logN, B)
        t = ctx.tower(x)
        got = ctx.checksum(t)
        rows = x.reshape(B * L, -1)
        w = 2 * np.arange(rows.shape[1], dtype=np.uint64) + np.uint64(1)
        want = np.stack([rows.sum(axis=1, dtype=np.uint64), (rows * w).sum(axis=1, dtype=np.uint64)], axis=1)
        assert np.array_equal(got, want)
        # the second word depends on the order of the words: two words of a row swapped change it, the plain sum stays
        y = x.copy()
        y[0, 0, [1, 2]] = y[0, 0, [2, 1]]
        if y[0, 0, 1] != y[0, 0, 2]:
            got2 = ctx.checksum(ctx.tower(y))
            assert got2[0, 0] == got[0, 0] and got2[0, 1] != got[0, 1]
        ctx.close()


@pytest.mark.parametrize("logN,L,B", [(5, 2, 2), (12, 2, 2), (13, 3, 2), (14, 2, 1), (15, 1, 2), (16, 2, 1), (17, 1, 1)])
def test_poly_mul(backend, oracle, logN, L, B):
    """fhe_poly_mul (c = a * b in Z_q[x]/(x^N + 1), COEFFICIENT in and out) against the oracle's INTT(NTT(a) o NTT(b)); two-pass
    rings take the fused row kernels (forward row pass of b, Hadamard product and inverse row pass in one kernel)"""
    o = oracle
    if is_emu(backend) and logN > 14 and not os.environ.get("FHE_TEST_BIG_EMU"):
        pytest.skip("emulator: keep the CPU suite short (FHE_TEST_BIG_EMU=1 runs these too)")
    rng = np.random.default_rng(77)
    N = 1 << logN
    q, psi = params(o, logN, L)
    ctx = fh.Context(backend, logN, q, psi)
    octx = o.orc_ctx_create(N, L, q, psi)
    a, b = libs.rand_tower(rng, q, N, B), libs.rand_tower(rng, q, N, B)
    a[0, :, :2] = 0
    b[0, :, 1] = q - np.uint64(1)
    wa, wb = a.copy(), b.copy()
    o.orc_ntt_fwd_tower(octx, wa, None, L, B, 0)
    o.orc_ntt_fwd_tower(octx, wb, None, L, B, 0)
    want = np.empty_like(a)
    for bb in range(B):
        for l in range(L):
            o.orc_vec_mul(want[bb, l], wa[bb, l], wb[bb, l], N, q[l])
    o.orc_ntt_inv_tower(octx, want, None, L, B, 0)
    ta, tb = ctx.tower(a, fmt=fh.COEFFICIENT), ctx.tower(b, fmt=fh.COEFFICIENT)
    got = ta.PolyMul(tb)
    assert np.array_equal(got.to_host(), want), f"polynomial product logN={logN}"
    assert np.array_equal(ta.to_host(), a) and np.array_equal(tb.to_host(), b), "operands must not change"
    o.orc_ctx_destroy(octx)
    ctx.close()
