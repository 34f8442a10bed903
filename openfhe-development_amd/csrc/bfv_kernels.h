// bfv_kernels.h — BFV-side basis conversions: ScaleAndRound family and the BEHZ trio.
//
// Replaces (src/core/include/lattice/hal/default/dcrtpoly-impl.h):
//   ScaleAndRound (DCRTPoly -> DCRTPoly)      :1513-1628     scale_round_kernel<true>
//   ApproxScaleAndRound                       :1470-1510     scale_round_kernel<false>
//   ScaleAndRoundPOverQ                       :1674-1689     p_over_q_kernel
//   FastBaseConvqToBskMontgomery (core)       :1731-1774     behz_q_to_bsk_kernel
//   FastRNSFloorq                             :1791-1840     behz_floorq_kernel
//   FastBaseConvSK                            :1845-1929     behz_conv_sk_kernel
// One coefficient per lane; limbs of that coefficient are 8-byte loads that are contiguous across lanes
// (coalesced per limb).  All tables are wave-uniform (scalar cache).  Integer results are exact residues; the one
// floating-point quantity (ScaleAndRound's nu) is accumulated in the reference's order with contraction disabled.
#ifndef FHE_BFV_KERNELS_H
#define FHE_BFV_KERNELS_H
#include "modarith.h"
#include "launch.h"
#include "ntt_kernels.h"

namespace fhe {

constexpr int kMaxBfvLimbs = 16;

// tower view: limb r of tower b lives at base + ((b*stride + first + r) << logN)
struct TowerView {
    uint64_t* p;
    uint32_t stride, first;
};
FHE_HD uint64_t* tv_at(const TowerView v, uint32_t b, uint32_t r, uint32_t logN, uint32_t ri) {
    return v.p + ((((uint64_t)b * v.stride + v.first + r)) << logN) + ri;
}

// ---- ScaleAndRound / ApproxScaleAndRound -------------------------------------------------------------
struct ScaleRoundArgs {
    TowerView in;        // input-basis limbs (sizeI rows)
    TowerView own;       // the output-basis limbs inside the same tower (sizeO rows)
    TowerView out;       // result (sizeO rows)
    const uint64_t* tab; // [sizeO][sizeI+1]
    const double* frac;  // [sizeI]   (exact variant)
    const uint64_t* o;   // [sizeO] output moduli
    const uint64_t* mu;  // [sizeO][2]
    uint32_t logN, batch, sizeI, sizeO;
};
template <bool EXACT>
FHE_GLOBAL void FHE_LAUNCH_BOUNDS(kThreads) scale_round_kernel(const ScaleRoundArgs g) {
    const uint64_t gid = (uint64_t)FHE_BID * kThreads + FHE_TID;
    if (gid >= ((uint64_t)g.batch << g.logN))
        return;
    const uint32_t b = (uint32_t)(gid >> g.logN), ri = (uint32_t)gid & ((1u << g.logN) - 1u);
    double nu = 0.5;
    if (EXACT) {
        for (uint32_t i = 0; i < g.sizeI; ++i)
            nu += FHE_ULOADF64(g.frac, i) * (double)(*tv_at(g.in, b, i, g.logN, ri));  // :1543-1548, i ascending
    }
    // isConvertableToNativeInt(nu): |nu| <= (double)(2^64-1) == 2^64   (utils/utilities.h:122-126)
    const bool small = EXACT && (nu <= 18446744073709551616.0);
    uint64_t alo = 0, ahi = 0;
    if (EXACT) {
        if (small)
            alo = (uint64_t)nu;
        else {  // static_cast<unsigned __int128>(nu): nu >= 2^64 is an integer-valued double
            ahi = (uint64_t)(nu * (1.0 / 18446744073709551616.0));
            alo = (uint64_t)(nu - (double)ahi * 18446744073709551616.0);
        }
    }
    for (uint32_t j = 0; j < g.sizeO; ++j) {
        const uint64_t* tj = g.tab + (uint64_t)j * (g.sizeI + 1);
        u128w acc{0, 0};
        for (uint32_t i = 0; i < g.sizeI; ++i)
            acc128(acc, *tv_at(g.in, b, i, g.logN, ri), FHE_ULOAD64(tj, i));
        acc128(acc, *tv_at(g.own, b, j, g.logN, ri), FHE_ULOAD64(tj, g.sizeI));
        const uint64_t oj = FHE_ULOAD64(g.o, j), mlo = FHE_ULOAD64(g.mu, 2 * j), mhi = FHE_ULOAD64(g.mu, 2 * j + 1);
        uint64_t v = barrett128(acc, oj, mlo, mhi);
        if (EXACT) {
            uint64_t a;
            if (small)
                a = alo >= oj ? alo % oj : alo;  // alpha.Mod(oj, mu) — exact remainder (:1566-1568)
            else
                a = barrett128(u128w{alo, ahi}, oj, mlo, mhi);  // :1586-1588
            v = add_mod(v, a, oj);
        }
        *tv_at(g.out, b, j, g.logN, ri) = v;
    }
}

// ---- ScaleAndRound -> NativePoly mod t (decryption, dcrtpoly-impl.h:1190-1467) and the BEHZ overload (:1631-1671) ----
struct ScaleRoundNativeArgs {
    TowerView in;            // sizeQ rows, COEFFICIENT
    uint64_t* out;           // [batch][N] residues mod t
    const uint64_t* q;       // [sizeQ] moduli of the rows (BEHZ overload)
    const TwPair* tabModt;   // [sizeQ] (value, Shoup precon mod t)   | BEHZ: tgammaQHatModq (precon mod q_i)
    const TwPair* tabBModt;  // [sizeQ] the "B" table                  | BEHZ: negInvqModtgamma (precon mod t*gamma)
    const double* frac;      // [sizeQ]
    const double* bfrac;     // [sizeQ]
    uint64_t t;              // BEHZ: t * gamma
    uint32_t logN, batch, sizeQ, qMSBHf;
    uint32_t pow2, split, nomod;  // the reference's branch conditions, evaluated on the host (:1198-1218 ff)
};
// the reference's ModMulFastConst (ubintnat.h:1464-1469): operand may exceed the modulus
FHE_HD uint64_t mod_mul_fast_const(uint64_t a, uint64_t b, uint64_t m, uint64_t bInv) {
    const uint64_t qq = mulhi64(a, bInv) + 1;
    const int64_t y   = (int64_t)(a * b - qq * m);
    return y >= 0 ? (uint64_t)y : (uint64_t)y + m;
}
FHE_GLOBAL void FHE_LAUNCH_BOUNDS(kThreads) scale_round_native_kernel(const ScaleRoundNativeArgs g) {
    const uint64_t gid = (uint64_t)FHE_BID * kThreads + FHE_TID;
    if (gid >= ((uint64_t)g.batch << g.logN))
        return;
    const uint32_t b = (uint32_t)(gid >> g.logN), ri = (uint32_t)gid & ((1u << g.logN) - 1u);
    const uint64_t t = g.t;
    double floatSum  = g.pow2 ? 0.5 : 0.0;
    uint64_t intSum  = 0;
    for (uint32_t i = 0; i < g.sizeQ; ++i) {
        const uint64_t v = *tv_at(g.in, b, i, g.logN, ri);
        const TwPair ta  = g.tabModt[i];
        if (!g.split) {
            floatSum += (double)v * FHE_ULOADF64(g.frac, i);
            intSum += g.nomod ? v * ta.w : mod_mul_fast_const(v, ta.w, t, ta.wp);
        }
        else {
            const TwPair tbm = g.tabBModt[i];
            const uint64_t hi = v >> g.qMSBHf, lo = v - (hi << g.qMSBHf);
            floatSum += (double)lo * FHE_ULOADF64(g.frac, i);
            floatSum += (double)hi * FHE_ULOADF64(g.bfrac, i);
            intSum += g.nomod ? lo * ta.w : mod_mul_fast_const(lo, ta.w, t, ta.wp);
            intSum += g.nomod ? hi * tbm.w : mod_mul_fast_const(hi, tbm.w, t, tbm.wp);
        }
    }
    uint64_t r;
    if (g.pow2) {
        intSum += (uint64_t)floatSum;
        r = intSum & (t - 1);
    }
    else {
        const double td = (double)t, tInv = 1. / td;
        floatSum += (double)intSum;
        floatSum -= td * (double)(uint64_t)(floatSum * tInv);
        r = (uint64_t)(floatSum + 0.5);
    }
    g.out[((uint64_t)b << g.logN) + ri] = r;
}
FHE_GLOBAL void FHE_LAUNCH_BOUNDS(kThreads) scale_round_behz_decrypt_kernel(const ScaleRoundNativeArgs g) {
    const uint64_t gid = (uint64_t)FHE_BID * kThreads + FHE_TID;
    if (gid >= ((uint64_t)g.batch << g.logN))
        return;
    const uint32_t b = (uint32_t)(gid >> g.logN), ri = (uint32_t)gid & ((1u << g.logN) - 1u);
    const uint64_t tgamma = g.t, gammaMinus1 = (1u << 26) - 1;
    uint64_t s = 0;
    for (uint32_t i = 0; i < g.sizeQ; ++i) {
        const TwPair ta = g.tabModt[i], tb2 = g.tabBModt[i];
        const uint64_t a = mod_mul_fast_const(*tv_at(g.in, b, i, g.logN, ri), ta.w, g.q[i], ta.wp);
        s                = add_mod(s, mod_mul_fast_const(a, tb2.w, tgamma, tb2.wp), tgamma);
    }
    s += s & gammaMinus1;
    g.out[((uint64_t)b << g.logN) + ri] = s >> 26;
}

// ---- ScaleAndRoundPOverQ ----------------------------------------------------------------------------
constexpr int kMaxPOverQ = 64;
struct POverQArgs {
    TowerView x;    // [sizeQ+1] rows, the last one modulo pLast
    TowerView out;  // [sizeQ]
    uint64_t q[kMaxPOverQ];    // moduli, by value in the kernel arguments (no device staging: the call stays asynchronous)
    TwPair pInv[kMaxPOverQ];   // [p^-1]_{q_i}
    uint64_t pLast;
    uint32_t logN, batch, sizeQ;
};
FHE_GLOBAL void FHE_LAUNCH_BOUNDS(kThreads) p_over_q_kernel(const POverQArgs g) {
    const uint64_t gid = (uint64_t)FHE_BID * kThreads + FHE_TID;
    if (gid >= ((uint64_t)g.batch << g.logN))
        return;
    const uint32_t b = (uint32_t)(gid >> g.logN), ri = (uint32_t)gid & ((1u << g.logN) - 1u);
    const uint64_t last = *tv_at(g.x, b, g.sizeQ, g.logN, ri), halfP = g.pLast >> 1;
    for (uint32_t i = 0; i < g.sizeQ; ++i) {
        const uint64_t qn = g.q[i];
        uint64_t v        = last;  // SwitchModulus(pLast -> q_i), mubintvecnat.cpp:109-122
        if (qn > g.pLast)
            v += (v > halfP) ? (qn - g.pLast) : 0;
        else {
            uint64_t bv = (v > halfP) ? (g.pLast - qn) : 0, av = v;
            if (av >= qn)
                av %= qn;
            if (bv >= qn)
                bv %= qn;
            v = (av < bv) ? av + qn - bv : av - bv;
        }
        const TwPair c = g.pInv[i];
        *tv_at(g.out, b, i, g.logN, ri) = mul_shoup(sub_mod(*tv_at(g.x, b, i, g.logN, ri), v, qn), c.w, c.wp, qn);
    }
}

// ---- BEHZ --------------------------------------------------------------------------------------------
// Every vector table is padded to kMaxBfvLimbs entries and every matrix is stored [target][kMaxBfvLimbs] (one target's
// row contiguous), so that the kernels read them unconditionally through the scalar cache (FHE_ULOAD64) and issue the
// residue loads of a coefficient back to back (clamped row index) before any arithmetic.
struct BehzTables {           // bfvrns-cryptoparameters.cpp:673-850; all device resident
    const uint64_t *q, *bsk;                 // [16] (padding = 1)
    const uint64_t *muQ, *muBsk;             // [16][2]
    const TwPair* mtQHatInv;                 // [16]   [mtilde (Q/q_i)^-1]_{q_i}
    const uint64_t* QHatModbsk;              // [numBsk][16]   [Q/q_i]_{bsk_j}
    const uint64_t* QHatModmt;               // [16]
    const TwPair *QModbsk, *mtInvModbsk;     // [16]
    const TwPair* tQHatInv;                  // [16]
    const uint64_t* qInvModbsk;              // [numBsk][16]   [q_i^-1]_{bsk_j}
    const TwPair* tQInvModbsk;               // [16]
    const TwPair* BHatInv;                   // [16]
    const uint64_t* BHatModmsk;              // [16]
    const uint64_t* BHatModq;                // [numQ][16]     [B/b_i]_{q_j}
    const TwPair* BModq;                     // [16]
    TwPair BInvModmsk;
    uint64_t negQInvModmt;
    uint64_t mskMu;                          // ComputeMu(msk)
    uint32_t mskMsb;
    uint32_t numQ, numBsk;
    uint32_t W;                              // table stride: 16 (the register-resident kernels) or kBehzWideLimbs (the wide ones below)
};
struct BehzArgs {
    TowerView inQ, inBsk;   // source limbs
    TowerView outQ, outBsk; // destination limbs
    BehzTables tb;
    uint32_t logN, batch;
};
FHE_HD TwPair uload_pair(const TwPair* p, uint32_t i) {
    const uint64_t* w = reinterpret_cast<const uint64_t*>(p);
    return TwPair{FHE_ULOAD64(w, 2 * i), FHE_ULOAD64(w, 2 * i + 1)};
}
// residues of one coefficient: rows 0..n-1 of the view, all loads issued before use (row index clamped)
FHE_HD void load_rows(uint64_t (&x)[kMaxBfvLimbs], const TowerView v, uint32_t b, uint32_t n, uint32_t logN, uint32_t ri) {
#pragma unroll
    for (int i = 0; i < kMaxBfvLimbs; ++i)
        x[i] = *tv_at(v, b, (uint32_t)i < n ? (uint32_t)i : n - 1u, logN, ri);
}
// sum_i y_i * row[i] mod m for the first n entries of a [.][kMaxBfvLimbs] table row (128-bit sum, then Barrett)
template <bool SPLIT30 = false>
FHE_HD uint64_t dot_row_mod(const uint64_t (&y)[kMaxBfvLimbs], const uint64_t* row, uint32_t n, uint64_t m, uint64_t mulo,
                            uint64_t muhi) {
    uint64_t h[kMaxBfvLimbs];
#pragma unroll
    for (int i = 0; i < kMaxBfvLimbs; ++i)
        h[i] = FHE_ULOAD64(row, i);
    // chunks of <= 8 products (y_i < 2^60, table entries < m) with one 64-bit Barrett reduction each (sum8, modarith.h)
    const uint32_t k = 64u - (uint32_t)__builtin_clzll(m);
    uint64_t v       = 0;
#pragma unroll
    for (int c0 = 0; c0 < kMaxBfvLimbs; c0 += 8) {
        if (c0 && c0 >= (int)n)
            break;
        uint64_t r;
        if (SPLIT30) {  // experimental (FHE_CONV_SUM8=2): both factors split at 30 bits, no carry bookkeeping (sum8s)
            sum8s s;
            sum8s_clear(s);
#pragma unroll
            for (int i = c0; i < c0 + 8 && i < kMaxBfvLimbs; ++i)
                if (i < (int)n) {
                    uint32_t y0, y1, h0, h1;
                    split30(y[i], y0, y1);
                    split30(h[i], h0, h1);
                    sum8s_add(s, y0, y1, h0, h1);
                }
            r = sum8s_reduce(s, m, k, mulo, muhi);
        }
        else {
            sum8 s;
            sum8_clear(s);
#pragma unroll
            for (int i = c0; i < c0 + 8 && i < kMaxBfvLimbs; ++i)
                if (i < (int)n)
                    sum8_add(s, y[i], h[i]);
            r = sum8_reduce(s, m, k, mulo, muhi);
        }
        v = c0 ? add_mod(v, r, m) : r;
    }
    return v;
}

// core of FastBaseConvqToBskMontgomery: inQ (COEFF) -> outBsk (COEFF)
template <bool SPLIT30 = false>
FHE_GLOBAL void FHE_LAUNCH_BOUNDS(kThreads) behz_q_to_bsk_kernel(const BehzArgs g) {
    const uint64_t gid = (uint64_t)FHE_BID * kThreads + FHE_TID;
    if (gid >= ((uint64_t)g.batch << g.logN))
        return;
    const uint32_t b = (uint32_t)(gid >> g.logN), ri = (uint32_t)gid & ((1u << g.logN) - 1u);
    const uint64_t mtilde = (uint64_t)1 << 16, half = mtilde >> 1, mask = mtilde - 1;
    uint64_t y[kMaxBfvLimbs];
    load_rows(y, g.inQ, b, g.tb.numQ, g.logN, ri);
    uint64_t rm = 0;
#pragma unroll
    for (int i = 0; i < kMaxBfvLimbs; ++i) {
        const TwPair c    = uload_pair(g.tb.mtQHatInv, i);
        const uint64_t qi = FHE_ULOAD64(g.tb.q, i), hm = FHE_ULOAD64(g.tb.QHatModmt, i);
        if (i < (int)g.tb.numQ) {
            y[i] = mul_shoup(y[i], c.w, c.wp, qi);
            rm += y[i] * hm;  // plain 64-bit wrap-around, :1741
        }
    }
    rm &= mask;
    rm *= g.tb.negQInvModmt;
    rm &= mask;
    for (uint32_t j = 0; j < g.tb.numBsk; ++j) {
        const uint64_t bj = FHE_ULOAD64(g.tb.bsk, j);
        const TwPair cq = uload_pair(g.tb.QModbsk, j), cm = uload_pair(g.tb.mtInvModbsk, j);
        const uint64_t v = dot_row_mod<SPLIT30>(y, g.tb.QHatModbsk + (uint64_t)j * kMaxBfvLimbs, g.tb.numQ, bj,
                                       FHE_ULOAD64(g.tb.muBsk, 2 * j), FHE_ULOAD64(g.tb.muBsk, 2 * j + 1));
        uint64_t r       = rm;
        if (rm >= half)
            r += bj - mtilde;  // centred remainder, :1767-1768
        r = mul_shoup(r, cq.w, cq.wp, bj);
        r = add_mod(r, v, bj);
        *tv_at(g.outBsk, b, j, g.logN, ri) = mul_shoup(r, cm.w, cm.wp, bj);
    }
}

// FastRNSFloorq, in place on the Q and Bsk limbs (COEFF)
template <bool SPLIT30 = false>
FHE_GLOBAL void FHE_LAUNCH_BOUNDS(kThreads) behz_floorq_kernel(const BehzArgs g) {
    const uint64_t gid = (uint64_t)FHE_BID * kThreads + FHE_TID;
    if (gid >= ((uint64_t)g.batch << g.logN))
        return;
    const uint32_t b = (uint32_t)(gid >> g.logN), ri = (uint32_t)gid & ((1u << g.logN) - 1u);
    uint64_t y[kMaxBfvLimbs], xb[kMaxBfvLimbs];
    load_rows(y, g.inQ, b, g.tb.numQ, g.logN, ri);
    load_rows(xb, g.inBsk, b, g.tb.numBsk, g.logN, ri);
#pragma unroll
    for (int i = 0; i < kMaxBfvLimbs; ++i) {
        const TwPair c    = uload_pair(g.tb.tQHatInv, i);
        const uint64_t qi = FHE_ULOAD64(g.tb.q, i);
        if (i < (int)g.tb.numQ) {
            y[i] = mul_shoup(y[i], c.w, c.wp, qi);
            *tv_at(g.outQ, b, i, g.logN, ri) = y[i];  // the reference updates the Q limbs in place (:1810-1816)
        }
    }
#pragma unroll
    for (int j = 0; j < kMaxBfvLimbs; ++j) {
        if (j < (int)g.tb.numBsk) {
            const uint64_t bj = FHE_ULOAD64(g.tb.bsk, j);
            const TwPair c    = uload_pair(g.tb.tQInvModbsk, j);
            const uint64_t s  = dot_row_mod<SPLIT30>(y, g.tb.qInvModbsk + (uint64_t)j * kMaxBfvLimbs, g.tb.numQ, bj,
                                            FHE_ULOAD64(g.tb.muBsk, 2 * j), FHE_ULOAD64(g.tb.muBsk, 2 * j + 1));
            const uint64_t v  = mul_shoup(xb[j], c.w, c.wp, bj);
            *tv_at(g.outBsk, b, j, g.logN, ri) = sub_mod(v, s, bj);
        }
    }
}

// FastBaseConvSK: inBsk (COEFF) -> outQ (COEFF)
template <bool SPLIT30 = false>
FHE_GLOBAL void FHE_LAUNCH_BOUNDS(kThreads) behz_conv_sk_kernel(const BehzArgs g) {
    const uint64_t gid = (uint64_t)FHE_BID * kThreads + FHE_TID;
    if (gid >= ((uint64_t)g.batch << g.logN))
        return;
    const uint32_t b = (uint32_t)(gid >> g.logN), ri = (uint32_t)gid & ((1u << g.logN) - 1u);
    const uint32_t numB = g.tb.numBsk - 1;
    const uint64_t msk = FHE_ULOAD64(g.tb.bsk, numB), mskHalf = msk >> 1;
    uint64_t y[kMaxBfvLimbs];
    load_rows(y, g.inBsk, b, numB, g.logN, ri);
    const uint64_t xsk = *tv_at(g.inBsk, b, numB, g.logN, ri);
    uint64_t alpha = 0;
#pragma unroll
    for (int i = 0; i < kMaxBfvLimbs; ++i) {
        const TwPair c    = uload_pair(g.tb.BHatInv, i);
        const uint64_t bi = FHE_ULOAD64(g.tb.bsk, i), hm = FHE_ULOAD64(g.tb.BHatModmsk, i);
        if (i < (int)numB) {
            y[i] = mul_shoup(y[i], c.w, c.wp, bi);
            // alpha += y_i * [B/b_i]_{msk} mod msk with fully reducing ModMul/ModAddEq (:1878-1881)
            const uint64_t yi = y[i] >= msk ? y[i] % msk : y[i];
            alpha = add_mod(alpha, mul_mod_barrett(yi, hm, msk, g.tb.mskMu, (int)g.tb.mskMsb), msk);
        }
    }
    alpha = sub_mod(alpha, xsk, msk);
    alpha = mul_shoup(alpha, g.tb.BInvModmsk.w, g.tb.BInvModmsk.wp, msk);
    for (uint32_t j = 0; j < g.tb.numQ; ++j) {
        const uint64_t qj = FHE_ULOAD64(g.tb.q, j);
        const TwPair c    = uload_pair(g.tb.BModq, j);
        const uint64_t v  = dot_row_mod<SPLIT30>(y, g.tb.BHatModq + (uint64_t)j * kMaxBfvLimbs, numB, qj,
                                        FHE_ULOAD64(g.tb.muQ, 2 * j), FHE_ULOAD64(g.tb.muQ, 2 * j + 1));
        uint64_t a        = alpha;
        if (a > mskHalf)
            a = (a < msk) ? a + qj - msk : a - msk;  // ModSubFast(alpha, msk, q_j) with 64-bit wrap (:1917-1918)
        a = mul_shoup(a, c.w, c.wp, qj);
        *tv_at(g.outQ, b, j, g.logN, ri) = sub_mod(v, a, qj);
    }
}


// ---- BEHZ with 16 ... 63 Q limbs (round 5) ---------------------------------------------------------------------------------------
// The kernels above keep one coefficient's y_i in 16 registers.  Deep BFV parameter sets (the reference's
// UTBFVRNS TestMultiplicativeDepthLimitation: 17 ... 70 Q limbs on toy rings) have more: the same arithmetic with the y_i in a
// per-lane array (private memory), tables of stride kBehzWideLimbs, run-time loops.  Every sum is exact (integer sums reduced in
// chunks of 8 products, modular addition of the chunks), so the residues are the reference's whatever the chunking; throughput is
// not a goal here — the members just must not fall back to the host mirror.
constexpr int kBehzWideLimbs = 128;
FHE_HD uint64_t dot_row_mod_wide(const uint64_t* y, const uint64_t* row, uint32_t n, uint64_t m, uint64_t mulo, uint64_t muhi) {
    const uint32_t k = 64u - (uint32_t)__builtin_clzll(m);
    uint64_t v       = 0;
    for (uint32_t c0 = 0; c0 < n; c0 += 8) {
        sum8 s;
        sum8_clear(s);
        for (uint32_t i = c0; i < c0 + 8 && i < n; ++i)
            sum8_add(s, y[i], FHE_ULOAD64(row, i));
        const uint64_t r = sum8_reduce(s, m, k, mulo, muhi);
        v                = c0 ? add_mod(v, r, m) : r;
    }
    return v;
}
FHE_GLOBAL void FHE_LAUNCH_BOUNDS(kThreads) behz_q_to_bsk_wide_kernel(const BehzArgs g) {
    const uint64_t gid = (uint64_t)FHE_BID * kThreads + FHE_TID;
    if (gid >= ((uint64_t)g.batch << g.logN))
        return;
    const uint32_t b = (uint32_t)(gid >> g.logN), ri = (uint32_t)gid & ((1u << g.logN) - 1u);
    const uint64_t mtilde = (uint64_t)1 << 16, half = mtilde >> 1, mask = mtilde - 1;
    uint64_t y[kBehzWideLimbs];
    uint64_t rm = 0;
    for (uint32_t i = 0; i < g.tb.numQ; ++i) {
        const TwPair c = uload_pair(g.tb.mtQHatInv, i);
        y[i]           = mul_shoup(*tv_at(g.inQ, b, i, g.logN, ri), c.w, c.wp, FHE_ULOAD64(g.tb.q, i));
        rm += y[i] * FHE_ULOAD64(g.tb.QHatModmt, i);  // plain 64-bit wrap-around, :1741
    }
    rm &= mask;
    rm *= g.tb.negQInvModmt;
    rm &= mask;
    for (uint32_t j = 0; j < g.tb.numBsk; ++j) {
        const uint64_t bj = FHE_ULOAD64(g.tb.bsk, j);
        const TwPair cq = uload_pair(g.tb.QModbsk, j), cm = uload_pair(g.tb.mtInvModbsk, j);
        const uint64_t v = dot_row_mod_wide(y, g.tb.QHatModbsk + (uint64_t)j * g.tb.W, g.tb.numQ, bj, FHE_ULOAD64(g.tb.muBsk, 2 * j),
                                            FHE_ULOAD64(g.tb.muBsk, 2 * j + 1));
        uint64_t r = rm;
        if (rm >= half)
            r += bj - mtilde;  // centred remainder, :1767-1768
        r = mul_shoup(r, cq.w, cq.wp, bj);
        r = add_mod(r, v, bj);
        *tv_at(g.outBsk, b, j, g.logN, ri) = mul_shoup(r, cm.w, cm.wp, bj);
    }
}
FHE_GLOBAL void FHE_LAUNCH_BOUNDS(kThreads) behz_floorq_wide_kernel(const BehzArgs g) {
    const uint64_t gid = (uint64_t)FHE_BID * kThreads + FHE_TID;
    if (gid >= ((uint64_t)g.batch << g.logN))
        return;
    const uint32_t b = (uint32_t)(gid >> g.logN), ri = (uint32_t)gid & ((1u << g.logN) - 1u);
    uint64_t y[kBehzWideLimbs];
    for (uint32_t i = 0; i < g.tb.numQ; ++i) {
        const TwPair c = uload_pair(g.tb.tQHatInv, i);
        y[i]           = mul_shoup(*tv_at(g.inQ, b, i, g.logN, ri), c.w, c.wp, FHE_ULOAD64(g.tb.q, i));
    }
    for (uint32_t j = 0; j < g.tb.numBsk; ++j) {  // (the Bsk limbs first: in place, outQ aliases inQ only row by row)
        const uint64_t bj = FHE_ULOAD64(g.tb.bsk, j);
        const TwPair c    = uload_pair(g.tb.tQInvModbsk, j);
        const uint64_t s  = dot_row_mod_wide(y, g.tb.qInvModbsk + (uint64_t)j * g.tb.W, g.tb.numQ, bj, FHE_ULOAD64(g.tb.muBsk, 2 * j),
                                            FHE_ULOAD64(g.tb.muBsk, 2 * j + 1));
        const uint64_t v  = mul_shoup(*tv_at(g.inBsk, b, j, g.logN, ri), c.w, c.wp, bj);
        *tv_at(g.outBsk, b, j, g.logN, ri) = sub_mod(v, s, bj);
    }
    for (uint32_t i = 0; i < g.tb.numQ; ++i)
        *tv_at(g.outQ, b, i, g.logN, ri) = y[i];  // the reference updates the Q limbs in place (:1810-1816)
}
FHE_GLOBAL void FHE_LAUNCH_BOUNDS(kThreads) behz_conv_sk_wide_kernel(const BehzArgs g) {
    const uint64_t gid = (uint64_t)FHE_BID * kThreads + FHE_TID;
    if (gid >= ((uint64_t)g.batch << g.logN))
        return;
    const uint32_t b = (uint32_t)(gid >> g.logN), ri = (uint32_t)gid & ((1u << g.logN) - 1u);
    const uint32_t numB = g.tb.numBsk - 1;
    const uint64_t msk = FHE_ULOAD64(g.tb.bsk, numB), mskHalf = msk >> 1;
    uint64_t y[kBehzWideLimbs];
    const uint64_t xsk = *tv_at(g.inBsk, b, numB, g.logN, ri);
    uint64_t alpha     = 0;
    for (uint32_t i = 0; i < numB; ++i) {
        const TwPair c = uload_pair(g.tb.BHatInv, i);
        y[i]           = mul_shoup(*tv_at(g.inBsk, b, i, g.logN, ri), c.w, c.wp, FHE_ULOAD64(g.tb.bsk, i));
        const uint64_t yi = y[i] >= msk ? y[i] % msk : y[i];  // fully reducing ModMul / ModAddEq (:1878-1881)
        alpha = add_mod(alpha, mul_mod_barrett(yi, FHE_ULOAD64(g.tb.BHatModmsk, i), msk, g.tb.mskMu, (int)g.tb.mskMsb), msk);
    }
    alpha = sub_mod(alpha, xsk, msk);
    alpha = mul_shoup(alpha, g.tb.BInvModmsk.w, g.tb.BInvModmsk.wp, msk);
    for (uint32_t j = 0; j < g.tb.numQ; ++j) {
        const uint64_t qj = FHE_ULOAD64(g.tb.q, j);
        const TwPair c    = uload_pair(g.tb.BModq, j);
        const uint64_t v  = dot_row_mod_wide(y, g.tb.BHatModq + (uint64_t)j * g.tb.W, numB, qj, FHE_ULOAD64(g.tb.muQ, 2 * j),
                                            FHE_ULOAD64(g.tb.muQ, 2 * j + 1));
        uint64_t a = alpha;
        if (a > mskHalf)
            a = (a < msk) ? a + qj - msk : a - msk;  // ModSubFast(alpha, msk, q_j) with 64-bit wrap (:1917-1918)
        a = mul_shoup(a, c.w, c.wp, qj);
        *tv_at(g.outQ, b, j, g.logN, ri) = sub_mod(v, a, qj);
    }
}

}  // namespace fhe
#endif
