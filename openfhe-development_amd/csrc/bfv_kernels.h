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
            nu += g.frac[i] * (double)(*tv_at(g.in, b, i, g.logN, ri));  // :1543-1548, i ascending
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
            acc128(acc, *tv_at(g.in, b, i, g.logN, ri), tj[i]);
        acc128(acc, *tv_at(g.own, b, j, g.logN, ri), tj[g.sizeI]);
        const uint64_t oj = g.o[j], mlo = g.mu[2 * j], mhi = g.mu[2 * j + 1];
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

// ---- ScaleAndRoundPOverQ ----------------------------------------------------------------------------
struct POverQArgs {
    TowerView x;    // [sizeQ+1] rows, the last one modulo pLast
    TowerView out;  // [sizeQ]
    const uint64_t* q;       // unused (kept for ABI stability of the struct)
    const TwPair* qPairs;    // [sizeQ] moduli in .w
    const TwPair* pInv;      // [sizeQ] [p^-1]_{q_i}
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
        const uint64_t qn = g.qPairs[i].w;
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
struct BehzTables {           // bfvrns-cryptoparameters.cpp:673-850; all device resident
    const uint64_t *q, *bsk;                 // [numQ], [numBsk]
    const uint64_t *muQ, *muBsk;             // [.][2]
    const TwPair* mtQHatInv;                 // [numQ]   [mtilde (Q/q_i)^-1]_{q_i}
    const uint64_t* QHatModbsk;              // [numQ][numBsk]
    const uint64_t* QHatModmt;               // [numQ]
    const TwPair *QModbsk, *mtInvModbsk;     // [numBsk]
    const TwPair* tQHatInv;                  // [numQ]
    const uint64_t* qInvModbsk;              // [numQ][numBsk]
    const TwPair* tQInvModbsk;               // [numBsk]
    const TwPair* BHatInv;                   // [numB]
    const uint64_t* BHatModmsk;              // [numB]
    const uint64_t* BHatModq;                // [numB][numQ]
    const TwPair* BModq;                     // [numQ]
    TwPair BInvModmsk;
    uint64_t negQInvModmt;
    uint64_t mskMu;                          // ComputeMu(msk)
    uint32_t mskMsb;
    uint32_t numQ, numBsk;
};
struct BehzArgs {
    TowerView inQ, inBsk;   // source limbs
    TowerView outQ, outBsk; // destination limbs
    BehzTables tb;
    uint32_t logN, batch;
};

// core of FastBaseConvqToBskMontgomery: inQ (COEFF) -> outBsk (COEFF)
FHE_GLOBAL void FHE_LAUNCH_BOUNDS(kThreads) behz_q_to_bsk_kernel(const BehzArgs g) {
    const uint64_t gid = (uint64_t)FHE_BID * kThreads + FHE_TID;
    if (gid >= ((uint64_t)g.batch << g.logN))
        return;
    const uint32_t b = (uint32_t)(gid >> g.logN), ri = (uint32_t)gid & ((1u << g.logN) - 1u);
    const uint64_t mtilde = (uint64_t)1 << 16, half = mtilde >> 1, mask = mtilde - 1;
    uint64_t y[kMaxBfvLimbs];
    uint64_t rm = 0;
#pragma unroll
    for (int i = 0; i < kMaxBfvLimbs; ++i) {
        if (i < (int)g.tb.numQ) {
            const TwPair c = g.tb.mtQHatInv[i];
            y[i]           = mul_shoup(*tv_at(g.inQ, b, i, g.logN, ri), c.w, c.wp, g.tb.q[i]);
            rm += y[i] * g.tb.QHatModmt[i];  // plain 64-bit wrap-around, :1741
        }
    }
    rm &= mask;
    rm *= g.tb.negQInvModmt;
    rm &= mask;
    for (uint32_t j = 0; j < g.tb.numBsk; ++j) {
        const uint64_t bj = g.tb.bsk[j];
        u128w acc{0, 0};
#pragma unroll
        for (int i = 0; i < kMaxBfvLimbs; ++i)
            if (i < (int)g.tb.numQ)
                acc128(acc, y[i], g.tb.QHatModbsk[(uint64_t)i * g.tb.numBsk + j]);
        const uint64_t v = barrett128(acc, bj, g.tb.muBsk[2 * j], g.tb.muBsk[2 * j + 1]);
        uint64_t r       = rm;
        if (rm >= half)
            r += bj - mtilde;  // centred remainder, :1767-1768
        const TwPair cq = g.tb.QModbsk[j], cm = g.tb.mtInvModbsk[j];
        r = mul_shoup(r, cq.w, cq.wp, bj);
        r = add_mod(r, v, bj);
        *tv_at(g.outBsk, b, j, g.logN, ri) = mul_shoup(r, cm.w, cm.wp, bj);
    }
}

// FastRNSFloorq, in place on the Q and Bsk limbs (COEFF)
FHE_GLOBAL void FHE_LAUNCH_BOUNDS(kThreads) behz_floorq_kernel(const BehzArgs g) {
    const uint64_t gid = (uint64_t)FHE_BID * kThreads + FHE_TID;
    if (gid >= ((uint64_t)g.batch << g.logN))
        return;
    const uint32_t b = (uint32_t)(gid >> g.logN), ri = (uint32_t)gid & ((1u << g.logN) - 1u);
    uint64_t y[kMaxBfvLimbs];
#pragma unroll
    for (int i = 0; i < kMaxBfvLimbs; ++i) {
        if (i < (int)g.tb.numQ) {
            const TwPair c = g.tb.tQHatInv[i];
            y[i]           = mul_shoup(*tv_at(g.inQ, b, i, g.logN, ri), c.w, c.wp, g.tb.q[i]);
            *tv_at(g.outQ, b, i, g.logN, ri) = y[i];  // the reference updates the Q limbs in place (:1810-1816)
        }
    }
    for (uint32_t j = 0; j < g.tb.numBsk; ++j) {
        const uint64_t bj = g.tb.bsk[j];
        u128w acc{0, 0};
#pragma unroll
        for (int i = 0; i < kMaxBfvLimbs; ++i)
            if (i < (int)g.tb.numQ)
                acc128(acc, y[i], g.tb.qInvModbsk[(uint64_t)i * g.tb.numBsk + j]);
        const uint64_t s = barrett128(acc, bj, g.tb.muBsk[2 * j], g.tb.muBsk[2 * j + 1]);
        const TwPair c   = g.tb.tQInvModbsk[j];
        const uint64_t v = mul_shoup(*tv_at(g.inBsk, b, j, g.logN, ri), c.w, c.wp, bj);
        *tv_at(g.outBsk, b, j, g.logN, ri) = sub_mod(v, s, bj);
    }
}

// FastBaseConvSK: inBsk (COEFF) -> outQ (COEFF)
FHE_GLOBAL void FHE_LAUNCH_BOUNDS(kThreads) behz_conv_sk_kernel(const BehzArgs g) {
    const uint64_t gid = (uint64_t)FHE_BID * kThreads + FHE_TID;
    if (gid >= ((uint64_t)g.batch << g.logN))
        return;
    const uint32_t b = (uint32_t)(gid >> g.logN), ri = (uint32_t)gid & ((1u << g.logN) - 1u);
    const uint32_t numB = g.tb.numBsk - 1;
    const uint64_t msk = g.tb.bsk[numB], mskHalf = msk >> 1;
    uint64_t y[kMaxBfvLimbs];
    uint64_t alpha = 0;
#pragma unroll
    for (int i = 0; i < kMaxBfvLimbs; ++i) {
        if (i < (int)numB) {
            const TwPair c = g.tb.BHatInv[i];
            y[i]           = mul_shoup(*tv_at(g.inBsk, b, i, g.logN, ri), c.w, c.wp, g.tb.bsk[i]);
            // alpha += y_i * [B/b_i]_{msk} mod msk with fully reducing ModMul/ModAddEq (:1878-1881)
            const uint64_t yi = y[i] >= msk ? y[i] % msk : y[i];
            alpha = add_mod(alpha, mul_mod_barrett(yi, g.tb.BHatModmsk[i], msk, g.tb.mskMu, (int)g.tb.mskMsb), msk);
        }
    }
    alpha = sub_mod(alpha, *tv_at(g.inBsk, b, numB, g.logN, ri), msk);
    alpha = mul_shoup(alpha, g.tb.BInvModmsk.w, g.tb.BInvModmsk.wp, msk);
    for (uint32_t j = 0; j < g.tb.numQ; ++j) {
        const uint64_t qj = g.tb.q[j];
        u128w acc{0, 0};
#pragma unroll
        for (int i = 0; i < kMaxBfvLimbs; ++i)
            if (i < (int)numB)
                acc128(acc, y[i], g.tb.BHatModq[(uint64_t)i * g.tb.numQ + j]);
        const uint64_t v = barrett128(acc, qj, g.tb.muQ[2 * j], g.tb.muQ[2 * j + 1]);
        uint64_t a       = alpha;
        if (a > mskHalf)
            a = (a < msk) ? a + qj - msk : a - msk;  // ModSubFast(alpha, msk, q_j) with 64-bit wrap (:1917-1918)
        const TwPair c = g.tb.BModq[j];
        a              = mul_shoup(a, c.w, c.wp, qj);
        *tv_at(g.outQ, b, j, g.logN, ri) = sub_mod(v, a, qj);
    }
}

}  // namespace fhe
#endif
