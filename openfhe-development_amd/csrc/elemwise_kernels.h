// elemwise_kernels.h — per-coefficient tower operations (HBM-bound streams).
//
// Replaces the per-limb OpenMP loops of DCRTPolyImpl::operator+=, -=, *=, Times(vector<NativeInteger>),
// Negate (src/core/include/lattice/hal/default/dcrtpoly-impl.h:347-408, 582-620; dcrtpoly.h:131-189)
// and the NativeVectorT kernels they call (src/core/lib/math/hal/intnat/mubintvecnat.cpp:132-142,
// 229-339), the EVALUATION/COEFFICIENT automorphisms (poly-impl.h:345-376), the centred modulus
// switch (mubintvecnat.cpp:109-122) and the tensor product of LeveledSHEBase::EvalMultCore
// (src/pke/lib/schemebase/base-leveledshe.cpp:607-644).
//
// Work decomposition: a workgroup owns 4096 consecutive words of the [rows][N] tower array (one
// "tile", the same granule as the NTT), each lane moves 16-byte pairs, so every wave instruction is a
// fully coalesced 1 KiB access.  For N >= 4096 a tile lies inside one limb and the modulus is uniform.
#ifndef FHE_ELEMWISE_KERNELS_H
#define FHE_ELEMWISE_KERNELS_H
#include "modarith.h"
#include "launch.h"
#include "ntt_kernels.h"

namespace fhe {

struct LimbConst {  // per-context-limb constants for exact a*b mod q
    uint64_t q;
    uint64_t mu;   // floor(2^(2*msb+3)/q)    (ComputeMu, ubintnat.h:642-647)
    uint32_t msb;  // bit length of q
    uint32_t pad;
};

enum ElemOp : int {
    OP_ADD = 0,        // out = a + b
    OP_SUB = 1,        // out = a - b
    OP_MUL = 2,        // out = a * b
    OP_NEG = 3,        // out = -a
    OP_MUL_CONST = 4,  // out = a * c[row]         (c as Shoup pair)
    OP_SUB_MUL_CONST = 5,  // out = (a - b) * c[row]   (ApproxModDown tail, dcrtpoly-impl.h:1002)
    OP_MUL_CONST_ADD = 6,  // out = a * c[row] + b     (DropLastElementAndScale tail, :707-708)
    OP_MULT_ACC = 7,   // out = out + a * b        (inner-product accumulate)
    OP_COPY = 8,
    OP_SUB_MUL_CONST_ACC = 9,  // out = out + (a - b) * c[row]   (ApproxModDown tail fused with EvalMult's `+= ks`)
    OP_ADD_CONST = 10,     // out = a + c[row]         (PolyImpl::Plus(Integer) in EVALUATION / Minus(Integer) with c = q - c)
    OP_ADD_CONST_AT0 = 11, // out = a + c[row] at coefficient 0 only (PolyImpl::Plus(Integer) in COEFFICIENT, poly-impl.h:213-214)
    OP_TIMES_QOVERT = 12,  // out = ((a * pre) mod preMod) * c[row]   (DCRTPolyImpl::TimesQovert, dcrtpoly-impl.h:868-885)
};

struct ElemArgs {
    uint64_t* out;
    const uint64_t* a;
    const uint64_t* b;
    const LimbConst* lc;   // [ctxLimbs]
    const TwPair* consts;  // [nLimbs] per tower row-in-tower constant (Shoup pair), may be null
    uint32_t logN, nLimbs, rows;
    uint32_t aStride, aFirst;  // aStride != 0: operand a is a [batch][aStride][N] view, rows aFirst.. of each tower
    uint32_t bStride, bFirst;  // same for b
    uint32_t oStride, oFirst;  // same for out
    uint32_t cvFirst = 0;      // elemwise_cv_kernel: this launch takes rows [cvFirst, cvFirst + kConstVecLimbs) of every tower
    TwPair pre = {0, 0};       // OP_TIMES_QOVERT only: the first factor (Shoup pair modulo preMod) ...
    uint64_t preMod = 0;       // ... and its modulus (the plaintext modulus t)
    // != 0: consecutive towers of the operand are this many WORDS apart (signed: towers allocated on their own, e.g. the two
    // elements of a ciphertext, any distance apart in either direction); overrides the row stride
    int64_t aDelta = 0, bDelta = 0, oDelta = 0;
    LimbSel sel;
};

template <int OP>
FHE_HD uint64_t elem_apply(uint64_t o, uint64_t a, uint64_t b, const LimbConst lc, const TwPair c) {
    const uint64_t q = lc.q;
    switch (OP) {
        case OP_ADD:
            return add_mod(a, b, q);
        case OP_SUB:
            return sub_mod(a, b, q);
        case OP_MUL:
            return mul_mod_barrett(a, b, q, lc.mu, (int)lc.msb);
        case OP_NEG:
            return a == 0 ? 0 : q - a;
        case OP_MUL_CONST:
            return mul_shoup(a, c.w, c.wp, q);
        case OP_SUB_MUL_CONST:
            return mul_shoup(sub_mod(a, b, q), c.w, c.wp, q);
        case OP_MUL_CONST_ADD:
            return add_mod(mul_shoup(a, c.w, c.wp, q), b, q);
        case OP_MULT_ACC:
            return add_mod(o, mul_mod_barrett(a, b, q, lc.mu, (int)lc.msb), q);
        case OP_SUB_MUL_CONST_ACC:
            return add_mod(o, mul_shoup(sub_mod(a, b, q), c.w, c.wp, q), q);
        case OP_ADD_CONST:
        case OP_ADD_CONST_AT0:
            return add_mod(a, c.w, q);
        case OP_TIMES_QOVERT:  // (a already holds (x * NegQModt) mod t, see elemwise_body; :882 ModMulFastEq = generalized Barrett)
            return mul_mod_barrett(a, c.w, q, lc.mu, (int)lc.msb);
        default:
            return a;
    }
}

// per-call constant vector passed BY VALUE in the kernel arguments (fhe_mul_const / fhe_mult_acc: the caller's host
// constants need no device staging buffer, so the call stays asynchronous and capturable into a HIP graph)
// (kernel arguments are limited to 4 KiB: the vector carries kConstVecLimbs rows; a tower with more rows takes one launch per window of
// rows — `cvFirst` of ElemArgs — and the rows outside the window are left to the other launches)
constexpr int kConstVecLimbs = 128;
struct ConstVec {
    TwPair c[kConstVecLimbs];
};

template <int OP, typename ConstAt, typename InWindow>
FHE_DEV void elemwise_body(const ElemArgs& g, ConstAt constAt, InWindow inWindow) {
    const uint32_t t          = FHE_TID;
    const uint64_t base       = (uint64_t)FHE_BID << kTileLog;
    const uint64_t totalWords = (uint64_t)g.rows << g.logN;
    constexpr bool needB = (OP == OP_ADD || OP == OP_SUB || OP == OP_MUL || OP == OP_SUB_MUL_CONST ||
                            OP == OP_MUL_CONST_ADD || OP == OP_MULT_ACC || OP == OP_SUB_MUL_CONST_ACC);
    constexpr bool needC = (OP == OP_MUL_CONST || OP == OP_SUB_MUL_CONST || OP == OP_MUL_CONST_ADD || OP == OP_SUB_MUL_CONST_ACC ||
                            OP == OP_ADD_CONST || OP == OP_ADD_CONST_AT0 || OP == OP_TIMES_QOVERT);
    constexpr bool needO = (OP == OP_MULT_ACC || OP == OP_SUB_MUL_CONST_ACC);
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const uint64_t off = base + (((uint64_t)m * kThreads + t) << 1);
        if (off >= totalWords)
            continue;
        const uint32_t row  = (uint32_t)(off >> g.logN);
        const uint32_t rit  = row % g.nLimbs;
        if (!inWindow(rit))
            continue;
        const LimbConst lc  = g.lc[g.sel.idx[rit]];
        TwPair c            = {0, 0};
        if (needC)
            c = constAt(rit);
        const uint32_t tb   = row / g.nLimbs;
        const uint64_t ri   = off & (((uint64_t)1 << g.logN) - 1u);
        const uint64_t inTower = ((uint64_t)rit << g.logN) + ri;
        const uint64_t aoff = g.aDelta ? (uint64_t)((int64_t)tb * g.aDelta) + inTower
                                       : g.aStride ? ((((uint64_t)tb * g.aStride + g.aFirst + rit) << g.logN) + ri) : off;
        uint64_t a0 = g.a[aoff], a1 = g.a[aoff + 1];
        if (OP == OP_TIMES_QOVERT) {  // :881 xi.ModMulFastConstEq(NegQModt, t, NegQModtPrecon)
            a0 = mul_shoup(a0, g.pre.w, g.pre.wp, g.preMod);
            a1 = mul_shoup(a1, g.pre.w, g.pre.wp, g.preMod);
        }
        uint64_t b0 = 0, b1 = 0, o0 = 0, o1 = 0;
        if (needB) {
            const uint64_t boff = g.bDelta ? (uint64_t)((int64_t)tb * g.bDelta) + inTower
                                           : g.bStride ? ((((uint64_t)tb * g.bStride + g.bFirst + rit) << g.logN) + ri) : off;
            b0 = g.b[boff];
            b1 = g.b[boff + 1];
        }
        const uint64_t ooff = g.oDelta ? (uint64_t)((int64_t)tb * g.oDelta) + inTower
                                       : g.oStride ? ((((uint64_t)tb * g.oStride + g.oFirst + rit) << g.logN) + ri) : off;
        if (needO) {
            o0 = g.out[ooff];
            o1 = g.out[ooff + 1];
        }
        if (OP == OP_ADD_CONST_AT0) {  // (off is even: only the first word of the pair can be coefficient 0)
            g.out[ooff]     = elem_apply<OP>(o0, a0, b0, lc, ri == 0 ? c : TwPair{0, 0});
            g.out[ooff + 1] = a1;
            continue;
        }
        g.out[ooff]     = elem_apply<OP>(o0, a0, b0, lc, c);
        g.out[ooff + 1] = elem_apply<OP>(o1, a1, b1, lc, c);
    }
}
template <int OP>
FHE_GLOBAL void FHE_LAUNCH_BOUNDS(kThreads) elemwise_kernel(const ElemArgs g) {
    elemwise_body<OP>(g, [&](uint32_t rit) { return g.consts[rit]; }, [](uint32_t) { return true; });
}
template <int OP>
FHE_GLOBAL void FHE_LAUNCH_BOUNDS(kThreads) elemwise_cv_kernel(const ElemArgs g, const ConstVec cv) {
    elemwise_body<OP>(g, [&](uint32_t rit) { return cv.c[rit - g.cvFirst]; },
                      [&](uint32_t rit) { return rit - g.cvFirst < (uint32_t)kConstVecLimbs; });
}

// ---- linear combination of towers with per-limb constants: out = sum_i c_i (.) x_i  [+ out] ------------------------------------------
// pke's weighted sums (internalEvalLinearWSumMutable, ckksrns-advancedshe.cpp:97-136: the inner loops of the Chebyshev evaluation of
// bootstrapping) multiply every term by its constant and add the products one by one: 2n - 1 tower-sized launches moving 5n - 3
// towers.  One launch reads each term once and writes the sum once (n + 1 towers).  Exact modular arithmetic: any order of the sum
// gives the reference's residues.  Every x_i is a dense [batch][nLimbs][N] tower of its own allocation.
constexpr int kMaxLinTerms = 16;
struct LinCombArgs {
    uint64_t* out;
    const uint64_t* x[kMaxLinTerms];
    const TwPair* consts;  // [nTerms][nLimbs] Shoup pairs (device)
    const uint64_t* q;     // [ctxLimbs]
    uint32_t nTerms, logN, nLimbs, rows, accumulate;
    LimbSel sel;
};
FHE_GLOBAL void FHE_LAUNCH_BOUNDS(kThreads) lincomb_kernel(const LinCombArgs g) {
    const uint32_t t          = FHE_TID;
    const uint64_t base       = (uint64_t)FHE_BID << kTileLog;
    const uint64_t totalWords = (uint64_t)g.rows << g.logN;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const uint64_t off = base + (((uint64_t)m * kThreads + t) << 1);
        if (off >= totalWords)
            continue;
        const uint32_t rit  = (uint32_t)(off >> g.logN) % g.nLimbs;
        const uint64_t q    = g.q[g.sel.idx[rit]];
        const uint64_t twoq = q << 1, nq = 0 - q;
        uint64_t s0 = 0, s1 = 0;  // lazily below 2q
        for (uint32_t i = 0; i < g.nTerms; ++i) {
            const TwPair c    = g.consts[(size_t)i * g.nLimbs + rit];
            const uint64_t a0 = g.x[i][off], a1 = g.x[i][off + 1];
            s0 = csub(s0 + mul_shoup_lazy_nq(a0, c.w, c.wp, nq), twoq);
            s1 = csub(s1 + mul_shoup_lazy_nq(a1, c.w, c.wp, nq), twoq);
        }
        s0 = csub(s0, q), s1 = csub(s1, q);
        if (g.accumulate) {
            s0 = add_mod(g.out[off], s0, q);
            s1 = add_mod(g.out[off + 1], s1, q);
        }
        g.out[off]     = s0;
        g.out[off + 1] = s1;
    }
}

// ---- EvalMultCore tensor product: d0 = a0*b0, d1 = a0*b1 + a1*b0, d2 = a1*b1 ------------------------
struct TensorArgs {
    const uint64_t *a0, *a1, *b0, *b1;
    uint64_t *d0, *d1, *d2;
    const LimbConst* lc;
    uint32_t logN, nLimbs, rows;
    LimbSel sel;
};
FHE_GLOBAL void FHE_LAUNCH_BOUNDS(kThreads) tensor_kernel(const TensorArgs g) {
    const uint32_t t          = FHE_TID;
    const uint64_t base       = (uint64_t)FHE_BID << kTileLog;
    const uint64_t totalWords = (uint64_t)g.rows << g.logN;
#pragma unroll 2
    for (int m = 0; m < 16; ++m) {
        const uint64_t off = base + (uint64_t)m * kThreads + t;
        if (off >= totalWords)
            continue;
        const uint32_t row = (uint32_t)(off >> g.logN);
        const LimbConst lc = g.lc[g.sel.idx[row % g.nLimbs]];
        const uint64_t x0 = g.a0[off], x1 = g.a1[off], y0 = g.b0[off], y1 = g.b1[off];
        const int msb = (int)lc.msb;
        g.d0[off] = mul_mod_barrett(x0, y0, lc.q, lc.mu, msb);
        g.d1[off] = add_mod(mul_mod_barrett(x0, y1, lc.q, lc.mu, msb), mul_mod_barrett(x1, y0, lc.q, lc.mu, msb), lc.q);
        g.d2[off] = mul_mod_barrett(x1, y1, lc.q, lc.mu, msb);
    }
}

// LeveledSHEBase::EvalSquareCore for a 2-element ciphertext (base-leveledshe.cpp:646-664):
// d0 = a0*a0, d1 = a0*a1 then d1 += d1, d2 = a1*a1
struct TensorSqArgs {
    const uint64_t *a0, *a1;
    uint64_t *d0, *d1, *d2;
    const LimbConst* lc;
    uint32_t logN, nLimbs, rows;
    LimbSel sel;
};
FHE_GLOBAL void FHE_LAUNCH_BOUNDS(kThreads) tensor_square_kernel(const TensorSqArgs g) {
    const uint32_t t          = FHE_TID;
    const uint64_t base       = (uint64_t)FHE_BID << kTileLog;
    const uint64_t totalWords = (uint64_t)g.rows << g.logN;
#pragma unroll 2
    for (int m = 0; m < 16; ++m) {
        const uint64_t off = base + (uint64_t)m * kThreads + t;
        if (off >= totalWords)
            continue;
        const uint32_t row = (uint32_t)(off >> g.logN);
        const LimbConst lc = g.lc[g.sel.idx[row % g.nLimbs]];
        const uint64_t x0 = g.a0[off], x1 = g.a1[off];
        const int msb = (int)lc.msb;
        const uint64_t m01 = mul_mod_barrett(x0, x1, lc.q, lc.mu, msb);
        g.d0[off] = mul_mod_barrett(x0, x0, lc.q, lc.mu, msb);
        g.d1[off] = add_mod(m01, m01, lc.q);
        g.d2[off] = mul_mod_barrett(x1, x1, lc.q, lc.mu, msb);
    }
}

// ---- automorphism -------------------------------------------------------------------------------
// EVALUATION: out[j] = in[precomp[j]]  (poly-impl.h:366-376), precomp[bitrev(j)] = bitrev(((2j+1)k mod 2N)>>1)
// computed on the fly (nbtheory2.cpp:264-275) so no table has to be shipped: for output index jr,
// j = bitrev(jr), idx = (((2j+1)*k) mod 2N) >> 1, source = bitrev(idx).
// COEFFICIENT: out[(j*k) mod N] = ((j*k)>>logN)&1 ? q - in[j] : in[j]   (poly-impl.h:359-362)
struct AutoArgs {
    uint64_t* out;
    const uint64_t* in;
    const uint64_t* q;  // [ctxLimbs]
    uint32_t logN, nLimbs, rows, k, evalFormat;
    LimbSel sel;
};
FHE_HD uint32_t bitrev32(uint32_t x, uint32_t nbits) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __brev(x) >> (32 - nbits);
#else
    uint32_t r = 0;
    for (uint32_t i = 0; i < nbits; ++i)
        r |= ((x >> i) & 1u) << (nbits - 1 - i);
    return r;
#endif
}
// source index of the output index whose bit reversal is j under the EVALUATION-format automorphism k (the map above)
FHE_HD uint32_t automorph_source(uint32_t j, uint32_t k, uint32_t logN) {
    const uint32_t idx = (((2u * j + 1u) * k) & ((2u << logN) - 1u)) >> 1;
    return bitrev32(idx, logN);
}
FHE_GLOBAL void FHE_LAUNCH_BOUNDS(kThreads) automorph_kernel(const AutoArgs g) {
    const uint32_t t          = FHE_TID;
    const uint64_t base       = (uint64_t)FHE_BID << kTileLog;
    const uint64_t totalWords = (uint64_t)g.rows << g.logN;
    const uint32_t N = 1u << g.logN, mask = N - 1u;
#pragma unroll 4
    for (int m = 0; m < 16; ++m) {
        const uint64_t off = base + (uint64_t)m * kThreads + t;
        if (off >= totalWords)
            continue;
        const uint64_t rowBase = off & ~(uint64_t)mask;
        const uint32_t jr      = (uint32_t)off & mask;
        if (g.evalFormat) {
            g.out[off] = g.in[rowBase + automorph_source(bitrev32(jr, g.logN), g.k, g.logN)];
        }
        else {
            // gather form of the scatter in the reference: out[jk mod N] = +-in[j]  <=>  j = jr * k^-1 mod 2N
            // (k odd => invertible mod 2N); host passes kInv in g.k? no: keep scatter semantics exactly:
            const uint32_t jk  = jr * g.k;  // this lane is SOURCE index jr
            const uint32_t row = (uint32_t)(off >> g.logN);
            const uint64_t q   = g.q[g.sel.idx[row % g.nLimbs]];
            const uint64_t v   = g.in[off];
            g.out[rowBase + (jk & mask)] = ((jk >> g.logN) & 1u) ? q - v : v;
        }
    }
}

// out (+)= sum_s Automorphism_{k_s}(in_s), EVALUATION format: the `first += ...AutomorphismTransform(...)` and
// `EvalAddExtInPlace(result, EvalFastRotationExt(...))` accumulations of the BSGS linear transform
// (ckksrns-fhe.cpp:1868-1876) over all outer steps in one pass (modular addition is exact, so the order is free).
// k_s = 1 is the identity (an outer step without rotation).
constexpr int kMaxAutoSum = 32;
struct AutoSumArgs {
    uint64_t* out;
    const uint64_t* in[kMaxAutoSum];
    uint32_t k[kMaxAutoSum];
    const uint64_t* q;  // [ctxLimbs]
    uint32_t logN, nLimbs, rows, nSrc, accumulate;
    LimbSel sel;
};
FHE_GLOBAL void FHE_LAUNCH_BOUNDS(kThreads) automorph_sum_kernel(const AutoSumArgs g) {
    const uint32_t t          = FHE_TID;
    const uint64_t base       = (uint64_t)FHE_BID << kTileLog;
    const uint64_t totalWords = (uint64_t)g.rows << g.logN;
    const uint32_t N = 1u << g.logN, mask = N - 1u;
#pragma unroll 2
    for (int m = 0; m < 16; ++m) {
        const uint64_t off = base + (uint64_t)m * kThreads + t;
        if (off >= totalWords)
            continue;
        const uint64_t rowBase = off & ~(uint64_t)mask;
        const uint32_t j       = bitrev32((uint32_t)off & mask, g.logN);
        const uint64_t q       = g.q[g.sel.idx[(uint32_t)(off >> g.logN) % g.nLimbs]];
        uint64_t v             = g.accumulate ? g.out[off] : 0;
        for (uint32_t s = 0; s < g.nSrc; ++s) {
            v = add_mod(v, g.in[s][rowBase + automorph_source(j, g.k[s], g.logN)], q);
        }
        g.out[off] = v;
    }
}

// ---- centred modulus switch of ONE source limb into every limb of a tower (a9) ----------------------
// out[b][i][r] = SwitchModulus(src[b][r] : qSrc -> q_i)  (mubintvecnat.cpp:109-122); used by
// DropLastElementAndScale (dcrtpoly-impl.h:703-704) and ModRaise (dcrtpoly-impl.h:87-93).
// Optionally fused with the per-limb constant multiply that follows it in the rescale (tmp *= const).
struct SwitchModArgs {
    uint64_t* out;        // [batch][nLimbs][N]
    const uint64_t* src;  // [batch][srcStrideLimbs][N], limb srcLimbPos of each tower is the source
    const uint64_t* q;    // [ctxLimbs]
    const TwPair* consts; // [nLimbs] or null
    uint32_t logN, nLimbs, rows, srcStrideLimbs, srcLimbPos, srcCtxLimb;
    LimbSel sel;
};
FHE_GLOBAL void FHE_LAUNCH_BOUNDS(kThreads) switch_modulus_kernel(const SwitchModArgs g) {
    const uint32_t t          = FHE_TID;
    const uint64_t base       = (uint64_t)FHE_BID << kTileLog;
    const uint64_t totalWords = (uint64_t)g.rows << g.logN;
    const uint32_t mask       = (1u << g.logN) - 1u;
    const uint64_t qs         = g.q[g.srcCtxLimb];
    const uint64_t halfQ      = qs >> 1;
#pragma unroll 4
    for (int m = 0; m < 16; ++m) {
        const uint64_t off = base + (uint64_t)m * kThreads + t;
        if (off >= totalWords)
            continue;
        const uint32_t row = (uint32_t)(off >> g.logN);
        const uint32_t b = row / g.nLimbs, rit = row % g.nLimbs;
        const uint64_t qn = g.q[g.sel.idx[rit]];
        uint64_t v = g.src[(((uint64_t)b * g.srcStrideLimbs + g.srcLimbPos) << g.logN) + ((uint32_t)off & mask)];
        v = switch_modulus_word(v, qs, halfQ, qn);
        if (g.consts) {
            const TwPair c = g.consts[rit];
            v              = mul_shoup(v, c.w, c.wp, qn);
        }
        g.out[off] = v;
    }
}

// ---- DCRTPolyImpl::CRTDecompose (dcrtpoly-impl.h:230-285; the digit decomposition of KeySwitchBV): the nW digits of ONE source limb,
// each lifted (centred, PolyImpl::SwitchModulus) into every limb of its own tower; COEFFICIENT in, COEFFICIENT out (the caller
// transforms all towers in one launch).  Digit w of a word = its bits [w * baseBits, (w + 1) * baseBits) (PolyImpl::BaseDecompose,
// poly-impl.h:524-547 -> GetDigitAtIndexForBase, ubintnat.h:1721-1729); baseBits == 0: the word itself, one tower (:237-251).
struct CrtDigitsArgs {
    uint64_t* out;        // [nW][nLimbs][N]: the towers of this source limb
    const uint64_t* src;  // [N]: the source limb
    const uint64_t* q;    // [ctxLimbs]
    uint32_t logN, nLimbs, nW, baseBits, srcPos, srcCtxLimb;
    LimbSel sel;
};
FHE_GLOBAL void FHE_LAUNCH_BOUNDS(kThreads) crt_digits_kernel(const CrtDigitsArgs g) {
    const uint32_t t          = FHE_TID;
    const uint64_t base       = (uint64_t)FHE_BID << kTileLog;
    const uint64_t totalWords = ((uint64_t)g.nW * g.nLimbs) << g.logN;
    const uint32_t mask       = (1u << g.logN) - 1u;
    const uint64_t qs = g.q[g.srcCtxLimb], halfQ = qs >> 1;
    const uint64_t dmask = g.baseBits ? (((uint64_t)1 << g.baseBits) - 1u) : ~(uint64_t)0;
#pragma unroll 4
    for (int m = 0; m < 16; ++m) {
        const uint64_t off = base + (uint64_t)m * kThreads + t;
        if (off >= totalWords)
            continue;
        const uint32_t row = (uint32_t)(off >> g.logN);
        const uint32_t w = row / g.nLimbs, k = row % g.nLimbs;
        uint64_t v = (g.src[(uint32_t)off & mask] >> (w * g.baseBits)) & dmask;  // (w * baseBits < 64: checked by the host)
        if (k != g.srcPos)
            v = switch_modulus_word(v, qs, halfQ, g.q[g.sel.idx[k]]);
        g.out[off] = v;
    }
}

// ---- DCRTPolyImpl::SetValuesModSwitch (dcrtpoly-impl.h:630-647): one COEFFICIENT limb modulo qFrom scaled to a modulus qTo through
// double precision, out[j] = uint64(floor(0.5 + double(x[j]) * (double(qTo) / double(qFrom)))) mod qTo — the reference's expression,
// one rounding per operation (no contraction: the library is compiled with -ffp-contract=off)
struct ModSwitchRoundArgs {
    const uint64_t* x;
    uint64_t* out;
    double ratio;
    uint64_t qTo;
    uint64_t words;
};
FHE_GLOBAL void FHE_LAUNCH_BOUNDS(kThreads) mod_switch_round_kernel(const ModSwitchRoundArgs g) {
    const uint64_t i = (uint64_t)FHE_BID * kThreads + FHE_TID;
    if (i >= g.words)
        return;
    const double v = __builtin_floor(0.5 + (double)g.x[i] * g.ratio);
    g.out[i]       = (uint64_t)v % g.qTo;
}

// ---- whole-tower checksums ---------------------------------------------------------------------------------------------
// out[row] = { sum_i w_i, sum_i (2i + 1) * w_i } mod 2^64 over the row's N words w_0 .. w_{N-1}, for every limb-row of x[rows][N]: a
// parity check of EVERY tower of a resident batch costs one read of the batch (the oracle's words of the seed towers are summed on
// the host).  The second word is POSITION-DEPENDENT (odd weights): right residues in a wrong coefficient order — the classic layout
// failure of a transform — change it (round 3's second word was an xor, which they would not).
// One workgroup per 4096-word tile; the tile's partial results go to the row's two words with atomics (out is zeroed first).
struct ChecksumArgs {
    const uint64_t* x;
    uint64_t* out;  // [rows][2]
    uint32_t logN, rows;
};
FHE_GLOBAL void FHE_LAUNCH_BOUNDS(kThreads) checksum_kernel(const ChecksumArgs g) {
    const uint32_t t          = FHE_TID;
    const uint32_t tileLog    = g.logN < (uint32_t)kTileLog ? g.logN : (uint32_t)kTileLog;  // small rings: one workgroup per row
    const uint64_t base       = (uint64_t)FHE_BID << tileLog;
    const uint64_t totalWords = (uint64_t)g.rows << g.logN;
    uint64_t s = 0, x = 0;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const uint32_t in  = ((uint32_t)m * kThreads + t) << 1;
        const uint64_t off = base + in;
        if (in >= (1u << tileLog) || off >= totalWords)
            continue;
        const uint64_t a = g.x[off], b = g.x[off + 1];
        const uint64_t i = off & (((uint64_t)1 << g.logN) - 1u);  // index of `a` in its row
        s += a + b;
        x += a * (2u * i + 1u) + b * (2u * i + 3u);
    }
    FHE_SHARED_U64(red, 2 * kThreads);
    red[t]            = s;
    red[kThreads + t] = x;
    FHE_SYNC();
    for (uint32_t w = kThreads / 2; w >= 1; w >>= 1) {
        if (t < w) {
            red[t] += red[t + w];
            red[kThreads + t] += red[kThreads + t + w];
        }
        FHE_SYNC();
    }
    if (t == 0 && base < totalWords) {
        const uint64_t row = base >> g.logN;
#ifdef FHE_EMU
        g.out[2 * row] += red[0];
        g.out[2 * row + 1] += red[kThreads];
#else
        atomicAdd((unsigned long long*)&g.out[2 * row], (unsigned long long)red[0]);
        atomicAdd((unsigned long long*)&g.out[2 * row + 1], (unsigned long long)red[kThreads]);
#endif
    }
}


// rows [first, first + nRows) of every tower of out[batch][stride][N] set to zero (round 5).  hipMemset2DAsync runs this shape — 16 towers
// x 8 MiB, 8-row pitch gaps — at ~130 GB/s (1 ms per call in the lockstep bootstrap's census: 5 % of its kernel time for the P rows of
// KeySwitchExt); one 32 KiB tile per workgroup with 16-byte stores runs at copy speed.
struct ZeroRowsArgs {
    uint64_t* out;
    uint32_t logN, stride, first, nRows, batch;
};
FHE_GLOBAL void FHE_LAUNCH_BOUNDS(kThreads) zero_rows_kernel(const ZeroRowsArgs g) {
    const uint64_t words = ((uint64_t)g.batch * g.nRows) << g.logN;
    const uint64_t base  = (uint64_t)FHE_BID * kTile;
#pragma unroll
    for (int k = 0; k < 8; ++k) {  // 16 words per lane as 8 two-word pieces, 256 lanes side by side
        const uint64_t w = base + ((uint64_t)k * kThreads + FHE_TID) * 2;
        if (w >= words)
            continue;
        const uint64_t row = w >> g.logN, col = w & (((uint64_t)1 << g.logN) - 1u);
        const uint64_t tb = row / g.nRows, r = row % g.nRows;
        uint64_t* d = g.out + (((tb * g.stride + g.first + r)) << g.logN) + col;
        d[0] = 0;
        d[1] = 0;
    }
}

}  // namespace fhe
#endif
