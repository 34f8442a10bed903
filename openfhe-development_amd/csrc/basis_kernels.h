// basis_kernels.h — CRT basis conversion and the HYBRID key-switch inner product.
//
// approx_switch_kernel replaces DCRTPolyImpl::ApproxSwitchCRTBasis (fast path,
// src/core/include/lattice/hal/default/dcrtpoly-impl.h:888-915): per coefficient
//   y_i   = x_i * [Qhat_i^-1]_{q_i}            (Shoup, ModMulFastConst)
//   out_j = Barrett128( sum_i y_i * [Qhat_i]_{p_j} )   (128-bit accumulate, BarrettUint128ModUint64,
//                                                      utils/utilities-int.h:60-99)
// switch_exact_kernel adds the floating-point overflow count of SwitchCRTBasis (:1008-1085).
// ks_inner_product_kernel replaces KeySwitchHYBRID::EvalFastKeySwitchCoreExt
// (src/pke/lib/keyswitch/keyswitch-hybrid.cpp:402-435).
//
// Mapping: one coefficient per lane (consecutive lanes = consecutive coefficients => coalesced
// reads/writes of every limb), all source limbs of that coefficient are reduced to y_i in registers,
// output limbs are produced in chunks of OUT_CHUNK accumulators; the conversion constants are
// wave-uniform and come through the scalar cache.
#ifndef FHE_BASIS_KERNELS_H
#define FHE_BASIS_KERNELS_H
#include "modarith.h"
#include "launch.h"
#include "ntt_kernels.h"
#include "elemwise_kernels.h"
#include "ntt_static.h"  // FHE_PINNED_ASM, reduce192_uniform (generated gfx950 code)

namespace fhe {


struct ConvTables {           // device-resident, built once per (source basis, target basis)
    const TwPair* hatInv;     // [32]  [Qhat_i^-1]_{q_i} as Shoup pair (entries >= nSrc are padding)
    const uint64_t* hatMod;   // [nDst][NSRC]  [Qhat_i]_{p_j}: one target's row contiguous, padded to the kernel's NSRC
    const uint64_t* srcQ;     // [32]
    const uint64_t* dstQ;     // [nDst]
    const uint64_t* dstMu;    // [nDst][2]  floor(2^128/p_j) (lo,hi)
    const uint64_t* dstRed;   // [nDst][4]  {p_j, 2^64 mod p_j, its Shoup precon, floor(2^64/p_j)} for reduce192_uniform
    // exact variant only:
    const double* srcQInv;    // [32] 1.0/q_i
    const uint64_t* alphaMod; // [nSrc+1][nDst]  [alpha*Q]_{p_j}
};

struct ConvArgs {
    const uint64_t* in;   // [batch][inStride][N]; source limb i is row inFirst + i
    uint64_t* out;        // [batch][outStride][N]; target limb j is row outFirst + j
    ConvTables tb;
    uint32_t logN, batch;
    uint32_t nSrc, nDst;
    uint32_t inStride, inFirst, outStride, outFirst;
    // chunked plans (more than 32 source limbs: one launch per chunk of <= 32, CHUNK instantiation only)
    uint32_t acc;              // this chunk's sums are added to what the previous chunks left in `out`
    uint32_t last;             // exact variant: the last chunk subtracts [alpha*Q]_{p_j}, alpha counted over ALL source limbs
    uint32_t nSrcAll;          // source limbs of the whole plan
    const uint64_t* inAll;     // source limb 0 of the whole plan (same layout as `in`)
    const TwPair* allHatInv;   // [nSrcAll]
    const uint64_t* allSrcQ;   // [nSrcAll]
    const double* allQInv;     // [nSrcAll]
};

// NSRC = compile-time upper bound of the number of source limbs (8, 16 or 32): y_i live in registers, the output limbs
// are produced one at a time as column sums over chunks of <= 8 products with one 64-bit Barrett reduction each, so the
// kernel needs few registers (high occupancy) and computes every y_i once.
// SUM8 = 2 (default since round 2: EvalMult +2.8 %, BFV +1 % on MI355X, profiles/r02_sweeps.md): both factors split at 30
// bits, no carry bookkeeping at all (sum8s, modarith.h); SUM8 = 1 (FHE_CONV_SUM8=1): plain 64-bit columns whose low one
// counts its carries (sum8).  Both are exact; any exact reduction equals BarrettUint128ModUint64 (utilities-int.h:60-99).
// CHUNK: one chunk of a plan with more than 32 source limbs (the reference's loop dcrtpoly-impl.h:895-915 has no bound): the chunk's
// sums are exact residues, so out_j = sum over chunks mod p_j; the overflow count of the exact variant is accumulated over all
// source limbs in the reference's order (i ascending) by the last chunk, which recomputes every y_i for it.
template <int NSRC, bool EXACT, int SUM8 = 2, bool CHUNK = false>
FHE_GLOBAL void FHE_LAUNCH_BOUNDS(kThreads) switch_basis_kernel(const ConvArgs g) {
    const uint32_t N     = 1u << g.logN;
    const uint64_t gid   = (uint64_t)FHE_BID * kThreads + FHE_TID;  // over batch*N coefficients
    if (gid >= ((uint64_t)g.batch << g.logN))
        return;
    const uint32_t b  = (uint32_t)(gid >> g.logN);
    const uint32_t ri = (uint32_t)gid & (N - 1u);
    const uint64_t* in = g.in + (((uint64_t)b * g.inStride + g.inFirst) << g.logN) + ri;
    uint64_t* out      = g.out + (((uint64_t)b * g.outStride + g.outFirst) << g.logN) + ri;

    // Every table is padded to NSRC entries on the host, and the source residues are read with a clamped row index,
    // so that all loads of a phase are unconditional: the compiler issues them back to back (one wide s_load per
    // table row, NSRC global loads in flight) instead of one load + wait per uniform branch.
    uint64_t xin[NSRC];
#pragma unroll
    for (int i = 0; i < NSRC; ++i) {
        const uint32_t row = (uint32_t)i < g.nSrc ? (uint32_t)i : g.nSrc - 1u;
        xin[i]             = in[(uint64_t)row << g.logN];
    }
    uint64_t y[NSRC];
    // overflow count of the exact variant: nu = 0.5 + sum_i y_i/q_i in double, i ascending, one rounding per
    // multiply and per add (dcrtpoly-impl.h:1056-1063); compiled with -ffp-contract=off
    double nu = 0.5;
    const uint64_t* hp = reinterpret_cast<const uint64_t*>(g.tb.hatInv);
#pragma unroll
    for (int i = 0; i < NSRC; ++i) {
        const uint64_t w = FHE_ULOAD64(hp, 2 * i), wp = FHE_ULOAD64(hp, 2 * i + 1), qi = FHE_ULOAD64(g.tb.srcQ, i);
        const double qinv = EXACT ? FHE_ULOADF64(g.tb.srcQInv, i) : 0.0;
        y[i] = 0;
        if (i < (int)g.nSrc) {
            y[i] = mul_shoup(xin[i], w, wp, qi);
            if (EXACT)
                nu += (double)y[i] * qinv;
        }
    }
    if (CHUNK && EXACT && g.last) {
        nu = 0.5;
        const uint64_t* hpa = reinterpret_cast<const uint64_t*>(g.allHatInv);
        const uint64_t* ina = g.inAll + (((uint64_t)b * g.inStride) << g.logN) + ri;
        for (uint32_t i = 0; i < g.nSrcAll; ++i) {
            const uint64_t yi = mul_shoup(ina[(uint64_t)i << g.logN], hpa[2 * i], hpa[2 * i + 1], g.allSrcQ[i]);
            nu += (double)yi * g.allQInv[i];
        }
    }
    const uint32_t alpha = EXACT && (!CHUNK || g.last) ? (uint32_t)nu : 0u;

    for (uint32_t j = 0; j < g.nDst; ++j) {
        uint64_t h[NSRC];
#pragma unroll
        for (int i = 0; i < NSRC; ++i)
            h[i] = FHE_ULOAD64(g.tb.hatMod, (uint64_t)j * NSRC + i);
        const uint64_t p = FHE_ULOAD64(g.tb.dstQ, j);
        const uint64_t mulo = FHE_ULOAD64(g.tb.dstMu, 2 * j), muhi = FHE_ULOAD64(g.tb.dstMu, 2 * j + 1);
        if (SUM8 == 2) {
            const uint32_t k = 64u - (uint32_t)__builtin_clzll(p);
            uint64_t v       = 0;
#pragma unroll
            for (int c0 = 0; c0 < NSRC; c0 += 8) {
                if (c0 && c0 >= (int)g.nSrc)
                    break;
                sum8s s8;
                sum8s_clear(s8);
#pragma unroll
                for (int i = c0; i < c0 + 8 && i < NSRC; ++i) {
                    uint32_t y0, y1, h0, h1;
                    split30(y[i], y0, y1);  // (loop-invariant over j: hoisted by the compiler; y[i] = 0 beyond nSrc)
                    split30(h[i], h0, h1);  // wave-uniform: scalar unit
                    sum8s_add(s8, y0, y1, h0, h1);
                }
                const uint64_t rj = sum8s_reduce(s8, p, k, mulo, muhi);
                v                 = c0 ? add_mod(v, rj, p) : rj;
            }
            if (CHUNK && g.acc)
                v = add_mod(v, out[(uint64_t)j << g.logN], p);
            if (EXACT && (!CHUNK || g.last))
                v = sub_mod(v, g.tb.alphaMod[(uint64_t)alpha * g.nDst + j], p);
            out[(uint64_t)j << g.logN] = v;
            continue;
        }
        {
            static_assert(SUM8 == 1 || SUM8 == 2, "SUM8 selects one of the two column-sum forms");
            const uint32_t k = 64u - (uint32_t)__builtin_clzll(p);
            uint64_t v       = 0;
#pragma unroll
            for (int c0 = 0; c0 < NSRC; c0 += 8) {
                if (c0 && c0 >= (int)g.nSrc)
                    break;
                sum8 s8;
                sum8_clear(s8);
#pragma unroll
                for (int i = c0; i < c0 + 8 && i < NSRC; ++i)
                    sum8_add(s8, y[i], h[i]);  // y[i] = 0 beyond nSrc
                const uint64_t rj = sum8_reduce(s8, p, k, mulo, muhi);
                v                 = c0 ? add_mod(v, rj, p) : rj;
            }
            if (CHUNK && g.acc)
                v = add_mod(v, out[(uint64_t)j << g.logN], p);
            if (EXACT && (!CHUNK || g.last))
                v = sub_mod(v, g.tb.alphaMod[(uint64_t)alpha * g.nDst + j], p);
            out[(uint64_t)j << g.logN] = v;
        }
    }
}

// ---- HYBRID inner product ------------------------------------------------------------------------
// For every ciphertext b, output limb i in [0, sizeQl+sizeP), coefficient r:
//   out0 = sum_j digit_j[i] * keyB_j[idx(i)],  out1 = sum_j digit_j[i] * keyA_j[idx(i)],
//   idx(i) = i < sizeQl ? i : i + (sizeQ - sizeQl)          (keyswitch-hybrid.cpp:425)
// digit_j[i] is read from the ModUp output buffer of digit j, except for the digit's own limbs which are
// taken straight from the (EVALUATION-format) input c — the reference copies them (:371-372), we do not.
constexpr int kMaxDigits = 8;
struct KsInnerArgs {
    const uint64_t* c;                 // [batch][sizeQl][N] EVAL
    const uint64_t* digits[kMaxDigits];  // digit j complement: [batch][nc_j][N] EVAL
    const uint64_t* keyB;              // [numPartQ][sizeQ+sizeP][N]
    const uint64_t* keyA;
    uint64_t* out0;                    // [batch][sizeQl+sizeP][N]
    uint64_t* out1;
    const LimbConst* lc;               // [ctxLimbs]; ctx limbs: Q then P
    const uint64_t* mu128;             // [ctxLimbs][2]
    const uint64_t* red;               // [ctxLimbs] redM | redR << 32: the quotient estimate of the NTT kernels (fhe_ctx_create)
    uint32_t logN, batch, sizeQl, sizeQ, sizeP, numDigits, alpha;
    uint32_t nc[kMaxDigits];           // complement size of digit j = sizeQl - size_j + sizeP
    uint32_t j0, acc;                  // more than kMaxDigits digits: this launch covers digits j0 .. j0+numDigits-1 and, for j0 > 0,
                                       // adds its (exact) sums to what the previous launches left in out0 / out1
};
FHE_GLOBAL void FHE_LAUNCH_BOUNDS(kThreads) ks_inner_product_kernel(const KsInnerArgs g) {
    // one workgroup = 4096 consecutive coefficients of one (b, i) output row.  The key rows of a (limb, tile) group are
    // shared by the whole batch: workgroups go round-robin over the 8 XCDs, so the group's `batch` workgroups take
    // consecutive slots of ONE XCD and the key tile is fetched into one L2 once (not once per XCD)
    const uint32_t t          = FHE_TID;
    const uint32_t tilesPerRow = (1u << g.logN) >> kTileLog ? ((1u << g.logN) >> kTileLog) : 1u;
    const uint32_t sizeQlP    = g.sizeQl + g.sizeP;
    const uint32_t xcd = FHE_BID & 7u, slot = FHE_BID >> 3;
    const uint32_t b   = slot % g.batch;
    const uint32_t grp = (slot / g.batch) * 8u + xcd;
    const uint32_t tr = grp % tilesPerRow;
    const uint32_t i  = grp / tilesPerRow;
    if (i >= sizeQlP)
        return;
    const uint32_t idx  = i < g.sizeQl ? i : i + (g.sizeQ - g.sizeQl);
    const LimbConst lc  = g.lc[idx];
    const uint64_t mulo = g.mu128[2 * idx], muhi = g.mu128[2 * idx + 1];
    // The ModUp digits arrive in the NTT's lazy range (< 16q, ks_precompute_run).  One quotient estimate k = (d.hi * redM) >> (32 + redR)
    // (floor(d / q) or one less for any 64-bit d once q has 36 bits, fhe_ctx_create) and one conditional subtraction make them residues
    // (moduli below 36 bits: four conditional subtractions); after that the <= kMaxDigits = 8 products per sum fit the 64-bit column
    // sums with one 64-bit Barrett reduction (sum8, modarith.h) -- half the instructions of a 192-bit accumulator + 128-bit Barrett per
    // output, which is what makes this kernel HBM-bound.  (The generated MAC / reduction of the conversion kernel were tried here and
    // lost: 1.61 ms instead of 1.14 ms per launch at config 3's shape, the pinned temporaries cost occupancy.)
    const uint64_t redc = g.red[idx];
    const uint32_t redM = (uint32_t)redc, redR = (uint32_t)(redc >> 32);
    const uint32_t N    = 1u << g.logN;
    const uint32_t rEnd = ((tr + 1u) << kTileLog) < N ? ((tr + 1u) << kTileLog) : N;
    const uint64_t q = lc.q;
    for (uint32_t r = (tr << kTileLog) + t; r < rEnd; r += kThreads) {
        sum8 s0, s1;
        sum8_clear(s0);
        sum8_clear(s1);
        for (uint32_t j = 0; j < g.numDigits; ++j) {
            const uint32_t start = (g.j0 + j) * g.alpha;
            const uint32_t sz    = sizeQlP - g.nc[j];
            uint64_t d;
            if (i >= start && i < start + sz)
                d = g.c[(((uint64_t)b * g.sizeQl + i) << g.logN) + r];
            else {
                const uint32_t pos = i < start ? i : i - sz;
                d = g.digits[j][(((uint64_t)b * g.nc[j] + pos) << g.logN) + r];
            }
            if (redR != 255u)
                d -= (uint64_t)((uint32_t)(((d >> 32) * redM) >> 32) >> redR) * q;
            else {
                d = csub(d, q << 3);
                d = csub(d, q << 2);
                d = csub(d, q << 1);
            }
            d = csub(d, q);
            const uint64_t koff = (((uint64_t)(g.j0 + j) * (g.sizeQ + g.sizeP) + idx) << g.logN) + r;
            sum8_add(s0, d, g.keyB[koff]);
            sum8_add(s1, d, g.keyA[koff]);
        }
        const uint64_t ooff = (((uint64_t)b * sizeQlP + i) << g.logN) + r;
        uint64_t v0 = sum8_reduce(s0, q, lc.msb, mulo, muhi), v1 = sum8_reduce(s1, q, lc.msb, mulo, muhi);
        if (g.acc) {
            v0 = add_mod(v0, g.out0[ooff], q);
            v1 = add_mod(v1, g.out1[ooff], q);
        }
        g.out0[ooff] = v0;
        g.out1[ooff] = v1;
    }
}

// The same inner product of ONE digit decomposition with SEVERAL keys (the baby-step rotations of the BSGS linear transform:
// EvalFastRotationExt(ct, idx_j, digits, addFirst = true) for every j, ckksrns-fhe.cpp:1842-1850, ckksrns-leveledshe.cpp:534-582):
// a lane makes the digits' residues of its coefficient canonical ONCE and walks the keys, so the digits are read once (not once per
// rotation).  first != null: out0's Q_l rows get `+ first * firstC[i]` (cTilda[0] += c0 * [P]_{q_i}, ckksrns-leveledshe.cpp:561-570) in
// the same store.  Exact sums and an exact reduction: the words equal ks_inner_product_kernel's followed by that element-wise pass.
constexpr int kMaxMultiKeys = 16;
struct KsInnerMultiArgs {
    const uint64_t* c;                    // [batch][sizeQl][N] EVAL
    const uint64_t* digits[kMaxDigits];   // digit j complement: [batch][nc_j][N] EVAL, lazy range
    const uint64_t* keyB[kMaxMultiKeys];  // key t: [numPartQ][sizeQ+sizeP][N]
    const uint64_t* keyA[kMaxMultiKeys];
    uint64_t* out0[kMaxMultiKeys];        // result of key t: [batch][sizeQl+sizeP][N]
    uint64_t* out1[kMaxMultiKeys];
    const uint64_t* first;                // [batch][sizeQl][N] or null
    const TwPair* firstC;                 // [sizeQl]
    const LimbConst* lc;
    const uint64_t* mu128;
    const uint64_t* red;
    uint32_t logN, batch, sizeQl, sizeQ, sizeP, numDigits, alpha, nKeys;
    uint32_t nc[kMaxDigits];
};
// ND = compile-time bound of the number of digits (their residues live in registers); CPL = adjacent coefficients per lane (2: 16-byte
// accesses); PF: the next key's residues are loaded before the sums of this key are computed (software pipelining inside the wave)
struct alignas(16) U64x2 {
    uint64_t a, b;
};
template <int CPL>
FHE_HD void ld_cpl(const uint64_t* p, uint64_t (&v)[CPL]) {
    if constexpr (CPL == 2) {
        const U64x2 w = *reinterpret_cast<const U64x2*>(p);
        v[0] = w.a, v[1] = w.b;
    }
    else
        v[0] = p[0];
}
template <int CPL>
FHE_HD void st_cpl(uint64_t* p, const uint64_t (&v)[CPL]) {
    if constexpr (CPL == 2)
        *reinterpret_cast<U64x2*>(p) = U64x2{v[0], v[1]};
    else
        p[0] = v[0];
}
template <int ND, int CPL = 1, bool PF = true>
FHE_GLOBAL void FHE_LAUNCH_BOUNDS(kThreads) ks_inner_multi_kernel(const KsInnerMultiArgs g) {
    const uint32_t t           = FHE_TID;
    const uint32_t tilesPerRow = (1u << g.logN) >> kTileLog ? ((1u << g.logN) >> kTileLog) : 1u;
    const uint32_t sizeQlP     = g.sizeQl + g.sizeP;
    const uint32_t xcd = FHE_BID & 7u, slot = FHE_BID >> 3;  // (same XCD-aware order as ks_inner_product_kernel: key tiles shared by the batch)
    const uint32_t b   = slot % g.batch;
    const uint32_t grp = (slot / g.batch) * 8u + xcd;
    const uint32_t tr = grp % tilesPerRow;
    const uint32_t i  = grp / tilesPerRow;
    if (i >= sizeQlP)
        return;
    const uint32_t idx  = i < g.sizeQl ? i : i + (g.sizeQ - g.sizeQl);
    const LimbConst lc  = g.lc[idx];
    const uint64_t mulo = g.mu128[2 * idx], muhi = g.mu128[2 * idx + 1];
    const uint64_t redc = g.red[idx];
    const uint32_t redM = (uint32_t)redc, redR = (uint32_t)(redc >> 32);
    const bool addFirst = g.first != nullptr && i < g.sizeQl;
    TwPair fc{0, 0};
    if (addFirst)
        fc = g.firstC[i];
    const uint32_t N    = 1u << g.logN;
    const uint32_t rEnd = ((tr + 1u) << kTileLog) < N ? ((tr + 1u) << kTileLog) : N;
    const uint64_t q = lc.q;
    const uint64_t ooff0 = ((uint64_t)b * sizeQlP + i) << g.logN;
    // where digit j's residues of this row live (wave-uniform): the digit's own limbs in the input c, the others in its ModUp buffer
    const uint64_t* src[ND];
    uint64_t koffJ[ND];  // word offset of digit j's row inside a key element
#pragma unroll
    for (int j = 0; j < ND; ++j) {
        const uint32_t jj    = (uint32_t)j < g.numDigits ? (uint32_t)j : g.numDigits - 1u;  // (padding re-reads the last digit; its products are skipped)
        const uint32_t start = jj * g.alpha;
        const uint32_t sz    = sizeQlP - g.nc[jj];
        const bool own       = i >= start && i < start + sz;
        const uint32_t pos   = i < start ? i : i - sz;
        src[j] = own ? g.c + (((uint64_t)b * g.sizeQl + i) << g.logN) : g.digits[jj] + (((uint64_t)b * g.nc[jj] + pos) << g.logN);
        koffJ[j] = ((uint64_t)jj * (g.sizeQ + g.sizeP) + idx) << g.logN;
    }
    const uint64_t* firstRow = addFirst ? g.first + (((uint64_t)b * g.sizeQl + i) << g.logN) : nullptr;
    for (uint32_t r = (tr << kTileLog) + CPL * t; r < rEnd; r += CPL * kThreads) {
        uint64_t d[ND][CPL];
#pragma unroll
        for (int j = 0; j < ND; ++j)  // (all loads of the coefficient first, back to back: digits, c0, the first key's residues)
            ld_cpl<CPL>(src[j] + r, d[j]);
        uint64_t fraw[CPL] = {};
        if (addFirst)
            ld_cpl<CPL>(firstRow + r, fraw);
        uint64_t kb[ND][CPL], ka[ND][CPL], nb[ND][CPL], na[ND][CPL];
#pragma unroll
        for (int j = 0; j < ND; ++j) {
            ld_cpl<CPL>(g.keyB[0] + koffJ[j] + r, kb[j]);
            ld_cpl<CPL>(g.keyA[0] + koffJ[j] + r, ka[j]);
        }
#pragma unroll
        for (int j = 0; j < ND; ++j)
#pragma unroll
            for (int u = 0; u < CPL; ++u) {
                uint64_t v = d[j][u];
                if (redR != 255u)
                    v -= (uint64_t)((uint32_t)(((v >> 32) * redM) >> 32) >> redR) * q;
                else {
                    v = csub(v, q << 3);
                    v = csub(v, q << 2);
                    v = csub(v, q << 1);
                }
                d[j][u] = csub(v, q);
            }
        uint64_t f[CPL];
#pragma unroll
        for (int u = 0; u < CPL; ++u)
            f[u] = addFirst ? mul_shoup(fraw[u], fc.w, fc.wp, q) : 0;
        for (uint32_t kk = 0; kk < g.nKeys; ++kk) {
            if (PF) {
                if (kk + 1u < g.nKeys) {
                    const uint64_t* kB = g.keyB[kk + 1u];
                    const uint64_t* kA = g.keyA[kk + 1u];
#pragma unroll
                    for (int j = 0; j < ND; ++j) {
                        ld_cpl<CPL>(kB + koffJ[j] + r, nb[j]);
                        ld_cpl<CPL>(kA + koffJ[j] + r, na[j]);
                    }
                }
            }
            else if (kk) {
                const uint64_t* kB = g.keyB[kk];
                const uint64_t* kA = g.keyA[kk];
#pragma unroll
                for (int j = 0; j < ND; ++j) {
                    ld_cpl<CPL>(kB + koffJ[j] + r, kb[j]);
                    ld_cpl<CPL>(kA + koffJ[j] + r, ka[j]);
                }
            }
            uint64_t v0[CPL], v1[CPL];
#pragma unroll
            for (int u = 0; u < CPL; ++u) {
                sum8 s0, s1;
                sum8_clear(s0);
                sum8_clear(s1);
#pragma unroll
                for (int j = 0; j < ND; ++j)
                    if ((uint32_t)j < g.numDigits) {
                        sum8_add(s0, d[j][u], kb[j][u]);
                        sum8_add(s1, d[j][u], ka[j][u]);
                    }
                v0[u] = sum8_reduce(s0, q, lc.msb, mulo, muhi);
                v1[u] = sum8_reduce(s1, q, lc.msb, mulo, muhi);
                if (addFirst)
                    v0[u] = add_mod(v0[u], f[u], q);
            }
            if (PF && kk + 1u < g.nKeys) {
#pragma unroll
                for (int j = 0; j < ND; ++j)
#pragma unroll
                    for (int u = 0; u < CPL; ++u)
                        kb[j][u] = nb[j][u], ka[j][u] = na[j][u];
            }
            st_cpl<CPL>(g.out0[kk] + ooff0 + r, v0);
            st_cpl<CPL>(g.out1[kk] + ooff0 + r, v1);
        }
    }
}

// ---- the same inner product over towers that live in separate allocations (the HAL backend of DCRTPoly: every digit and
// every key element is its own tower) ----
//   out_e[b][i] = sum_t x_t[b][i] * k_{e,t}[keyRow[i]]   (e = 0, 1; t < nTerms <= 8; all operands canonical residues)
struct InnerRowsArgs {
    const uint64_t* x[kMaxDigits];   // [batch][rows][N]
    const uint64_t* k0[kMaxDigits];  // [keyRows][N]
    const uint64_t* k1[kMaxDigits];  // may all be null (one output)
    uint64_t* out0;                  // [batch][rows][N]
    uint64_t* out1;
    const LimbConst* lc;
    const uint64_t* mu128;
    uint32_t logN, batch, rows, nTerms;
    uint32_t acc;                    // more than 8 terms: later launches add their sums to out0 / out1
    uint8_t keyRow[kMaxLimbs];
    LimbSel sel;
};
FHE_GLOBAL void FHE_LAUNCH_BOUNDS(kThreads) inner_rows_kernel(const InnerRowsArgs g) {
    // one workgroup = one 4096-word tile of one output row; 2 words per thread and step (16-byte accesses)
    const uint32_t t           = FHE_TID;
    const uint32_t N           = 1u << g.logN;
    const uint32_t tilesPerRow = N >> kTileLog ? (N >> kTileLog) : 1u;
    const uint32_t tr          = FHE_BID % tilesPerRow;
    const uint32_t row         = FHE_BID / tilesPerRow;  // b * rows + i
    const uint32_t i           = row % g.rows;
    const uint32_t idx         = g.sel.idx[i];
    const LimbConst lc         = g.lc[idx];
    const uint64_t mulo = g.mu128[2 * idx], muhi = g.mu128[2 * idx + 1];
    const uint64_t q    = lc.q;
    const uint32_t rEnd = ((tr + 1u) << kTileLog) < N ? ((tr + 1u) << kTileLog) : N;
    const uint64_t xoff = (uint64_t)row << g.logN;
    const uint64_t koff = (uint64_t)g.keyRow[i] << g.logN;
    const bool two      = g.k1[0] != nullptr;
    for (uint32_t r = (tr << kTileLog) + 2u * t; r < rEnd; r += 2u * kThreads) {
        sum8 a0, a1, b0, b1;
        sum8_clear(a0);
        sum8_clear(a1);
        sum8_clear(b0);
        sum8_clear(b1);
        for (uint32_t j = 0; j < g.nTerms; ++j) {
            const uint64_t x0 = g.x[j][xoff + r], x1 = g.x[j][xoff + r + 1];
            sum8_add(a0, x0, g.k0[j][koff + r]);
            sum8_add(a1, x1, g.k0[j][koff + r + 1]);
            if (two) {
                sum8_add(b0, x0, g.k1[j][koff + r]);
                sum8_add(b1, x1, g.k1[j][koff + r + 1]);
            }
        }
        uint64_t v0 = sum8_reduce(a0, q, lc.msb, mulo, muhi), v1 = sum8_reduce(a1, q, lc.msb, mulo, muhi);
        if (g.acc) {
            v0 = add_mod(v0, g.out0[xoff + r], q);
            v1 = add_mod(v1, g.out0[xoff + r + 1], q);
        }
        g.out0[xoff + r]     = v0;
        g.out0[xoff + r + 1] = v1;
        if (two) {
            v0 = sum8_reduce(b0, q, lc.msb, mulo, muhi), v1 = sum8_reduce(b1, q, lc.msb, mulo, muhi);
            if (g.acc) {
                v0 = add_mod(v0, g.out1[xoff + r], q);
                v1 = add_mod(v1, g.out1[xoff + r + 1], q);
            }
            g.out1[xoff + r]     = v0;
            g.out1[xoff + r + 1] = v1;
        }
    }
}

// ---- baby-step/giant-step inner sums (double hoisting) --------------------------------------------
// inner_i[e][b][l][r] = sum_j rot_j[e][b][l][r] * diag_{i,j}[l][r]  over the extended basis Q_l u P, for ALL outer steps i
// in one pass: the EvalMultExt / EvalAddExtInPlace chains of FHECKKSRNS::EvalLinearTransform (ckksrns-fhe.cpp:1855-1859,
// 2723-2740).  A lane keeps the nIn rotated residues of its coefficient in registers and walks the outer steps, so every
// rotated ciphertext is read once (not once per outer step); each sum is kept in a 192-bit accumulator and reduced once
// (an exact reduction of the exact sum equals the reference's chain of ModMul / ModAdd results).
// Workgroup order is XCD-aware: the 2*batch workgroups that need the same plaintext rows (one limb tile, every element
// and ciphertext of the batch) get consecutive slots on ONE XCD, so the diagonals are fetched once per XCD L2.
constexpr int kMaxBsgsIn = 16;  // inner rotations per launch (more are accumulated by further launches)
struct BsgsInnerArgs {
    const uint64_t* rot;          // [nIn][2][batch][sizeQl+sizeP][N] EVAL, canonical (already offset to the chunk's first rotation);
                                  // rot_j is stored BEFORE its automorphism: the kernel reads it through the index map of k[j]
    const uint64_t* const* diag;  // DEVICE table [nOut][nInPad]: plaintext rows [sizeQl+sizeP][N] EVAL; absent terms and the
                                  // padding up to a multiple of the kernel's NIN point at rows of zeros
    uint64_t* out;                // [2][nOut][batch][sizeQl+sizeP][N]
    const LimbConst* lc;          // [ctxLimbs]; ctx limbs: Q then P
    const uint64_t* mu128;        // [ctxLimbs][2]
    uint32_t logN, batch, sizeQl, sizeQ, sizeP, nIn, nInPad, j0, nOut, accumulate;
    uint32_t k[kMaxBsgsIn];       // automorphism index of rotation j0 + j (1: none): the last step of EvalFastRotationExt
                                  // (AutomorphismTransform, ckksrns-leveledshe.cpp:572-579) is this kernel's gather, out[r] = in[map_k(r)]
};
// CPL = coefficients per lane (adjacent: CPL = 2 gives 16-byte accesses); the plaintext residues of outer step i+1 are
// loaded before the sums of step i are computed (software pipelining: the loads of a wave overlap its own arithmetic).
template <int NIN, int CPL>
FHE_GLOBAL void FHE_LAUNCH_BOUNDS(kThreads) bsgs_inner_kernel(const BsgsInnerArgs g) {
    const uint32_t t           = FHE_TID;
    const uint32_t N           = 1u << g.logN;
    const uint32_t tilesPerRow = (N >> kTileLog) ? (N >> kTileLog) : 1u;
    const uint32_t sizeQlP     = g.sizeQl + g.sizeP;
    const uint32_t nGroups     = sizeQlP * tilesPerRow, per = 2u * g.batch;
    const uint32_t xcd = FHE_BID & 7u, slot = FHE_BID >> 3;
    const uint32_t grp = (slot / per) * 8u + xcd, local = slot % per;
    if (grp >= nGroups)
        return;
    const uint32_t e = local / g.batch, b = local % g.batch;
    const uint32_t l = grp / tilesPerRow, tr = grp % tilesPerRow;
    const uint32_t idx  = l < g.sizeQl ? l : l + (g.sizeQ - g.sizeQl);
    const LimbConst lc  = g.lc[idx];
    const uint64_t mulo = g.mu128[2 * idx], muhi = g.mu128[2 * idx + 1];
    const uint64_t rotStride = ((uint64_t)2 * g.batch * sizeQlP) << g.logN;  // words between rot_j and rot_{j+1}
    const uint64_t rotOff    = (((uint64_t)e * g.batch + b) * sizeQlP + l) << g.logN;
    const uint64_t outStride = ((uint64_t)g.batch * sizeQlP) << g.logN;      // words between inner_i and inner_{i+1}
    const uint64_t outOff    = ((((uint64_t)e * g.nOut) * g.batch + b) * sizeQlP + l) << g.logN;
    const uint32_t rEnd      = ((tr + 1u) << kTileLog) < N ? ((tr + 1u) << kTileLog) : N;
    for (uint32_t r = (tr << kTileLog) + CPL * t; r < rEnd; r += CPL * kThreads) {
        // all loads unconditional (absent terms point at a row of zeros, missing rotations re-read the last one and meet
        // a zero row too): the compiler issues the NIN loads of a phase back to back
        uint64_t x[NIN][CPL], y[NIN][CPL], yn[NIN][CPL];
        uint32_t jb[CPL];
#pragma unroll
        for (int u = 0; u < CPL; ++u)
            jb[u] = bitrev32(r + (uint32_t)u, g.logN);
#pragma unroll
        for (int j = 0; j < NIN; ++j) {
            const uint32_t jj  = (uint32_t)j < g.nIn ? (uint32_t)j : g.nIn - 1u;
            const uint64_t* xp = g.rot + ((uint64_t)jj * rotStride + rotOff);  // uniform base + 32-bit lane offset
            const uint32_t kj  = g.k[jj];
#pragma unroll
            for (int u = 0; u < CPL; ++u)  // (the map permutes inside aligned blocks: a wave reads exactly one 512-byte segment)
                x[j][u] = xp[automorph_source(jb[u], kj, g.logN)];
        }
        const uint32_t lr = (l << g.logN) + r;  // word offset inside a plaintext: below 2^23
        {
            const uint64_t* const* drow = g.diag + g.j0;
#pragma unroll
            for (int j = 0; j < NIN; ++j)
                ld_cpl<CPL>(drow[j] + lr, y[j]);
        }
        for (uint32_t i = 0; i < g.nOut; ++i) {
            {  // next outer step's plaintext residues (the last step re-reads its own: no branch around the loads)
                const uint32_t in = i + 1u < g.nOut ? i + 1u : i;
                const uint64_t* const* drow = g.diag + (uint64_t)in * g.nInPad + g.j0;
#pragma unroll
                for (int j = 0; j < NIN; ++j)
                    ld_cpl<CPL>(drow[j] + lr, yn[j]);
            }
            uint64_t v[CPL] = {};
#pragma unroll
            for (int j0 = 0; j0 < NIN; j0 += 8) {  // one exact reduction per 8 terms (sum8)
#pragma unroll
                for (int u = 0; u < CPL; ++u) {
                    sum8 s;
                    sum8_clear(s);
#pragma unroll
                    for (int j = j0; j < j0 + 8 && j < NIN; ++j)
                        sum8_add(s, x[j][u], y[j][u]);
                    const uint64_t rj = sum8_reduce(s, lc.q, lc.msb, mulo, muhi);
                    v[u]              = j0 ? add_mod(v[u], rj, lc.q) : rj;
                }
            }
            uint64_t* op = g.out + (outOff + (uint64_t)i * outStride) + r;
            if (g.accumulate) {
                uint64_t ov[CPL];
                ld_cpl<CPL>(op, ov);
#pragma unroll
                for (int u = 0; u < CPL; ++u)
                    v[u] = add_mod(ov[u], v[u], lc.q);
            }
            st_cpl<CPL>(op, v);
#pragma unroll
            for (int j = 0; j < NIN; ++j)
#pragma unroll
                for (int u = 0; u < CPL; ++u)
                    y[j][u] = yn[j][u];
        }
    }
}

}  // namespace fhe
#endif
