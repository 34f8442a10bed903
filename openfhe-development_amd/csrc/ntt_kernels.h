// ntt_kernels.h — batched negacyclic NTT over RNS towers for gfx950 (MI355X).
//
// Replaces NumberTheoreticTransformNat::ForwardTransformToBitReverseInPlace /
// InverseTransformFromBitReverseInPlace (src/core/include/math/hal/intnat/transformnat-impl.h:303-374,
// 512-625) as dispatched per limb by DCRTPolyImpl::SwitchFormat (dcrtpoly-impl.h:1932-1940).
// Same transform, same twiddle tables (Table[bitrev(i)] = psi^i, :714-756), same output order
// (forward: natural -> bit-reversed, inverse: bit-reversed -> natural with 1/N folded into the last
// stage), outputs are canonical residues in [0,q)  => bit-identical results.
//
// Design (MI355X-first, not the reference's loop nest):
//  * one 64-bit residue per lane, one "tile" of 4096 residues (32 KiB) per 256-thread workgroup,
//    16 residues per lane held in VGPRs;
//  * a transform of 2^logN points is split into one pass (logN <= 12: the whole limb lives in one
//    tile) or two passes over HBM (a strided "column" pass of T1 stages and a contiguous "row" pass
//    of T2 stages, logN = T1 + T2), i.e. 2 reads + 2 writes of the data per transform;
//  * inside a pass the stages are executed as register-resident radix-16 steps (4 butterfly stages on
//    the 16 values of a lane) with an XOR-swizzled LDS exchange between steps;
//  * Harvey lazy butterflies: values live in [0,4q) (forward) / [0,2q) (inverse), one conditional
//    subtraction per butterfly, canonicalised only when the last stage stores to HBM.  Between the two
//    passes the intermediate tower in HBM is in the lazy range (never visible to callers);
//  * twiddles are (w, floor(w*2^64/q)) pairs fetched as one 16-byte load; the first step of the
//    column pass uses wave-uniform twiddles (scalar loads), the row pass relies on L2 reuse across the
//    batch (work is ordered batch-fastest per XCD).
//
// The file is also compiled by the test-only lane emulator (tests/emu/), hence the FHE_* macros.
#ifndef FHE_NTT_KERNELS_H
#define FHE_NTT_KERNELS_H
#include "modarith.h"
#include "launch.h"

namespace fhe {

constexpr int kTileLog  = 12;
constexpr int kTile     = 1 << kTileLog;  // residues per workgroup tile
constexpr int kThreads  = 256;            // 4 waves
constexpr int kMaxLimbs = 256;  // rows of one tower / limbs of one context (the limb map travels by value: one byte per row)

struct alignas(16) TwPair {
    uint64_t w, wp;  // wp = floor(w * 2^64 / q)   (PrepModMulConst, ubintnat.h:1437-1444)
};

struct LimbSel {
    uint8_t idx[kMaxLimbs];  // tower row -> context limb
};

struct NttStep {
    int8_t fI;   // position of the 4-bit register field inside the 12-bit tile index
    int8_t Fj;   // position of the same field inside the coefficient index j
    int8_t bHi;  // butterfly stages act on field bits bHi..bLo (bHi < bLo: data movement only)
    int8_t bLo;
    // fast kernel only (forward transform): lazy-reduction schedule of this step
    //   0: no correction (bounds still fit)      1: all 16 values are brought back below 8q before the step
    uint8_t mode;
    uint8_t levels;
    uint8_t uniformTw;  // 1: the twiddle index of this step does not depend on the lane (scalar loads)
    uint8_t pad;
};

struct NttPassArgs {
    const uint64_t* xin;  // source of the pass's first load ([batch][inStride][N] view, see inStride)
    uint64_t* x;          // [rows][N] destination (and source when xin == x)
    const TwPair* tw;     // [ctxLimbs][N], forward or inverse table
    const TwPair* twRow;  // [ctxLimbs][N/4096][15][256] lane-major copy for the row pass's last/first step (ntt_static.h), or null
    const uint64_t* q;    // [ctxLimbs]
    const uint64_t* red;  // [ctxLimbs] quotient-estimate constants of the static kernels: redM | redR << 32 (ntt_static.h)
    const TwPair* fin;    // inverse only: [ctxLimbs][2] = {N^-1, Table_inv[1]*N^-1}
    uint32_t logN;
    uint32_t T;           // stages in this pass (tile = 2^T points x 2^(12-T) transforms)
    uint32_t nLimbs;      // limbs per tower (row % nLimbs selects sel.idx[])
    uint32_t rows;        // batch * nLimbs
    uint32_t batch;
    uint32_t nSteps;
    uint32_t canonLevels; // fast forward kernel: outputs of the canon step are < 2^canonLevels * q
    uint32_t canonStep;   // index of the step after whose stages values are canonicalised to [0,q); >= nSteps: never
    uint32_t xcdSwizzle;  // 1: remap blockIdx so that an XCD keeps one (limb, tile) pair across the batch
    uint32_t inStride;    // 0: xin is dense like x; else towers of xin are inStride rows apart and the
    uint32_t inFirst;     //    transformed rows start at row inFirst of each tower
    uint32_t outStride;   // 0: x is dense; else x is a [batch][outStride][N] view, rows outFirst.. of each tower
    uint32_t outFirst;    //    (applies to every access of a.x, i.e. stores and in-place reloads)
    NttStep steps[6];
    LimbSel sel;
    // Optional epilogue of the pass that stores the transform's result (static forward kernels only): instead of the
    // plain store, out = (A - r) * C [+ out]  — ApproxModDown's last line (dcrtpoly-impl.h:1002) fused with EvalMult's
    // `+=` (base-leveledshe.cpp:210-211), so that the converted tower never goes to HBM.  epiMode 0: off,
    // 1: out = (A - r) * C, 2: out += (A - r) * C.  Towers [0, epiSplit) go to epiOut0, the rest to epiOut1 (dense
    // [towers][nLimbs][N]); A is a [towers][epiAStride][N] view (rows epiAFirst..), C one Shoup pair per tower row.
    uint32_t epiMode, epiSplit, epiAStride, epiAFirst;
    const uint64_t* epiA;
    const TwPair* epiC;
    uint64_t *epiOut0, *epiOut1;
    // Optional prologue of a forward transform's first pass (static column kernels only, PRO instances): every limb of a tower
    // is loaded from ONE row of xin (row inFirst of the tower's inStride rows), a COEFFICIENT limb modulo q[proSrcLimb], and
    // brought to the limb's own modulus on the way in (SwitchModulus, mubintvecnat.cpp:109-122) — the `tmp[i] = lastPoly;
    // tmp[i].SwitchModulus(q_i)` of DropLastElementAndScale (dcrtpoly-impl.h:703-704) never goes to HBM.  0: off.
    uint32_t proMode, proSrcLimb;
    // != 0: consecutive towers of the pass's first load / of the epilogue's operand A are this many WORDS apart (signed: towers
    // allocated on their own — the two elements of a ciphertext); overrides inStride / epiAStride (static kernels only)
    int64_t inDelta, epiADelta;
};
// SwitchModulus of one residue (mubintvecnat.cpp:109-122): v modulo qs, centred, to the modulus qn
FHE_HD uint64_t switch_modulus_word(uint64_t v, uint64_t qs, uint64_t halfQs, uint64_t qn) {
    if (qn > qs)
        return v + ((v > halfQs) ? (qn - qs) : 0);
    // ModSubEq semantics (ubintnat.h:889-899): operands reduced mod qn first
    uint64_t bv = (v > halfQs) ? (qs - qn) : 0;
    uint64_t av = v;
    if (av >= qn)
        av %= qn;
    if (bv >= qn)
        bv %= qn;
    return (av < bv) ? av + qn - bv : av - bv;
}

// LDS word index swizzle: conflict-free ds_read_b64/ds_write_b64 for every register-field position
// (sigma is GF(2)-linear: sigma(a ^ b) = sigma(a) ^ sigma(b))
FHE_HD uint32_t lds_sigma(uint32_t I) {
    return I ^ ((I >> 4) & 31u);
}

// ---- butterflies -------------------------------------------------------------------------------
// forward (Cooley-Tukey), lazy: inputs in [0,4q) -> outputs in [0,4q)
FHE_HD void bfly_fwd(uint64_t& a, uint64_t& b, const TwPair w, uint64_t nq, uint64_t twoq) {
    uint64_t X = csub(a, twoq);
    uint64_t T = mul_shoup_lazy_nq(b, w.w, w.wp, nq);
    a          = X + T;
    b          = X - T + twoq;
}
// inverse (Gentleman-Sande), lazy: inputs in [0,2q) -> outputs in [0,2q)
FHE_HD void bfly_inv(uint64_t& a, uint64_t& b, const TwPair w, uint64_t nq, uint64_t twoq) {
    uint64_t u = a, v = b;
    a          = csub(u + v, twoq);
    b          = mul_shoup_lazy_nq(u - v + twoq, w.w, w.wp, nq);
}
// last inverse stage: lower output *N^-1, upper output *(w1*N^-1)  (transformnat-impl.h:598-624)
FHE_HD void bfly_inv_last(uint64_t& a, uint64_t& b, const TwPair nInv, const TwPair w1nInv, uint64_t nq,
                          uint64_t twoq) {
    uint64_t u = a, v = b;
    a          = mul_shoup_lazy_nq(u + v, nInv.w, nInv.wp, nq);
    b          = mul_shoup_lazy_nq(u - v + twoq, w1nInv.w, w1nInv.wp, nq);
}

// ---- the pass kernel ---------------------------------------------------------------------------
// LAYOUT_A: strided "column" pass: tile index I = p*C + c, coefficient j = p*(N>>T) + cb*C + c
// else     : contiguous "row" pass: tile = 4096 consecutive words of the [rows][N] array
template <bool LAYOUT_A, bool INVERSE>
FHE_GLOBAL void FHE_LAUNCH_BOUNDS(kThreads) ntt_pass_kernel(const NttPassArgs a) {
    FHE_SHARED_U64(lds, kTile);

    const uint32_t t    = FHE_TID;
    const uint32_t logN = a.logN;
    const uint32_t N    = 1u << logN;
    const uint32_t T    = a.T;

    // ---- which tile? ----
    uint32_t tile = FHE_BID;
    const uint32_t tilesPerRow = (N >= (uint32_t)kTile) ? (N >> kTileLog) : 1u;
    if (a.xcdSwizzle) {
        // blockIdx = xcd + 8*i ; i = pairIdx*batch + b ; pair = pairIdx*8 + xcd ; pair = limb*tilesPerRow + tr
        const uint32_t xcd = tile & 7u, i = tile >> 3;
        const uint32_t b = i % a.batch, pairIdx = i / a.batch;
        const uint32_t pair = pairIdx * 8u + xcd;
        const uint32_t limb = pair / tilesPerRow, tr = pair % tilesPerRow;
        tile = (b * a.nLimbs + limb) * tilesPerRow + tr;
    }

    // ---- tile -> memory geometry ----
    const uint32_t logC = kTileLog - T;          // transforms per tile
    const uint32_t S    = N >> T;                // LAYOUT_A: row stride of the point index (columns)
    uint64_t gbase;                              // word offset of tile index 0
    uint32_t jbase = 0;                          // LAYOUT_A: column offset inside the row
    uint32_t rowA  = 0;
    if (LAYOUT_A) {
        rowA  = tile / tilesPerRow;
        jbase = (tile % tilesPerRow) << logC;
        gbase = (uint64_t)rowA << logN;
    }
    else {
        gbase = (uint64_t)tile << kTileLog;
    }
    const uint64_t totalWords = (uint64_t)a.rows << logN;

    uint64_t r[16];
    uint64_t q = 0, twoq = 0, nq = 0;
    const TwPair* tw = nullptr;
    uint32_t limb    = 0;

    for (uint32_t si = 0; si < a.nSteps; ++si) {
        const NttStep st    = a.steps[si];
        const uint32_t fI   = (uint32_t)st.fI;
        const uint32_t Ib   = ((t >> fI) << (fI + 4)) | (t & ((1u << fI) - 1u));
        // word offset (within the whole array) of register k:  off(k) = off0 + k*kstride
        uint64_t off0;
        uint64_t kstride;
        uint32_t j0;  // coefficient index of register 0
        if (LAYOUT_A) {
            const uint32_t p0 = Ib >> logC, c0 = Ib & ((1u << logC) - 1u);
            j0      = p0 * S + jbase + c0;
            off0    = gbase + j0;
            kstride = (fI >= logC) ? ((uint64_t)S << (fI - logC)) : ((uint64_t)1 << fI);
        }
        else {
            off0    = gbase + Ib;
            j0      = (uint32_t)(off0 & (N - 1u));
            kstride = (uint64_t)1 << fI;
        }
        const bool inRange = LAYOUT_A ? true : (off0 < totalWords);

        const uint32_t row = LAYOUT_A ? rowA : (uint32_t)(off0 >> logN);
        if (si == 0 || (!LAYOUT_A && logN < (uint32_t)kTileLog)) {
            // the limb is uniform per workgroup when N >= 4096; smaller rings pack several limbs into one
            // tile and the lane's limb follows the step's register mapping
            limb = a.sel.idx[(inRange ? row : 0u) % a.nLimbs];
            q    = a.q[limb];
            twoq = q << 1;
            nq   = 0 - q;
            tw   = a.tw + ((uint64_t)limb << logN);
        }
        if (si == 0) {
            if (a.inStride) {
                // strided source view: map every element's dense row to its row in the [batch][inStride][N] view
                // (with N < 4096 the 16 values of a lane can lie in different rows)
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const uint64_t d = off0 + k * kstride;
                    if (LAYOUT_A || d < totalWords) {
                        const uint32_t rr = (uint32_t)(d >> logN);
                        r[k] = a.xin[((((uint64_t)(rr / a.nLimbs) * a.inStride + a.inFirst + rr % a.nLimbs)) << logN) +
                                     (d & (N - 1u))];
                    }
                    else
                        r[k] = 0;
                }
            }
            else {
#pragma unroll
                for (int k = 0; k < 16; ++k)
                    r[k] = (LAYOUT_A || off0 + k * kstride < totalWords) ? a.xin[off0 + k * kstride] : 0;
            }
        }
        else {
            const uint32_t sb = lds_sigma(Ib);
#pragma unroll
            for (int k = 0; k < 16; ++k)
                r[k] = lds[sb ^ lds_sigma((uint32_t)k << fI)];
            FHE_SYNC();  // everyone has read before anyone overwrites
        }

        if (st.bHi >= st.bLo) {
            const uint32_t Fj    = (uint32_t)st.Fj;
            const uint32_t jhigh = j0 >> (Fj + 4);
            if (!INVERSE) {
#pragma unroll
                for (int b = 3; b >= 0; --b) {
                    if (b <= st.bHi && b >= st.bLo) {
                        const uint32_t s      = logN - 1u - (Fj + b);
                        const uint32_t twbase = (1u << s) + (jhigh << (3 - b));
#pragma unroll
                        for (int g = 0; g < (8 >> b); ++g) {
                            const TwPair w = tw[twbase + g];
#pragma unroll
                            for (int lo = 0; lo < (1 << b); ++lo) {
                                const int k0 = (g << (b + 1)) | lo;
                                bfly_fwd(r[k0], r[k0 | (1 << b)], w, nq, twoq);
                            }
                        }
                    }
                }
            }
            else {
#pragma unroll
                for (int b = 0; b <= 3; ++b) {
                    if (b <= st.bHi && b >= st.bLo) {
                        const uint32_t s      = logN - 1u - (Fj + b);
                        const uint32_t twbase = (1u << s) + (jhigh << (3 - b));
                        if (s == 0) {
                            const TwPair nInv = a.fin[2 * limb], w1n = a.fin[2 * limb + 1];
#pragma unroll
                            for (int lo = 0; lo < (1 << b); ++lo)  // b == 3 here, g == 0
                                bfly_inv_last(r[lo], r[lo | (1 << b)], nInv, w1n, nq, twoq);
                        }
                        else {
#pragma unroll
                            for (int g = 0; g < (8 >> b); ++g) {
                                const TwPair w = tw[twbase + g];
#pragma unroll
                                for (int lo = 0; lo < (1 << b); ++lo) {
                                    const int k0 = (g << (b + 1)) | lo;
                                    bfly_inv(r[k0], r[k0 | (1 << b)], w, nq, twoq);
                                }
                            }
                        }
                    }
                }
            }
        }

        if (si == a.canonStep) {
#pragma unroll
            for (int k = 0; k < 16; ++k)
                r[k] = INVERSE ? csub(r[k], q) : csub(csub(r[k], twoq), q);
        }
        if (si + 1 == a.nSteps) {
            if (a.outStride) {
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const uint64_t d = off0 + k * kstride;
                    if (LAYOUT_A || d < totalWords) {
                        const uint32_t rr = (uint32_t)(d >> logN);
                        a.x[((((uint64_t)(rr / a.nLimbs) * a.outStride + a.outFirst + rr % a.nLimbs)) << logN) + (d & (N - 1u))] = r[k];
                    }
                }
            }
            else {
#pragma unroll
                for (int k = 0; k < 16; ++k)
                    if (LAYOUT_A || off0 + k * kstride < totalWords)
                        a.x[off0 + k * kstride] = r[k];
            }
        }
        else {
            const uint32_t sb = lds_sigma(Ib);
#pragma unroll
            for (int k = 0; k < 16; ++k)
                lds[sb ^ lds_sigma((uint32_t)k << fI)] = r[k];
            FHE_SYNC();
        }
    }
}


// ================================================================================================
// Arithmetic shared by the kernels for rings with N >= 4096 (ntt_pass_full_kernel below, ntt_static.h):
//  * a 10-multiply Shoup butterfly with the x+T sum folded into the multiply-add chain (bfly_fwd_fast),
//  * forward transform: conditional subtractions only where the 64-bit headroom (16q) would otherwise
//    overflow (NttStep::mode), instead of one per butterfly.
// ================================================================================================

// hi64(y * wp): 4 multiply-adds
FHE_HD uint64_t mulhi64_mad(uint32_t yl, uint32_t yh, uint64_t wp) {
    const uint32_t pl = (uint32_t)wp, ph = (uint32_t)(wp >> 32);
    const uint64_t p0 = mul32x32(yl, pl);
    const uint64_t p1 = mad64(yh, pl, p0 >> 32);
    const uint64_t p2 = mad64(yl, ph, (uint32_t)p1);
    return mad64(yh, ph, p1 >> 32) + (p2 >> 32);
}
// returns (x + y*w - floor(y*wp/2^64)*q) mod 2^64  ==  x + T,  T in [0,2q)
FHE_HD uint64_t shoup_acc(uint64_t x, uint64_t y, const TwPair w, uint64_t nq) {
    const uint32_t yl = (uint32_t)y, yh = (uint32_t)(y >> 32);
    const uint64_t Q  = mulhi64_mad(yl, yh, w.wp);
    const uint32_t Ql = (uint32_t)Q, Qh = (uint32_t)(Q >> 32), wl = (uint32_t)w.w, wh = (uint32_t)(w.w >> 32);
    const uint32_t nql = (uint32_t)nq, nqh = (uint32_t)(nq >> 32);
    uint64_t C = mul32x32(yl, wh);  // cross terms: only their low 32 bits matter
    C          = mad64(yh, wl, C);
    C          = mad64(Ql, nqh, C);
    C          = mad64(Qh, nql, C);
    uint64_t L = mad64(yl, wl, x);  // low product accumulated straight onto x
    L          = mad64(Ql, nql, L);
    const uint32_t hi = (uint32_t)(L >> 32) + (uint32_t)C;
    return ((uint64_t)hi << 32) | (uint32_t)L;
}
// x in [0, 2m) -> [0, m) written as subtract-then-select (2 + 2 instructions)
FHE_HD uint64_t csub2(uint64_t x, uint64_t m) {
    const uint64_t d = x - m;
    return x < m ? x : d;
}
// forward butterfly of the static kernels' plain C++ path (host / emulator build, and the device build under
// FHE_NO_BFLY_ASM); the product kernels run the generated in-place gfx950 code of ntt_bfly_pinned.h instead
FHE_HD void bfly_fwd_fast(uint64_t& a, uint64_t& b, const TwPair w, uint64_t nq, uint64_t twoq) {
    const uint64_t X  = a;
    const uint64_t an = shoup_acc(X, b, w, nq);  // X + T
    b                 = (X << 1) + twoq - an;    // X - T + 2q
    a                 = an;
}

}  // namespace fhe
#endif
