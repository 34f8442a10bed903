// ntt_row8.h — the row pass of a two-pass transform at EIGHT residues per lane, tiles of 1 / 2 / 4 / 8 waves (round 6).
//
// Why: the 16-residues-per-lane row pass (ntt_static.h) needs 120-124 VGPRs and 38 KiB of LDS per 4096-word tile: four tiles per CU,
// six workgroup barriers per tile; 8.2-8.7 ms per pass against 5.5 ms of HBM time.  What round 6 measured (profiles/r06_sweeps.md):
//   * waves per SIMD are NOT the lever: this kernel at 512 threads per 4096-word tile (64 VGPRs, 8 waves per SIMD, ONE barrier per
//     tile) runs as fast as the 16-residue kernel, to the percent; a persistent, software-pipelined form of it was 10 % slower;
//   * a row pass takes  HBM time + half of its compute time  (12 / 11 / 10 / 9 stages: 8.34 / 8.23 / 8.00 / 7.60 ms, compute alone
//     5.95): what overlaps is bounded by the number of INDEPENDENT tiles a CU holds — a workgroup loads, computes and stores as one
//     customer, and four customers keep two equally loaded stations 70 % busy;
//   * so the tile shrinks instead: 2^(9+WB) words on 2^WB waves (WB = 0..3), eight residues per lane, <= 64 VGPRs and 4.5 KiB of LDS
//     per wave: 32 / 2^WB tiles per CU (WB = 2: eight 2048-word tiles of 256 threads).  The row pass has 9 + WB stages, the column
//     pass the rest (2^16: 5 + 11 — the column access pattern with the best HBM rate, profiles/r06_stridebench.json).
// Structure (unchanged from the first form): the tile index is  w:c:k:m  (WB, 3, 3, 3 bits).  Radix-8 steps act on the fields c (tile
// bits 6-8), k (3-5), m (0-2) and on w together with the top 3 - WB bits of c (one register field of 3 bits whose top WB bits carry
// butterflies).  Only that step mixes the waves: from then on WAVE w OWNS the 512 words whose top field is w, and the exchanges
// between the steps on c, k and m are private to a wave — LDS instructions of one wave execute in order, so they need NO workgroup
// barrier.  A tile meets ONE barrier (none at WB = 0).  Every exchange has its own additive address scheme
// F1(field1) + F2(field2) + F3(field3) inside the wave's region of 576 words, chosen so that both its write and its read pattern put
// the 32 lanes of a half-wave on 32 distinct 8-byte bank pairs (tools/occbench.hip, profiles/r06_occbench.json: 296 ns per 64 KiB
// exchange and CU with the skews, 351-438 without); register indices enter as immediates (`base + imm`), as in ntt_static.h.  HBM
// sees 512 contiguous bytes per wave instruction on both sides.
// Registers (ntt_bfly8_pinned.h): residues v[48:63], the one butterfly slot's temporaries v[36:47], v[0:35] for addresses and the
// per-lane twiddle pairs of a radix-8 step (at most 6 of its 7 at a time).
// Same transform as transformnat-impl.h:303-374 (forward, stages on tile bits T-1..0) / 512-625 (inverse, 0..T-1); twiddle of the
// stage on coefficient bit P for the pair (J, J + 2^P): Table[2^(logN-1-P) + (J >> (P+1))].
#ifndef FHE_NTT_ROW8_H
#define FHE_NTT_ROW8_H
#include "ntt_static.h"
#include "ntt_bfly8_pinned.h"

namespace fhe {
namespace r8 {

constexpr int kRegion = 576;  // LDS words per wave (the schemes below reach word 571)

// address schemes of the exchanges inside a wave's region (c, k, m = the fields at tile bits 6-8, 3-5, 0-2):
//   X1 (w-layout  <-> c-layout, through the barrier): 576 w + 64 c + 8 k + m
//   X2 (c-layout  <-> k-layout): 72 c + 8 k + m
//   X3 (k-layout  <-> m-layout): FC3(c) + 33 k + m
//   X4 (m-layout  <-> store layout = c-layout): FC4(c) + k + 36 m
FHE_HD constexpr uint32_t fc3(uint32_t c) { return 8u * (c & 3u) + 264u * (c >> 2); }
FHE_HD constexpr uint32_t fc4(uint32_t c) { return 8u * (c & 3u) + 288u * (c >> 2); }

// ---- butterflies: generated gfx950 code (device) or the same arithmetic in C++ with the bounds checked (emulator) ----------
template <bool UNI, int B>
FHE_HD void fwd_stage(uint64_t (&r)[8], const TwPair (&w)[4], const BflyConst c, uint32_t (&bnd)[8]) {
#ifdef FHE_PINNED_ASM
    (void)bnd;
    if constexpr (UNI) {
        if constexpr (B == 0) stage_fwd_s_b0(r, w, c);
        if constexpr (B == 1) stage_fwd_s_b1(r, w, c);
        if constexpr (B == 2) stage_fwd_s_b2(r, w, c);
    }
    else {
        if constexpr (B == 0) stage_fwd_v_b0(r, w, c);
        if constexpr (B == 1) stage_fwd_v_b1(r, w, c);
        if constexpr (B == 2) stage_fwd_v_b2(r, w, c);
    }
#else
    const uint64_t nq = ((uint64_t)c.nqh << 32) | c.nql;
    for (int g = 0; g < (4 >> B); ++g)
        for (int lo = 0; lo < (1 << B); ++lo) {
            const int k0 = (g << (B + 1)) | lo, k1 = k0 | (1 << B);
            FHE_BOUND_CHECK(bnd[k0] + (uint32_t)kFwdGrow <= 16u, "row8: a forward butterfly whose `a` input may exceed 13q");
            FHE_BOUND_CHECK((unsigned __int128)r[k0] < (unsigned __int128)bnd[k0] * c.q, "row8: a residue above its scheduled bound");
            const uint64_t a = r[k0], T = shoup_trunc(r[k1], w[g], nq);
            FHE_BOUND_CHECK(T < c.threeq, "row8: a truncated Shoup product of 3q or more");
            r[k0]   = a + T;
            r[k1]   = a - T + c.threeq;
            bnd[k0] = bnd[k1] = bnd[k0] + (uint32_t)kFwdGrow;
        }
#endif
}
// the 4 residues whose index has bit B clear (the `a` inputs of the stage on bit B) below 2q
template <int B>
FHE_HD void red4_a(uint64_t (&r)[8], const BflyConst c, uint32_t (&bnd)[8]) {
#ifdef FHE_PINNED_ASM
    (void)bnd;
    if constexpr (B == 0) red4_a0(r, c);
    if constexpr (B == 1) red4_a1(r, c);
    if constexpr (B == 2) red4_a2(r, c);
#else
    for (int k = 0; k < 8; ++k)
        if (!((k >> B) & 1)) {
            r[k] = red_estimate(r[k], c);
            FHE_BOUND_CHECK(r[k] < c.twoq, "row8: a quotient-estimate reduction that left 2q or more");
            bnd[k] = 2;
        }
#endif
}
FHE_HD void red_all(uint64_t (&r)[8], const BflyConst c) {
#ifdef FHE_PINNED_ASM
    red8(r, c);
#else
    for (int k = 0; k < 8; ++k) {
        r[k] = red_estimate(r[k], c);
        FHE_BOUND_CHECK(r[k] < c.twoq, "row8: a quotient-estimate reduction that left 2q or more");
    }
#endif
}
FHE_HD void csub_all(uint64_t (&r)[8], uint64_t m) {
#ifdef FHE_PINNED_ASM
    csub8(r, m);
#else
    for (int k = 0; k < 8; ++k)
        r[k] = csub2(r[k], m);
#endif
}

// the lazy-inverse plan of a step with NB stages (the TOP NB register bits: 3-NB..2), closing reductions or not: tables of ntt_bfly8_pinned.h
template <int NB, bool LAZY>
struct InvPlan;
#define FHE_R8_PLAN(NB, LAZY, TAG)                                                      \
    template <>                                                                         \
    struct InvPlan<NB, LAZY> {                                                          \
        static constexpr const RedOp (&pre)[3][4]        = kInvPre##TAG;                \
        static constexpr const unsigned char (&K)[3][4]  = kInvK##TAG;                  \
        static constexpr const RedOp (&end)[8]           = kInvEnd##TAG;                \
        static constexpr const unsigned char (&out)[8]   = kInvOut##TAG;                \
    };
FHE_R8_PLAN(3, false, Full) FHE_R8_PLAN(3, true, FullLazy) FHE_R8_PLAN(2, false, Two) FHE_R8_PLAN(2, true, TwoLazy)
FHE_R8_PLAN(1, false, One) FHE_R8_PLAN(1, true, OneLazy)
#undef FHE_R8_PLAN

#ifndef FHE_PINNED_ASM
FHE_HD void apply_red_op(uint64_t (&r)[8], const RedOp op, const BflyConst c, uint32_t (&bnd)[8]) {
    if (op.kind == 1) {
        FHE_BOUND_CHECK(bnd[op.k] <= 2u * op.m, "row8: a conditional subtraction of less than half the bound");
        r[op.k]   = csub2(r[op.k], (uint64_t)op.m * c.q);
        bnd[op.k] = op.m;
    }
    else if (op.kind == 2) {
        r[op.k] = red_estimate(r[op.k], c);
        FHE_BOUND_CHECK(r[op.k] < c.twoq, "row8: a quotient-estimate reduction that left 2q or more");
        bnd[op.k] = 2;
    }
}
#endif
// inverse stage on register bit B of a step with NB stages: the plan's reductions, then  a' = u + v (not reduced),
// b' = shoup_trunc(u - v + K q, w) < 3q
template <bool UNI, int NB, int B>
FHE_HD void inv_stage(uint64_t (&r)[8], const TwPair (&w)[4], const BflyConst c, uint32_t (&bnd)[8]) {
#ifdef FHE_PINNED_ASM
    (void)bnd;
#define FHE_R8_INV(TAG, NBB, NAME, BB) if constexpr (NB == NBB && B == BB) stage_invl_##TAG##_##NAME##_b##BB(r, w, c);
    if constexpr (UNI) {
        FHE_R8_INV(s, 3, full, 0) FHE_R8_INV(s, 3, full, 1) FHE_R8_INV(s, 3, full, 2) FHE_R8_INV(s, 2, two, 1) FHE_R8_INV(s, 2, two, 2)
        FHE_R8_INV(s, 1, one, 2)
    }
    else {
        FHE_R8_INV(v, 3, full, 0) FHE_R8_INV(v, 3, full, 1) FHE_R8_INV(v, 3, full, 2) FHE_R8_INV(v, 2, two, 1) FHE_R8_INV(v, 2, two, 2)
        FHE_R8_INV(v, 1, one, 2)
    }
#undef FHE_R8_INV
#else
    using P = InvPlan<NB, true>;  // (the stages' reductions and constants do not depend on the closing reductions)
    const uint64_t nq = ((uint64_t)c.nqh << 32) | c.nql;
    for (int i = 0; i < 4; ++i)
        apply_red_op(r, P::pre[B][i], c, bnd);
    int j = 0;
    for (int g = 0; g < (4 >> B); ++g)
        for (int lo = 0; lo < (1 << B); ++lo, ++j) {
            const int k0 = (g << (B + 1)) | lo, k1 = k0 | (1 << B);
            const uint32_t K = P::K[B][j];
            FHE_BOUND_CHECK(bnd[k1] <= K && bnd[k0] + K <= 16u, "row8: an inverse butterfly outside its planned bounds");
            FHE_BOUND_CHECK((unsigned __int128)r[k0] < (unsigned __int128)bnd[k0] * c.q &&
                                (unsigned __int128)r[k1] < (unsigned __int128)bnd[k1] * c.q,
                            "row8: an inverse residue above its planned bound");
            const uint64_t u = r[k0], v = r[k1];
            r[k0] = u + v;
            r[k1] = shoup_trunc(u - v + (uint64_t)K * c.q, w[g], nq);
            FHE_BOUND_CHECK(r[k1] < c.threeq, "row8: a truncated Shoup product of 3q or more");
            bnd[k0] += bnd[k1];
            bnd[k1] = 3;
        }
#endif
}
template <int NB>
FHE_HD void inv_end(uint64_t (&r)[8], const BflyConst c, uint32_t (&bnd)[8]) {
#ifdef FHE_PINNED_ASM
    (void)bnd;
    if constexpr (NB == 3) inv_end_full(r, c);
    if constexpr (NB == 2) inv_end_two(r, c);
    if constexpr (NB == 1) inv_end_one(r, c);
#else
    using P = InvPlan<NB, false>;
    for (int i = 0; i < 8; ++i)
        apply_red_op(r, P::end[i], c, bnd);
    for (int k = 0; k < 8; ++k)
        FHE_BOUND_CHECK(bnd[k] == P::out[k] && bnd[k] <= 3u, "row8: an inverse step that does not end below 3q");
#endif
}

// ---- one radix-8 step: NB stages on the TOP NB register bits (2..3-NB) of the field whose bit 0 sits at coefficient bit F ---------
// hi = J >> (F + 3) of the lane's residues (the coefficient-index bits above the field); UNI: hi is wave-uniform (scalar loads).
// Twiddle of stage b, group g: Table[2^(logN-1-(F+b)) + (hi << (2-b)) + g].  Forward runs b = 2..3-NB, inverse b = 3-NB..2.
template <bool UNI, int B>
FHE_HD void load_tw(TwPair (&w)[4], const TwPair* tw, uint32_t hi, uint32_t F, uint32_t logN) {
    const uint32_t s = logN - 1u - (F + (uint32_t)B);
    // a uniform base (SGPR pair) and a 32-bit byte offset per lane: the loads take the `saddr + voffset + imm` form, one VGPR of
    // address per stage (a limb's table is 16 N <= 2 MiB)
    const TwPair* ub = tw + ((size_t)1 << s);
#pragma unroll
    for (int g = 0; g < (4 >> B); ++g) {
        if constexpr (UNI) {
            const uint64_t* p = reinterpret_cast<const uint64_t*>(ub + ((size_t)hi << (2 - B)) + g);
#if defined(__HIP_DEVICE_COMPILE__)
            // (the address is opaque before this point: the scalar loads are issued HERE, not hoisted to the top of the tile where
            // their 4..16 SGPRs would be spilled to lanes of a VGPR and read back with VALU instructions)
            asm volatile("" : "+s"(p));
#endif
            w[g] = TwPair{FHE_ULOAD64(p, 0), FHE_ULOAD64(p, 1)};
        }
        else {
            const uint32_t off = (hi << (6 - B)) + 16u * (uint32_t)g;
            w[g]               = *reinterpret_cast<const TwPair*>(reinterpret_cast<const char*>(ub) + off);
        }
    }
}
// forward: SWEEP brings the 4 `a` inputs of the first stage (register bit 2) below 2q first (bound in: anything below 16q)
template <bool UNI, int NB, bool SWEEP>
FHE_HD void fwd_step(uint64_t (&r)[8], const TwPair* tw, uint32_t hi, uint32_t F, uint32_t logN, const BflyConst c, uint32_t inBound) {
    uint32_t bnd[8];
    for (int k = 0; k < 8; ++k)
        bnd[k] = inBound;
    // per-lane twiddles: at most 6 pairs in registers at a time (the last stage's 4 pairs are fetched once the first stage's pair is dead)
    TwPair w2[4], w1[4], w0[4];
    if constexpr (NB >= 1) load_tw<UNI, 2>(w2, tw, hi, F, logN);
    if constexpr (NB >= 2) load_tw<UNI, 1>(w1, tw, hi, F, logN);
    if constexpr (SWEEP && NB >= 1)
        red4_a<2>(r, c, bnd);
    if constexpr (NB >= 1) fwd_stage<UNI, 2>(r, w2, c, bnd);
    if constexpr (NB >= 3) load_tw<UNI, 0>(w0, tw, hi, F, logN);
    if constexpr (NB >= 2) fwd_stage<UNI, 1>(r, w1, c, bnd);
    if constexpr (NB >= 3) fwd_stage<UNI, 0>(r, w0, c, bnd);
}
template <bool UNI, int NB, bool LAZY>
FHE_HD void inv_step(uint64_t (&r)[8], const TwPair* tw, uint32_t hi, uint32_t F, uint32_t logN, const BflyConst c) {
    uint32_t bnd[8];
    for (int k = 0; k < 8; ++k)
        bnd[k] = 3;
    TwPair w2[4], w1[4], w0[4];
    if constexpr (NB >= 3) load_tw<UNI, 0>(w0, tw, hi, F, logN);
    if constexpr (NB >= 2) load_tw<UNI, 1>(w1, tw, hi, F, logN);
    if constexpr (NB >= 1 && NB < 3) load_tw<UNI, 2>(w2, tw, hi, F, logN);
    if constexpr (NB >= 3) inv_stage<UNI, NB, 0>(r, w0, c, bnd);
    if constexpr (NB >= 3) load_tw<UNI, 2>(w2, tw, hi, F, logN);
    if constexpr (NB >= 2) inv_stage<UNI, NB, 1>(r, w1, c, bnd);
    if constexpr (NB >= 1) inv_stage<UNI, NB, 2>(r, w2, c, bnd);
    if constexpr (!LAZY && NB >= 1)
        inv_end<NB>(r, c, bnd);
}

// forward lazy-range schedule of the pass (units of q): a radix-8 step of NB stages adds 3 NB; a step whose stages would pass 16
// sweeps first (its first stage's `a` inputs below 2q).  Steps: w (WB stages), c, k, m (3 each).
template <int WB, int BIN>
struct FwdSched {
    static constexpr int TA = WB;
    static constexpr int stages(int i) { return i == 0 ? TA : 3; }
    static constexpr int before(int i) {
        int b = BIN;
        for (int j = 0; j < i; ++j)
            b = (b + 3 * stages(j) <= 16) ? b + 3 * stages(j) : 2 + 3 * stages(j);
        return b;
    }
    static constexpr bool sweep(int i) { return before(i) + 3 * stages(i) > 16; }
    static constexpr int out = before(4);
};

// MODE as in ntt_static.h: forward 9 (or 1): bound class of the pass input (below 2q / canonical); inverse 0: the column pass follows
// (the last step's closing reductions are left to it: residues below 16q), 1: this pass ends the transform (not instantiated yet).
struct TileAt {
    uint32_t tb, rit, tr;  // tower of the batch, row inside the tower, tile inside the row
};
// blockIdx -> tile: with xcdSwizzle an XCD (blockIdx mod 8) keeps one (limb, tile of the ring) across the batch, so that the workgroups
// it runs at a time share one slice of the twiddle tables in its L2 (as ntt_static.h)
FHE_DEV TileAt tile_at(const NttPassArgs& a, uint32_t vt, uint32_t trLog, bool swz) {
    uint32_t tile = vt;
    if (swz) {
        const uint32_t xcd = vt & 7u, i = vt >> 3;
        const uint32_t b = i % a.batch, pairIdx = i / a.batch;
        const uint32_t pair = pairIdx * 8u + xcd;
        tile = ((b * a.nLimbs + (pair >> trLog)) << trLog) + (pair & ((1u << trLog) - 1u));
    }
    const uint32_t row = tile >> trLog;
    return TileAt{row / a.nLimbs, row % a.nLimbs, tile & ((1u << trLog) - 1u)};
}

#if defined(__HIP_DEVICE_COMPILE__)
// the value is opaque to the compiler from here on: what is derived from it is computed HERE (not early and then kept in a register for
// the whole tile: the kernel has 64 VGPRs and they are all spoken for)
#define FHE_R8_OPAQUE_V(x) asm volatile("" : "+v"(x))
#else
#define FHE_R8_OPAQUE_V(x) ((void)0)
#endif
// lane coordinates, derived afresh where an exchange or a step needs them: l = lane of the wave, (lhi, llo) = its two 3-bit fields
#define FHE_R8_LANE()                 \
    uint32_t tq_ = t;                 \
    FHE_R8_OPAQUE_V(tq_);             \
    const uint32_t l = tq_ & 63u, lhi = l >> 3, llo = l & 7u; \
    (void)lhi, (void)llo

// WB: the tile has 2^WB waves and 2^(9+WB) words; the pass has T = 9 + WB stages.
template <bool INV, int WB, int MODE>
FHE_DEV void ntt_row8_core(const NttPassArgs& a, uint64_t* lds) {
    static_assert(WB >= 0 && WB <= 3, "row8: 1, 2, 4 or 8 waves per tile");
    static_assert(!(INV && MODE != 0), "row8: the inverse pass that ends a transform is ntt_static.h's");
    constexpr uint32_t tileLog = 9u + (uint32_t)WB;
    constexpr uint32_t threads = 64u << WB;
    constexpr int WLOW         = 3 - WB;  // register bits of the w-step's field that belong to c (no butterflies on them)
    const uint32_t t     = FHE_TID;
    const uint32_t wv    = WB ? FHE_UNIFORM(t >> 6) : 0u;
    const uint32_t logN  = a.logN;
    const uint32_t trLog = logN - tileLog;
    uint64_t* reg = lds + (size_t)wv * kRegion;  // this wave's region
    uint64_t r[8];
    // (xcdSwizzle of the args is the host's answer for 4096-word tiles; smaller tiles only add factors of two)
    const TileAt at      = tile_at(a, FHE_BID, trLog, a.xcdSwizzle != 0);
    const uint32_t jbase = at.tr << tileLog;
    const uint32_t rit   = at.rit;
    const uint64_t inRow  = a.inStride ? ((uint64_t)at.tb * a.inStride + a.inFirst + rit) : ((uint64_t)at.tb * a.nLimbs + rit);
    const uint64_t outRow = a.outStride ? ((uint64_t)at.tb * a.outStride + a.outFirst + rit) : ((uint64_t)at.tb * a.nLimbs + rit);
    const uint32_t limb = FHE_UNIFORM(a.sel.idx[rit]);
    const uint64_t q    = FHE_ULOAD64(a.q, limb);
    const uint64_t twoq = q << 1, nq = 0 - q;
    const TwPair* tw    = a.tw + ((uint64_t)limb << logN);
    const uint64_t redc = FHE_ULOAD64(a.red, limb);
    const BflyConst c{(uint32_t)nq, (uint32_t)(nq >> 32), q, twoq, 0 - twoq, twoq + q, (uint32_t)redc, (uint32_t)(redc >> 32)};
    const uint64_t* src = (a.inDelta ? a.xin + (int64_t)at.tb * a.inDelta + ((uint64_t)(a.inFirst + rit) << logN) : a.xin + (inRow << logN)) + jbase;
    uint64_t* dst       = a.x + (outRow << logN) + jbase;
    // X1 word of register rr in the w-layout: the register is (w : top 3 - WB bits of c); lane t = (low WB bits of c, k, m)
    auto x1 = [](int rr) { return (uint32_t)kRegion * ((uint32_t)rr >> WLOW) + (64u << WB) * ((uint32_t)rr & ((1u << WLOW) - 1u)); };

    if constexpr (!INV) {
        using S = FwdSched<WB, (MODE == 9 ? 2 : MODE)>;
        if constexpr (WB > 0) {
            // w-layout: register rr holds word threads * rr + t of the tile (512 contiguous bytes per wave instruction); uniform twiddles
            {
                FHE_R8_LANE();
                const uint64_t* s0 = src + tq_;
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    r[k] = FHE_GLD(&s0[threads * (uint32_t)k]);
            }
            fwd_step<true, WB, S::sweep(0)>(r, tw, at.tr, 6 + WB, logN, c, S::sweep(0) ? 2u : (uint32_t)S::before(0));
            FHE_R8_LANE();
            uint64_t* Lw = lds + tq_;  // X1: 576 w + 64 c + 8 k + m
#pragma unroll
            for (int k = 0; k < 8; ++k)
                FHE_LDS_ST(Lw[x1(k)], r[k]);
            FHE_SSYNC();
            const uint64_t* Lr = reg + l;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                FHE_LDS_LD(r[k], Lr[64 * k]);
        }
        else {
            // one wave per tile: its 512 words straight from memory in the c-layout
            FHE_R8_LANE();
            const uint64_t* s0 = src + l;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                r[k] = FHE_GLD(&s0[64u * (uint32_t)k]);
        }
        // c-layout: lane = (k, m), register = c; wave-uniform twiddles
        fwd_step<true, 3, S::sweep(1)>(r, tw, (jbase >> 9) + wv, 6, logN, c, S::sweep(1) ? 2u : (uint32_t)S::before(1));
        {   // X2: 72 c + 8 k + m
            FHE_R8_LANE();
            FHE_WAVE_SYNC();
            uint64_t* Lw = reg + l;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                FHE_LDS_ST(Lw[72 * k], r[k]);
            FHE_WAVE_SYNC();
            const uint64_t* Lr = reg + 72u * lhi + llo;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                FHE_LDS_LD(r[k], Lr[8 * k]);
            // k-layout: lane = (c, m), register = k
            fwd_step<false, 3, S::sweep(2)>(r, tw, (jbase >> 6) + (wv << 3) + lhi, 3, logN, c, S::sweep(2) ? 2u : (uint32_t)S::before(2));
        }
        {   // X3: FC3(c) + 33 k + m
            FHE_R8_LANE();
            FHE_WAVE_SYNC();
            uint64_t* Lw = reg + fc3(lhi) + llo;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                FHE_LDS_ST(Lw[33 * k], r[k]);
            FHE_WAVE_SYNC();
            const uint64_t* Lr = reg + fc3(lhi) + 33u * llo;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                FHE_LDS_LD(r[k], Lr[k]);
            // m-layout: lane = (c, k), register = m: the lane's 8 consecutive coefficients
            fwd_step<false, 3, S::sweep(3)>(r, tw, (jbase >> 3) + (wv << 6) + l, 0, logN, c, S::sweep(3) ? 2u : (uint32_t)S::before(3));
        }
        if (a.canonStep != 0xffffffffu) {
            red_all(r, c);
            csub_all(r, q);
        }
        {   // X4: FC4(c) + k + 36 m, read back in the c-layout (= the store layout)
            FHE_R8_LANE();
            FHE_WAVE_SYNC();
            uint64_t* Lw = reg + fc4(lhi) + llo;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                FHE_LDS_ST(Lw[36 * k], r[k]);
            FHE_WAVE_SYNC();
            const uint64_t* Lr = reg + lhi + 36u * llo;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                FHE_LDS_LD(r[k], Lr[fc4((uint32_t)k)]);
            uint64_t* d0 = dst + (wv << 9) + l;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                FHE_GST(&d0[64u * (uint32_t)k], r[k]);
        }
    }
    else {
        {   // c-layout load (512 contiguous bytes per wave instruction), X4 backwards into the m-layout
            FHE_R8_LANE();
            const uint64_t* s0 = src + (wv << 9) + l;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                r[k] = FHE_GLD(&s0[64u * (uint32_t)k]);
            uint64_t* Lw = reg + lhi + 36u * llo;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                FHE_LDS_ST(Lw[fc4((uint32_t)k)], r[k]);
            FHE_WAVE_SYNC();
            const uint64_t* Lr = reg + fc4(lhi) + llo;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                FHE_LDS_LD(r[k], Lr[36 * k]);
            inv_step<false, 3, false>(r, tw, (jbase >> 3) + (wv << 6) + l, 0, logN, c);
        }
        {   // X3 backwards
            FHE_R8_LANE();
            FHE_WAVE_SYNC();
            uint64_t* Lw = reg + fc3(lhi) + 33u * llo;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                FHE_LDS_ST(Lw[k], r[k]);
            FHE_WAVE_SYNC();
            const uint64_t* Lr = reg + fc3(lhi) + llo;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                FHE_LDS_LD(r[k], Lr[33 * k]);
            inv_step<false, 3, false>(r, tw, (jbase >> 6) + (wv << 3) + lhi, 3, logN, c);
        }
        {   // X2 backwards
            FHE_R8_LANE();
            FHE_WAVE_SYNC();
            uint64_t* Lw = reg + 72u * lhi + llo;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                FHE_LDS_ST(Lw[8 * k], r[k]);
            FHE_WAVE_SYNC();
            const uint64_t* Lr = reg + l;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                FHE_LDS_LD(r[k], Lr[72 * k]);
        }
        // the column pass that follows reduces on the way in (anything below 16q): the LAST step with stages skips its closing reductions
        inv_step<true, 3, (WB == 0)>(r, tw, (jbase >> 9) + wv, 6, logN, c);
        if constexpr (WB > 0) {
            FHE_R8_LANE();
            {   // X1 backwards, through the barrier
                FHE_WAVE_SYNC();
                uint64_t* Lw = reg + l;
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    FHE_LDS_ST(Lw[64 * k], r[k]);
                FHE_SSYNC();
                const uint64_t* Lr = lds + tq_;
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    FHE_LDS_LD(r[k], Lr[x1(k)]);
            }
            inv_step<true, WB, true>(r, tw, at.tr, 6 + WB, logN, c);
            uint64_t* d0 = dst + tq_;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                FHE_GST(&d0[threads * (uint32_t)k], r[k]);
        }
        else {
            FHE_R8_LANE();
            uint64_t* d0 = dst + l;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                FHE_GST(&d0[64u * (uint32_t)k], r[k]);
        }
    }
}
#undef FHE_R8_LANE

// 64 VGPRs: 8 waves per SIMD, 32 / 2^WB tiles per CU
template <bool INV, int WB, int MODE>
FHE_GLOBAL void FHE_LAUNCH_BOUNDS2(64 << WB, 8) ntt_row8_kernel(const NttPassArgs a) {
    FHE_SHARED_U64(lds, (1 << WB) * kRegion);
    ntt_row8_core<INV, WB, MODE>(a, lds);
}

}  // namespace r8
}  // namespace fhe
#endif
