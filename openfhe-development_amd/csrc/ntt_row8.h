// ntt_row8.h — the row pass of a two-pass transform at EIGHT residues per lane (round 6).
//
// Why: the 16-residues-per-lane row pass (ntt_static.h) needs 120-124 VGPRs and 38 KiB of LDS per 256-thread workgroup: four
// workgroups = 16 waves per CU by both limits, six workgroup barriers per tile; it ran at 8.2-8.7 ms per pass against 5.5 ms of
// HBM time and 5.7 ms of integer issue (profiles/r05_sweeps.md 1a/1b: throughput was set by having four customers for two equally
// loaded stations).  This kernel keeps the same tile (4096 consecutive words of one limb = 32 KiB), the same butterflies, tables and
// lazy ranges, and changes the residency:
//   * 512 threads x 8 residues per lane; <= 64 VGPRs per lane (ntt_bfly8_pinned.h: residues v[48:63], one butterfly slot's
//     temporaries v[36:47], v[0:35] for addresses and the 7 per-lane twiddle pairs of a radix-8 step) => 8 waves per SIMD,
//     32 waves (4 workgroups) per CU, 36 KiB of LDS per workgroup;
//   * the 12-bit tile index is  w:c:k:m  (3 bits each).  Radix-8 steps act on the fields w (tile bits 9-11), c (6-8), k (3-5),
//     m (0-2).  Only the step on w mixes the eight waves: from then on WAVE w OWNS the 512 words whose top field is w, and the
//     exchanges between the steps on c, k and m are private to a wave — LDS instructions of one wave execute in order, so they
//     need NO workgroup barrier.  A tile meets ONE barrier (ntt_static.h: six);
//   * every exchange has its own additive address scheme  F1(field1) + F2(field2) + F3(field3)  inside the wave's region of 576
//     words, chosen so that both its write and its read pattern put the 32 lanes of a half-wave on 32 distinct 8-byte bank pairs
//     (tools/occbench.hip, profiles/r06_occbench.json: 296 ns per 64 KiB exchange and CU with the skews, 351-438 without);
//     register indices enter as immediates (`base + imm`), as in ntt_static.h;
//   * HBM sees 512 contiguous bytes per wave instruction on both sides (forward: load in the w-layout, store after a last
//     wave-private exchange; inverse: the mirror image).
// Same transform as transformnat-impl.h:303-374 (forward, stages on tile bits 11..0) / 512-625 (inverse, 0..11); twiddle of the
// stage on coefficient bit P for the pair (J, J + 2^P): Table[2^(logN-1-P) + (J >> (P+1))].
#ifndef FHE_NTT_ROW8_H
#define FHE_NTT_ROW8_H
#include "ntt_static.h"
#include "ntt_bfly8_pinned.h"

namespace fhe {
namespace r8 {

constexpr int kThreads8 = 512;
constexpr int kRegion   = 576;           // LDS words per wave (the schemes below reach word 571)
constexpr int kLdsWords = 8 * kRegion;   // 36 KiB

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

// the lazy-inverse plan of a step with NB stages (register bits 0..NB-1), closing reductions or not: tables of ntt_bfly8_pinned.h
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
        FHE_R8_INV(s, 3, full, 0) FHE_R8_INV(s, 3, full, 1) FHE_R8_INV(s, 3, full, 2) FHE_R8_INV(s, 2, two, 0) FHE_R8_INV(s, 2, two, 1)
        FHE_R8_INV(s, 1, one, 0)
    }
    else {
        FHE_R8_INV(v, 3, full, 0) FHE_R8_INV(v, 3, full, 1) FHE_R8_INV(v, 3, full, 2) FHE_R8_INV(v, 2, two, 0) FHE_R8_INV(v, 2, two, 1)
        FHE_R8_INV(v, 1, one, 0)
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

// ---- one radix-8 step: NB stages on the register bits 0..NB-1 of the field at tile bit F -----------------------------------
// hi = J >> (F + 3) of the lane's residues (the coefficient-index bits above the field); UNI: hi is wave-uniform (scalar loads).
// Twiddle of stage b, group g: Table[2^(logN-1-(F+b)) + (hi << (2-b)) + g].  Forward runs b = NB-1..0, inverse b = 0..NB-1.
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
// forward: `sweep` brings the 4 `a` inputs of the first stage below 2q first (bound in: anything below 16q)
template <bool UNI, int NB, bool SWEEP>
FHE_HD void fwd_step(uint64_t (&r)[8], const TwPair* tw, uint32_t hi, uint32_t F, uint32_t logN, const BflyConst c, uint32_t inBound) {
    uint32_t bnd[8];
    for (int k = 0; k < 8; ++k)
        bnd[k] = inBound;
    // per-lane twiddles: at most 6 pairs in registers at a time (the last stage's 4 pairs are fetched once the first stage's pair is dead)
    TwPair w2[4], w1[4], w0[4];
    if constexpr (NB >= 3) load_tw<UNI, 2>(w2, tw, hi, F, logN);
    if constexpr (NB >= 2) load_tw<UNI, 1>(w1, tw, hi, F, logN);
    if constexpr (NB >= 1 && NB < 3) load_tw<UNI, 0>(w0, tw, hi, F, logN);
    if constexpr (SWEEP && NB >= 1)
        red4_a<NB - 1>(r, c, bnd);
    if constexpr (NB >= 3) fwd_stage<UNI, 2>(r, w2, c, bnd);
    if constexpr (NB >= 3) load_tw<UNI, 0>(w0, tw, hi, F, logN);
    if constexpr (NB >= 2) fwd_stage<UNI, 1>(r, w1, c, bnd);
    if constexpr (NB >= 1) fwd_stage<UNI, 0>(r, w0, c, bnd);
}
template <bool UNI, int NB, bool LAZY>
FHE_HD void inv_step(uint64_t (&r)[8], const TwPair* tw, uint32_t hi, uint32_t F, uint32_t logN, const BflyConst c) {
    uint32_t bnd[8];
    for (int k = 0; k < 8; ++k)
        bnd[k] = 3;
    TwPair w2[4], w1[4], w0[4];
    if constexpr (NB >= 1) load_tw<UNI, 0>(w0, tw, hi, F, logN);
    if constexpr (NB >= 2) load_tw<UNI, 1>(w1, tw, hi, F, logN);
    if constexpr (NB >= 1) inv_stage<UNI, NB, 0>(r, w0, c, bnd);
    if constexpr (NB >= 3) load_tw<UNI, 2>(w2, tw, hi, F, logN);
    if constexpr (NB >= 2) inv_stage<UNI, NB, 1>(r, w1, c, bnd);
    if constexpr (NB >= 3) inv_stage<UNI, NB, 2>(r, w2, c, bnd);
    if constexpr (!LAZY && NB >= 1)
        inv_end<NB>(r, c, bnd);
}

// forward lazy-range schedule of the pass (units of q): a radix-8 step of NB stages adds 3 NB; a step whose stages would pass 16
// sweeps first (its first stage's `a` inputs below 2q).  Steps: w (TA = T - 9 stages), c, k, m (3 each).
template <int T, int BIN>
struct FwdSched {
    static constexpr int TA = T - 9;
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
//
// PIPE: the workgroup is PERSISTENT (grid = 3 per CU) and software-pipelined: the loads of its next tile are issued before the
// butterflies of the current one (16 more VGPRs: 80 per lane, 6 waves per SIMD, 3 workgroups per CU), the stores of the current tile
// drain under the next one.  Why: with one tile per workgroup a CU holds four tiles by the register file AND by the 32-wave limit,
// whatever the lane count per tile, and each of them spends 2.6 us loading and 2.3 us storing without computing — 8.1 ms per pass for
// this kernel and for ntt_static.h's alike (profiles/r06_sweeps.md: residency in waves is not the lever, tiles in flight are).  The
// counter of vector-memory operations is in order, so waiting for the prefetched loads (issued BEFORE the previous tile's stores)
// does not wait for those stores.  Two barriers per tile: the cross-wave exchange X1, and one that keeps a fast wave's next X1 (or,
// inverse, its next wave-private exchange) out of regions a slow wave still reads.
struct TileAt {
    uint32_t tb, rit, tr;  // tower of the batch, row inside the tower, tile inside the row
};
FHE_DEV TileAt tile_at(const NttPassArgs& a, uint32_t vt, uint32_t trLog) {
    uint32_t tile = vt;
    if (a.xcdSwizzle) {
        const uint32_t xcd = vt & 7u, i = vt >> 3;
        const uint32_t b = i % a.batch, pairIdx = i / a.batch;
        const uint32_t pair = pairIdx * 8u + xcd;
        tile = ((b * a.nLimbs + (pair >> trLog)) << trLog) + (pair & ((1u << trLog) - 1u));
    }
    const uint32_t row = tile >> trLog;
    return TileAt{row / a.nLimbs, row % a.nLimbs, tile & ((1u << trLog) - 1u)};
}
FHE_DEV const uint64_t* tile_src(const NttPassArgs& a, const TileAt& p) {
    const uint32_t logN = a.logN;
    const uint64_t inRow = a.inStride ? ((uint64_t)p.tb * a.inStride + a.inFirst + p.rit) : ((uint64_t)p.tb * a.nLimbs + p.rit);
    return (a.inDelta ? a.xin + (int64_t)p.tb * a.inDelta + ((uint64_t)(a.inFirst + p.rit) << logN) : a.xin + (inRow << logN)) +
           ((size_t)p.tr << kTileLog);
}

#if defined(__HIP_DEVICE_COMPILE__)
// the value is opaque to the compiler from here on: what is derived from it is computed HERE (not hoisted out of the tile loop and
// kept in a register for the whole tile: the kernel has 64 / 80 VGPRs and ~100 SGPRs, and they are all spoken for)
#define FHE_R8_OPAQUE_V(x) asm volatile("" : "+v"(x))
#define FHE_R8_OPAQUE_S(x) asm volatile("" : "+s"(x))
#else
#define FHE_R8_OPAQUE_V(x) ((void)0)
#define FHE_R8_OPAQUE_S(x) ((void)0)
#endif
// lane coordinates, derived afresh where an exchange or a step needs them: l = lane of the wave, (lhi, llo) = its two 3-bit fields
#define FHE_R8_LANE()                 \
    uint32_t tq_ = t;                 \
    FHE_R8_OPAQUE_V(tq_);             \
    const uint32_t l = tq_ & 63u, lhi = l >> 3, llo = l & 7u; \
    (void)lhi, (void)llo

template <bool INV, int T, int MODE, bool PIPE>
FHE_DEV void ntt_row8_core(const NttPassArgs& a, uint64_t* lds) {
    static_assert(T >= 9 && T <= 12, "row8: 9..12 stages");
    static_assert(!(INV && MODE != 0), "row8: the inverse pass that ends a transform is ntt_static.h's");
    constexpr int TA = T - 9;
    constexpr bool loadW = !INV && TA > 0;  // first load in the w-layout (lane t, word t + 512 k), else in the c-layout
    constexpr uint32_t ldStep = loadW ? 512u : 64u;
    const uint32_t t     = FHE_TID;
    const uint32_t wv    = FHE_UNIFORM(t >> 6);
    const uint32_t logN  = a.logN;
    const uint32_t trLog = logN - (uint32_t)kTileLog;
    const uint32_t nTiles = a.rows << trLog;
    const uint32_t stride = PIPE ? FHE_NBLK : 0u;
    uint64_t* reg = lds + (size_t)wv * kRegion;  // this wave's region
    uint64_t r[8], nx[8];
    uint32_t vt = FHE_BID;
    TileAt at   = tile_at(a, vt, trLog);
    // word of register 0 in the load layout; register k: + ldStep k
    auto load_tile = [&](uint64_t (&v)[8]) {
        FHE_R8_LANE();
        const uint64_t* s0 = tile_src(a, at) + (loadW ? tq_ : (wv << 9) + l);
#pragma unroll
        for (int k = 0; k < 8; ++k)
            v[k] = FHE_GLD(&s0[ldStep * (uint32_t)k]);
    };
    if constexpr (PIPE)
        load_tile(nx);
    do {
        const uint32_t jbase = at.tr << kTileLog;
        const uint32_t rit   = at.rit;
        const uint64_t outRow = a.outStride ? ((uint64_t)at.tb * a.outStride + a.outFirst + rit) : ((uint64_t)at.tb * a.nLimbs + rit);
        const uint32_t limb = FHE_UNIFORM(a.sel.idx[rit]);
        const uint64_t q    = FHE_ULOAD64(a.q, limb);
        const uint64_t twoq = q << 1, nq = 0 - q;
        const TwPair* tw    = a.tw + ((uint64_t)limb << logN);
        const uint64_t redc = FHE_ULOAD64(a.red, limb);
        const BflyConst c{(uint32_t)nq, (uint32_t)(nq >> 32), q, twoq, 0 - twoq, twoq + q, (uint32_t)redc, (uint32_t)(redc >> 32)};
        uint64_t* dst = a.x + (outRow << logN) + jbase;
        if constexpr (PIPE) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                r[k] = nx[k];
            if (vt + stride < nTiles) {  // (uniform) the next tile's loads, before this tile's butterflies
                at = tile_at(a, vt + stride, trLog);
                load_tile(nx);
            }
        }
        else
            load_tile(r);
        // hi* = the coefficient-index bits above a field, for the lane's residues in that step's layout
        if constexpr (!INV) {
            using S = FwdSched<T, (MODE == 9 ? 2 : MODE)>;
            if constexpr (TA > 0) {
                // w-layout: lane t = (c, k, m), register = w (loaded with 512 contiguous bytes per wave instruction); uniform twiddles
                fwd_step<true, TA, S::sweep(0)>(r, tw, jbase >> 12, 9, logN, c, S::sweep(0) ? 2u : (uint32_t)S::before(0));
                if constexpr (PIPE)
                    FHE_SSYNC();  // every wave has left the previous tile's exchanges
                FHE_R8_LANE();
                uint64_t* Lw = lds + tq_;  // X1: 576 w + (64 c + 8 k + m)
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    FHE_LDS_ST(Lw[kRegion * k], r[k]);
                FHE_SSYNC();
                const uint64_t* Lr = reg + l;
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    FHE_LDS_LD(r[k], Lr[64 * k]);
            }
            // (no stage on the w field: the wave's 512 words came straight from memory in the c-layout)
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
                fwd_step<false, 3, S::sweep(2)>(r, tw, (jbase >> 6) + (wv << 3) + lhi, 3, logN, c,
                                                S::sweep(2) ? 2u : (uint32_t)S::before(2));
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
                fwd_step<false, 3, S::sweep(3)>(r, tw, (jbase >> 3) + (wv << 6) + l, 0, logN, c,
                                                S::sweep(3) ? 2u : (uint32_t)S::before(3));
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
                if constexpr (PIPE && TA > 0)
                    FHE_SSYNC();  // every wave has read the previous tile's X1
                FHE_R8_LANE();
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
            inv_step<true, 3, (TA == 0)>(r, tw, (jbase >> 9) + wv, 6, logN, c);
            if constexpr (TA > 0) {
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
                        FHE_LDS_LD(r[k], Lr[kRegion * k]);
                }
                inv_step<true, TA, true>(r, tw, jbase >> 12, 9, logN, c);
                uint64_t* d0 = dst + tq_;
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    FHE_GST(&d0[512u * (uint32_t)k], r[k]);
            }
            else {
                FHE_R8_LANE();
                uint64_t* d0 = dst + (wv << 9) + l;
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    FHE_GST(&d0[64u * (uint32_t)k], r[k]);
            }
        }
        vt += stride;
    } while (PIPE && vt < nTiles);
}
#undef FHE_R8_LANE

// one tile per workgroup: <= 64 VGPRs, 8 waves per SIMD, 4 workgroups per CU
template <bool INV, int T, int MODE>
FHE_GLOBAL void FHE_LAUNCH_BOUNDS2(kThreads8, 8) ntt_row8_kernel(const NttPassArgs a) {
    FHE_SHARED_U64(lds, kLdsWords);
    ntt_row8_core<INV, T, MODE, false>(a, lds);
}
// persistent and software-pipelined: <= 80 VGPRs, 6 waves per SIMD, 3 workgroups per CU (launch with kPipePerCu workgroups per CU)
constexpr uint32_t kPipePerCu = 3;
template <bool INV, int T, int MODE>
FHE_GLOBAL void FHE_LAUNCH_BOUNDS2(kThreads8, 6) ntt_row8_pipe_kernel(const NttPassArgs a) {
    FHE_SHARED_U64(lds, kLdsWords);
    ntt_row8_core<INV, T, MODE, true>(a, lds);
}

}  // namespace r8
}  // namespace fhe
#endif
