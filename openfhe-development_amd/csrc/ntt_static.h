// ntt_static.h — production NTT pass kernels for rings with N >= 4096: the step plan of a pass is a compile-time
// constant (template parameters), so that
//  * every loop over steps / stages is straight-line code,
//  * the LDS exchange uses `base + immediate` addresses: the tile index I is stored at word I + (I >> 4) (one pad word
//    per 16), which is additive over the disjoint bit fields of (lane part, register part) and conflict-free for
//    ds_read_b64 / ds_write_b64 at every field position; two buffers alternate, one barrier per exchange,
//  * the 16 residues of a lane stay in v[32:63] and the butterflies / conditional subtractions run in place on them
//    (ntt_bfly_pinned.h, generated and simulated by tools/gen_ntt_asm.py),
//  * wave-uniform twiddles and all limb constants are scalar operands.
// Same transform, tables, stage order and lazy-reduction bounds as ntt_pass_full_kernel (ntt_kernels.h), which remains
// the fallback for pass shapes that are not instantiated here; reference: transformnat-impl.h:303-374, 512-625.
#ifndef FHE_NTT_STATIC_H
#define FHE_NTT_STATIC_H
#include "ntt_kernels.h"
#include "ntt_inv_plan16.h"

#if defined(__HIP_DEVICE_COMPILE__) && !defined(FHE_NO_BFLY_ASM)
#include "ntt_bfly_pinned.h"
#define FHE_PINNED_ASM 1
#endif

namespace fhe {

#ifndef FHE_PINNED_ASM
struct BflyConst {
    uint32_t nql, nqh;
    uint64_t q, twoq, ntwoq, threeq;
    uint32_t redM, redR;
};
struct BflyZero {
    uint32_t z0, z1;
};
#endif
// Round 4 lazy ranges (tools/gen_ntt_asm.py): a forward butterfly with the truncated Shoup quotient adds at most 3q to the
// bound of its `a` input; a step whose stages would push a bound past 16q (< 2^64) first brings the 8 `a` inputs of its
// first stage below 2q with one quotient estimate each.
constexpr int kFwdGrow = 3;

// ---- the generated code's arithmetic restated in C++ (round 5; the path of the lane emulator and of FHE_NO_BFLY_ASM builds).  Until
// round 4 this path used exact quotients and `% 2q`: the CPU suite could not see a wrong lazy bound, a wrong pairing of SPlan with
// schedule_fwd, or a bad quotient estimate.  Now it computes what ntt_bfly_pinned.h computes (tools/gen_ntt_asm.py: fwd_stream,
// red_stream) and checks the bounds the schedule promises, value by value.
// y * w - Q' * q  (mod 2^64) with the TRUNCATED Shoup quotient Q' = y_h p_h + floor((y_h p_l + y_l p_h) / 2^32): in [0, 3q)
FHE_HD uint64_t shoup_trunc(uint64_t y, const TwPair w, uint64_t nq) {
    const uint64_t yl = (uint32_t)y, yh = y >> 32, pl = (uint32_t)w.wp, ph = w.wp >> 32;
    const unsigned __int128 mid = (unsigned __int128)(yh * pl) + (unsigned __int128)(yl * ph);
    const uint64_t Q            = yh * ph + (uint64_t)(mid >> 32);
    return y * w.w + Q * nq;
}
// any 64-bit x -> x - k q in [0, 2q): k = ((x >> 32) * redM >> 32) >> redR  (red_stream); limbs below 2^35 (redR == 255) take the ladder of
// three conditional subtractions, which needs x < 16q (red_slow_stream)
FHE_HD uint64_t red_estimate(uint64_t x, const BflyConst c) {
    if (c.redR == 255u)
        return csub2(csub2(csub2(x, c.q << 3), c.q << 2), c.twoq);
    const uint64_t nq = ((uint64_t)c.nqh << 32) | c.nql;
    const uint32_t k  = (uint32_t)(((x >> 32) * (uint64_t)c.redM) >> 32) >> c.redR;
    return x + (uint64_t)k * nq;
}
#if defined(FHE_EMU)
#define FHE_BOUND_CHECK(cond, what)                                                                  \
    do {                                                                                             \
        if (!(cond)) {                                                                               \
            fprintf(stderr, "ntt_static: lazy-range violation in the emulator: %s\n", what);       \
            abort();                                                                                 \
        }                                                                                            \
    } while (0)
#else
#define FHE_BOUND_CHECK(cond, what) ((void)0)
#endif

// compile-time plan of one pass: the same grouping of the T stages into register-resident steps as plan_pass()
template <bool LA, bool INV, int T>
struct SPlan {
    static constexpr int logC = kTileLog - T;
    static constexpr int nst  = (T + 3) / 4;
    static constexpr int ish  = LA ? logC : 0;
    static constexpr int size(int i) { return T / nst + (i < T % nst ? 1 : 0); }
    static constexpr int before(int i) {
        int s = 0;
        for (int j = 0; j < i; ++j)
            s += size(j);
        return s;
    }
    static constexpr int fp(int i) {
        if (!INV) {
            const int top = T - 1 - before(i);
            return top - 3 > 0 ? top - 3 : 0;
        }
        const int bot = before(i);
        return bot < T - 4 ? bot : T - 4;
    }
    static constexpr int bHi(int i) { return !INV ? (T - 1 - before(i)) - fp(i) : before(i) - fp(i) + size(i) - 1; }
    static constexpr int bLo(int i) { return !INV ? bHi(i) - size(i) + 1 : before(i) - fp(i); }
    static constexpr int fI(int i) { return fp(i) + ish; }
    static constexpr bool stageFirst = !LA && fI(0) < 4;        // coalesced staging through LDS before the first step
    static constexpr bool stageLast  = !LA && fI(nst - 1) < 4;  // ... and after the last one
    // forward lazy-reduction schedule (schedule_fwd): bound of the values, in units of q, before step i
    static constexpr int boundBefore(int i, int bin) {
        int b = bin;
        for (int j = 0; j < i; ++j)
            b = (b + kFwdGrow * size(j) <= 16) ? b + kFwdGrow * size(j) : 2 + kFwdGrow * size(j);
        return b;
    }
    static constexpr bool sweep(int i, int bin) { return boundBefore(i, bin) + kFwdGrow * size(i) > 16; }
    static constexpr int outBound(int bin) { return boundBefore(nst, bin); }
};

#ifdef FHE_ABL_NOLDS  // timing experiment: the LDS exchange moves nothing (results are wrong)
#define FHE_LDS_ST(slot, val) ((void)(slot))
#define FHE_LDS_LD(dst, slot) ((void)(slot))
#else
#define FHE_LDS_ST(slot, val) ((slot) = (val))
#define FHE_LDS_LD(dst, slot) ((dst) = (slot))
#endif
#ifdef FHE_ABL_NOSYNC  // timing experiment: no workgroup barriers (results are wrong)
#define FHE_SSYNC() ((void)0)
#else
#define FHE_SSYNC() FHE_SYNC()
#endif
#ifdef FHE_ABL_NOLOAD  // timing experiment: the tile is made up from its address, no HBM load (results are wrong)
#define FHE_GLD(p) ((uint64_t)(uintptr_t)(p))
#else
#define FHE_GLD(p) (*(p))
#endif
#ifdef FHE_ABL_NOSTORE  // timing experiment: the pass's result stays in registers (stores behind a test that never holds)
#define FHE_GST(p, v) do { if ((v) == 0x123456789abcdefull) *(p) = (v); } while (0)
#else
#define FHE_GST(p, v) (*(p) = (v))
#endif
constexpr int kLdsPadWords = kTile + (kTile >> 4);
FHE_HD constexpr uint32_t lds_pad(uint32_t I) {
    return I + (I >> 4);
}

// Inverse stages are LAZY in the sum output (a' = u + v is not reduced): bounds add along a residue's chain of sum outputs and return to
// 3q with every product output, the constant K of u - v + K q is the bound of v, a pair whose bounds would pass 16q is reduced first by
// the cheapest means, and the step ends with the reductions that bring every residue below 3q again (tools/gen_ntt_asm.py inv_plan).
// The C++ build (lane emulator, FHE_NO_BFLY_ASM) FOLLOWS THE GENERATED PLAN (ntt_inv_plan16.h: the same tables the gfx950 blocks were
// emitted from) op by op and checks every bound it promises, value by value — a wrong plan fails the CPU suite (round 6; until then this
// path was a round-3 restatement with its own bound rule and exact quotients).
FHE_HD void apply_plan_op(uint64_t (&r)[16], const plan16::RedOp op, const BflyConst c, uint32_t (&bnd)[16]) {
    if (op.kind == 1) {
        FHE_BOUND_CHECK(bnd[op.k] <= 2u * op.m, "inverse plan: a conditional subtraction of less than half the bound");
        r[op.k]   = csub2(r[op.k], (uint64_t)op.m * c.q);
        bnd[op.k] = op.m;
    }
    else if (op.kind == 2) {
        r[op.k] = red_estimate(r[op.k], c);
        FHE_BOUND_CHECK(r[op.k] < c.twoq, "inverse plan: a quotient-estimate reduction that left 2q or more");
        bnd[op.k] = 2;
    }
}
FHE_HD void inv_lazy_stage_cpp(int B, int BLO, uint64_t (&r)[16], const TwPair (&w)[8], const BflyConst c, uint32_t (&bnd)[16]) {
    const uint64_t nq = ((uint64_t)c.nqh << 32) | c.nql;
    for (int i = 0; i < 16; ++i)
        apply_plan_op(r, plan16::kPre[BLO][B][i], c, bnd);
    int j = 0;
    for (int g = 0; g < (8 >> B); ++g)
        for (int lo = 0; lo < (1 << B); ++lo, ++j) {
            const int k0 = (g << (B + 1)) | lo, k1 = k0 | (1 << B);
            const uint32_t K = plan16::kK[BLO][B][j];
            FHE_BOUND_CHECK(bnd[k1] <= K && bnd[k0] + K <= 16u, "inverse plan: a butterfly outside its planned bounds");
            FHE_BOUND_CHECK((unsigned __int128)r[k0] < (unsigned __int128)bnd[k0] * c.q && (unsigned __int128)r[k1] < (unsigned __int128)bnd[k1] * c.q,
                            "inverse plan: a residue above its planned bound");
            const uint64_t u = r[k0], v = r[k1];
            r[k0] = u + v;
            r[k1] = shoup_trunc(u - v + (uint64_t)K * c.q, w[g], nq);
            FHE_BOUND_CHECK(r[k1] < c.threeq, "inverse plan: a truncated Shoup product of 3q or more");
            bnd[k0] += bnd[k1];
            bnd[k1] = 3;
        }
}
FHE_HD void inv_lazy_end_cpp(int BLO, int BHI, uint64_t (&r)[16], const BflyConst c, uint32_t (&bnd)[16]) {
    for (int i = 0; i < 16; ++i)
        apply_plan_op(r, plan16::kEnd[BLO][BHI][i], c, bnd);
    for (int k = 0; k < 16; ++k)
        FHE_BOUND_CHECK(bnd[k] == plan16::kOut[BLO][BHI][k] && bnd[k] <= 3u, "inverse plan: a step that does not end below 3q");
}

template <bool INV, bool UNI, int B, int BLO>
FHE_HD void run_stage(uint64_t (&r)[16], const TwPair (&w)[8], const BflyConst c, const BflyZero z, uint32_t (&bnd)[16]) {
#ifdef FHE_PINNED_ASM
    (void)bnd;
    if constexpr (!INV) {
        if constexpr (UNI) {
            if constexpr (B == 0) stage_fwd_s_b0(r, w, c, z);
            if constexpr (B == 1) stage_fwd_s_b1(r, w, c, z);
            if constexpr (B == 2) stage_fwd_s_b2(r, w, c, z);
            if constexpr (B == 3) stage_fwd_s_b3(r, w, c, z);
        }
        else {
            if constexpr (B == 0) stage_fwd_v_b0(r, w, c, z);
            if constexpr (B == 1) stage_fwd_v_b1(r, w, c, z);
            if constexpr (B == 2) stage_fwd_v_b2(r, w, c, z);
            if constexpr (B == 3) stage_fwd_v_b3(r, w, c, z);
        }
    }
    else {
#define FHE_INVL(TAG, BB, LO) if constexpr (B == BB && BLO == LO) stage_invl_##TAG##_b##BB##_lo##LO(r, w, c, z);
        if constexpr (UNI) {
            FHE_INVL(s, 0, 0) FHE_INVL(s, 1, 0) FHE_INVL(s, 2, 0) FHE_INVL(s, 3, 0) FHE_INVL(s, 1, 1) FHE_INVL(s, 2, 1)
            FHE_INVL(s, 3, 1) FHE_INVL(s, 2, 2) FHE_INVL(s, 3, 2) FHE_INVL(s, 3, 3)
        }
        else {
            FHE_INVL(v, 0, 0) FHE_INVL(v, 1, 0) FHE_INVL(v, 2, 0) FHE_INVL(v, 3, 0) FHE_INVL(v, 1, 1) FHE_INVL(v, 2, 1)
            FHE_INVL(v, 3, 1) FHE_INVL(v, 2, 2) FHE_INVL(v, 3, 2) FHE_INVL(v, 3, 3)
        }
#undef FHE_INVL
    }
#else
    (void)z;
    if (INV) {
        inv_lazy_stage_cpp(B, BLO, r, w, c, bnd);
        return;
    }
    // fwd_stream: a' = a + T, b' = a - T + 3q with T = shoup_trunc(b, w) in [0, 3q); `bnd` = the bound (units of q) the schedule
    // promises for every residue: the butterfly needs bound(a) + 3 <= 16 (16q < 2^64 for q < 2^60), whatever b is
    const uint64_t nq = ((uint64_t)c.nqh << 32) | c.nql;
    for (int g = 0; g < (8 >> B); ++g)
        for (int lo = 0; lo < (1 << B); ++lo) {
            const int k0 = (g << (B + 1)) | lo, k1 = k0 | (1 << B);
            FHE_BOUND_CHECK(bnd[k0] + (uint32_t)kFwdGrow <= 16u, "a forward butterfly whose `a` input may exceed 13q");
            FHE_BOUND_CHECK((unsigned __int128)r[k0] < (unsigned __int128)bnd[k0] * c.q, "a residue above its scheduled bound");
            const uint64_t a = r[k0], T = shoup_trunc(r[k1], w[g], nq);
            FHE_BOUND_CHECK(T < c.threeq, "a truncated Shoup product of 3q or more");
            r[k0]   = a + T;
            r[k1]   = a - T + c.threeq;
            bnd[k0] = bnd[k1] = bnd[k0] + (uint32_t)kFwdGrow;
        }
#endif
}
// end of a lazy inverse step: every residue back below 2q
template <int BLO, int BHI>
FHE_HD void run_inv_lazy_end(uint64_t (&r)[16], const BflyConst c, uint32_t (&bnd)[16]) {
#ifdef FHE_PINNED_ASM
    (void)bnd;
#define FHE_INVE(LO, HI) if constexpr (BLO == LO && BHI == HI) inv_lazy_end_lo##LO##_hi##HI(r, c);
    FHE_INVE(0, 0) FHE_INVE(0, 1) FHE_INVE(0, 2) FHE_INVE(0, 3) FHE_INVE(1, 1) FHE_INVE(1, 2) FHE_INVE(1, 3) FHE_INVE(2, 2)
    FHE_INVE(2, 3) FHE_INVE(3, 3)
#undef FHE_INVE
#else
    inv_lazy_end_cpp(BLO, BHI, r, c, bnd);
#endif
}

// last inverse stage (s == 0, always field bit 3): lower output * N^-1, upper output * (w1 * N^-1).  Its inputs carry the
// lazy bounds of the step (see run_stage): the pair's K and the 16q -> 8q correction follow the same plan.
template <int BLO>
FHE_HD void run_last_inv_stage(uint64_t (&r)[16], const TwPair nInv, const TwPair w1n, const BflyConst c, const BflyZero z,
                               uint32_t (&bnd)[16]) {
    const uint64_t q = c.q;
#ifdef FHE_PINNED_ASM
    (void)bnd;
    if constexpr (BLO == 0) inv_lazy_pre_b3_lo0(r, c);
    if constexpr (BLO == 1) inv_lazy_pre_b3_lo1(r, c);
    if constexpr (BLO == 2) inv_lazy_pre_b3_lo2(r, c);
    if constexpr (BLO == 3) inv_lazy_pre_b3_lo3(r, c);
#pragma unroll
    for (int lo = 0; lo < 8; ++lo) {
        const uint64_t u = r[lo], v = r[lo | 8];
        r[lo]            = u + v;
        r[lo | 8]        = u - v + (uint64_t)kInvLazyK[BLO][3][lo] * q;
    }
    mul2_s_0(r, nInv, w1n, c, z);
    mul2_s_1(r, nInv, w1n, c, z);
    mul2_s_2(r, nInv, w1n, c, z);
    mul2_s_3(r, nInv, w1n, c, z);
    mul2_s_4(r, nInv, w1n, c, z);
    mul2_s_5(r, nInv, w1n, c, z);
    mul2_s_6(r, nInv, w1n, c, z);
    mul2_s_7(r, nInv, w1n, c, z);
#else
    (void)z;
    // the generated plan of the step's last stage (register bit 3): its reductions, then  (u + v) * N^-1  and  (u - v + K q) * (w1 N^-1)
    // with EXACT quotients (results below 2q), as mul2_s_* compute them
    const uint64_t nq = ((uint64_t)c.nqh << 32) | c.nql;
    for (int i = 0; i < 16; ++i)
        apply_plan_op(r, plan16::kPre[BLO][3][i], c, bnd);
    for (int lo = 0; lo < 8; ++lo) {
        const uint32_t K = plan16::kK[BLO][3][lo];
        FHE_BOUND_CHECK(bnd[lo | 8] <= K && bnd[lo] + K <= 16u, "inverse plan: the last stage outside its planned bounds");
        FHE_BOUND_CHECK((unsigned __int128)r[lo] < (unsigned __int128)bnd[lo] * q && (unsigned __int128)r[lo | 8] < (unsigned __int128)bnd[lo | 8] * q,
                        "inverse plan: a residue above its planned bound before the last stage");
        const uint64_t u = r[lo], v = r[lo | 8];
        r[lo]     = shoup_acc(0, u + v, nInv, nq);
        r[lo | 8] = shoup_acc(0, u - v + (uint64_t)K * q, w1n, nq);
        FHE_BOUND_CHECK(r[lo] < c.twoq && r[lo | 8] < c.twoq, "inverse plan: an exact Shoup product of 2q or more");
        bnd[lo] = bnd[lo | 8] = 2;
    }
#endif
}

// the 8 residues whose index has bit B clear (the `a` inputs of a stage on bit B) below 2q (any 64-bit value before)
template <int B>
FHE_HD void run_red8_a(uint64_t (&r)[16], const BflyConst c) {
#ifdef FHE_PINNED_ASM
    if constexpr (B == 0) red8_a0(r, c);
    if constexpr (B == 1) red8_a1(r, c);
    if constexpr (B == 2) red8_a2(r, c);
    if constexpr (B == 3) red8_a3(r, c);
#else
    for (int k = 0; k < 16; ++k)
        if (!((k >> B) & 1)) {
            r[k] = red_estimate(r[k], c);
            FHE_BOUND_CHECK(r[k] < c.twoq, "a quotient-estimate reduction that left 2q or more");
        }
#endif
}
// every residue below 2q (any 64-bit value before)
FHE_HD void run_red16(uint64_t (&r)[16], const BflyConst c) {
#ifdef FHE_PINNED_ASM
    red16(r, c);
#else
    for (int k = 0; k < 16; ++k) {
        r[k] = red_estimate(r[k], c);
        FHE_BOUND_CHECK(r[k] < c.twoq, "a quotient-estimate reduction that left 2q or more");
    }
#endif
}
// residues I and I|8 times the Shoup pair `cw`, lazily reduced to [0,2q)
template <int I>
FHE_HD void epi_mul2(uint64_t (&r)[16], const TwPair cw, const BflyConst c, const BflyZero z) {
#ifdef FHE_PINNED_ASM
    if constexpr (I == 0) mul2_s_0(r, cw, cw, c, z);
    if constexpr (I == 1) mul2_s_1(r, cw, cw, c, z);
    if constexpr (I == 2) mul2_s_2(r, cw, cw, c, z);
    if constexpr (I == 3) mul2_s_3(r, cw, cw, c, z);
    if constexpr (I == 4) mul2_s_4(r, cw, cw, c, z);
    if constexpr (I == 5) mul2_s_5(r, cw, cw, c, z);
    if constexpr (I == 6) mul2_s_6(r, cw, cw, c, z);
    if constexpr (I == 7) mul2_s_7(r, cw, cw, c, z);
#else
    (void)z;
    const uint64_t nq = ((uint64_t)c.nqh << 32) | c.nql;
    r[I]     = shoup_acc(0, r[I], cw, nq);
    r[I | 8] = shoup_acc(0, r[I | 8], cw, nq);
#endif
}
FHE_HD void run_csub16(uint64_t (&r)[16], uint64_t m) {
#ifdef FHE_PINNED_ASM
    csub16(r, m);
#else
    for (int k = 0; k < 16; ++k)
        r[k] = csub2(r[k], m);
#endif
}

// Twiddles of one step.  `rowLane` (row passes only, may be null) points at this lane's entry of the lane-major copy
// of the twiddles of the step whose register field sits at tile bit 0: there a lane's 15 twiddles are distinct from
// every other lane's, and in the standard table (index 2^s + (j >> 4) * 2^(3-b) + g) neighbouring lanes are 2^(3-b)
// entries apart — one 16-byte piece per cache line and instruction.  The copy stores, per tile, slot (2^(3-b) - 1 + g)
// of all 256 lanes contiguously, so each load instruction of a wave reads 1 KiB of consecutive memory.
struct TwSrc {
    const TwPair* tw;       // standard table of the limb (Table[bitrev(i)] = psi^i)
    const TwPair* rowLane;  // lane-major copy: slot s of this lane at rowLane[s * kThreads]
    const TwPair* fin;      // inverse: {N^-1, Table_inv[1] * N^-1}
    const TwPair* shared;   // LDS copy of the twiddles of the row pass's "shared" step (below), or null
    uint32_t sharedLane;    // this lane's group index in that step: t >> fp
};
constexpr int kRowTwSlots = 15;

// "Shared" step of a row pass: register field at tile bits fp..fp+3 with 4 <= fp < 8.  Its twiddles depend on the
// lane only through t >> fp (2^(8-fp) <= 16 groups), and for one stage b the tile needs ONE contiguous run of
// groups * (8 >> b) table entries starting at 2^s + (tileBase >> (fp+4)) * 2^(3-b).  The workgroup copies the runs of
// the step's stages into LDS with one 16-byte load per lane at kernel start (<= 240 entries); the lanes then read
// their (up to 15) twiddles as ds_read_b128 instead of 15 global loads of which 16+ lanes fetch the same address.
template <int T, bool INV>
struct SharedStep {
    using P = SPlan<false, INV, T>;
    static constexpr int find() {
        for (int i = 0; i < P::nst; ++i)
            if (P::fp(i) >= 4 && P::fp(i) + 4 < kTileLog)
                return i;
        return -1;
    }
    static constexpr int I      = find();
    static constexpr bool any   = I >= 0;
    static constexpr int fp     = any ? P::fp(any ? I : 0) : 4;
    static constexpr int groups = 1 << (8 - fp);
    static constexpr bool active(int b) { return any && b <= P::bHi(any ? I : 0) && b >= P::bLo(any ? I : 0); }
    static constexpr int count(int b) { return active(b) ? groups * (8 >> b) : 0; }
    static constexpr int offset(int b) {  // LDS entry offset of stage b's run
        int o = 0;
        for (int k = 0; k < b; ++k)
            o += count(k);
        return o;
    }
    static constexpr int total = count(0) + count(1) + count(2) + count(3);
};
constexpr int kSharedTwWords = 2 * 256;  // LDS words reserved for the shared step's twiddles (<= 240 TwPairs)

template <bool LA, bool INV, int T, int I, int B>
struct StageInfo {
    using P = SPlan<LA, INV, T>;
    static constexpr bool active = B <= P::bHi(I) && B >= P::bLo(I);
    static constexpr int fp      = P::fp(I);
    static constexpr bool uni    = LA ? (fp + 4 == T) : (fp + 4 >= kTileLog);
};

// issue the loads of stage B's twiddles (8 >> B of them)
template <bool LA, bool INV, int T, int I, int B, bool ENDS>
FHE_HD void load_stage_tw(TwPair (&w)[8], const TwSrc ts, uint32_t j0, uint32_t logN) {
    using S = StageInfo<LA, INV, T, I, B>;
    using P = SPlan<LA, INV, T>;
    if constexpr (S::active) {
        // the transform's last inverse stage (s == 0) is the top stage of the pass that ends the transform
        if constexpr (INV && ENDS && I == P::nst - 1 && B == P::bHi(I)) {
            static_assert(B == 3, "the last inverse stage sits at field bit 3");
            const uint64_t* fp64 = reinterpret_cast<const uint64_t*>(ts.fin);
            w[0] = TwPair{FHE_ULOAD64(fp64, 0), FHE_ULOAD64(fp64, 1)};
            w[1] = TwPair{FHE_ULOAD64(fp64, 2), FHE_ULOAD64(fp64, 3)};
        }
        else {
#ifdef FHE_ABL_NOBFLY  // timing experiment: no twiddle loads, no butterflies (results are wrong)
            return;
#endif
            // twiddle index = 2^s + (j >> (Fj + 4)) * 2^(3-B) + g,  s = logN - 1 - (Fj + B),  Fj = fp + (LA ? logN - T : 0)
            const uint32_t Fj = (uint32_t)S::fp + (LA ? logN - (uint32_t)T : 0u);
            const uint32_t s  = logN - 1u - (Fj + (uint32_t)B);
            uint32_t jhigh    = j0 >> (Fj + 4u);
            if constexpr (S::uni)
                jhigh = FHE_UNIFORM(jhigh);
            const TwPair* base = ts.tw + ((size_t)1 << s);
            const uint32_t off = jhigh << (3 - B);
#pragma unroll
            for (int g = 0; g < (8 >> B); ++g) {
#ifdef FHE_ABL_NOTW  // timing experiment: every butterfly uses the limb's first twiddle (results are wrong)
                const uint64_t* p0 = reinterpret_cast<const uint64_t*>(ts.tw + 1);
                w[g]               = TwPair{FHE_ULOAD64(p0, 0) + (S::uni ? 0 : (j0 & 1)), FHE_ULOAD64(p0, 1)};
                continue;
#endif
                if constexpr (S::uni) {
                    const uint64_t* p = reinterpret_cast<const uint64_t*>(base + off + g);
                    w[g]              = TwPair{FHE_ULOAD64(p, 0), FHE_ULOAD64(p, 1)};
                }
                else if constexpr (!LA && S::fp == 0) {
                    if (ts.rowLane)
                        w[g] = ts.rowLane[(size_t)((1 << (3 - B)) - 1 + g) * kThreads];
                    else
                        w[g] = base[off + g];
                }
                else if constexpr (!LA && SharedStep<T, INV>::any && SharedStep<T, INV>::I == I) {
                    using SS = SharedStep<T, INV>;
                    w[g]     = ts.shared[SS::offset(B) + ts.sharedLane * (8 >> B) + g];  // LDS (ds_read_b128)
                }
                else
                    w[g] = base[off + g];
            }
        }
    }
}

template <bool LA, bool INV, int T, int I, int B, bool ENDS, bool LAZYOUT>
FHE_HD void exec_stage(uint64_t (&r)[16], const TwPair (&w)[8], const BflyConst c, const BflyZero z, uint32_t (&bnd)[16]) {
    using S = StageInfo<LA, INV, T, I, B>;
    using P = SPlan<LA, INV, T>;
    if constexpr (S::active) {
        if constexpr (INV && ENDS && I == P::nst - 1 && B == P::bHi(I))
            run_last_inv_stage<P::bLo(I)>(r, w[0], w[1], c, z, bnd);
        else {
#ifdef FHE_ABL_NOBFLY
            return;
#endif
            run_stage<INV, S::uni, B, P::bLo(I)>(r, w, c, z, bnd);
            // a lazy inverse step ends with the reductions back below 3q (after its last stage); the last step of a row
            // pass that the column pass follows leaves them to that (HBM-bound) pass
            if constexpr (INV && B == P::bHi(I) && !(LAZYOUT && I == P::nst - 1))
                run_inv_lazy_end<P::bLo(I), P::bHi(I)>(r, c, bnd);
        }
    }
}

// one register-resident step I of the plan: its stages, highest field bit first (forward) / lowest first (inverse);
// the twiddle loads of a stage are issued before the butterflies of the previous stage (the asm blocks are
// scheduling barriers, so the order written here is the order executed)
template <bool LA, bool INV, int T, int I, bool ENDS, bool LAZYOUT>
FHE_HD void run_step(uint64_t (&r)[16], const TwSrc ts, uint32_t j0, uint32_t logN, const BflyConst c, const BflyZero z,
                     uint32_t inBound = 2) {
    TwPair w0[8], w1[8], w2[8], w3[8];
    // lazy bounds in units of q (C++ build only): an inverse step starts below 2q; a forward step starts with what the schedule
    // promises for the `a` inputs of its first stage (inBound: 2 behind a sweep, else SPlan::boundBefore)
    uint32_t bnd[16];
    for (int k = 0; k < 16; ++k)
        bnd[k] = INV ? 3u : inBound;
    constexpr int B0 = INV ? 0 : 3, B1 = INV ? 1 : 2, B2 = INV ? 2 : 1, B3 = INV ? 3 : 0;
    load_stage_tw<LA, INV, T, I, B0, ENDS>(w0, ts, j0, logN);
    load_stage_tw<LA, INV, T, I, B1, ENDS>(w1, ts, j0, logN);
    exec_stage<LA, INV, T, I, B0, ENDS, LAZYOUT>(r, w0, c, z, bnd);
    load_stage_tw<LA, INV, T, I, B2, ENDS>(w2, ts, j0, logN);
    exec_stage<LA, INV, T, I, B1, ENDS, LAZYOUT>(r, w1, c, z, bnd);
    load_stage_tw<LA, INV, T, I, B3, ENDS>(w3, ts, j0, logN);
    exec_stage<LA, INV, T, I, B2, ENDS, LAZYOUT>(r, w2, c, z, bnd);
    exec_stage<LA, INV, T, I, B3, ENDS, LAZYOUT>(r, w3, c, z, bnd);
}

// lane geometry of a step whose register field sits at tile-index bit fI: tile index of register 0 and the
// coefficient index / stride of the lane's 16 values
template <bool LA, int T, int FI>
FHE_HD void lane_geom_s(uint32_t t, uint32_t S, uint32_t& Ib, uint32_t& jrel, uint64_t& kstride) {
    constexpr int logC = kTileLog - T;
    Ib = ((t >> FI) << (FI + 4)) | (t & ((1u << FI) - 1u));
    if (LA) {
        const uint32_t p0 = Ib >> logC, c0 = Ib & ((1u << logC) - 1u);
        jrel    = p0 * S + c0;
        kstride = (FI >= logC) ? ((uint64_t)S << (FI >= logC ? FI - logC : 0)) : ((uint64_t)1 << FI);
    }
    else {
        jrel    = Ib;
        kstride = (uint64_t)1 << FI;
    }
}

// MODE, forward: an upper bound of the pass input in units of q — 1 = canonical, 9 = the lazy output of a 4-stage
// column pass, 16 = anything below 16q (the lazy-reduction schedule is derived from it).  MODE, inverse: 1 = this pass ends the transform
// (its top stage is the transform's last stage, with N^-1 folded in), 0 = it does not.
// One LDS buffer with a barrier on either side of an exchange (34 KiB, 4 workgroups per CU); the double-buffered form
// (68 KiB, 2 workgroups per CU) was measured slower (profiles/r01_sweeps.md).
// RAWOUT: the pass ends with its residues in the registers of the last step's lane layout (no staging exchange, no store);
// RAWIN: the pass starts from residues already in the registers of its first step's lane layout (no load, no staging).
// Both exist for the fused polynomial product (poly_mul kernels below): a forward row pass whose last step and an inverse
// row pass whose first step both act on tile bit 0 hold the same 16 consecutive residues per lane.
// PRO: the first load takes every limb of a tower from one COEFFICIENT row modulo another limb's modulus and switches it to the
// limb's own modulus (NttPassArgs::proMode) — forward column passes only.
template <bool LA, bool INV, int T, int MODE, bool EPI = false, bool RAWIN = false, bool RAWOUT = false, bool PRO = false>
FHE_DEV void ntt_static_core(const NttPassArgs& a, uint32_t bid, uint64_t* lds, uint64_t (&r)[16]) {
    using P = SPlan<LA, INV, T>;
    static_assert(!(RAWIN || RAWOUT) || !LA, "register hand-over exists for row passes only");
    static_assert(!PRO || (LA && !INV && !RAWIN), "the load prologue exists for forward column passes only");
    const uint32_t t    = FHE_TID;
    const uint32_t logN = a.logN;
    const uint32_t N    = 1u << logN;
    const uint32_t tilesPerRow = N >> kTileLog;
    uint32_t tile = bid;
    if (a.xcdSwizzle) {
        const uint32_t xcd = tile & 7u, i = tile >> 3;
        const uint32_t b = i % a.batch, pairIdx = i / a.batch;
        const uint32_t pair = pairIdx * 8u + xcd;
        tile = (b * a.nLimbs + pair / tilesPerRow) * tilesPerRow + pair % tilesPerRow;
    }
    constexpr int logC = kTileLog - T;
    const uint32_t S    = N >> T;
    const uint32_t row  = tile / tilesPerRow, tr = tile % tilesPerRow;
    const uint32_t jbase = LA ? (tr << logC) : (tr << kTileLog);
    const uint32_t tb = row / a.nLimbs, rit = row % a.nLimbs;
    const uint64_t inRow  = PRO ? ((uint64_t)tb * a.inStride + a.inFirst)
                                : a.inStride ? ((uint64_t)tb * a.inStride + a.inFirst + rit) : (uint64_t)row;
    const uint64_t outRow = a.outStride ? ((uint64_t)tb * a.outStride + a.outFirst + rit) : (uint64_t)row;
    const uint32_t limb = FHE_UNIFORM(a.sel.idx[rit]);
    const uint64_t q    = FHE_ULOAD64(a.q, limb);
    const uint64_t twoq = q << 1, nq = 0 - q;
    TwSrc ts;
    ts.tw      = a.tw + ((uint64_t)limb << logN);
    ts.fin     = a.fin + 2 * (size_t)limb;
    ts.rowLane = nullptr;
    if (!LA && a.twRow)  // [limb][tile of the ring][slot][lane]
        ts.rowLane = a.twRow + ((size_t)limb * tilesPerRow + tr) * (kRowTwSlots * kThreads) + t;
    // twiddles of the shared step: one 16-byte global load per lane now, written to LDS behind the data loads
    TwPair* sharedLds = reinterpret_cast<TwPair*>(lds + kLdsPadWords);
    ts.shared         = sharedLds;
    ts.sharedLane     = 0;
    uint64_t sharedW = 0, sharedWp = 0;  // (two scalars: a conditionally assigned 16-byte struct lands in scratch)
    using SS = SharedStep<LA ? 12 : T, INV>;
    constexpr bool useShared = !LA && SS::any;
    if constexpr (useShared) {
        ts.sharedLane = t >> SS::fp;
        if (t < (uint32_t)SS::total) {
            // entry t belongs to the run of stage b with offset(b) <= t < offset(b) + count(b)
            uint32_t b = 0, ob = 0;
            if (SS::count(1) && t >= (uint32_t)SS::offset(1))
                b = 1, ob = (uint32_t)SS::offset(1);
            if (SS::count(2) && t >= (uint32_t)SS::offset(2))
                b = 2, ob = (uint32_t)SS::offset(2);
            if (SS::count(3) && t >= (uint32_t)SS::offset(3))
                b = 3, ob = (uint32_t)SS::offset(3);
            if (!SS::count(0) && b == 0)  // the step has no stage 0: its first run belongs to the lowest active stage
                b = SS::count(1) ? 1 : SS::count(2) ? 2 : 3;
            const uint32_t s_b = logN - 1u - ((uint32_t)SS::fp + b);
            const size_t idx   = ((size_t)1 << s_b) + ((size_t)(jbase >> (SS::fp + 4)) << (3u - b)) + (t - ob);
            const TwPair v     = ts.tw[idx];
            sharedW = v.w, sharedWp = v.wp;
        }
    }
    const uint64_t redc = FHE_ULOAD64(a.red, limb);  // {redM, redR} of the limb (fhe_ctx_create)
    const BflyConst c{(uint32_t)nq, (uint32_t)(nq >> 32), q, twoq, 0 - twoq, twoq + q, (uint32_t)redc, (uint32_t)(redc >> 32)};
    BflyZero z{0, 0};
#ifdef FHE_PINNED_ASM
    asm volatile("v_mov_b32 %0, 0\n\tv_mov_b32 %1, 0" : "={v65}"(z.z0), "={v81}"(z.z1));
#endif
    const bool canonOut = a.canonStep != 0xffffffffu;
    // forward: bound class of the pass input (instance name 9 = what a column pass leaves: below 2q since round 4);
    // inverse row pass that does not end the transform: its last step's reductions happen in the column pass
    constexpr int BIN      = INV ? 0 : (MODE == 9 ? 2 : MODE);
    constexpr bool LAZYOUT = INV && !LA && MODE == 0;
    const uint64_t* src = a.inDelta ? a.xin + (int64_t)tb * a.inDelta + ((uint64_t)(a.inFirst + (PRO ? 0u : rit)) << logN) + jbase
                                    : a.xin + (inRow << logN) + jbase;
    uint64_t* dst       = a.x + (outRow << logN) + jbase;

    uint32_t Ib, jrel;
    uint64_t ks;
    // final store of the pass, with the optional fused epilogue (NttPassArgs::epiMode)
    auto store_result = [&](uint64_t (&v)[16], uint32_t jr, uint64_t kstr) {
        if constexpr (!EPI) {
#pragma unroll
            for (int k = 0; k < 16; ++k)
                FHE_GST(&dst[jr + k * kstr], v[k]);
        }
        else {
            const uint64_t* cw = reinterpret_cast<const uint64_t*>(a.epiC + rit);
            const uint64_t cW = FHE_ULOAD64(cw, 0), cWp = FHE_ULOAD64(cw, 1);
            const uint64_t* A  = a.epiADelta ? a.epiA + (int64_t)tb * a.epiADelta + ((uint64_t)(a.epiAFirst + rit) << logN) + jbase
                                             : a.epiA + ((((uint64_t)tb * a.epiAStride + a.epiAFirst + rit)) << logN) + jbase;
            const bool second  = tb >= a.epiSplit;
            uint64_t* O        = (second ? a.epiOut1 : a.epiOut0) +
                          ((((uint64_t)(second ? tb - a.epiSplit : tb) * a.nLimbs + rit)) << logN) + jbase;
            const bool acc = a.epiMode == 2u;
            const TwPair cpair{cW, cWp};
            // groups of 4 residues {2g, 2g+1, 2g+8, 2g+9} keep the extra live registers small; the multiplication
            // by C runs in place on the pinned residues (ntt_bfly_pinned.h, mul2_s_*)
#define FHE_EPI_GROUP(G)                                                                          \
    {                                                                                             \
        constexpr int ks4[4] = {2 * G, 2 * G + 1, 2 * G + 8, 2 * G + 9};                          \
        uint64_t av[4], ov[4] = {0, 0, 0, 0};                                                     \
        _Pragma("unroll") for (int k = 0; k < 4; ++k) av[k] = A[jr + ks4[k] * kstr];              \
        if (acc) {                                                                                \
            _Pragma("unroll") for (int k = 0; k < 4; ++k) ov[k] = O[jr + ks4[k] * kstr];          \
        }                                                                                         \
        _Pragma("unroll") for (int k = 0; k < 4; ++k) v[ks4[k]] = av[k] + twoq - v[ks4[k]];       \
        epi_mul2<2 * G>(v, cpair, c, z);                                                          \
        epi_mul2<2 * G + 1>(v, cpair, c, z);                                                      \
        _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                           \
            uint64_t x = csub2(v[ks4[k]], q);                                                     \
            if (acc)                                                                              \
                x = add_mod(ov[k], x, q);                                                         \
            O[jr + ks4[k] * kstr] = x;                                                            \
        }                                                                                         \
    }
            FHE_EPI_GROUP(0)
            FHE_EPI_GROUP(1)
            FHE_EPI_GROUP(2)
            FHE_EPI_GROUP(3)
#undef FHE_EPI_GROUP
        }
    };

    auto pro_switch = [&](uint64_t (&v)[16]) {
        const uint64_t qs = FHE_ULOAD64(a.q, a.proSrcLimb), halfQs = qs >> 1;
#pragma unroll
        for (int k = 0; k < 16; ++k)
            v[k] = switch_modulus_word(v[k], qs, halfQs, q);
    };

#define FHE_SHARED_TW_TO_LDS()                                             \
    if constexpr (useShared) {                                             \
        if (t < (uint32_t)SS::total)                                       \
            sharedLds[t] = TwPair{sharedW, sharedWp};                      \
        /* the shared step must sit behind at least one barrier */          \
        if constexpr (SS::I == 0 && !P::stageFirst)                        \
            FHE_SSYNC();                                                   \
    }

    // ---- first load ----
    if constexpr (RAWIN) {
        FHE_SSYNC();  // every wave has left the previous pass's exchange buffer and shared twiddles
        if constexpr (useShared) {
            if (t < (uint32_t)SS::total)
                sharedLds[t] = TwPair{sharedW, sharedWp};
            if constexpr (SS::I == 0)
                FHE_SSYNC();
        }
    }
    else if constexpr (P::stageFirst) {
        lane_geom_s<LA, T, 8>(t, S, Ib, jrel, ks);
#pragma unroll
        for (int k = 0; k < 16; ++k)
            r[k] = FHE_GLD(&src[jrel + k * ks]);
        if constexpr (PRO)
            pro_switch(r);
        FHE_SHARED_TW_TO_LDS()
        uint64_t* L = lds + lds_pad(Ib);
#pragma unroll
        for (int k = 0; k < 16; ++k)
            FHE_LDS_ST(L[lds_pad((uint32_t)k << 8)], r[k]);
        FHE_SSYNC();
    }

#define FHE_STATIC_STEP(I)                                                                                        \
    if constexpr (I < P::nst) {                                                                                   \
        constexpr int fI = P::fI(I);                                                                              \
        lane_geom_s<LA, T, fI>(t, S, Ib, jrel, ks);                                                               \
        if constexpr (I == 0 && RAWIN) {                                                                          \
            /* residues already in r[] in this step's layout */                                                   \
        }                                                                                                         \
        else if constexpr (I == 0 && !P::stageFirst) {                                                            \
            _Pragma("unroll") for (int k = 0; k < 16; ++k) r[k] = FHE_GLD(&src[jrel + k * ks]);                   \
            if constexpr (PRO)                                                                                    \
                pro_switch(r);                                                                                    \
            FHE_SHARED_TW_TO_LDS()                                                                                \
        }                                                                                                         \
        else {                                                                                                    \
            const uint64_t* L = lds + lds_pad(Ib);                                           \
            _Pragma("unroll") for (int k = 0; k < 16; ++k) FHE_LDS_LD(r[k], L[lds_pad((uint32_t)k << fI)]);                  \
        }                                                                                                         \
        /* an inverse column pass takes what the lazy row pass left (anything below 16q) */                        \
        if constexpr (I == 0 && INV && LA)                                                                        \
            run_red16(r, c);                                                                                      \
        /* lazy reduction: a forward butterfly's outputs are bounded by its `a` input + 3q whatever the `b` input  \
           (< 2^64) is, so only the 8 `a` inputs of the step's first stage go back below 2q */                      \
        if constexpr (!INV && P::sweep(I, BIN))                                                                  \
            run_red8_a<P::bHi(I)>(r, c);                                                                          \
        run_step<LA, INV, T, I, (INV && MODE == 1), LAZYOUT>(r, ts, jbase + jrel, logN, c, z,                      \
                                                             (uint32_t)(P::sweep(I, BIN) ? 2 : P::boundBefore(I, BIN)));  \
        if constexpr (I == P::nst - 1) {                                                                          \
            if (canonOut) {                                                                                       \
                if constexpr (INV)                                                                                \
                    run_csub16(r, q);                                                                             \
                else {                                                                                            \
                    run_red16(r, c);                                                                              \
                    if constexpr (!EPI) /* the epilogue takes values below 2q (A + 2q - r, then a Shoup product) */ \
                        run_csub16(r, q);                                                                         \
                }                                                                                                 \
            }                                                                                                     \
            else if constexpr (!INV && LA)                                                                        \
                run_red16(r, c); /* the (HBM-bound) column pass hands the row pass values below 2q */             \
        }                                                                                                         \
        if constexpr (I == P::nst - 1 && RAWOUT) {                                                                \
            /* the caller takes the residues from r[] */                                                          \
        }                                                                                                         \
        else if constexpr (I == P::nst - 1 && !P::stageLast) {                                                    \
            store_result(r, jrel, ks);                                                                            \
        }                                                                                                         \
        else {                                                                                                    \
            if constexpr (I > 0 || P::stageFirst || RAWIN)                                                        \
                FHE_SSYNC(); /* every lane has finished reading the buffer */                                      \
            uint64_t* L = lds + lds_pad(Ib);                                                 \
            _Pragma("unroll") for (int k = 0; k < 16; ++k) FHE_LDS_ST(L[lds_pad((uint32_t)k << fI)], r[k]);                  \
            FHE_SSYNC();                                                                                           \
        }                                                                                                         \
    }
    FHE_STATIC_STEP(0)
    FHE_STATIC_STEP(1)
    FHE_STATIC_STEP(2)
#undef FHE_STATIC_STEP
#undef FHE_SHARED_TW_TO_LDS

    // ---- last store ----
    if constexpr (P::stageLast && !RAWOUT) {
        lane_geom_s<LA, T, 8>(t, S, Ib, jrel, ks);
        const uint64_t* L = lds + lds_pad(Ib);
#pragma unroll
        for (int k = 0; k < 16; ++k)
            FHE_LDS_LD(r[k], L[lds_pad((uint32_t)k << 8)]);
        store_result(r, jrel, ks);
    }
}

template <bool LA, bool INV, int T, int MODE, bool EPI = false, bool PRO = false>
FHE_DEV void ntt_static_body(const NttPassArgs& a, uint32_t bid, uint64_t* lds) {
    uint64_t r[16];
    ntt_static_core<LA, INV, T, MODE, EPI, false, false, PRO>(a, bid, lds, r);
}

template <bool LA, bool INV, int T, int MODE, bool EPI = false, bool PRO = false>
FHE_GLOBAL void FHE_LAUNCH_BOUNDS(kThreads) ntt_static_kernel(const NttPassArgs a) {
    // a single-step pass without staging (the 4-stage column pass) never touches LDS: do not reserve any, so that
    // more workgroups fit on a CU
    using P = SPlan<LA, INV, T>;
    constexpr bool needsLds = P::nst > 1 || P::stageFirst || P::stageLast;
#ifndef FHE_ABL_LDSPAD  // timing experiment: extra LDS words per workgroup (fewer workgroups per CU)
#define FHE_ABL_LDSPAD 0
#endif
    FHE_SHARED_U64(lds, needsLds ? kLdsPadWords + kSharedTwWords + FHE_ABL_LDSPAD : 1);
    ntt_static_body<LA, INV, T, MODE, EPI, PRO>(a, FHE_BID, lds);
}


// ---- fused negacyclic polynomial product  c = a * b  (a, b, c in COEFFICIENT form; SURVEY.md 8(d) "fused fwd o mul o inv") -----
// Per limb the reference computes INTT(NTT(a) o NTT(b)) with three transforms and a Hadamard product, each a round trip
// through memory.  Here the contiguous (row) passes are fused around the product:
//   column pass of a and of b (two launches of the plain column kernel, out of place into the workspace)
//   poly_mul_row_a_kernel   forward row pass of a; the tile's transform is stored in the last step's lane layout
//                           (word k*256 + t of the tile = register k of lane t): a private order, fully coalesced
//   poly_mul_row_b_kernel   forward row pass of b, canonical residues in registers; product with a's transform (same lane
//                           layout: NTT(a)[i] * NTT(b)[i] for the same i); inverse row pass straight from the registers;
//                           the forward pass's last exchange and store, the Hadamard kernel and the inverse pass's first
//                           load and exchange never happen
//   inverse column pass of c (plain kernel)
// A forward row pass ends and an inverse row pass begins on tile bit 0, i.e. with the same 16 consecutive residues per lane,
// which is what makes the register hand-over possible (transformnat-impl.h:352-373 is the forward transform's last, unit-
// stride stage, :541-567 the inverse's first).
struct PolyMulArgs {
    NttPassArgs fwd;       // forward row pass (of a: in place on the workspace; of b: input = the workspace, x unused)
    NttPassArgs inv;       // inverse row pass into c (poly_mul_row_b_kernel only)
    uint64_t* aEval;       // workspace of a: [rows][N], tiles in lane layout after poly_mul_row_a_kernel
    const LimbConst* lc;   // [ctxLimbs] Barrett constants of the product
};
template <int T>
FHE_GLOBAL void FHE_LAUNCH_BOUNDS(kThreads) poly_mul_row_a_kernel(const PolyMulArgs g) {
    FHE_SHARED_U64(lds, kLdsPadWords + kSharedTwWords);
    uint64_t r[16];
    ntt_static_core<false, false, T, 9, false, false, true>(g.fwd, FHE_BID, lds, r);  // canonical (fwd.canonStep set)
    uint64_t* dst = g.aEval + ((uint64_t)FHE_BID << kTileLog) + FHE_TID;
#pragma unroll
    for (int k = 0; k < 16; ++k)
        dst[(uint32_t)k * kThreads] = r[k];
}
template <int T>
FHE_GLOBAL void FHE_LAUNCH_BOUNDS2(kThreads, 4) poly_mul_row_b_kernel(const PolyMulArgs g) {  // 4 waves per SIMD: <= 128 VGPRs
    FHE_SHARED_U64(lds, kLdsPadWords + kSharedTwWords);
    uint64_t r[16];
    ntt_static_core<false, false, T, 9, false, false, true>(g.fwd, FHE_BID, lds, r);
    {
        const uint32_t tilesPerRow = 1u << (g.fwd.logN - (uint32_t)kTileLog);
        const uint32_t row         = FHE_BID / tilesPerRow;
        const LimbConst lc         = g.lc[FHE_UNIFORM(g.fwd.sel.idx[row % g.fwd.nLimbs])];
        const uint64_t* av         = g.aEval + ((uint64_t)FHE_BID << kTileLog) + FHE_TID;
        // groups of 4 keep the extra live registers small (the kernel must stay within 128 VGPRs: 4 waves per SIMD)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            uint64_t x[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                x[k] = av[(uint32_t)(4 * g4 + k) * kThreads];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                r[4 * g4 + k] = mul_mod_barrett(x[k], r[4 * g4 + k], lc.q, lc.mu, (int)lc.msb);
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" ::: "memory");  // (keeps the next group's loads behind this group's products)
#endif
        }
    }
    ntt_static_core<false, true, T, 0, false, true, false>(g.inv, FHE_BID, lds, r);
}

}  // namespace fhe
#endif
