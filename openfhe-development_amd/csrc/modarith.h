// modarith.h — 64-bit residue arithmetic for the DCRTPoly hot path on gfx950.
//
// Every value is a residue of a prime q < 2^60 held in one 64-bit word, one word per lane
// (reference data model: NativeIntegerT<uint64_t>, src/core/include/math/hal/intnat/ubintnat.h).
// Final results are always the canonical residue in [0,q), so they are bit-identical to the
// reference's ModMulFastConst / ModMulFast / ModAddFast / ModSubFast outputs (ubintnat.h:737-757,
// 911-930, 1348-1361, 1464-1469); intermediate values use lazy ranges [0,2q) / [0,4q), which the
// 4 spare bits of a 60-bit modulus allow.
//
// The same header compiles for the device (hipcc) and for the host (g++, used by the host-side
// table builders and by the test-only lane emulator under tests/emu/).
#ifndef FHE_MODARITH_H
#define FHE_MODARITH_H
#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP_DEVICE_COMPILE__)
#include <hip/hip_runtime.h>
#define FHE_HD __host__ __device__ __forceinline__
#else
#define FHE_HD inline
#endif

namespace fhe {

FHE_HD uint64_t mulhi64(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}

// x in [0, 2m) -> [0, m)
FHE_HD uint64_t csub(uint64_t x, uint64_t m) {
    return x >= m ? x - m : x;
}

// 32x32+64 -> 64 multiply-add: the only integer multiplier CDNA4 has (v_mad_u64_u32, measured 5.1 cycles per
// wave64 on a SIMD; v_mul_hi_u32 is slower at 7.7).  Spelled as inline asm on the device so that hipcc cannot
// re-select v_mul_hi_u32 / v_mul_lo_u32 + v_add3 for parts of the chain.
FHE_HD uint64_t mad64(uint32_t a, uint32_t b, uint64_t c) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(FHE_NO_MAD_ASM)
    uint64_t d;
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c) : "vcc");
    return d;
#else
    return (uint64_t)a * b + c;
#endif
}
FHE_HD uint64_t mul32x32(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(FHE_NO_MAD_ASM)
    uint64_t d;
    asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(d) : "v"(a), "v"(b) : "vcc");
    return d;
#else
    return (uint64_t)a * b;
#endif
}

// Shoup multiplication by a constant w with precomputed wp = floor(w * 2^64 / q)
// (ubintnat.h:1437-1444).  LAZY form: any y < 2^64, result in [0, 2q), congruent to y*w.
// Ten multiply-adds: Q = hi64(y*wp) (4), then lo64(y*w + Q*(-q)) (2 for the low word with 64-bit carry,
// 4 for the cross terms accumulated into the high word).  nq = -q mod 2^64.
FHE_HD uint64_t mul_shoup_lazy_nq(uint64_t y, uint64_t w, uint64_t wp, uint64_t nq) {
    const uint32_t yl = (uint32_t)y, yh = (uint32_t)(y >> 32), pl = (uint32_t)wp, ph = (uint32_t)(wp >> 32);
    const uint64_t p0 = mul32x32(yl, pl);
    const uint64_t p1 = mad64(yh, pl, p0 >> 32);
    const uint64_t p2 = mad64(yl, ph, (uint32_t)p1);
    const uint64_t Q  = mad64(yh, ph, p1 >> 32) + (p2 >> 32);
    const uint32_t Ql = (uint32_t)Q, Qh = (uint32_t)(Q >> 32), wl = (uint32_t)w, wh = (uint32_t)(w >> 32);
    const uint32_t nql = (uint32_t)nq, nqh = (uint32_t)(nq >> 32);
    uint64_t acc = mul32x32(yl, wl);
    acc          = mad64(Ql, nql, acc);
    uint64_t h   = acc >> 32;
    h            = mad64(yl, wh, h);
    h            = mad64(yh, wl, h);
    h            = mad64(Ql, nqh, h);
    h            = mad64(Qh, nql, h);
    return (acc & 0xffffffffull) | (h << 32);
}
FHE_HD uint64_t mul_shoup_lazy(uint64_t y, uint64_t w, uint64_t wp, uint64_t q) {
    return mul_shoup_lazy_nq(y, w, wp, 0 - q);
}
// canonical result in [0,q): same value as ModMulFastConst (ubintnat.h:1464-1469)
FHE_HD uint64_t mul_shoup(uint64_t y, uint64_t w, uint64_t wp, uint64_t q) {
    return csub(mul_shoup_lazy(y, w, wp, q), q);
}

FHE_HD uint64_t add_mod(uint64_t a, uint64_t b, uint64_t q) {  // ModAddFast, ubintnat.h:737-743
    return csub(a + b, q);
}
FHE_HD uint64_t sub_mod(uint64_t a, uint64_t b, uint64_t q) {  // ModSubFast, ubintnat.h:911-921
    return a >= b ? a - b : a + q - b;
}

// 128-bit value as two words
struct u128w {
    uint64_t lo, hi;
};
FHE_HD u128w mul128(uint64_t a, uint64_t b) {  // Mul128, utils/utilities-int.h:47-49
    u128w r;
    r.lo = a * b;
    r.hi = mulhi64(a, b);
    return r;
}
FHE_HD void acc128(u128w& s, uint64_t a, uint64_t b) {  // s += a*b
    uint64_t lo = a * b;
    uint64_t hi = mulhi64(a, b);
    s.lo += lo;
    s.hi += hi + (s.lo < lo);
}

// Column-wise 64x64 multiply-accumulate: three 64-bit column sums with explicit carry words, no shifts or
// zero-extended operands in the inner loop (8 instructions per MAC: 4 v_mad_u64_u32 + 4 carry adds).
//   value = c0 + (c1 << 32) + (c2 << 64) + carries: k0 counts 2^64 overflows of c0, k1 of c1, k2 of c2.
struct mac192 {
    uint64_t c0, c1, c2;
    uint32_t k0, k1, k2;
};
FHE_HD void mac192_clear(mac192& m) {
    m.c0 = m.c1 = m.c2 = 0;
    m.k0 = m.k1 = m.k2 = 0;
}
FHE_HD void mac192_add(mac192& m, uint64_t a, uint64_t b) {
    const uint32_t al = (uint32_t)a, ah = (uint32_t)(a >> 32), bl = (uint32_t)b, bh = (uint32_t)(b >> 32);
    uint64_t t;
    t = (uint64_t)al * bl + m.c0;
    m.k0 += t < m.c0;
    m.c0 = t;
    t    = (uint64_t)al * bh + m.c1;
    m.k1 += t < m.c1;
    m.c1 = t;
    t    = (uint64_t)ah * bl + m.c1;
    m.k1 += t < m.c1;
    m.c1 = t;
    t    = (uint64_t)ah * bh + m.c2;
    m.k2 += t < m.c2;
    m.c2 = t;
}
// mac192_add with a wave-uniform multiplier b (a table word in SGPRs): on gfx950 the carries come straight from the
// multiply-adds' carry-outs (7 VALU instructions; hipcc's version of the C++ above is 13: multiply, 64-bit add,
// compare, add-with-carry per column).  Two instructions separate every carry write from its reader.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(FHE_NO_MAD_ASM)
__device__ __forceinline__ void mac192_add_uniform(mac192& m, uint64_t a, uint64_t b) {
    const uint32_t al = (uint32_t)a, ah = (uint32_t)(a >> 32), bl = (uint32_t)b, bh = (uint32_t)(b >> 32);
    asm volatile(
        "v_mad_u64_u32 %[c0], s[42:43], %[al], %[bl], %[c0]\n\t"
        "v_mad_u64_u32 %[c1], s[44:45], %[al], %[bh], %[c1]\n\t"
        "v_mad_u64_u32 %[c2], s[40:41], %[ah], %[bh], %[c2]\n\t"
        "v_addc_co_u32_e64 %[k0], s[40:41], %[k0], 0, s[42:43]\n\t"
        "v_mad_u64_u32 %[c1], s[46:47], %[ah], %[bl], %[c1]\n\t"
        "v_addc_co_u32_e64 %[k1], s[40:41], %[k1], 0, s[44:45]\n\t"
        "s_nop 0\n\t"
        "v_addc_co_u32_e64 %[k1], s[40:41], %[k1], 0, s[46:47]\n\t"
        : [c0] "+v"(m.c0), [c1] "+v"(m.c1), [c2] "+v"(m.c2), [k0] "+v"(m.k0), [k1] "+v"(m.k1)
        : [al] "v"(al), [ah] "v"(ah), [bl] "s"(bl), [bh] "s"(bh)
        : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47");
}
#else
FHE_HD void mac192_add_uniform(mac192& m, uint64_t a, uint64_t b) {
    mac192_add(m, a, b);
}
#endif
// fold the columns into a 128-bit value (the true sum must be < 2^128)
struct u128w;
FHE_HD void mac192_fold(const mac192& m, uint64_t& lo, uint64_t& hi) {
    // total = c0 + c1*2^32 + c2*2^64 + k0*2^64 + k1*2^96 + k2*2^128(=0 by assumption)
    const uint64_t c1lo = m.c1 << 32, c1hi = (m.c1 >> 32) + ((uint64_t)m.k1 << 32);
    lo = m.c0 + c1lo;
    hi = m.c2 + c1hi + (uint64_t)m.k0 + (lo < c1lo);
}

// ---- sums of at most 8 products of residues below 2^60 (the reference's MAX_MODULUS_SIZE; fhe_ctx_create enforces it) ----
// With x = xh*2^32 + xl, y = yh*2^32 + yl: xl*yh, xh*yl < 2^60 and xh*yh < 2^56, so the 16 middle products and the 8 high
// products of a chunk accumulate in plain 64-bit words; only the low column needs a carry count (7 instructions per term).
struct sum8 {
    uint64_t c0, c1, c2;
    uint32_t k0;
};
FHE_HD void sum8_clear(sum8& s) {
    s.c0 = s.c1 = s.c2 = 0;
    s.k0 = 0;
}
FHE_HD void sum8_add(sum8& s, uint64_t x, uint64_t y) {
    const uint32_t xl = (uint32_t)x, xh = (uint32_t)(x >> 32), yl = (uint32_t)y, yh = (uint32_t)(y >> 32);
    const uint64_t lo = (uint64_t)xl * yl + s.c0;
    s.k0 += lo < s.c0;
    s.c0 = lo;
    s.c1 += (uint64_t)xl * yh;
    s.c1 += (uint64_t)xh * yl;
    s.c2 += (uint64_t)xh * yh;
}
// The exact residue mod q (k = bit length of q <= 60) of a sum S of <= 8 products a_i*b_i with a_i < 2^60 and b_i < q
// (so S < 2^(k+63); in particular <= 8 products of residues of q):
//   s    = floor(S / 2^(k-1)) < 2^64,   mu' = floor(2^(k+63) / q) = floor(mu128 / 2^(65-k)) < 2^64
//   qhat = floor(s * mu' / 2^64):  S/q - 3 < qhat <= S/q   (one unit each from the two floors inside and the one outside)
//   r    = S - qhat*q in [0, 3q), 3q < 2^62: the low words suffice; two conditional subtractions make it canonical.
// Any exact reduction equals the reference's ModMul / ModAdd chain (mubintvecnat.cpp:229-339) and its
// BarrettUint128ModUint64 (utils/utilities-int.h:60-99).
FHE_HD uint64_t sum8_reduce(const sum8& s, uint64_t q, uint32_t k, uint64_t mu_lo, uint64_t mu_hi) {
    const uint64_t m = s.c1 << 32, lo = s.c0 + m;
    const uint64_t hi = s.c2 + (s.c1 >> 32) + s.k0 + (lo < m);
    const uint32_t sh = 65u - k;  // 5..61 (k = 60..4)
    const uint64_t mu = (mu_lo >> sh) | (mu_hi << (64u - sh));
    const uint64_t sv = (lo >> (k - 1u)) | (hi << sh);
    uint64_t r        = lo - mulhi64(sv, mu) * q;
    r -= r >= q ? q : 0;
    r -= r >= q ? q : 0;
    return r;
}

// The same sums with both factors split at 30 bits (x = x1*2^30 + x0 for x < 2^60): every partial product is below 2^60,
// so <= 8 terms accumulate in three plain 64-bit columns with NO carry bookkeeping at all (4 multiply-adds per term);
// S = c0 + c1*2^30 + c2*2^60 is assembled once.  A wave-uniform factor is split once per table entry, a per-lane factor
// once per loaded residue.  This is the DEFAULT form of every conversion kernel since round 2 (measured on the MI355X: 83 instead of
// 103 VALU instructions per output residue; FHE_CONV_SUM8=1 selects the carry-counted sum8 above for A/B runs).
struct sum8s {
    uint64_t c0, c1, c2;
};
FHE_HD void sum8s_clear(sum8s& s) { s.c0 = s.c1 = s.c2 = 0; }
FHE_HD void split30(uint64_t x, uint32_t& x0, uint32_t& x1) {
    x0 = (uint32_t)x & 0x3fffffffu;
    x1 = (uint32_t)(x >> 30);
}
FHE_HD void sum8s_add(sum8s& s, uint32_t x0, uint32_t x1, uint32_t y0, uint32_t y1) {
    s.c0 += (uint64_t)x0 * y0;
    s.c1 += (uint64_t)x0 * y1;
    s.c1 += (uint64_t)x1 * y0;
    s.c2 += (uint64_t)x1 * y1;
}
FHE_HD uint64_t sum8s_reduce(const sum8s& s, uint64_t q, uint32_t k, uint64_t mu_lo, uint64_t mu_hi) {
    const uint64_t t1 = s.c1 << 30, l1 = s.c0 + t1, t2 = s.c2 << 60, lo = l1 + t2;
    const uint64_t hi = (s.c1 >> 34) + (s.c2 >> 4) + (l1 < t1) + (lo < t2);
    const uint32_t sh = 65u - k;
    const uint64_t mu = (mu_lo >> sh) | (mu_hi << (64u - sh));
    const uint64_t sv = (lo >> (k - 1u)) | (hi << sh);
    uint64_t r        = lo - mulhi64(sv, mu) * q;
    r -= r >= q ? q : 0;
    r -= r >= q ? q : 0;
    return r;
}

// a (128-bit) mod q with mu = floor(2^128/q) given as (mu_lo, mu_hi).
// Same quotient estimate as BarrettUint128ModUint64 (utils/utilities-int.h:60-99): the low word of
// floor(a*mu / 2^128), then r = a_lo - quot*q and final corrective subtractions.
FHE_HD uint64_t barrett128(u128w a, uint64_t q, uint64_t mu_lo, uint64_t mu_hi) {
    uint64_t left_hi = mulhi64(a.lo, mu_lo);
    uint64_t mid_lo  = a.lo * mu_hi;
    uint64_t mid_hi  = mulhi64(a.lo, mu_hi);
    uint64_t tmp1    = mid_lo + left_hi;
    uint64_t tmp2    = mid_hi + (tmp1 < mid_lo);
    uint64_t m2_lo   = a.hi * mu_lo;
    uint64_t m2_hi   = mulhi64(a.hi, mu_lo);
    uint64_t carry   = (uint64_t)(m2_lo + tmp1) < m2_lo;
    uint64_t quot    = a.hi * mu_hi + tmp2 + m2_hi + carry;
    uint64_t r       = a.lo - quot * q;
    // quot is the exact floor(a*mu/2^128) with mu = floor(2^128/q), so floor(a/q) - quot is 0 or 1 (a < 2^128) and
    // r < 2q; the reference's `while (r >= q) r -= q` therefore runs at most once.  Two conditional subtractions keep a
    // margin and avoid a divergent loop.
    return csub(csub(r, q), q);
}

// exact a*b mod q for a,b < q < 2^60 via the 128-bit Barrett above (any exact product equals the
// reference's ModMulFast result)
FHE_HD uint64_t mul_mod(uint64_t a, uint64_t b, uint64_t q, uint64_t mu_lo, uint64_t mu_hi) {
    return barrett128(mul128(a, b), q, mu_lo, mu_hi);
}

// Single-word Barrett for a*b with a,b < q < 2^62: mu64 = floor(2^(2*nb)/q) style is avoided; instead
// use the reference's own generalized Barrett (ubintnat.h:1348-1361) with mu = floor(2^(2*msb+3)/q):
//   t = (a*b) >> (msb-2);  est = (t*mu) >> (msb+5);  r = a*b - est*q;  r in [0, 2q)
FHE_HD uint64_t mul_mod_barrett(uint64_t a, uint64_t b, uint64_t q, uint64_t mu, int msb) {
    uint64_t lo = a * b, hi = mulhi64(a, b);
    int n       = msb - 2;
    uint64_t t  = (lo >> n) | (hi << (64 - n));  // low 64 bits of (prod >> n); msb>=3 so 0<n<64
    // est = (t*mu) >> (n+7)   (128-bit product, shift n+7 in (7, 69))
    uint64_t plo = t * mu, phi = mulhi64(t, mu);
    int sh       = n + 7;
    uint64_t est = sh < 64 ? ((plo >> sh) | (phi << (64 - sh))) : (phi >> (sh - 64));
    uint64_t r   = lo - est * q;
    return csub(r, q);
}

}  // namespace fhe
#endif
