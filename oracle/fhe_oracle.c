/*
 * fhe_oracle.c — TEST INFRASTRUCTURE ONLY (see fhe_oracle.h).
 *
 * Plain-C restatement of the reference's CPU algorithms for the lbcrypto::DCRTPoly hot path.
 * Each function cites the reference file:line (relative to /root/reference/) it follows.
 * Written from the algorithm description in SURVEY.md Appendix A; loop structure follows the
 * reference so that operation ORDER (which matters for the double-precision paths) is the same.
 *
 * Parity status: PINNED (tests/test_oracle_golden.py, tests/test_oracle_vs_ref.py).
 */
#include "fhe_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* scratch arrays of the per-coefficient loops: limbs of one tower (the reference has no bound; the tests go to 256) */
#define ORC_MAX_LIMBS 256

typedef unsigned __int128 u128;

/* ------------------------------------------------------------------------------------------
 * a1: scalar arithmetic
 * ---------------------------------------------------------------------------------------- */
uint64_t orc_mulmod(uint64_t a, uint64_t b, uint64_t q) {
    return (uint64_t)(((u128)a * b) % q);
}

uint64_t orc_powmod(uint64_t a, uint64_t e, uint64_t q) {
    uint64_t r = 1 % q;
    a %= q;
    while (e) {
        if (e & 1)
            r = orc_mulmod(r, a, q);
        a = orc_mulmod(a, a, q);
        e >>= 1;
    }
    return r;
}

uint64_t orc_invmod(uint64_t a, uint64_t q) {
    /* q prime on this path (all moduli are NTT primes); Fermat */
    return orc_powmod(a % q, q - 2, q);
}

/* nbtheory.h:169-186 — GetMSB: index of the most significant set bit, 1-based; 0 for x == 0 */
uint32_t orc_get_msb(uint64_t x) {
    uint32_t r = 0;
    while (x) {
        ++r;
        x >>= 1;
    }
    return r;
}

/* ubintnat.h:642-647 — mu = floor(2^(2*msb+3) / q) */
uint64_t orc_compute_mu(uint64_t q) {
    u128 t = (u128)1 << (2 * orc_get_msb(q) + 3);
    return (uint64_t)(t / q);
}

/* ubintnat.h:1348-1361 — generalized Barrett, alpha = n+3 */
uint64_t orc_mod_mul_fast(uint64_t a, uint64_t b, uint64_t q, uint64_t mu) {
    int64_t n = (int64_t)orc_get_msb(q) - 2;
    u128 prod = (u128)a * b;
    u128 rv   = prod;
    u128 t    = (u128)(uint64_t)(prod >> n) * mu; /* RShiftD keeps the low 64 bits of (prod >> n) */
    rv -= (u128)q * (u128)(t >> (n + 7));
    uint64_t r = (uint64_t)rv;
    if (r >= q)
        r -= q;
    return r;
}

/* ubintnat.h:1437-1444 — Shoup precomputation floor(b * 2^64 / q) */
uint64_t orc_prep_mod_mul_const(uint64_t b, uint64_t q) {
    return (uint64_t)((((u128)b) << 64) / q);
}

/* ubintnat.h:1464-1469 — Shoup multiplication, result in [0,q) for a < q */
uint64_t orc_mod_mul_fast_const(uint64_t a, uint64_t b, uint64_t q, uint64_t bPrecon) {
    uint64_t qq    = (uint64_t)(((u128)a * bPrecon) >> 64) + 1;
    int64_t yprime = (int64_t)(a * b - qq * q);
    return (uint64_t)(yprime >= 0 ? yprime : yprime + (int64_t)q);
}

/* ubintnat.h:737-743 */
uint64_t orc_mod_add_fast(uint64_t a, uint64_t b, uint64_t q) {
    uint64_t r = a + b;
    if (r >= q)
        r -= q;
    return r;
}

/* ubintnat.h:911-921 */
uint64_t orc_mod_sub_fast(uint64_t a, uint64_t b, uint64_t q) {
    if (a < b)
        return a + q - b;
    return a - b;
}

/* a2: mu128 = floor(2^128 / q) as the reference computes it from BigInteger(1)<<128
 * (rns-cryptoparameters.cpp:288-293).  2^128 / q = floor((2^128 - 1) / q) unless q | 2^128, impossible for odd q. */
void orc_barrett_mu128(uint64_t q, uint64_t mu[2]) {
    u128 m = (~(u128)0) / q;
    mu[0]  = (uint64_t)m;
    mu[1]  = (uint64_t)(m >> 64);
}

/* utilities-int.h:60-99 — BarrettUint128ModUint64 */
uint64_t orc_barrett128(uint64_t a_lo, uint64_t a_hi, uint64_t q, uint64_t mu_lo, uint64_t mu_hi) {
    uint64_t left_hi = (uint64_t)(((u128)a_lo * mu_lo) >> 64);
    u128 middle      = (u128)a_lo * mu_hi;
    uint64_t mid_lo  = (uint64_t)middle;
    uint64_t mid_hi  = (uint64_t)(middle >> 64);
    uint64_t tmp1    = mid_lo + left_hi;
    uint64_t carry   = tmp1 < mid_lo;
    uint64_t tmp2    = mid_hi + carry;
    middle           = (u128)a_hi * mu_lo;
    mid_lo           = (uint64_t)middle;
    mid_hi           = (uint64_t)(middle >> 64);
    carry            = (uint64_t)(mid_lo + tmp1) < mid_lo;
    left_hi          = mid_hi + carry;
    tmp1             = a_hi * mu_hi + tmp2 + left_hi;
    uint64_t r       = a_lo - tmp1 * q;
    while (r >= q)
        r -= q;
    return r;
}

/* ------------------------------------------------------------------------------------------
 * number theory (parameter reproduction)
 * ---------------------------------------------------------------------------------------- */
/* Deterministic Miller-Rabin for 64-bit inputs.  The reference uses a probabilistic
 * MillerRabinPrimalityTest (nbtheory-impl.h:120-160); both agree on every 64-bit input
 * that matters (a deterministic witness set has no false positives or negatives < 2^64). */
int orc_is_prime(uint64_t n) {
    static const uint64_t small[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
    if (n < 2)
        return 0;
    for (size_t i = 0; i < sizeof(small) / sizeof(small[0]); ++i) {
        if (n == small[i])
            return 1;
        if (n % small[i] == 0)
            return 0;
    }
    uint64_t d = n - 1;
    int s      = 0;
    while ((d & 1) == 0) {
        d >>= 1;
        ++s;
    }
    for (size_t i = 0; i < sizeof(small) / sizeof(small[0]); ++i) {
        uint64_t x = orc_powmod(small[i], d, n);
        if (x == 1 || x == n - 1)
            continue;
        int comp = 1;
        for (int r = 1; r < s; ++r) {
            x = orc_mulmod(x, x, n);
            if (x == n - 1) {
                comp = 0;
                break;
            }
        }
        if (comp)
            return 0;
    }
    return 1;
}

/* nbtheory-impl.h:329-347 */
uint64_t orc_first_prime(uint32_t nBits, uint64_t m) {
    uint64_t q    = (uint64_t)1 << nBits;
    uint64_t r    = q % m;
    uint64_t qNew = q + 1 - r;
    if (r > 0)
        qNew += m;
    while (!orc_is_prime(qNew))
        qNew += m;
    return qNew;
}

/* nbtheory-impl.h:349-373 */
uint64_t orc_last_prime(uint32_t nBits, uint64_t m) {
    uint64_t q    = (uint64_t)1 << nBits;
    uint64_t r    = q % m;
    uint64_t qNew = q + 1 - r;
    if (r < 2)
        qNew -= m;
    while (!orc_is_prime(qNew))
        qNew -= m;
    return qNew;
}

/* nbtheory-impl.h:375-383 */
uint64_t orc_next_prime(uint64_t q, uint64_t m) {
    uint64_t qNew = q + m;
    while (!orc_is_prime(qNew))
        qNew += m;
    return qNew;
}

/* nbtheory-impl.h:385-393 */
uint64_t orc_previous_prime(uint64_t q, uint64_t m) {
    uint64_t qNew = q - m;
    while (!orc_is_prime(qNew))
        qNew -= m;
    return qNew;
}

/* nbtheory-impl.h:183-231 — the MINIMUM primitive m-th root of unity mod q (m a power of two).
 * The reference finds one primitive root from a random generator and then cycles through all
 * powers coprime to m keeping the smallest; the minimum does not depend on the starting root,
 * so any primitive root serves as the start. */
uint64_t orc_root_of_unity(uint64_t m, uint64_t q) {
    if ((q - 1) % m != 0)
        return 0;
    uint64_t e = (q - 1) / m;
    uint64_t root = 0;
    for (uint64_t g = 2; g < q; ++g) {
        uint64_t r = orc_powmod(g, e, q);
        /* primitive m-th root (m power of two) iff r^(m/2) == -1 */
        if (orc_powmod(r, m / 2, q) == q - 1) {
            root = r;
            break;
        }
    }
    /* all primitive roots = odd powers of root */
    uint64_t sq    = orc_mulmod(root, root, q);
    uint64_t x     = root;
    uint64_t minRU = x;
    for (uint64_t i = 1; i < m / 2; ++i) {
        x = orc_mulmod(x, sq, q);
        if (x < minRU)
            minRU = x;
    }
    return minRU;
}

/* nbtheory.h:135-157 */
uint32_t orc_reverse_bits(uint32_t x, uint32_t nbits) {
    uint32_t r = 0;
    for (uint32_t i = 0; i < nbits; ++i)
        r |= ((x >> i) & 1u) << (nbits - 1 - i);
    return r;
}

/* nbtheory2.cpp:264-275 */
void orc_precompute_auto_map(uint32_t n, uint32_t k, uint32_t* precomp) {
    uint32_t m    = n << 1;
    uint32_t logm = orc_get_msb(m) - 1;
    uint32_t logn = logm - 1;
    for (uint32_t j = 0; j < n; ++j) {
        uint32_t jTmp   = (j << 1) + 1;
        uint32_t prod   = jTmp * k; /* 32-bit wrap-around as in the reference */
        uint32_t idx    = (prod - ((prod >> logm) << logm)) >> 1;
        uint32_t jrev   = orc_reverse_bits(j, logn);
        uint32_t idxrev = orc_reverse_bits(idx, logn);
        precomp[jrev]   = idxrev;
    }
}

/* nbtheory2.cpp:243-262 */
uint32_t orc_find_automorphism_index_2n_complex(int32_t i, uint32_t m) {
    if (i == 0)
        return 1;
    if (i == (int32_t)m - 1)
        return (uint32_t)i;
    uint64_t g0 = 5;
    if (i < 0) {
        /* 5^-1 mod m, m power of two: Newton iteration */
        uint64_t inv = 1;
        for (int it = 0; it < 6; ++it)
            inv = (inv * (2 - 5 * inv)) & (m - 1);
        g0 = inv;
    }
    uint64_t g  = g0;
    uint32_t iu = (uint32_t)(i < 0 ? -i : i);
    for (uint32_t j = 1; j < iu; ++j)
        g = (g * g0) & (m - 1);
    return (uint32_t)g;
}

/* ildcrtparams.h:100-117 — LastPrime then PreviousPrime chain, roots via RootOfUnity
 * (ilparams.h ctor -> RootOfUnity(order, modulus)) */
void orc_dcrt_params(uint32_t order, uint32_t nLimbs, uint32_t bits, uint64_t* q, uint64_t* psi) {
    uint64_t cur = orc_last_prime(bits, order);
    for (uint32_t i = 0; i < nLimbs; ++i) {
        if (i > 0)
            cur = orc_previous_prime(cur, order);
        q[i]   = cur;
        psi[i] = orc_root_of_unity(order, cur);
    }
}

/* ------------------------------------------------------------------------------------------
 * a3..a5: NTT
 * ---------------------------------------------------------------------------------------- */
/* transformnat-impl.h:714-756 */
void orc_ntt_precompute(uint64_t q, uint64_t psi, uint32_t N, uint64_t* tbl, uint64_t* tblPrecon,
                        uint64_t* tblInv, uint64_t* tblInvPrecon, uint64_t* nInv, uint64_t* nInvPrecon) {
    uint32_t msb    = orc_get_msb(N - 1);
    uint64_t psiInv = orc_invmod(psi, q);
    uint64_t x = 1, xinv = 1;
    for (uint32_t i = 0; i < N; ++i) {
        uint32_t iinv      = orc_reverse_bits(i, msb);
        tbl[iinv]          = x;
        tblPrecon[iinv]    = orc_prep_mod_mul_const(x, q);
        x                  = orc_mulmod(x, psi, q);
        tblInv[iinv]       = xinv;
        tblInvPrecon[iinv] = orc_prep_mod_mul_const(xinv, q);
        xinv               = orc_mulmod(xinv, psiInv, q);
    }
    /* TableCOI[msb] = (2^msb)^-1 = N^-1  (:743-750) */
    *nInv       = orc_invmod((uint64_t)N % q, q);
    *nInvPrecon = orc_prep_mod_mul_const(*nInv, q);
}

/* transformnat-impl.h:303-374 (GNUC branch) */
void orc_ntt_fwd(uint64_t* x, uint32_t N, uint64_t q, const uint64_t* tbl, const uint64_t* tblPrecon) {
    uint32_t n = N >> 1;
    uint32_t t = n, logt = orc_get_msb(n);
    for (uint32_t m = 1; m < n; m <<= 1, t >>= 1, --logt) {
        for (uint32_t i = 0; i < m; ++i) {
            uint64_t omega = tbl[i + m], pre = tblPrecon[i + m];
            uint32_t j1 = i << logt, j2 = j1 + t;
            for (; j1 < j2; ++j1) {
                uint64_t of = orc_mod_mul_fast_const(x[j1 + t], omega, q, pre);
                uint64_t lo = x[j1];
                uint64_t hi = lo + of;
                if (hi >= q)
                    hi -= q;
                if (lo < of)
                    lo += q;
                lo -= of;
                x[j1]     = hi;
                x[j1 + t] = lo;
            }
        }
    }
    for (uint32_t i = 0; i < (n << 1); i += 2) {
        uint64_t omega = tbl[(i >> 1) + n], pre = tblPrecon[(i >> 1) + n];
        uint64_t of = orc_mod_mul_fast_const(x[i + 1], omega, q, pre);
        uint64_t lo = x[i];
        uint64_t hi = lo + of;
        if (hi >= q)
            hi -= q;
        if (lo < of)
            lo += q;
        lo -= of;
        x[i]     = hi;
        x[i + 1] = lo;
    }
}

/* transformnat-impl.h:512-625 (GNUC branch) */
void orc_ntt_inv(uint64_t* x, uint32_t N, uint64_t q, const uint64_t* tblInv, const uint64_t* tblInvPrecon,
                 uint64_t nInv, uint64_t nInvPrecon) {
    uint32_t n         = N;
    uint64_t omega1Inv = orc_mod_mul_fast_const(tblInv[1], nInv, q, nInvPrecon);
    uint64_t pre1Inv   = orc_prep_mod_mul_const(omega1Inv, q);
    if (n > 2) {
        for (uint32_t i = 0; i < n; i += 2) {
            uint64_t omega = tblInv[(i + n) >> 1], pre = tblInvPrecon[(i + n) >> 1];
            uint64_t lo = x[i], hi = x[i + 1];
            uint64_t of = lo;
            if (of < hi)
                of += q;
            of -= hi;
            lo += hi;
            if (lo >= q)
                lo -= q;
            x[i]     = lo;
            x[i + 1] = orc_mod_mul_fast_const(of, omega, q, pre);
        }
    }
    uint32_t t = 2, logt = 2;
    for (uint32_t m = n >> 2; m > 1; m >>= 1, t <<= 1, ++logt) {
        for (uint32_t i = 0; i < m; ++i) {
            uint64_t omega = tblInv[i + m], pre = tblInvPrecon[i + m];
            uint32_t j1 = i << logt, j2 = j1 + t;
            for (; j1 < j2; ++j1) {
                uint64_t lo = x[j1], hi = x[j1 + t];
                uint64_t of = lo;
                if (of < hi)
                    of += q;
                of -= hi;
                lo += hi;
                if (lo >= q)
                    lo -= q;
                x[j1]     = lo;
                x[j1 + t] = orc_mod_mul_fast_const(of, omega, q, pre);
            }
        }
    }
    uint32_t j2 = n >> 1;
    for (uint32_t j1 = 0; j1 < j2; ++j1) {
        uint64_t lo = x[j1], hi = x[j1 + j2];
        uint64_t of = lo;
        if (of < hi)
            of += q;
        of -= hi;
        lo += hi;
        if (lo >= q)
            lo -= q;
        x[j1]      = lo;
        x[j1 + j2] = orc_mod_mul_fast_const(of, omega1Inv, q, pre1Inv);
    }
    for (uint32_t i = 0; i < j2; ++i)
        x[i] = orc_mod_mul_fast_const(x[i], nInv, q, nInvPrecon);
}

struct orc_ctx {
    uint32_t N, L;
    uint64_t *q, *psi;
    uint64_t *tbl, *pre, *tblInv, *preInv; /* [L][N] */
    uint64_t *nInv, *nInvPre;
};

orc_ctx* orc_ctx_create(uint32_t N, uint32_t nLimbs, const uint64_t* q, const uint64_t* psi) {
    orc_ctx* c = (orc_ctx*)calloc(1, sizeof(orc_ctx));
    c->N       = N;
    c->L       = nLimbs;
    c->q       = (uint64_t*)malloc(sizeof(uint64_t) * nLimbs);
    c->psi     = (uint64_t*)malloc(sizeof(uint64_t) * nLimbs);
    c->nInv    = (uint64_t*)malloc(sizeof(uint64_t) * nLimbs);
    c->nInvPre = (uint64_t*)malloc(sizeof(uint64_t) * nLimbs);
    size_t sz  = sizeof(uint64_t) * (size_t)nLimbs * N;
    c->tbl     = (uint64_t*)malloc(sz);
    c->pre     = (uint64_t*)malloc(sz);
    c->tblInv  = (uint64_t*)malloc(sz);
    c->preInv  = (uint64_t*)malloc(sz);
    memcpy(c->q, q, sizeof(uint64_t) * nLimbs);
    memcpy(c->psi, psi, sizeof(uint64_t) * nLimbs);
#pragma omp parallel for schedule(dynamic)
    for (uint32_t i = 0; i < nLimbs; ++i)
        orc_ntt_precompute(q[i], psi[i], N, c->tbl + (size_t)i * N, c->pre + (size_t)i * N,
                           c->tblInv + (size_t)i * N, c->preInv + (size_t)i * N, &c->nInv[i], &c->nInvPre[i]);
    return c;
}

void orc_ctx_destroy(orc_ctx* c) {
    if (!c)
        return;
    free(c->q);
    free(c->psi);
    free(c->nInv);
    free(c->nInvPre);
    free(c->tbl);
    free(c->pre);
    free(c->tblInv);
    free(c->preInv);
    free(c);
}
uint32_t orc_ctx_n(const orc_ctx* c) { return c->N; }
uint32_t orc_ctx_limbs(const orc_ctx* c) { return c->L; }
uint64_t orc_ctx_modulus(const orc_ctx* c, uint32_t limb) { return c->q[limb]; }

static void ctx_fwd(const orc_ctx* c, uint64_t* x, uint32_t limb) {
    orc_ntt_fwd(x, c->N, c->q[limb], c->tbl + (size_t)limb * c->N, c->pre + (size_t)limb * c->N);
}
static void ctx_inv(const orc_ctx* c, uint64_t* x, uint32_t limb) {
    orc_ntt_inv(x, c->N, c->q[limb], c->tblInv + (size_t)limb * c->N, c->preInv + (size_t)limb * c->N,
                c->nInv[limb], c->nInvPre[limb]);
}

/* dcrtpoly-impl.h:1932-1940 — SwitchFormat: omp parallel for over limbs, per polynomial */
void orc_ntt_fwd_tower(const orc_ctx* c, uint64_t* x, const uint32_t* limbIdx, uint32_t nSel, uint32_t batch,
                       int nThreads) {
#ifdef _OPENMP
    if (nThreads <= 0)
        nThreads = omp_get_max_threads();
#endif
    for (uint32_t b = 0; b < batch; ++b) {
#pragma omp parallel for num_threads(nThreads)
        for (uint32_t i = 0; i < nSel; ++i)
            ctx_fwd(c, x + ((size_t)b * nSel + i) * c->N, limbIdx ? limbIdx[i] : i);
    }
}

void orc_ntt_inv_tower(const orc_ctx* c, uint64_t* x, const uint32_t* limbIdx, uint32_t nSel, uint32_t batch,
                       int nThreads) {
#ifdef _OPENMP
    if (nThreads <= 0)
        nThreads = omp_get_max_threads();
#endif
    for (uint32_t b = 0; b < batch; ++b) {
#pragma omp parallel for num_threads(nThreads)
        for (uint32_t i = 0; i < nSel; ++i)
            ctx_inv(c, x + ((size_t)b * nSel + i) * c->N, limbIdx ? limbIdx[i] : i);
    }
}

/* ------------------------------------------------------------------------------------------
 * a7: element-wise ops
 * ---------------------------------------------------------------------------------------- */
/* mubintvecnat.cpp:229-236 */
void orc_vec_add(uint64_t* out, const uint64_t* a, const uint64_t* b, size_t n, uint64_t q) {
    for (size_t i = 0; i < n; ++i)
        out[i] = orc_mod_add_fast(a[i], b[i], q);
}
/* mubintvecnat.cpp:273-279 */
void orc_vec_sub(uint64_t* out, const uint64_t* a, const uint64_t* b, size_t n, uint64_t q) {
    for (size_t i = 0; i < n; ++i)
        out[i] = orc_mod_sub_fast(a[i], b[i], q);
}
/* mubintvecnat.cpp:325-339 — Barrett with mu computed per call */
void orc_vec_mul(uint64_t* out, const uint64_t* a, const uint64_t* b, size_t n, uint64_t q) {
    uint64_t mu = orc_compute_mu(q);
    for (size_t i = 0; i < n; ++i)
        out[i] = orc_mod_mul_fast(a[i], b[i], q, mu);
}
/* mubintvecnat.cpp:295-304 — Shoup with precon computed per call */
void orc_vec_mul_const(uint64_t* out, const uint64_t* a, uint64_t c, size_t n, uint64_t q) {
    if (c >= q)
        c %= q;
    uint64_t pre = orc_prep_mod_mul_const(c, q);
    for (size_t i = 0; i < n; ++i)
        out[i] = orc_mod_mul_fast_const(a[i], c, q, pre);
}
/* mubintvecnat.cpp:132-142 */
void orc_vec_mult_acc(uint64_t* acc, const uint64_t* v, uint64_t c, size_t n, uint64_t q) {
    if (c >= q)
        c %= q;
    uint64_t pre = orc_prep_mod_mul_const(c, q);
    for (size_t i = 0; i < n; ++i)
        acc[i] = orc_mod_add_fast(acc[i], orc_mod_mul_fast_const(v[i], c, q, pre), q);
}
/* PolyImpl::Plus(Integer) (poly-impl.h:211-218): ModAdd on every word in EVALUATION (mubintvecnat.cpp ModAdd: the constant
 * reduced first), ModAddAtIndex(0, .) in COEFFICIENT */
void orc_vec_add_const(uint64_t* out, const uint64_t* a, uint64_t c, size_t n, uint64_t q, int coeff0Only) {
    if (c >= q)
        c %= q;
    for (size_t i = 0; i < n; ++i)
        out[i] = (coeff0Only && i != 0) ? a[i] : orc_mod_add_fast(a[i], c, q);
}
/* PolyImpl::Minus(Integer) (poly-impl.h:221-225): ModSub on every word in both formats */
void orc_vec_sub_const(uint64_t* out, const uint64_t* a, uint64_t c, size_t n, uint64_t q) {
    if (c >= q)
        c %= q;
    for (size_t i = 0; i < n; ++i)
        out[i] = orc_mod_sub_fast(a[i], c, q);
}
/* keyswitch-hybrid.cpp:419-430 for one limb: out[r] = sum_t x_t[r] * k_t[r] (NativePoly products, operator+=) */
void orc_vec_inner_product(uint64_t* out, const uint64_t* const* x, const uint64_t* const* k, uint32_t nTerms, size_t n, uint64_t q) {
    for (size_t r = 0; r < n; ++r) {
        uint64_t acc = 0;
        for (uint32_t t = 0; t < nTerms; ++t)
            acc = orc_mod_add_fast(acc, orc_mulmod(x[t][r], k[t][r], q), q);
        out[r] = acc;
    }
}
/* dcrtpoly-impl.h:347-354 -> poly Negate -> q - v (0 stays 0 via ModSub(0, v)) */
void orc_vec_neg(uint64_t* out, const uint64_t* a, size_t n, uint64_t q) {
    for (size_t i = 0; i < n; ++i)
        out[i] = orc_mod_sub_fast(0, a[i], q) % q;
}

/* ------------------------------------------------------------------------------------------
 * a8: automorphism
 * ---------------------------------------------------------------------------------------- */
/* poly-impl.h:366-376 */
void orc_automorph_eval(uint64_t* out, const uint64_t* in, uint32_t N, const uint32_t* precomp) {
    for (uint32_t j = 0; j < N; ++j)
        out[j] = in[precomp[j]];
}
/* poly-impl.h:345-353 (EVALUATION branch without a table) */
void orc_automorph_eval_k(uint64_t* out, const uint64_t* in, uint32_t N, uint32_t k) {
    uint32_t logn = orc_get_msb(N) - 1;
    uint32_t mask = (1u << logn) - 1;
    uint32_t jk   = k;
    for (uint32_t j = 0; j < N; ++j, jk += 2 * k) {
        uint32_t jrev   = orc_reverse_bits(j, logn);
        uint32_t idxrev = orc_reverse_bits((jk >> 1) & mask, logn);
        out[jrev]       = in[idxrev];
    }
}
/* poly-impl.h:359-362 (COEFFICIENT branch; q - 0 = q is stored unreduced, as the reference does) */
void orc_automorph_coeff(uint64_t* out, const uint64_t* in, uint32_t N, uint32_t k, uint64_t q) {
    uint32_t logn = orc_get_msb(N) - 1;
    uint32_t mask = (1u << logn) - 1;
    uint32_t jk   = 0;
    for (uint32_t j = 0; j < N; ++j, jk += k)
        out[jk & mask] = ((jk >> logn) & 1u) ? q - in[j] : in[j];
}

/* ------------------------------------------------------------------------------------------
 * a9: centred modulus switch — mubintvecnat.cpp:109-122
 * ---------------------------------------------------------------------------------------- */
void orc_switch_modulus(uint64_t* v, size_t n, uint64_t oldq, uint64_t newq) {
    uint64_t halfQ = oldq >> 1;
    uint64_t diff  = (oldq > newq) ? (oldq - newq) : (newq - oldq);
    if (newq > oldq) {
        for (size_t i = 0; i < n; ++i)
            v[i] += (v[i] > halfQ) ? diff : 0;
    }
    else {
        /* ModSubEq (ubintnat.h:889-899): both operands reduced mod newq first */
        for (size_t i = 0; i < n; ++i) {
            uint64_t av = v[i];
            uint64_t bv = (v[i] > halfQ) ? diff : 0;
            if (av >= newq)
                av %= newq;
            if (bv >= newq)
                bv %= newq;
            v[i] = (av < bv) ? av + newq - bv : av - bv;
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * a10: ApproxSwitchCRTBasis fast path — dcrtpoly-impl.h:888-915
 * ---------------------------------------------------------------------------------------- */
void orc_approx_switch_crt_basis(const uint64_t* x, uint32_t sizeQ, uint32_t N, const uint64_t* q,
                                 const uint64_t* QHatInvModq, const uint64_t* QHatInvModqPrecon,
                                 const uint64_t* QHatModp, uint32_t sizeP, const uint64_t* p,
                                 const uint64_t* mu128, uint64_t* out) {
#pragma omp parallel for
    for (uint32_t ri = 0; ri < N; ++ri) {
        u128 sum[ORC_MAX_LIMBS];
        for (uint32_t j = 0; j < sizeP; ++j)
            sum[j] = 0;
        for (uint32_t i = 0; i < sizeQ; ++i) {
            uint64_t y = orc_mod_mul_fast_const(x[(size_t)i * N + ri], QHatInvModq[i], q[i], QHatInvModqPrecon[i]);
            for (uint32_t j = 0; j < sizeP; ++j)
                sum[j] += (u128)y * QHatModp[(size_t)i * sizeP + j];
        }
        for (uint32_t j = 0; j < sizeP; ++j)
            out[(size_t)j * N + ri] =
                orc_barrett128((uint64_t)sum[j], (uint64_t)(sum[j] >> 64), p[j], mu128[2 * j], mu128[2 * j + 1]);
    }
}

/* ------------------------------------------------------------------------------------------
 * a16: SwitchCRTBasis — dcrtpoly-impl.h:1008-1085.  The double accumulation order is part of
 * the result: nu starts at 0.5, i ascending, one rounding per multiply and per add (no FMA:
 * this file is compiled with -ffp-contract=off).
 * ---------------------------------------------------------------------------------------- */
void orc_switch_crt_basis(const uint64_t* x, uint32_t sizeQ, uint32_t N, const uint64_t* q,
                          const uint64_t* QHatInvModq, const uint64_t* QHatInvModqPrecon,
                          const uint64_t* QHatModp_pq, const uint64_t* alphaQModp, uint32_t sizeP,
                          const uint64_t* p, const uint64_t* mu128, const double* qInv, uint64_t* out) {
#pragma omp parallel for
    for (uint32_t ri = 0; ri < N; ++ri) {
        uint64_t y[ORC_MAX_LIMBS];
        double nu = 0.5;
        for (uint32_t i = 0; i < sizeQ; ++i) {
            y[i] = orc_mod_mul_fast_const(x[(size_t)i * N + ri], QHatInvModq[i], q[i], QHatInvModqPrecon[i]);
            nu += (double)y[i] * qInv[i];
        }
        size_t alpha           = (size_t)nu;
        const uint64_t* aQModp = alphaQModp + alpha * sizeP;
        for (uint32_t j = 0; j < sizeP; ++j) {
            u128 cur = 0;
            for (uint32_t i = 0; i < sizeQ; ++i)
                cur += (u128)y[i] * QHatModp_pq[(size_t)j * sizeQ + i];
            uint64_t v = orc_barrett128((uint64_t)cur, (uint64_t)(cur >> 64), p[j], mu128[2 * j], mu128[2 * j + 1]);
            out[(size_t)j * N + ri] = orc_mod_sub_fast(v, aQModp[j], p[j]);
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * HYBRID key switching tables — rns-cryptoparameters.cpp:80-350.
 * The reference computes these with BigInteger products/quotients; every stored value is a
 * residue of a product of moduli, so 64-bit modular products give the same numbers.
 * ---------------------------------------------------------------------------------------- */
struct orc_hybrid {
    uint32_t N, sizeQ, sizeP, numPartQ, alpha;
    orc_ctx* ctx; /* limbs [0,sizeQ) = Q, [sizeQ, sizeQ+sizeP) = P */
    uint64_t *q, *p;
    uint64_t *PInvModq, *PInvModqPrecon;       /* [sizeQ] */
    uint64_t *PHatInvModp, *PHatInvModpPrecon; /* [sizeP] */
    uint64_t* PHatModq;                        /* [sizeP][sizeQ] */
    uint64_t* muQ128;                          /* [sizeQ][2] modqBarrettMu */
};

static uint32_t ceil_div(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

/* rns-cryptoparameters.cpp:128-176 */
uint32_t orc_hybrid_select_p(uint32_t N, uint32_t sizeQ, const uint64_t* q, uint32_t numPartQ, uint32_t auxBits,
                             uint64_t* p, uint64_t* psiP) {
    uint32_t a = ceil_div(sizeQ, numPartQ);
    /* maxBits = max bit length of the composite digits: exact via long double log2 is unsafe,
     * so multiply out in arbitrary precision (little-endian 64-bit limbs). */
    uint32_t maxBits = 0;
    for (uint32_t j = 0; j < numPartQ; ++j) {
        uint64_t big[ORC_MAX_LIMBS + 16];
        uint32_t len = 1;
        big[0]       = 1;
        for (uint32_t i = a * j; i < (j + 1) * a && i < sizeQ; ++i) {
            uint64_t carry = 0;
            for (uint32_t k = 0; k < len; ++k) {
                u128 t = (u128)big[k] * q[i] + carry;
                big[k] = (uint64_t)t;
                carry  = (uint64_t)(t >> 64);
            }
            if (carry)
                big[len++] = carry;
        }
        uint32_t bits = 64 * (len - 1) + orc_get_msb(big[len - 1]);
        if (bits > maxBits)
            maxBits = bits;
    }
    uint32_t sizeP     = ceil_div(maxBits, auxBits);
    uint64_t primeStep = 2 * (uint64_t)N; /* CryptoParametersCKKSRNS::FindAuxPrimeStep, ckksrns-cryptoparameters.cpp:185-188 */
    uint64_t pPrev     = orc_first_prime(auxBits, primeStep);
    for (uint32_t i = 0; i < sizeP; ++i) {
        int foundInQ;
        do {
            p[i]     = orc_previous_prime(pPrev, primeStep);
            foundInQ = 0;
            for (uint32_t j = 0; j < sizeQ; ++j)
                if (p[i] == q[j])
                    foundInQ = 1;
            pPrev = p[i];
        } while (foundInQ);
        psiP[i] = orc_root_of_unity(2 * (uint64_t)N, p[i]);
    }
    return sizeP;
}

orc_hybrid* orc_hybrid_create(uint32_t N, uint32_t sizeQ, const uint64_t* q, const uint64_t* psiQ,
                              uint32_t sizeP, const uint64_t* p, const uint64_t* psiP, uint32_t numPartQ) {
    orc_hybrid* h = (orc_hybrid*)calloc(1, sizeof(orc_hybrid));
    h->N          = N;
    h->sizeQ      = sizeQ;
    h->sizeP      = sizeP;
    h->numPartQ   = numPartQ;
    h->alpha      = ceil_div(sizeQ, numPartQ); /* :92 */
    uint32_t L    = sizeQ + sizeP;
    uint64_t* m   = (uint64_t*)malloc(sizeof(uint64_t) * L);
    uint64_t* r   = (uint64_t*)malloc(sizeof(uint64_t) * L);
    memcpy(m, q, sizeof(uint64_t) * sizeQ);
    memcpy(m + sizeQ, p, sizeof(uint64_t) * sizeP);
    memcpy(r, psiQ, sizeof(uint64_t) * sizeQ);
    memcpy(r + sizeQ, psiP, sizeof(uint64_t) * sizeP);
    h->ctx = orc_ctx_create(N, L, m, r);
    h->q   = m;
    h->p   = m + sizeQ;
    free(r);

    h->PInvModq          = (uint64_t*)malloc(sizeof(uint64_t) * sizeQ);
    h->PInvModqPrecon    = (uint64_t*)malloc(sizeof(uint64_t) * sizeQ);
    h->PHatInvModp       = (uint64_t*)malloc(sizeof(uint64_t) * sizeP);
    h->PHatInvModpPrecon = (uint64_t*)malloc(sizeof(uint64_t) * sizeP);
    h->PHatModq          = (uint64_t*)malloc(sizeof(uint64_t) * sizeP * sizeQ);
    h->muQ128            = (uint64_t*)malloc(sizeof(uint64_t) * 2 * sizeQ);
    /* [P^-1]_{q_i}  (:205-212) */
    for (uint32_t i = 0; i < sizeQ; ++i) {
        uint64_t Pmod = 1;
        for (uint32_t j = 0; j < sizeP; ++j)
            Pmod = orc_mulmod(Pmod, p[j] % q[i], q[i]);
        h->PInvModq[i]       = orc_invmod(Pmod, q[i]);
        h->PInvModqPrecon[i] = orc_prep_mod_mul_const(h->PInvModq[i], q[i]);
        orc_barrett_mu128(q[i], h->muQ128 + 2 * i);
    }
    /* [(P/p_j)^-1]_{p_j}, [P/p_j]_{q_i}  (:214-230) */
    for (uint32_t j = 0; j < sizeP; ++j) {
        uint64_t hat = 1;
        for (uint32_t k = 0; k < sizeP; ++k)
            if (k != j)
                hat = orc_mulmod(hat, p[k] % p[j], p[j]);
        h->PHatInvModp[j]       = orc_invmod(hat, p[j]);
        h->PHatInvModpPrecon[j] = orc_prep_mod_mul_const(h->PHatInvModp[j], p[j]);
        for (uint32_t i = 0; i < sizeQ; ++i) {
            uint64_t v = 1;
            for (uint32_t k = 0; k < sizeP; ++k)
                if (k != j)
                    v = orc_mulmod(v, p[k] % q[i], q[i]);
            h->PHatModq[(size_t)j * sizeQ + i] = v;
        }
    }
    return h;
}

void orc_hybrid_destroy(orc_hybrid* h) {
    if (!h)
        return;
    orc_ctx_destroy(h->ctx);
    free(h->q);
    free(h->PInvModq);
    free(h->PInvModqPrecon);
    free(h->PHatInvModp);
    free(h->PHatInvModpPrecon);
    free(h->PHatModq);
    free(h->muQ128);
    free(h);
}
uint32_t orc_hybrid_alpha(const orc_hybrid* h) { return h->alpha; }
void orc_hybrid_get_PInvModq(const orc_hybrid* h, uint64_t* out) { memcpy(out, h->PInvModq, 8 * h->sizeQ); }
void orc_hybrid_get_PHatInvModp(const orc_hybrid* h, uint64_t* out) { memcpy(out, h->PHatInvModp, 8 * h->sizeP); }
void orc_hybrid_get_PHatModq(const orc_hybrid* h, uint64_t* out) {
    memcpy(out, h->PHatModq, 8 * (size_t)h->sizeP * h->sizeQ);
}

/* digit geometry at level sizeQl (keyswitch-hybrid.cpp:329-349): */
static uint32_t num_parts_at(const orc_hybrid* h, uint32_t sizeQl) {
    uint32_t n = ceil_div(sizeQl, h->alpha);
    return n > h->numPartQ ? h->numPartQ : n;
}
static uint32_t part_size_at(const orc_hybrid* h, uint32_t part, uint32_t sizeQl) {
    uint32_t np = num_parts_at(h, sizeQl);
    if (part == np - 1)
        return sizeQl - h->alpha * part;
    return h->alpha;
}

/* [ (Q_part^(l) / q_i)^-1 ]_{q_i} for the digit truncated to sizePartQl limbs (:297-318) */
uint32_t orc_hybrid_get_PartQlHatInvModq(const orc_hybrid* h, uint32_t part, uint32_t sizeQl, uint64_t* out) {
    uint32_t sz    = part_size_at(h, part, sizeQl);
    uint32_t start = h->alpha * part;
    for (uint32_t i = 0; i < sz; ++i) {
        uint64_t qi  = h->q[start + i];
        uint64_t hat = 1;
        for (uint32_t k = 0; k < sz; ++k)
            if (k != i)
                hat = orc_mulmod(hat, h->q[start + k] % qi, qi);
        out[i] = orc_invmod(hat, qi);
    }
    return sz;
}

/* complementary basis of digit `part` at level sizeQl (:253-283): all Q_l limbs outside the digit, then P */
static uint32_t compl_basis(const orc_hybrid* h, uint32_t part, uint32_t sizeQl, uint32_t* limbIdx) {
    uint32_t sz    = part_size_at(h, part, sizeQl);
    uint32_t start = h->alpha * part;
    uint32_t n     = 0;
    for (uint32_t i = 0; i < sizeQl; ++i)
        if (i < start || i >= start + sz)
            limbIdx[n++] = i;
    for (uint32_t j = 0; j < h->sizeP; ++j)
        limbIdx[n++] = h->sizeQ + j;
    return n;
}

/* [Q_part^(l)/q_i]_{c_j} over the complementary basis (:320-349) -> out[sizePartQl][sizeCompl] */
uint32_t orc_hybrid_get_PartQlHatModp(const orc_hybrid* h, uint32_t part, uint32_t sizeQl, uint64_t* out,
                                      uint64_t* complModuli) {
    uint32_t idx[ORC_MAX_LIMBS];
    uint32_t nc    = compl_basis(h, part, sizeQl, idx);
    uint32_t sz    = part_size_at(h, part, sizeQl);
    uint32_t start = h->alpha * part;
    const uint64_t* mod = h->q; /* q then p contiguous */
    for (uint32_t i = 0; i < sz; ++i)
        for (uint32_t j = 0; j < nc; ++j) {
            uint64_t cj = mod[idx[j]];
            uint64_t v  = 1;
            for (uint32_t k = 0; k < sz; ++k)
                if (k != i)
                    v = orc_mulmod(v, h->q[start + k] % cj, cj);
            out[(size_t)i * nc + j] = v;
        }
    if (complModuli)
        for (uint32_t j = 0; j < nc; ++j)
            complModuli[j] = mod[idx[j]];
    return nc;
}

/* keyswitch-hybrid.cpp:314-379 */
uint32_t orc_hybrid_precompute_digits(const orc_hybrid* h, const uint64_t* c, uint32_t sizeQl, uint64_t* digits) {
    const uint32_t N = h->N, sizeP = h->sizeP, sizeQlP = sizeQl + sizeP;
    const uint32_t np = num_parts_at(h, sizeQl);
    const uint64_t* mod = h->q;
    for (uint32_t part = 0; part < np; ++part) {
        uint32_t sz    = part_size_at(h, part, sizeQl);
        uint32_t start = h->alpha * part;
        uint32_t idx[ORC_MAX_LIMBS];
        uint32_t nc = compl_basis(h, part, sizeQl, idx);
        /* partsCt = digit limbs of c, to COEFFICIENT */
        uint64_t* partsCt = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)sz * N);
        memcpy(partsCt, c + (size_t)start * N, sizeof(uint64_t) * (size_t)sz * N);
#pragma omp parallel for
        for (uint32_t i = 0; i < sz; ++i)
            ctx_inv(h->ctx, partsCt + (size_t)i * N, start + i);
        uint64_t hatInv[ORC_MAX_LIMBS], hatInvPre[ORC_MAX_LIMBS], cm[ORC_MAX_LIMBS], mu[2 * ORC_MAX_LIMBS];
        uint64_t* hatModp = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)sz * nc);
        orc_hybrid_get_PartQlHatInvModq(h, part, sizeQl, hatInv);
        for (uint32_t i = 0; i < sz; ++i)
            hatInvPre[i] = orc_prep_mod_mul_const(hatInv[i], h->q[start + i]);
        orc_hybrid_get_PartQlHatModp(h, part, sizeQl, hatModp, cm);
        for (uint32_t j = 0; j < nc; ++j)
            orc_barrett_mu128(cm[j], mu + 2 * j);
        uint64_t* compl_ = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)nc * N);
        orc_approx_switch_crt_basis(partsCt, sz, N, h->q + start, hatInv, hatInvPre, hatModp, nc, cm, mu, compl_);
#pragma omp parallel for
        for (uint32_t j = 0; j < nc; ++j)
            ctx_fwd(h->ctx, compl_ + (size_t)j * N, idx[j]);
        /* assemble in basis Q_l ∪ P (:369-376) */
        uint64_t* dst = digits + (size_t)part * sizeQlP * N;
        for (uint32_t i = 0; i < start; ++i)
            memcpy(dst + (size_t)i * N, compl_ + (size_t)i * N, 8 * (size_t)N);
        for (uint32_t i = start; i < start + sz; ++i)
            memcpy(dst + (size_t)i * N, c + (size_t)i * N, 8 * (size_t)N);
        for (uint32_t i = start + sz; i < sizeQlP; ++i)
            memcpy(dst + (size_t)i * N, compl_ + (size_t)(i - sz) * N, 8 * (size_t)N);
        (void)mod;
        free(partsCt);
        free(hatModp);
        free(compl_);
    }
    return np;
}

/* keyswitch-hybrid.cpp:402-435 */
void orc_hybrid_inner_product(const orc_hybrid* h, const uint64_t* digits, uint32_t numPartQl, uint32_t sizeQl,
                              const uint64_t* keyB, const uint64_t* keyA, uint64_t* out0, uint64_t* out1) {
    const uint32_t N = h->N, sizeQlP = sizeQl + h->sizeP, sizeQP = h->sizeQ + h->sizeP;
    const uint32_t delta = h->sizeQ - sizeQl;
    memset(out0, 0, sizeof(uint64_t) * (size_t)sizeQlP * N);
    memset(out1, 0, sizeof(uint64_t) * (size_t)sizeQlP * N);
    for (uint32_t j = 0; j < numPartQl; ++j) {
#pragma omp parallel for
        for (uint32_t i = 0; i < sizeQlP; ++i) {
            uint32_t idx = (i >= sizeQl) ? i + delta : i;
            uint64_t qi  = h->q[idx]; /* q then p contiguous: idx in [0, sizeQ+sizeP) */
            uint64_t mu  = orc_compute_mu(qi);
            const uint64_t* cji = digits + ((size_t)j * sizeQlP + i) * N;
            const uint64_t* bji = keyB + ((size_t)j * sizeQP + idx) * N;
            const uint64_t* aji = keyA + ((size_t)j * sizeQP + idx) * N;
            uint64_t* e0 = out0 + (size_t)i * N;
            uint64_t* e1 = out1 + (size_t)i * N;
            for (uint32_t r = 0; r < N; ++r) {
                e0[r] = orc_mod_add_fast(e0[r], orc_mod_mul_fast(cji[r], bji[r], qi, mu), qi);
                e1[r] = orc_mod_add_fast(e1[r], orc_mod_mul_fast(cji[r], aji[r], qi, mu), qi);
            }
        }
    }
}

/* dcrtpoly-impl.h:966-1005 with t == 0 */
void orc_hybrid_approx_mod_down(const orc_hybrid* h, const uint64_t* x, uint32_t sizeQl, uint64_t* out) {
    const uint32_t N = h->N, sizeP = h->sizeP;
    uint64_t* partP = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)sizeP * N);
    memcpy(partP, x + (size_t)sizeQl * N, sizeof(uint64_t) * (size_t)sizeP * N);
#pragma omp parallel for
    for (uint32_t j = 0; j < sizeP; ++j)
        ctx_inv(h->ctx, partP + (size_t)j * N, h->sizeQ + j);
    /* PHatModq is [sizeP][sizeQ]; ApproxSwitchCRTBasis reads QHatModp[i][j] with i over P, j over Q_l:
     * pass a [sizeP][sizeQl] slice copy */
    uint64_t* hat = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)sizeP * sizeQl);
    for (uint32_t j = 0; j < sizeP; ++j)
        memcpy(hat + (size_t)j * sizeQl, h->PHatModq + (size_t)j * h->sizeQ, 8 * (size_t)sizeQl);
    uint64_t* sw = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)sizeQl * N);
    orc_approx_switch_crt_basis(partP, sizeP, N, h->p, h->PHatInvModp, h->PHatInvModpPrecon, hat, sizeQl, h->q,
                                h->muQ128, sw);
#pragma omp parallel for
    for (uint32_t i = 0; i < sizeQl; ++i) {
        uint64_t qi = h->q[i];
        uint64_t* s = sw + (size_t)i * N;
        ctx_fwd(h->ctx, s, i);
        /* (m_vectors[i] - switched) * PInvModq[i]  — PolyImpl::Times(NativeInteger) -> ModMul Shoup
         * (mubintvecnat.cpp:295-304) */
        for (uint32_t r = 0; r < N; ++r) {
            uint64_t d            = orc_mod_sub_fast(x[(size_t)i * N + r], s[r], qi);
            out[(size_t)i * N + r] = orc_mod_mul_fast_const(d, h->PInvModq[i], qi, h->PInvModqPrecon[i]);
        }
    }
    free(partP);
    free(hat);
    free(sw);
}

/* ApproxModDown with the BGV factors (dcrtpoly-impl.h:966-1005, t > 0): partP[j] *= t^-1 mod p_j (:981-983),
 * switched[i] *= t before its NTT (:996-999) */
void orc_hybrid_approx_mod_down_t(const orc_hybrid* h, const uint64_t* x, uint32_t sizeQl, uint64_t t, uint64_t* out) {
    const uint32_t N = h->N, sizeP = h->sizeP;
    uint64_t* partP = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)sizeP * N);
    memcpy(partP, x + (size_t)sizeQl * N, sizeof(uint64_t) * (size_t)sizeP * N);
    for (uint32_t j = 0; j < sizeP; ++j) {
        const uint64_t pj = h->p[j];
        ctx_inv(h->ctx, partP + (size_t)j * N, h->sizeQ + j);
        orc_vec_mul_const(partP + (size_t)j * N, partP + (size_t)j * N, orc_invmod(t % pj, pj), N, pj);
    }
    uint64_t* hat = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)sizeP * sizeQl);
    for (uint32_t j = 0; j < sizeP; ++j)
        memcpy(hat + (size_t)j * sizeQl, h->PHatModq + (size_t)j * h->sizeQ, 8 * (size_t)sizeQl);
    uint64_t* sw = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)sizeQl * N);
    orc_approx_switch_crt_basis(partP, sizeP, N, h->p, h->PHatInvModp, h->PHatInvModpPrecon, hat, sizeQl, h->q,
                                h->muQ128, sw);
    for (uint32_t i = 0; i < sizeQl; ++i) {
        uint64_t qi = h->q[i];
        uint64_t* s = sw + (size_t)i * N;
        orc_vec_mul_const(s, s, t % qi, N, qi);
        ctx_fwd(h->ctx, s, i);
        for (uint32_t r = 0; r < N; ++r) {
            uint64_t d             = orc_mod_sub_fast(x[(size_t)i * N + r], s[r], qi);
            out[(size_t)i * N + r] = orc_mod_mul_fast_const(d, h->PInvModq[i], qi, h->PInvModqPrecon[i]);
        }
    }
    free(partP);
    free(hat);
    free(sw);
}

/* keyswitch-hybrid.cpp:308-312 + :381-400 */
void orc_hybrid_key_switch(const orc_hybrid* h, const uint64_t* c, uint32_t sizeQl, const uint64_t* keyB,
                           const uint64_t* keyA, uint64_t* out0, uint64_t* out1) {
    const uint32_t N = h->N, sizeQlP = sizeQl + h->sizeP;
    uint32_t np      = num_parts_at(h, sizeQl);
    uint64_t* digits = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)np * sizeQlP * N);
    uint64_t* e0     = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)sizeQlP * N);
    uint64_t* e1     = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)sizeQlP * N);
    orc_hybrid_precompute_digits(h, c, sizeQl, digits);
    orc_hybrid_inner_product(h, digits, np, sizeQl, keyB, keyA, e0, e1);
    orc_hybrid_approx_mod_down(h, e0, sizeQl, out0);
    orc_hybrid_approx_mod_down(h, e1, sizeQl, out1);
    free(digits);
    free(e0);
    free(e1);
}

/* base-leveledshe.cpp:607-644 (EvalMultCore) + :201-214 (relinearise and add) */
void orc_ckks_eval_mult_relin(const orc_hybrid* h, const uint64_t* a0, const uint64_t* a1, const uint64_t* b0,
                              const uint64_t* b1, uint32_t sizeQl, const uint64_t* keyB, const uint64_t* keyA,
                              uint64_t* c0, uint64_t* c1) {
    const uint32_t N = h->N;
    size_t sz        = (size_t)sizeQl * N;
    uint64_t* d2     = (uint64_t*)malloc(sizeof(uint64_t) * sz);
    uint64_t* t      = (uint64_t*)malloc(sizeof(uint64_t) * sz);
    uint64_t* k0     = (uint64_t*)malloc(sizeof(uint64_t) * sz);
    uint64_t* k1     = (uint64_t*)malloc(sizeof(uint64_t) * sz);
#pragma omp parallel for
    for (uint32_t i = 0; i < sizeQl; ++i) {
        uint64_t qi = h->q[i];
        size_t o    = (size_t)i * N;
        orc_vec_mul(c0 + o, a0 + o, b0 + o, N, qi); /* d0 = a0*b0 */
        orc_vec_mul(c1 + o, a0 + o, b1 + o, N, qi); /* d1 = a0*b1 + a1*b0 */
        orc_vec_mul(t + o, a1 + o, b0 + o, N, qi);
        orc_vec_add(c1 + o, c1 + o, t + o, N, qi);
        orc_vec_mul(d2 + o, a1 + o, b1 + o, N, qi); /* d2 = a1*b1 */
    }
    orc_hybrid_key_switch(h, d2, sizeQl, keyB, keyA, k0, k1);
#pragma omp parallel for
    for (uint32_t i = 0; i < sizeQl; ++i) {
        uint64_t qi = h->q[i];
        size_t o    = (size_t)i * N;
        orc_vec_add(c0 + o, c0 + o, k0 + o, N, qi);
        orc_vec_add(c1 + o, c1 + o, k1 + o, N, qi);
    }
    free(d2);
    free(t);
    free(k0);
    free(k1);
}

/* base-leveledshe.cpp:381-422 / :432-463 */
void orc_eval_automorphism(const orc_hybrid* h, const uint64_t* c0, const uint64_t* c1, uint32_t sizeQl, uint32_t k,
                           const uint64_t* keyB, const uint64_t* keyA, uint64_t* out0, uint64_t* out1) {
    const uint32_t N = h->N;
    size_t sz        = (size_t)sizeQl * N;
    uint64_t* k0     = (uint64_t*)malloc(sizeof(uint64_t) * sz);
    uint64_t* k1     = (uint64_t*)malloc(sizeof(uint64_t) * sz);
    uint32_t* pre    = (uint32_t*)malloc(sizeof(uint32_t) * N);
    orc_hybrid_key_switch(h, c1, sizeQl, keyB, keyA, k0, k1);
    orc_precompute_auto_map(N, k, pre);
    for (uint32_t i = 0; i < sizeQl; ++i) {
        size_t o = (size_t)i * N;
        orc_vec_add(k0 + o, k0 + o, c0 + o, N, h->q[i]); /* ba[0] += cv[0] */
        orc_automorph_eval(out0 + o, k0 + o, N, pre);
        orc_automorph_eval(out1 + o, k1 + o, N, pre);
    }
    free(k0);
    free(k1);
    free(pre);
}

/* LeveledSHECKKSRNS::EvalFastRotationExt (ckksrns-leveledshe.cpp:534-582): cTilda = EvalFastKeySwitchCoreExt(digits of c1,
 * key) in the extended basis Q_l u P; addFirst: cTilda[0] += c0 * [P]_{q_i} on the Q_l limbs (KeySwitchExt,
 * keyswitch-hybrid.cpp:217-243); both elements through the automorphism.  out0/out1 [(sizeQl+sizeP)][N]. */
void orc_eval_fast_rotation_ext(const orc_hybrid* h, const uint64_t* c0, const uint64_t* c1, uint32_t sizeQl, uint32_t k,
                                int addFirst, const uint64_t* keyB, const uint64_t* keyA, uint64_t* out0, uint64_t* out1) {
    const uint32_t N = h->N, sizeQlP = sizeQl + h->sizeP;
    const uint32_t np = num_parts_at(h, sizeQl);
    uint64_t* digits  = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)np * sizeQlP * N);
    uint64_t* e0      = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)sizeQlP * N);
    uint64_t* e1      = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)sizeQlP * N);
    uint64_t* tmp     = (uint64_t*)malloc(sizeof(uint64_t) * N);
    uint32_t* pre     = (uint32_t*)malloc(sizeof(uint32_t) * N);
    orc_hybrid_precompute_digits(h, c1, sizeQl, digits);
    orc_hybrid_inner_product(h, digits, np, sizeQl, keyB, keyA, e0, e1);
    if (addFirst)
        for (uint32_t i = 0; i < sizeQl; ++i) {
            const uint64_t qi = h->q[i];
            uint64_t PModq    = 1; /* rns-cryptoparameters.cpp:200-203 */
            for (uint32_t j = 0; j < h->sizeP; ++j)
                PModq = orc_mulmod(PModq, h->p[j] % qi, qi);
            orc_vec_mul_const(tmp, c0 + (size_t)i * N, PModq, N, qi); /* TimesNoCheck(PModq) */
            orc_vec_add(e0 + (size_t)i * N, e0 + (size_t)i * N, tmp, N, qi);
        }
    orc_precompute_auto_map(N, k, pre);
    for (uint32_t i = 0; i < sizeQlP; ++i) {
        orc_automorph_eval(out0 + (size_t)i * N, e0 + (size_t)i * N, N, pre);
        orc_automorph_eval(out1 + (size_t)i * N, e1 + (size_t)i * N, N, pre);
    }
    free(digits), free(e0), free(e1), free(tmp), free(pre);
}

/* Baby-step/giant-step plaintext-matrix x ciphertext product with double hoisting: FHECKKSRNS::EvalLinearTransform
 * (ckksrns-fhe.cpp:1832-1882) and one level of EvalCoeffsToSlots / EvalSlotsToCoeffs (:1884-2198) share this shape.
 *   rot_j   = inK[j]  ? EvalFastRotationExt(ct, inK[j], digits(ct), addFirst = true) : KeySwitchExt(ct, true)      (:1845-1847, 1855)
 *   inner_i = sum_j rot_j * diag[i*nIn + j]       (EvalMultExt / EvalAddExtInPlace, :2723-2740; NULL diag = term absent)
 *   outK[i] == 0:  first += ApproxModDown(inner_i[0]);  outer[1] += inner_i[1]                               (:1861-1866, 1976-1979)
 *   else        :  d = KeySwitchDown(inner_i); first += Automorphism_k(d[0]);
 *                  outer += EvalFastRotationExt(d, k, digits(d), addFirst = false)                            (:1868-1876)
 *   result  = KeySwitchDown(outer); result[0] += first                                                        (:1879-1880)
 * Everything in EVALUATION format; c0,c1,out0,out1 [sizeQl][N]; diag rows [(sizeQl+sizeP)][N]; keys [numPartQ][sizeQ+sizeP][N]. */
void orc_ckks_bsgs_transform(const orc_hybrid* h, const uint64_t* c0, const uint64_t* c1, uint32_t sizeQl, uint32_t nIn,
                             const uint32_t* inK, const uint64_t* const* inKeyB, const uint64_t* const* inKeyA, uint32_t nOut,
                             const uint32_t* outK, const uint64_t* const* outKeyB, const uint64_t* const* outKeyA,
                             const uint64_t* const* diag, uint64_t* out0, uint64_t* out1) {
    const uint32_t N = h->N, sizeQlP = sizeQl + h->sizeP;
    const size_t ext = (size_t)sizeQlP * N, low = (size_t)sizeQl * N;
    uint64_t* rot    = (uint64_t*)calloc((size_t)nIn * 2 * ext, sizeof(uint64_t));
    uint64_t* inner  = (uint64_t*)malloc(sizeof(uint64_t) * 2 * ext);
    uint64_t* outer  = (uint64_t*)calloc(2 * ext, sizeof(uint64_t));
    uint64_t* tmpE   = (uint64_t*)malloc(sizeof(uint64_t) * 2 * ext);
    uint64_t* first  = (uint64_t*)calloc(low, sizeof(uint64_t));
    uint64_t* d      = (uint64_t*)malloc(sizeof(uint64_t) * 2 * low);
    uint64_t* tmp    = (uint64_t*)malloc(sizeof(uint64_t) * N);
    uint32_t* pre    = (uint32_t*)malloc(sizeof(uint32_t) * N);
    for (uint32_t j = 0; j < nIn; ++j) {
        uint64_t *r0 = rot + (size_t)j * 2 * ext, *r1 = r0 + ext;
        if (inK[j])
            orc_eval_fast_rotation_ext(h, c0, c1, sizeQl, inK[j], 1, inKeyB[j], inKeyA[j], r0, r1);
        else /* KeySwitchExt(ct, true), keyswitch-hybrid.cpp:217-243: c * [P]_{q_i} on the Q_l limbs, zeros on the P limbs */
            for (uint32_t i = 0; i < sizeQl; ++i) {
                const uint64_t qi = h->q[i];
                uint64_t PModq    = 1;
                for (uint32_t t = 0; t < h->sizeP; ++t)
                    PModq = orc_mulmod(PModq, h->p[t] % qi, qi);
                orc_vec_mul_const(r0 + (size_t)i * N, c0 + (size_t)i * N, PModq, N, qi);
                orc_vec_mul_const(r1 + (size_t)i * N, c1 + (size_t)i * N, PModq, N, qi);
            }
    }
    for (uint32_t i = 0; i < nOut; ++i) {
        memset(inner, 0, sizeof(uint64_t) * 2 * ext);
        for (uint32_t j = 0; j < nIn; ++j) {
            const uint64_t* a = diag[(size_t)i * nIn + j];
            if (!a)
                continue;
            for (uint32_t e = 0; e < 2; ++e)
                for (uint32_t l = 0; l < sizeQlP; ++l) {
                    const uint64_t m = l < sizeQl ? h->q[l] : h->p[l - sizeQl];
                    orc_vec_mul(tmp, rot + ((size_t)j * 2 + e) * ext + (size_t)l * N, a + (size_t)l * N, N, m);
                    orc_vec_add(inner + e * ext + (size_t)l * N, inner + e * ext + (size_t)l * N, tmp, N, m);
                }
        }
        if (!outK[i]) {
            orc_hybrid_approx_mod_down(h, inner, sizeQl, d); /* KeySwitchDownFirstElement */
            for (uint32_t l = 0; l < sizeQl; ++l)
                orc_vec_add(first + (size_t)l * N, first + (size_t)l * N, d + (size_t)l * N, N, h->q[l]);
            for (uint32_t l = 0; l < sizeQlP; ++l) {
                const uint64_t m = l < sizeQl ? h->q[l] : h->p[l - sizeQl];
                orc_vec_add(outer + ext + (size_t)l * N, outer + ext + (size_t)l * N, inner + ext + (size_t)l * N, N, m);
            }
        }
        else {
            orc_hybrid_approx_mod_down(h, inner, sizeQl, d);
            orc_hybrid_approx_mod_down(h, inner + ext, sizeQl, d + low);
            orc_precompute_auto_map(N, outK[i], pre);
            for (uint32_t l = 0; l < sizeQl; ++l) {
                orc_automorph_eval(tmp, d + (size_t)l * N, N, pre);
                orc_vec_add(first + (size_t)l * N, first + (size_t)l * N, tmp, N, h->q[l]);
            }
            orc_eval_fast_rotation_ext(h, d, d + low, sizeQl, outK[i], 0, outKeyB[i], outKeyA[i], tmpE, tmpE + ext);
            for (uint32_t e = 0; e < 2; ++e)
                for (uint32_t l = 0; l < sizeQlP; ++l) {
                    const uint64_t m = l < sizeQl ? h->q[l] : h->p[l - sizeQl];
                    orc_vec_add(outer + e * ext + (size_t)l * N, outer + e * ext + (size_t)l * N, tmpE + e * ext + (size_t)l * N, N, m);
                }
        }
    }
    orc_hybrid_approx_mod_down(h, outer, sizeQl, out0);
    orc_hybrid_approx_mod_down(h, outer + ext, sizeQl, out1);
    for (uint32_t l = 0; l < sizeQl; ++l)
        orc_vec_add(out0 + (size_t)l * N, out0 + (size_t)l * N, first + (size_t)l * N, N, h->q[l]);
    free(rot), free(inner), free(outer), free(tmpE), free(first), free(d), free(tmp), free(pre);
}

/* ------------------------------------------------------------------------------------------
 * a15: rescale
 * ---------------------------------------------------------------------------------------- */
/* ckksrns-cryptoparameters.cpp:60-81 for the level whose last limb is l = sizeQl-1:
 * QlQlInvModqlDivqlModq[i] = ((Q^(l-1))^-1 mod q_l * Q^(l-1) / q_l) mod q_i ,  qlInvModq[i] = q_l^-1 mod q_i.
 * With Q' = Q^(l-1) = prod_{k<l} q_k:  result = floor(Q' * inv / q_l), inv = Q'^-1 mod q_l.
 * Q'*inv = 1 + m*q_l for an integer m, hence floor(Q'*inv/q_l) = m = (Q'*inv - 1)/q_l, and
 * m mod q_i = (0*inv - 1) * q_l^-1 = -(q_l^-1) mod q_i  (Q' = 0 mod q_i for i < l). */
void orc_rescale_tables(const orc_ctx* c, uint32_t sizeQl, uint64_t* QlQlInvModqlDivqlModq, uint64_t* qlInvModq) {
    uint32_t l  = sizeQl - 1;
    uint64_t ql = c->q[l];
    for (uint32_t i = 0; i < l; ++i) {
        uint64_t qi   = c->q[i];
        uint64_t inv  = orc_invmod(ql % qi, qi);
        qlInvModq[i]  = inv;
        QlQlInvModqlDivqlModq[i] = (qi - inv) % qi;
    }
}

/* dcrtpoly-impl.h:693-712, m_format == EVALUATION */
void orc_drop_last_element_and_scale(const orc_ctx* c, const uint64_t* x, uint32_t sizeQl, uint64_t* out) {
    const uint32_t N = c->N, l = sizeQl - 1;
    uint64_t tabA[ORC_MAX_LIMBS], tabB[ORC_MAX_LIMBS];
    orc_rescale_tables(c, sizeQl, tabA, tabB);
    uint64_t* last = (uint64_t*)malloc(sizeof(uint64_t) * N);
    memcpy(last, x + (size_t)l * N, sizeof(uint64_t) * N);
    ctx_inv(c, last, l);
#pragma omp parallel for
    for (uint32_t i = 0; i < l; ++i) {
        uint64_t qi   = c->q[i];
        uint64_t* tmp = (uint64_t*)malloc(sizeof(uint64_t) * N);
        memcpy(tmp, last, sizeof(uint64_t) * N);
        orc_switch_modulus(tmp, N, c->q[l], qi);
        orc_vec_mul_const(tmp, tmp, tabA[i], N, qi);
        ctx_fwd(c, tmp, i);
        orc_vec_mul_const(out + (size_t)i * N, x + (size_t)i * N, tabB[i], N, qi);
        orc_vec_add(out + (size_t)i * N, out + (size_t)i * N, tmp, N, qi);
        free(tmp);
    }
    free(last);
}

/* DCRTPolyImpl::ModReduce (dcrtpoly-impl.h:736-755), BGV modulus switching by the last limb with plaintext modulus t:
 * delta = [last]_{COEFF} * (-t^-1 mod q_l); per remaining limb: x_i = (x_i + t * SwitchModulus(delta -> q_i)) * q_l^-1.
 * Tables as CryptoParametersBGVRNS builds them (bgvrns-cryptoparameters.cpp: negtInvModq, qlInvModq, tModqPrecon). */
void orc_mod_reduce(const orc_ctx* c, const uint64_t* x, uint32_t sizeQl, uint64_t t, int evalFormat, uint64_t* out) {
    const uint32_t N = c->N, l = sizeQl - 1;
    const uint64_t ql = c->q[l];
    const uint64_t negtInv = (ql - orc_invmod(t % ql, ql)) % ql;
    uint64_t* delta = (uint64_t*)malloc(sizeof(uint64_t) * N);
    memcpy(delta, x + (size_t)l * N, sizeof(uint64_t) * N);
    if (evalFormat)
        ctx_inv(c, delta, l);                        /* delta.SetFormat(COEFFICIENT)  :741 */
    orc_vec_mul_const(delta, delta, negtInv, N, ql); /* delta *= negtInvModq          :742 */
#pragma omp parallel for
    for (uint32_t i = 0; i < l; ++i) {
        const uint64_t qi   = c->q[i];
        const uint64_t qlInv = orc_invmod(ql % qi, qi);
        uint64_t* tmp = (uint64_t*)malloc(sizeof(uint64_t) * N);
        memcpy(tmp, delta, sizeof(uint64_t) * N);
        orc_switch_modulus(tmp, N, ql, qi);          /* :749 */
        if (evalFormat)
            ctx_fwd(c, tmp, i);                      /* :750-751 */
        orc_vec_mul_const(tmp, tmp, t % qi, N, qi);  /* tmp *= t                      :752 */
        orc_vec_add(out + (size_t)i * N, x + (size_t)i * N, tmp, N, qi);
        orc_vec_mul_const(out + (size_t)i * N, out + (size_t)i * N, qlInv, N, qi); /* :753 */
        free(tmp);
    }
    free(delta);
}

/* ------------------------------------------------------------------------------------------
 * a17: ScaleAndRound family.  Double-precision accumulation order is part of the result.
 * ---------------------------------------------------------------------------------------- */
/* dcrtpoly-impl.h:1513-1628 (HAVE_INT128 && NATIVEINT == 64 branch) */
void orc_scale_and_round(const uint64_t* x, uint32_t sizeI, uint32_t sizeO, uint32_t N, int outputFirst,
                         const uint64_t* tab, const double* frac, const uint64_t* o, const uint64_t* mu128,
                         uint64_t* out) {
    const uint32_t inputIndex  = outputFirst ? sizeO : 0;
    const uint32_t outputIndex = outputFirst ? 0 : sizeI;
#pragma omp parallel for
    for (uint32_t ri = 0; ri < N; ++ri) {
        double nu = 0.5;
        for (uint32_t i = 0; i < sizeI; ++i)
            nu += frac[i] * (double)x[(size_t)(i + inputIndex) * N + ri];
        /* isConvertableToNativeInt: |nu| <= (double)Max64BitValue  (utils/utilities.h:122-126) */
        const int small = fabs(nu) <= (double)UINT64_MAX;
        for (uint32_t j = 0; j < sizeO; ++j) {
            const uint64_t* tj = tab + (size_t)j * (sizeI + 1);
            u128 cur = 0;
            for (uint32_t i = 0; i < sizeI; ++i)
                cur += (u128)x[(size_t)(i + inputIndex) * N + ri] * tj[i];
            cur += (u128)x[(size_t)(outputIndex + j) * N + ri] * tj[sizeI];
            const uint64_t oj = o[j];
            uint64_t v = orc_barrett128((uint64_t)cur, (uint64_t)(cur >> 64), oj, mu128[2 * j], mu128[2 * j + 1]);
            uint64_t a;
            if (small) {
                uint64_t alpha = (uint64_t)nu;
                a = alpha >= oj ? alpha % oj : alpha;
            }
            else {
                u128 alpha = (u128)nu;
                a = orc_barrett128((uint64_t)alpha, (uint64_t)(alpha >> 64), oj, mu128[2 * j], mu128[2 * j + 1]);
            }
            out[(size_t)j * N + ri] = orc_mod_add_fast(v, a, oj);
        }
    }
}

/* dcrtpoly-impl.h:1470-1510 */
void orc_approx_scale_and_round(const uint64_t* x, uint32_t sizeQ, uint32_t sizeP, uint32_t N, const uint64_t* tab,
                                const uint64_t* p, const uint64_t* mu128, uint64_t* out) {
#pragma omp parallel for
    for (uint32_t ri = 0; ri < N; ++ri)
        for (uint32_t j = 0; j < sizeP; ++j) {
            const uint64_t* tj = tab + (size_t)j * (sizeQ + 1);
            u128 cur = 0;
            for (uint32_t i = 0; i < sizeQ; ++i)
                cur += (u128)x[(size_t)i * N + ri] * tj[i];
            cur += (u128)x[(size_t)(sizeQ + j) * N + ri] * tj[sizeQ];
            out[(size_t)j * N + ri] =
                orc_barrett128((uint64_t)cur, (uint64_t)(cur >> 64), p[j], mu128[2 * j], mu128[2 * j + 1]);
        }
}

/* dcrtpoly-impl.h:1674-1689 */
void orc_scale_and_round_p_over_q(const uint64_t* x, uint32_t sizeQ, uint32_t N, const uint64_t* q, uint64_t pLast,
                                  const uint64_t* pInvModq, uint64_t* out) {
    for (uint32_t i = 0; i < sizeQ; ++i) {
        uint64_t* tmp = (uint64_t*)malloc(sizeof(uint64_t) * N);
        memcpy(tmp, x + (size_t)sizeQ * N, sizeof(uint64_t) * N);
        orc_switch_modulus(tmp, N, pLast, q[i]);
        orc_vec_sub(out + (size_t)i * N, x + (size_t)i * N, tmp, N, q[i]);
        orc_vec_mul_const(out + (size_t)i * N, out + (size_t)i * N, pInvModq[i], N, q[i]);
        free(tmp);
    }
}

/* DCRTPolyImpl::TimesQovert (dcrtpoly-impl.h:868-885): xi.ModMulFastConstEq(NegQModt, t, precon) then xi.ModMulFastEq(tInvModq[i], q_i, mu)
 * on every word of limb i; x [L][N] in place */
void orc_times_q_over_t(uint64_t* x, uint32_t L, uint32_t N, const uint64_t* q, uint64_t t, uint64_t negQModt, const uint64_t* tInvModq) {
    const uint64_t pre = orc_prep_mod_mul_const(negQModt, t);
    for (uint32_t i = 0; i < L; ++i) {
        const uint64_t mu = orc_compute_mu(q[i]);
        for (uint32_t ri = 0; ri < N; ++ri) {
            uint64_t v = orc_mod_mul_fast_const(x[(size_t)i * N + ri], negQModt, t, pre);
            x[(size_t)i * N + ri] = orc_mod_mul_fast(v, tInvModq[i], q[i], mu);
        }
    }
}
/* DCRTPolyImpl::SetValuesModSwitch (dcrtpoly-impl.h:630-647), the arithmetic of the one-limb case: x [N] COEFFICIENT modulo qFrom ->
 * out [N] modulo qTo through double precision (:641-644) */
void orc_set_values_mod_switch(const uint64_t* x, uint32_t N, uint64_t qFrom, uint64_t qTo, uint64_t* out) {
    const double ratio = (double)qTo / (double)qFrom;
    for (uint32_t j = 0; j < N; ++j)
        out[j] = (uint64_t)floor(0.5 + (double)x[j] * ratio) % qTo;
}

/* DCRTPolyImpl::ScaleAndRound -> NativePoly mod t (dcrtpoly-impl.h:1190-1467; BFV HPS decryption).  The eight branches
 * differ in three switches: t a power of two (final `& (t-1)` vs the double-precision reduction :1375-1377), the
 * hi/lo split of every residue at bit qMSB/2 when qMSB + sizeQMSB >= 52 (:1247ff), and plain wrap-around products
 * vs ModMulFastConst modulo t.  Double-precision sums in the reference's order (i ascending; lo term, then hi term). */
static uint32_t msb64(uint64_t v) {
    uint32_t n = 0;
    while (v) {
        ++n;
        v >>= 1;
    }
    return n;
}
void orc_scale_and_round_native(const uint64_t* x, uint32_t sizeQ, uint32_t N, const uint64_t* q, uint64_t t,
                                const uint64_t* tabModt, const uint64_t* tabBModt, const double* frac, const double* bfrac,
                                uint64_t* out) {
    uint64_t qmax = q[0];
    for (uint32_t i = 1; i < sizeQ; ++i)
        if (q[i] > qmax)
            qmax = q[i];
    const uint32_t qMSB = msb64(qmax), tMSB = msb64(t), sizeQMSB = msb64(sizeQ), qMSBHf = qMSB >> 1;
    const int pow2  = (t & (t - 1)) == 0;
    const int split = !(qMSB + sizeQMSB < 52);
    int nomod;
    if (!split)
        nomod = pow2 ? (qMSB + sizeQMSB + tMSB < 63) : (qMSB + tMSB + sizeQMSB < 52);
    else
        nomod = pow2 ? (qMSBHf + tMSB + sizeQMSB < 62) : (qMSBHf + tMSB + sizeQMSB < 52);
    uint64_t pre[ORC_MAX_LIMBS], preB[ORC_MAX_LIMBS];
    for (uint32_t i = 0; i < sizeQ; ++i) {
        pre[i]  = orc_prep_mod_mul_const(tabModt[i], t);
        preB[i] = tabBModt ? orc_prep_mod_mul_const(tabBModt[i], t) : 0;
    }
    const double td = (double)t, tInv = 1. / td;
    for (uint32_t ri = 0; ri < N; ++ri) {
        double floatSum = pow2 ? 0.5 : 0.0;
        uint64_t intSum = 0;
        for (uint32_t i = 0; i < sizeQ; ++i) {
            const uint64_t v = x[(size_t)i * N + ri];
            if (!split) {
                floatSum += (double)v * frac[i];
                intSum += nomod ? v * tabModt[i] : orc_mod_mul_fast_const(v, tabModt[i], t, pre[i]);
            }
            else {
                const uint64_t hi = v >> qMSBHf, lo = v - (hi << qMSBHf);
                floatSum += (double)lo * frac[i];
                floatSum += (double)hi * bfrac[i];
                intSum += nomod ? lo * tabModt[i] : orc_mod_mul_fast_const(lo, tabModt[i], t, pre[i]);
                intSum += nomod ? hi * tabBModt[i] : orc_mod_mul_fast_const(hi, tabBModt[i], t, preB[i]);
            }
        }
        if (pow2) {
            intSum += (uint64_t)floatSum;
            out[ri] = intSum & (t - 1);
        }
        else {
            floatSum += (double)intSum;
            floatSum -= td * (double)(uint64_t)(floatSum * tInv);
            out[ri] = (uint64_t)(floatSum + 0.5);
        }
    }
}

/* DCRTPolyImpl::ScaleAndRound, BEHZ decryption overload (dcrtpoly-impl.h:1631-1671), gamma = 2^26 */
void orc_scale_and_round_behz_decrypt(const uint64_t* x, uint32_t sizeQ, uint32_t N, const uint64_t* q, uint64_t tgamma,
                                      const uint64_t* tgammaQHatModq, const uint64_t* negInvqModtgamma, uint64_t* out) {
    const uint64_t gammaMinus1 = (1u << 26) - 1;
    for (uint32_t k = 0; k < N; ++k) {
        uint64_t s = 0;
        for (uint32_t i = 0; i < sizeQ; ++i) {
            const uint64_t a = orc_mod_mul_fast_const(x[(size_t)i * N + k], tgammaQHatModq[i], q[i],
                                                      orc_prep_mod_mul_const(tgammaQHatModq[i], q[i]));
            const uint64_t b = orc_mod_mul_fast_const(a, negInvqModtgamma[i], tgamma,
                                                      orc_prep_mod_mul_const(negInvqModtgamma[i], tgamma));
            s = orc_mod_add_fast(s, b, tgamma);
        }
        s += s & gammaMinus1;
        out[k] = s >> 26;
    }
}

/* ------------------------------------------------------------------------------------------
 * a18: BEHZ.  Tables as bfvrns-cryptoparameters.cpp:673-850 builds them (residues of products => modular
 * arithmetic; the one genuinely multi-precision step is the msk size check B*msk >= 2n*t*Q).
 * ---------------------------------------------------------------------------------------- */
struct orc_behz {
    uint32_t N, numQ, numBsk;
    uint64_t t, *q, *bsk, *psiBsk;
    uint64_t *mtQHatInv, *mtQHatInvPre; /* [numQ]  [mtilde*(Q/q_i)^-1]_{q_i} */
    uint64_t* QHatModbsk;               /* [numQ][numBsk] */
    uint64_t* QHatModmt;                /* [numQ] */
    uint64_t *QModbsk, *QModbskPre;     /* [numBsk] */
    uint64_t negQInvModmt;
    uint64_t *mtInvModbsk, *mtInvModbskPre; /* [numBsk] */
    uint64_t *tQHatInv, *tQHatInvPre;       /* [numQ] */
    uint64_t* qInvModbsk;                   /* [numQ][numBsk] */
    uint64_t *tQInvModbsk, *tQInvModbskPre; /* [numBsk] */
    uint64_t *BHatInv, *BHatInvPre;         /* [numB] */
    uint64_t* BHatModmsk;                   /* [numB] */
    uint64_t BInvModmsk, BInvModmskPre;
    uint64_t* BHatModq;                     /* [numB][numQ] */
    uint64_t *BModq, *BModqPre;             /* [numQ] */
    uint64_t *muBsk, *muQ;                  /* [.][2] */
};

static uint32_t mp_mul_small(uint64_t* big, uint32_t len, uint64_t m) {
    uint64_t carry = 0;
    for (uint32_t k = 0; k < len; ++k) {
        u128 t = (u128)big[k] * m + carry;
        big[k] = (uint64_t)t;
        carry  = (uint64_t)(t >> 64);
    }
    if (carry)
        big[len++] = carry;
    return len;
}
static int mp_less(const uint64_t* a, uint32_t la, const uint64_t* b, uint32_t lb) {
    if (la != lb)
        return la < lb;
    for (int k = (int)la - 1; k >= 0; --k)
        if (a[k] != b[k])
            return a[k] < b[k];
    return 0;
}
static uint64_t prod_mod_list(const uint64_t* m, uint32_t n, int skip, uint64_t mod) {
    uint64_t v = 1 % mod;
    for (uint32_t k = 0; k < n; ++k)
        if ((int)k != skip)
            v = orc_mulmod(v, m[k] % mod, mod);
    return v;
}

orc_behz* orc_behz_create(uint32_t N, uint32_t numQ, const uint64_t* q, uint64_t t) {
    orc_behz* h = (orc_behz*)calloc(1, sizeof(orc_behz));
    const uint64_t M = 2 * (uint64_t)N, mtilde = (uint64_t)1 << 16;
    const uint32_t numB = numQ, numBsk = numQ + 1;
    h->N = N, h->numQ = numQ, h->numBsk = numBsk, h->t = t;
    h->q      = (uint64_t*)malloc(8 * numQ);
    h->bsk    = (uint64_t*)malloc(8 * numBsk);
    h->psiBsk = (uint64_t*)malloc(8 * numBsk);
    memcpy(h->q, q, 8 * numQ);
    /* B = numQ primes below q.back(), then msk (:682-711) */
    uint64_t cur = q[numQ - 1];
    for (uint32_t i = 0; i < numB; ++i) {
        cur          = orc_previous_prime(cur, M);
        h->bsk[i]    = cur;
        h->psiBsk[i] = orc_root_of_unity(M, cur);
    }
    uint64_t msk = orc_previous_prime(h->bsk[numB - 1], M);
    uint32_t s   = orc_get_msb(msk);
    {
        uint64_t lhs[ORC_MAX_LIMBS + 16], rhs[ORC_MAX_LIMBS + 16];
        uint32_t ll, lr = 1;
        rhs[0] = 1;
        lr     = mp_mul_small(rhs, lr, M);
        lr     = mp_mul_small(rhs, lr, t);
        for (uint32_t i = 0; i < numQ; ++i)
            lr = mp_mul_small(rhs, lr, q[i]);
        for (;;) {
            ll     = 1;
            lhs[0] = 1;
            for (uint32_t i = 0; i < numB; ++i)
                ll = mp_mul_small(lhs, ll, h->bsk[i]);
            ll = mp_mul_small(lhs, ll, msk);
            if (!mp_less(lhs, ll, rhs, lr))
                break;
            msk = orc_next_prime(orc_first_prime(++s, M), M);
        }
    }
    h->bsk[numB]    = msk;
    h->psiBsk[numB] = orc_root_of_unity(M, msk);

    h->mtQHatInv      = (uint64_t*)malloc(8 * numQ);
    h->mtQHatInvPre   = (uint64_t*)malloc(8 * numQ);
    h->QHatModbsk     = (uint64_t*)malloc(8 * (size_t)numQ * numBsk);
    h->QHatModmt      = (uint64_t*)malloc(8 * numQ);
    h->QModbsk        = (uint64_t*)malloc(8 * numBsk);
    h->QModbskPre     = (uint64_t*)malloc(8 * numBsk);
    h->mtInvModbsk    = (uint64_t*)malloc(8 * numBsk);
    h->mtInvModbskPre = (uint64_t*)malloc(8 * numBsk);
    h->tQHatInv       = (uint64_t*)malloc(8 * numQ);
    h->tQHatInvPre    = (uint64_t*)malloc(8 * numQ);
    h->qInvModbsk     = (uint64_t*)malloc(8 * (size_t)numQ * numBsk);
    h->tQInvModbsk    = (uint64_t*)malloc(8 * numBsk);
    h->tQInvModbskPre = (uint64_t*)malloc(8 * numBsk);
    h->BHatInv        = (uint64_t*)malloc(8 * numB);
    h->BHatInvPre     = (uint64_t*)malloc(8 * numB);
    h->BHatModmsk     = (uint64_t*)malloc(8 * numB);
    h->BHatModq       = (uint64_t*)malloc(8 * (size_t)numB * numQ);
    h->BModq          = (uint64_t*)malloc(8 * numQ);
    h->BModqPre       = (uint64_t*)malloc(8 * numQ);
    h->muBsk          = (uint64_t*)malloc(16 * numBsk);
    h->muQ            = (uint64_t*)malloc(16 * numQ);
    for (uint32_t i = 0; i < numQ; ++i) {
        const uint64_t qi     = q[i];
        const uint64_t hatInv = orc_invmod(prod_mod_list(q, numQ, (int)i, qi), qi);
        h->tQHatInv[i]        = orc_mulmod(hatInv, t % qi, qi);                 /* :722-733 */
        h->tQHatInvPre[i]     = orc_prep_mod_mul_const(h->tQHatInv[i], qi);
        h->mtQHatInv[i]       = orc_mulmod(hatInv, mtilde % qi, qi);            /* :755-768 */
        h->mtQHatInvPre[i]    = orc_prep_mod_mul_const(h->mtQHatInv[i], qi);
        for (uint32_t j = 0; j < numBsk; ++j) {
            h->QHatModbsk[(size_t)i * numBsk + j] = prod_mod_list(q, numQ, (int)i, h->bsk[j]); /* :735-747 */
            h->qInvModbsk[(size_t)i * numBsk + j] = orc_invmod(qi % h->bsk[j], h->bsk[j]);      /* :749-755 */
        }
        /* Q/q_i mod 2^16: product of the other moduli mod 2^16 */
        uint64_t v = 1;
        for (uint32_t k = 0; k < numQ; ++k)
            if (k != i)
                v = (v * (q[k] & (mtilde - 1))) & (mtilde - 1);
        h->QHatModmt[i] = v;
        h->BModq[i]     = prod_mod_list(h->bsk, numB, -1, qi);                  /* :838-845 */
        h->BModqPre[i]  = orc_prep_mod_mul_const(h->BModq[i], qi);
        orc_barrett_mu128(qi, h->muQ + 2 * i);
    }
    {   /* [-Q^-1]_{mtilde} (:770-773): Q odd => invertible mod 2^16; Newton iteration */
        uint64_t Qm = 1;
        for (uint32_t k = 0; k < numQ; ++k)
            Qm = (Qm * (q[k] & (mtilde - 1))) & (mtilde - 1);
        uint64_t inv = 1;
        for (int it = 0; it < 5; ++it)
            inv = (inv * (2 - Qm * inv)) & (mtilde - 1);
        h->negQInvModmt = ((mtilde - 1) * inv) & (mtilde - 1);
    }
    for (uint32_t j = 0; j < numBsk; ++j) {
        const uint64_t bj    = h->bsk[j];
        h->QModbsk[j]        = prod_mod_list(q, numQ, -1, bj);                   /* :775-783 */
        h->QModbskPre[j]     = orc_prep_mod_mul_const(h->QModbsk[j], bj);
        h->mtInvModbsk[j]    = orc_invmod(mtilde % bj, bj);                      /* :785-793 */
        h->mtInvModbskPre[j] = orc_prep_mod_mul_const(h->mtInvModbsk[j], bj);
        h->tQInvModbsk[j]    = orc_mulmod(orc_invmod(h->QModbsk[j], bj), t % bj, bj); /* :795-804 */
        h->tQInvModbskPre[j] = orc_prep_mod_mul_const(h->tQInvModbsk[j], bj);
        orc_barrett_mu128(bj, h->muBsk + 2 * j);
    }
    for (uint32_t i = 0; i < numB; ++i) {
        const uint64_t bi = h->bsk[i];
        h->BHatInv[i]     = orc_invmod(prod_mod_list(h->bsk, numB, (int)i, bi), bi);   /* :806-817 */
        h->BHatInvPre[i]  = orc_prep_mod_mul_const(h->BHatInv[i], bi);
        h->BHatModmsk[i]  = prod_mod_list(h->bsk, numB, (int)i, msk);                 /* :829-834 */
        for (uint32_t j = 0; j < numQ; ++j)
            h->BHatModq[(size_t)i * numQ + j] = prod_mod_list(h->bsk, numB, (int)i, q[j]); /* :819-827 */
    }
    h->BInvModmsk    = orc_invmod(prod_mod_list(h->bsk, numB, -1, msk), msk);         /* :836-837 */
    h->BInvModmskPre = orc_prep_mod_mul_const(h->BInvModmsk, msk);
    return h;
}
void orc_behz_destroy(orc_behz* h) {
    if (!h)
        return;
    uint64_t* ptrs[] = {h->q, h->bsk, h->psiBsk, h->mtQHatInv, h->mtQHatInvPre, h->QHatModbsk, h->QHatModmt, h->QModbsk,
                        h->QModbskPre, h->mtInvModbsk, h->mtInvModbskPre, h->tQHatInv, h->tQHatInvPre, h->qInvModbsk,
                        h->tQInvModbsk, h->tQInvModbskPre, h->BHatInv, h->BHatInvPre, h->BHatModmsk, h->BHatModq,
                        h->BModq, h->BModqPre, h->muBsk, h->muQ};
    for (size_t i = 0; i < sizeof(ptrs) / sizeof(ptrs[0]); ++i)
        free(ptrs[i]);
    free(h);
}
uint32_t orc_behz_num_bsk(const orc_behz* h) { return h->numBsk; }
void orc_behz_get_bsk(const orc_behz* h, uint64_t* bsk, uint64_t* psiBsk) {
    memcpy(bsk, h->bsk, 8 * h->numBsk);
    memcpy(psiBsk, h->psiBsk, 8 * h->numBsk);
}

/* dcrtpoly-impl.h:1731-1774 (coefficient-domain core of FastBaseConvqToBskMontgomery) */
void orc_behz_q_to_bsk_montgomery(const orc_behz* h, const uint64_t* xq, uint64_t* outBsk) {
    const uint32_t N = h->N, numQ = h->numQ, numBsk = h->numBsk;
    const uint64_t mtilde = (uint64_t)1 << 16, half = mtilde >> 1, mask = mtilde - 1;
#pragma omp parallel for
    for (uint32_t k = 0; k < N; ++k) {
        uint64_t y[ORC_MAX_LIMBS];
        uint64_t rm = 0;
        for (uint32_t i = 0; i < numQ; ++i) {
            y[i] = orc_mod_mul_fast_const(xq[(size_t)i * N + k], h->mtQHatInv[i], h->q[i], h->mtQHatInvPre[i]);
            rm += y[i] * h->QHatModmt[i];
        }
        rm &= mask;
        rm *= h->negQInvModmt;
        rm &= mask;
        for (uint32_t j = 0; j < numBsk; ++j) {
            const uint64_t bj = h->bsk[j];
            u128 res          = 0;
            for (uint32_t i = 0; i < numQ; ++i)
                res += (u128)y[i] * h->QHatModbsk[(size_t)i * numBsk + j];
            uint64_t v = orc_barrett128((uint64_t)res, (uint64_t)(res >> 64), bj, h->muBsk[2 * j], h->muBsk[2 * j + 1]);
            uint64_t r = rm;
            if (rm >= half)
                r += bj - mtilde;
            r = orc_mod_mul_fast_const(r, h->QModbsk[j], bj, h->QModbskPre[j]);
            r = orc_mod_add_fast(r, v, bj);
            outBsk[(size_t)j * N + k] = orc_mod_mul_fast_const(r, h->mtInvModbsk[j], bj, h->mtInvModbskPre[j]);
        }
    }
}

/* dcrtpoly-impl.h:1791-1840 */
void orc_behz_fast_rns_floorq(const orc_behz* h, uint64_t* x) {
    const uint32_t N = h->N, numQ = h->numQ, numBsk = h->numBsk;
#pragma omp parallel for
    for (uint32_t k = 0; k < N; ++k) {
        for (uint32_t i = 0; i < numQ; ++i)
            x[(size_t)i * N + k] = orc_mod_mul_fast_const(x[(size_t)i * N + k], h->tQHatInv[i], h->q[i], h->tQHatInvPre[i]);
        for (uint32_t j = 0; j < numBsk; ++j) {
            const uint64_t bj = h->bsk[j];
            u128 aq           = 0;
            for (uint32_t i = 0; i < numQ; ++i)
                aq += (u128)x[(size_t)i * N + k] * h->qInvModbsk[(size_t)i * numBsk + j];
            uint64_t s = orc_barrett128((uint64_t)aq, (uint64_t)(aq >> 64), bj, h->muBsk[2 * j], h->muBsk[2 * j + 1]);
            uint64_t v = orc_mod_mul_fast_const(x[(size_t)(numQ + j) * N + k], h->tQInvModbsk[j], bj, h->tQInvModbskPre[j]);
            x[(size_t)(numQ + j) * N + k] = orc_mod_sub_fast(v, s, bj);
        }
    }
}

/* dcrtpoly-impl.h:1845-1929 */
void orc_behz_fast_base_conv_sk(const orc_behz* h, const uint64_t* x, uint64_t* outQ) {
    const uint32_t N = h->N, numQ = h->numQ, numBsk = h->numBsk, numB = numBsk - 1;
    const uint64_t msk = h->bsk[numB], mskHalf = msk >> 1;
#pragma omp parallel for
    for (uint32_t k = 0; k < N; ++k) {
        uint64_t b[ORC_MAX_LIMBS];
        uint64_t alpha = 0;
        for (uint32_t i = 0; i < numB; ++i) {
            b[i] = orc_mod_mul_fast_const(x[(size_t)(numQ + i) * N + k], h->BHatInv[i], h->bsk[i], h->BHatInvPre[i]);
            /* ModMul / ModAddEq with full reduction of both operands (ubintnat.h generic versions) */
            alpha = (alpha + orc_mulmod(b[i] % msk, h->BHatModmsk[i], msk)) % msk;
        }
        alpha = orc_mod_sub_fast(alpha, x[(size_t)(numQ + numB) * N + k], msk);
        alpha = orc_mod_mul_fast_const(alpha, h->BInvModmsk, msk, h->BInvModmskPre);
        for (uint32_t j = 0; j < numQ; ++j) {
            const uint64_t qj = h->q[j];
            u128 res          = 0;
            for (uint32_t i = 0; i < numB; ++i)
                res += (u128)b[i] * h->BHatModq[(size_t)i * numQ + j];
            uint64_t v = orc_barrett128((uint64_t)res, (uint64_t)(res >> 64), qj, h->muQ[2 * j], h->muQ[2 * j + 1]);
            uint64_t a = alpha;
            if (a > mskHalf)
                a = orc_mod_sub_fast(a, msk, qj);
            a = orc_mod_mul_fast_const(a, h->BModq[j], qj, h->BModqPre[j]);
            outQ[(size_t)j * N + k] = orc_mod_sub_fast(v, a, qj);
        }
    }
}

/* LeveledSHEBFVRNS::EvalMult, BEHZ branch (src/pke/lib/scheme/bfvrns/bfvrns-leveledshe.cpp:198-445):
 *   :302-325  FastBaseConvqToBskMontgomery on every input element, then SetFormat(EVALUATION)
 *   :327-372  tensor product over Q u Bsk (non-Karatsuba order: d0=a0*b0, d1=a0*b1 then += a1*b0, d2=a1*b1)
 *   :414-437  per product element: SetFormat(COEFFICIENT); FastRNSFloorq; FastBaseConvSK
 * ctxAll: limbs 0..numQ-1 = Q, numQ.. = Bsk.  Inputs [numQ][N] EVALUATION; outputs [numQ][N] COEFFICIENT. */
void orc_bfv_eval_mult_behz(const orc_behz* h, const orc_ctx* ctxAll, const uint64_t* a0, const uint64_t* a1,
                            const uint64_t* b0, const uint64_t* b1, uint64_t* d0, uint64_t* d1, uint64_t* d2) {
    const uint32_t N = h->N, numQ = h->numQ, numBsk = h->numBsk, tot = numQ + numBsk;
    const size_t qw = (size_t)numQ * N, aw = (size_t)tot * N;
    uint32_t* idxBsk = (uint32_t*)malloc(4 * numBsk);
    for (uint32_t j = 0; j < numBsk; ++j)
        idxBsk[j] = numQ + j;
    const uint64_t* in[4] = {a0, a1, b0, b1};
    uint64_t* ext[4];
    uint64_t* coef = (uint64_t*)malloc(8 * qw);
    for (int e = 0; e < 4; ++e) {
        ext[e] = (uint64_t*)malloc(8 * aw);
        memcpy(coef, in[e], 8 * qw);
        orc_ntt_inv_tower(ctxAll, coef, NULL, numQ, 1, 1);                 /* dcrtpoly-impl.h:1708-1712 */
        orc_behz_q_to_bsk_montgomery(h, coef, ext[e] + qw);
        orc_ntt_fwd_tower(ctxAll, ext[e] + qw, idxBsk, numBsk, 1, 1);
        memcpy(ext[e], in[e], 8 * qw);                                     /* Q limbs keep their NTT form */
    }
    uint64_t* prod[3];
    uint64_t* tmp = (uint64_t*)malloc(8 * (size_t)N);
    for (int e = 0; e < 3; ++e)
        prod[e] = (uint64_t*)malloc(8 * aw);
    for (uint32_t l = 0; l < tot; ++l) {
        const uint64_t m = orc_ctx_modulus(ctxAll, l);
        const size_t o   = (size_t)l * N;
        orc_vec_mul(prod[0] + o, ext[0] + o, ext[2] + o, N, m);
        orc_vec_mul(prod[1] + o, ext[0] + o, ext[3] + o, N, m);
        orc_vec_mul(tmp, ext[1] + o, ext[2] + o, N, m);
        orc_vec_add(prod[1] + o, prod[1] + o, tmp, N, m);
        orc_vec_mul(prod[2] + o, ext[1] + o, ext[3] + o, N, m);
    }
    uint64_t* out[3] = {d0, d1, d2};
    for (int e = 0; e < 3; ++e) {
        orc_ntt_inv_tower(ctxAll, prod[e], NULL, tot, 1, 1);
        orc_behz_fast_rns_floorq(h, prod[e]);
        orc_behz_fast_base_conv_sk(h, prod[e], out[e]);
        free(prod[e]);
    }
    for (int e = 0; e < 4; ++e)
        free(ext[e]);
    free(tmp), free(coef), free(idxBsk);
}

/* DCRTPolyImpl::ExpandCRTBasis / ExpandCRTBasisReverseOrder (dcrtpoly-impl.h:1088-1148).  ctxQP: limbs 0..nQ-1 = Q,
 * nQ.. = P.  x [nQ][N] in `inEval`; out [(nQ+nP)][N] in `resultEval`, Q rows first (reverse: P rows first).  Tables as
 * for orc_switch_crt_basis. */
void orc_expand_crt_basis(const orc_ctx* ctxQP, uint32_t nQ, uint32_t nP, const uint64_t* x, int inEval,
                          const uint64_t* QHatInvModq, const uint64_t* QHatInvModqPrecon, const uint64_t* QHatModp_pq,
                          const uint64_t* alphaQModp, const uint64_t* muP128, const double* qInv, int resultEval,
                          int reverse, uint64_t* out) {
    const uint32_t N = ctxQP->N;
    const size_t qw = (size_t)nQ * N, pw = (size_t)nP * N;
    uint64_t* coef = (uint64_t*)malloc(8 * qw);
    memcpy(coef, x, 8 * qw);
    if (inEval)
        for (uint32_t i = 0; i < nQ; ++i)
            ctx_inv(ctxQP, coef + (size_t)i * N, i);
    uint64_t* partP = (uint64_t*)malloc(8 * pw);
    orc_switch_crt_basis(coef, nQ, N, ctxQP->q, QHatInvModq, QHatInvModqPrecon, QHatModp_pq, alphaQModp, nP, ctxQP->q + nQ,
                         muP128, qInv, partP);
    uint64_t* qpart = reverse ? out + pw : out;
    uint64_t* ppart = reverse ? out : out + qw;
    memcpy(qpart, (resultEval && inEval) ? x : coef, 8 * qw); /* :1104-1105 */
    memcpy(ppart, partP, 8 * pw);
    if (resultEval) {
        if (!inEval)
            for (uint32_t i = 0; i < nQ; ++i)
                ctx_fwd(ctxQP, qpart + (size_t)i * N, i);
        for (uint32_t j = 0; j < nP; ++j)
            ctx_fwd(ctxQP, ppart + (size_t)j * N, nQ + j);
    }
    free(coef);
    free(partP);
}

/* DCRTPolyImpl::ApproxModUp (dcrtpoly-impl.h:935-963).  ctxQP: limbs 0..nQ-1 = Q, nQ.. = P.  x [nQ][N] in `inEval`;
 * out [(nQ+nP)][N] EVALUATION: the stored EVALUATION copy of the Q limbs is moved back (:943-951), the P part is
 * ApproxSwitchCRTBasis of the coefficient form (:948), every limb goes to EVALUATION (:958-960).  QHatModp [nQ][nP]. */
void orc_approx_mod_up(const orc_ctx* ctxQP, uint32_t nQ, uint32_t nP, const uint64_t* x, int inEval,
                       const uint64_t* QHatInvModq, const uint64_t* QHatInvModqPrecon, const uint64_t* QHatModp,
                       const uint64_t* muP128, uint64_t* out) {
    const uint32_t N = ctxQP->N;
    const size_t qw = (size_t)nQ * N;
    uint64_t* coef = (uint64_t*)malloc(8 * qw);
    memcpy(coef, x, 8 * qw);
    if (inEval)
        for (uint32_t i = 0; i < nQ; ++i)
            ctx_inv(ctxQP, coef + (size_t)i * N, i);
    orc_approx_switch_crt_basis(coef, nQ, N, ctxQP->q, QHatInvModq, QHatInvModqPrecon, QHatModp, nP, ctxQP->q + nQ, muP128,
                                out + qw);
    memcpy(out, inEval ? x : coef, 8 * qw);
    for (uint32_t i = 0; i < nQ + nP; ++i)
        if (!(inEval && i < nQ)) /* SetFormat(EVALUATION) is a no-op on limbs that already are */
            ctx_fwd(ctxQP, out + (size_t)i * N, i);
    free(coef);
}

/* DCRTPolyImpl::ExpandCRTBasisQlHat (dcrtpoly-impl.h:1167-1187): x [sizeQl][N] -> out [sizeQ][N], limbs below sizeQl times
 * QlHatModq[i] (ModMulFastConst), the appended limbs zero */
void orc_expand_crt_basis_ql_hat(const uint64_t* x, uint32_t sizeQl, uint32_t N, const uint64_t* q, const uint64_t* QlHatModq,
                                 uint32_t sizeQ, uint64_t* out) {
    for (uint32_t i = 0; i < sizeQl; ++i) {
        const uint64_t pre = orc_prep_mod_mul_const(QlHatModq[i], q[i]);
        for (uint32_t r = 0; r < N; ++r)
            out[(size_t)i * N + r] = orc_mod_mul_fast_const(x[(size_t)i * N + r], QlHatModq[i], q[i], pre);
    }
    memset(out + (size_t)sizeQl * N, 0, 8 * (size_t)(sizeQ - sizeQl) * N);
}

/* LeveledSHEBase::EvalSquareCore for a 2-element ciphertext (base-leveledshe.cpp:646-664): towers [nLimbs][N] over q[] */
void orc_eval_square_core(const uint64_t* a0, const uint64_t* a1, uint32_t nLimbs, uint32_t N, const uint64_t* q, uint64_t* d0,
                          uint64_t* d1, uint64_t* d2) {
    for (uint32_t i = 0; i < nLimbs; ++i) {
        const size_t o = (size_t)i * N;
        orc_vec_mul(d0 + o, a0 + o, a0 + o, N, q[i]); /* cv[0] * cv[0] */
        orc_vec_mul(d1 + o, a0 + o, a1 + o, N, q[i]); /* cv[0] * cv[1] */
        orc_vec_add(d1 + o, d1 + o, d1 + o, N, q[i]); /* cvr.back() += cvr.back() */
        orc_vec_mul(d2 + o, a1 + o, a1 + o, N, q[i]); /* cv[1] * cv[1] */
    }
}

/* the "ModRaise" constructor DCRTPolyImpl(const PolyType&, params) (dcrtpoly-impl.h:87-93): x [N] modulo q[0] ->
 * out [nLimbs][N], limb 0 a copy, every other limb SwitchModulus(q[0] -> q[i]) */
void orc_mod_raise(const uint64_t* x, uint32_t N, const uint64_t* q, uint32_t nLimbs, uint64_t* out) {
    for (uint32_t i = 0; i < nLimbs; ++i) {
        memcpy(out + (size_t)i * N, x, 8 * (size_t)N);
        if (i)
            orc_switch_modulus(out + (size_t)i * N, N, q[0], q[i]);
    }
}

/* DCRTPolyImpl::CRTDecompose(baseBits) (dcrtpoly-impl.h:230-285; the digit decomposition of KeySwitchBV, keyswitch-bv.cpp:254).
 * x [nLimbs][N] in COEFFICIENT format.  baseBits == 0 (:237-251): tower i of the result = limb i of x switched (centred,
 * PolyImpl::SwitchModulus) into every other modulus, limb i itself kept, EVALUATION.  baseBits > 0 (:254-284): limb i is cut into
 * ceil(msb(q_i) / baseBits) digits of baseBits bits (PolyImpl::BaseDecompose, poly-impl.h:524-547 -> GetDigitAtIndexForBase,
 * ubintnat.h:1721-1729: bit by bit, least significant digit first), every digit switched into every other modulus, then SwitchFormat;
 * towers in the order (limb 0's digits, limb 1's digits, ...).  Returns the number of towers; out (may be null: count only)
 * [towers][nLimbs][N] EVALUATION. */
uint32_t orc_crt_decompose(const orc_ctx* c, const uint64_t* x, uint32_t nLimbs, uint32_t baseBits, uint64_t* out) {
    const uint32_t N = c->N;
    uint32_t towers = 0;
    for (uint32_t i = 0; i < nLimbs; ++i) {
        const uint32_t nBits = orc_get_msb(c->q[i]);
        const uint32_t nW = baseBits == 0 ? 1 : (nBits / baseBits + (nBits % baseBits != 0));
        for (uint32_t w = 0; w < nW && out; ++w) {
            uint64_t* T = out + (size_t)(towers + w) * nLimbs * N;
            for (uint32_t k = 0; k < nLimbs; ++k) {
                uint64_t* row = T + (size_t)k * N;
                for (uint32_t r = 0; r < N; ++r) {
                    const uint64_t v = x[(size_t)i * N + r];
                    if (baseBits == 0)
                        row[r] = v;
                    else {
                        uint64_t digit = 0; /* GetDigitAtIndexForBase(w + 1, 1 << baseBits): bits newIndex .. of the value, 1-based */
                        uint32_t newIndex = 1 + w * baseBits;
                        for (uint64_t b = 1; b < ((uint64_t)1 << baseBits); b <<= 1, ++newIndex)
                            digit += ((v >> (newIndex - 1)) & 1u) * b; /* (newIndex - 1 < 64 for every window the callers ask for) */
                        row[r] = digit;
                    }
                }
                if (k != i)
                    orc_switch_modulus(row, N, c->q[i], c->q[k]);
            }
            orc_ntt_fwd_tower(c, T, NULL, nLimbs, 1, 0);
        }
        towers += nW;
    }
    return towers;
}

/* DCRTPolyImpl::FastExpandCRTBasisPloverQ (dcrtpoly-impl.h:1151-1164), COEFFICIENT: x [nQ][N] over Q ->
 * out [(nQl+nPl)][N] = [Ql | Pl]; tables named as in CRTBasisExtensionPrecomputations. */
void orc_fast_expand_crt_basis_p_over_q(const uint64_t* x, uint32_t nQ, uint32_t N, const uint64_t* q,
                                        const uint64_t* mPlQHatInvModq, const uint64_t* mPlQHatInvModqPrecon,
                                        const uint64_t* qInvModp /*[nQ][nPl]*/, uint32_t nPl, const uint64_t* pl,
                                        const uint64_t* muPl128, const uint64_t* PlHatInvModp, const uint64_t* PlHatInvModpPrecon,
                                        const uint64_t* PlHatModq_qp /*[nQl][nPl]*/, const uint64_t* alphaPlModq /*[nPl+1][nQl]*/,
                                        uint32_t nQl, const uint64_t* ql, const uint64_t* muQl128, const double* pInv,
                                        uint64_t* out) {
    uint64_t* partPl = out + (size_t)nQl * N;
    orc_approx_switch_crt_basis(x, nQ, N, q, mPlQHatInvModq, mPlQHatInvModqPrecon, qInvModp, nPl, pl, muPl128, partPl);
    orc_switch_crt_basis(partPl, nPl, N, pl, PlHatInvModp, PlHatInvModpPrecon, PlHatModq_qp, alphaPlModq, nQl, ql, muQl128,
                         pInv, out);
}

/* ------------------------------------------------------------------------------------------
 * f3: sampled towers (csrc/sampler_kernels.h) — PARITY UNPINNED against the reference's WORDS by construction.
 * The reference draws uniform residues (math/discreteuniformgenerator.h:55-77), Peikert-inversion Gaussian integers
 * (math/discretegaussiangenerator-impl.h:75-115) and ternary values from ONE sequential Blake2 stream per thread; a device sampler
 * needs a counter-based generator, so for a given seed the words are not the reference's ("gives up bit-parity with
 * Blake2; keep optional").  What IS restated from the reference, line for line: the table of Initialize() (:75-89), the inversion rule of
 * GenerateInt() (:101-107: seed = U - 0.5, tmp = |seed| - a/2, 0 if tmp <= 0, else 1 + lower_bound index, sign of seed), the range
 * [0, modulus) of the uniform generator and the storage of a negative integer k as q - |k| (dcrtpoly-impl.h:126-150).  The generator
 * is Philox4x32-10 (Salmon et al., SC'11) — pinned by the Random123 known-answer vectors in tests/test_sampler.py — with
 * key = seed, counter = (element.lo, element.hi, draw, stream).
 * ---------------------------------------------------------------------------------------- */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0, c1 = n1, c2 = n2, c3 = n3;
        k0 += 0x9E3779B9u, k1 += 0xBB67AE85u;
    }
    out[0] = c0, out[1] = c1, out[2] = c2, out[3] = c3;
}
static void orc_philox_draw(uint64_t e, uint32_t d, uint32_t stream, uint64_t seed, uint32_t r[4]) {
    const uint32_t ctr[4] = {(uint32_t)e, (uint32_t)(e >> 32), d, stream}, key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    orc_philox4x32_10(ctr, key, r);
}
/* out[batch][nLimbs][N]: every word uniform in [0, q[l]) */
void orc_sample_uniform(uint64_t* out, const uint64_t* q, uint32_t nLimbs, uint32_t batch, size_t N, uint64_t seed, uint32_t stream) {
    for (uint64_t e = 0; e < (uint64_t)batch * nLimbs * N; ++e) {
        const uint64_t ql = q[(e / N) % nLimbs];
        const unsigned bits = 64u - (unsigned)__builtin_clzll(ql);
        const uint64_t mask = bits >= 64 ? ~0ull : ((1ull << bits) - 1ull);
        for (uint32_t d = 0;; ++d) {
            uint32_t r[4];
            orc_philox_draw(e, d, stream, seed, r);
            const uint64_t x0 = (((uint64_t)r[1] << 32) | r[0]) & mask, x1 = (((uint64_t)r[3] << 32) | r[2]) & mask;
            if (x0 < ql) { out[e] = x0; break; }
            if (x1 < ql) { out[e] = x1; break; }
        }
    }
}
/* the table of DiscreteGaussianGeneratorImpl::Initialize (discretegaussiangenerator-impl.h:75-89); returns its length (<= cap), *a */
uint32_t orc_dgg_table(double sigma, double* vals, uint32_t cap, double* a) {
    const double M = 12.00610553538285;
    const int64_t fin = (int64_t)ceil(sigma * M);
    if (fin > (int64_t)cap)
        return 0;
    const double variance = 2 * sigma * sigma;
    double cusum = 0.0;
    for (int64_t x = 1; x <= fin; ++x)
        vals[x - 1] = (cusum += exp(-((double)(x * x) / variance)));
    *a = 1.0 / (2 * cusum + 1.0);
    for (int64_t x = 0; x < fin; ++x)
        vals[x] *= *a;
    return (uint32_t)fin;
}
static void orc_store_signed(uint64_t* out, const uint64_t* q, uint32_t nLimbs, size_t N, uint32_t tb, size_t j, int64_t k) {
    for (uint32_t l = 0; l < nLimbs; ++l)
        out[((size_t)tb * nLimbs + l) * N + j] = k < 0 ? q[l] - (uint64_t)(-k) : (uint64_t)k;
}
/* GenerateInt (:101-107) per coefficient, the integer stored modulo every limb; ints (may be NULL) receives the signed integers */
void orc_sample_gaussian(uint64_t* out, int64_t* ints, const uint64_t* q, uint32_t nLimbs, uint32_t batch, size_t N, double sigma,
                         uint64_t seed, uint32_t stream) {
    double vals[4096], a = 0.0;
    const uint32_t n = orc_dgg_table(sigma, vals, 4096, &a);
    for (uint64_t e = 0; e < (uint64_t)batch * N; ++e) {
        uint32_t r[4];
        orc_philox_draw(e, 0, stream, seed, r);
        const uint64_t m = ((((uint64_t)r[1] << 32) | r[0]) >> 11);
        const double s   = (double)m * (1.0 / 9007199254740992.0) - 0.5;
        const double tmp = fabs(s) - a / 2;
        int64_t k = 0;
        if (tmp > 0.0) {
            uint32_t lo = 0, hi = n; /* std::lower_bound */
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (vals[mid] < tmp)
                    lo = mid + 1;
                else
                    hi = mid;
            }
            if (lo >= n)
                lo = n - 1;
            k = s > 0.0 ? (int64_t)lo + 1 : -((int64_t)lo + 1);
        }
        if (ints)
            ints[e] = k;
        orc_store_signed(out, q, nLimbs, N, (uint32_t)(e / N), (size_t)(e % N), k);
    }
}
/* uniform in {-1, 0, 1} (TernaryUniformGeneratorImpl::GenerateVector, h = 0): two bits until they are not 3 */
void orc_sample_ternary(uint64_t* out, int64_t* ints, const uint64_t* q, uint32_t nLimbs, uint32_t batch, size_t N, uint64_t seed,
                        uint32_t stream) {
    for (uint64_t e = 0; e < (uint64_t)batch * N; ++e) {
        int64_t k = 2;
        for (uint32_t d = 0; k == 2; ++d) {
            uint32_t r[4];
            orc_philox_draw(e, d, stream, seed, r);
            for (int w = 0; w < 4 && k == 2; ++w)
                for (int i = 0; i < 16 && k == 2; ++i) {
                    const uint32_t t = (r[w] >> (2 * i)) & 3u;
                    if (t != 3u)
                        k = (int64_t)t - 1;
                }
        }
        if (ints)
            ints[e] = k;
        orc_store_signed(out, q, nLimbs, N, (uint32_t)(e / N), (size_t)(e % N), k);
    }
}
