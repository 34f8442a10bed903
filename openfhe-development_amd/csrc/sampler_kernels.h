// sampler_kernels.h — uniform, discrete-Gaussian and ternary residue towers generated ON the device.
//
// Replaces, as an OPTION, the host loops of DiscreteUniformGeneratorImpl::GenerateVector (math/discreteuniformgenerator.h:55-77:
// every entry uniform in [0, modulus)), DiscreteGaussianGeneratorImpl::GenerateIntVector (math/discretegaussiangenerator-impl.h:75-115:
// Peikert's inversion over the table Initialize() builds) and TernaryUniformGeneratorImpl::GenerateVector (h = 0: every entry uniform in
// {-1, 0, 1}) behind the sampling constructors of DCRTPolyImpl (dcrtpoly-impl.h:126-205), which key generation calls once per digit and
// key (keyswitch-hybrid.cpp:96-103): a bootstrapping key set is 5-6 GB of uniform words sampled on the host and uploaded.
//
// The reference draws from ONE sequential Blake2 stream per thread; a device sampler needs a counter-based generator, so the WORDS differ
// from the reference's for the same seed ("gives up bit-parity with Blake2; keep optional") — the distributions are the
// reference's.  Generator: Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11), key = the
// caller's 64-bit seed, counter = (element index, draw number, sampler stream id): every element owns a sub-stream, so the result does not
// depend on the launch geometry and the oracle restates it word for word (oracle/fhe_oracle.c orc_sample_*).
#ifndef FHE_SAMPLER_KERNELS_H
#define FHE_SAMPLER_KERNELS_H
#include "ntt_kernels.h"

namespace fhe {

struct Philox4 {
    uint32_t v[4];
};
FHE_HD Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0, c1 = n1, c2 = n2, c3 = n3;
        k0 += 0x9E3779B9u, k1 += 0xBB67AE85u;
    }
    return Philox4{{c0, c1, c2, c3}};
}

struct SampleArgs {
    uint64_t* out;        // [batch][nLimbs][N]
    const uint64_t* q;    // [ctxLimbs]
    uint32_t logN, nLimbs, batch;
    uint64_t seed;        // Philox key
    uint32_t stream;      // sampler stream id (counter word 3): one per sampled tower set
    uint32_t kind;        // 0 uniform, 1 discrete Gaussian, 2 ternary
    const double* cdf;    // Gaussian: the reference's table m_vals (Initialize()), length cdfLen; a = 1 / (2 * cusum + 1)
    uint32_t cdfLen;
    double a;
    LimbSel sel;
};

// uniform in [0, q): candidates of bitlen(q) bits until one is below q (at least every second is); draw d of element e uses counter
// (e.lo, e.hi, d, stream) and yields two 64-bit candidates
FHE_HD uint64_t sample_uniform_word(uint64_t e, uint64_t q, uint64_t seed, uint32_t stream) {
    const uint32_t bits = 64u - (uint32_t)__builtin_clzll(q);
    const uint64_t mask = bits >= 64 ? ~0ull : ((1ull << bits) - 1ull);
    for (uint32_t d = 0;; ++d) {
        const Philox4 r = philox4x32_10((uint32_t)e, (uint32_t)(e >> 32), d, stream, (uint32_t)seed, (uint32_t)(seed >> 32));
        const uint64_t x0 = (((uint64_t)r.v[1] << 32) | r.v[0]) & mask, x1 = (((uint64_t)r.v[3] << 32) | r.v[2]) & mask;
        if (x0 < q)
            return x0;
        if (x1 < q)
            return x1;
    }
}
// Peikert's inversion, discretegaussiangenerator-impl.h:101-107: seed = U[0,1) - 0.5, tmp = |seed| - a / 2, 0 if tmp <= 0, else
// (1 + index of the first table entry >= tmp) with the sign of seed.  U = 53 random bits * 2^-53.
FHE_HD int64_t sample_gaussian_int(uint64_t e, const double* cdf, uint32_t n, double a, uint64_t seed, uint32_t stream) {
    const Philox4 r  = philox4x32_10((uint32_t)e, (uint32_t)(e >> 32), 0u, stream, (uint32_t)seed, (uint32_t)(seed >> 32));
    const uint64_t m = ((((uint64_t)r.v[1] << 32) | r.v[0]) >> 11);
    const double s   = (double)m * (1.0 / 9007199254740992.0) - 0.5;
    const double tmp = (s < 0 ? -s : s) - a / 2;
    if (tmp <= 0.0)
        return 0;
    uint32_t lo = 0, hi = n;  // std::lower_bound
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (cdf[mid] < tmp)
            lo = mid + 1;
        else
            hi = mid;
    }
    if (lo >= n)
        lo = n - 1;  // (the reference throws here: probability below 2^-100)
    const int64_t v = (int64_t)lo + 1;
    return s > 0.0 ? v : -v;
}
// uniform in {-1, 0, 1}: two random bits until they are not 3 (sixteen tries per 32-bit word)
FHE_HD int64_t sample_ternary_int(uint64_t e, uint64_t seed, uint32_t stream) {
    for (uint32_t d = 0;; ++d) {
        const Philox4 r = philox4x32_10((uint32_t)e, (uint32_t)(e >> 32), d, stream, (uint32_t)seed, (uint32_t)(seed >> 32));
        for (int w = 0; w < 4; ++w)
            for (int i = 0; i < 16; ++i) {
                const uint32_t t = (r.v[w] >> (2 * i)) & 3u;
                if (t != 3u)
                    return (int64_t)t - 1;
            }
    }
}

// one lane per coefficient: uniform towers draw per (tower, limb, coefficient); Gaussian / ternary towers draw ONE integer per
// (tower, coefficient) and store it modulo every limb (dcrtpoly-impl.h:126-150: negative k as q - |k|)
FHE_GLOBAL void FHE_LAUNCH_BOUNDS(kThreads) sample_kernel(const SampleArgs a) {
    const uint32_t N   = 1u << a.logN;
    const uint64_t gid = (uint64_t)FHE_BID * kThreads + FHE_TID;
    if (a.kind == 0) {
        const uint64_t total = (uint64_t)a.batch * a.nLimbs * N;
        if (gid >= total)
            return;
        const uint32_t row = (uint32_t)(gid >> a.logN);
        const uint64_t q   = a.q[a.sel.idx[row % a.nLimbs]];
        a.out[gid]         = sample_uniform_word(gid, q, a.seed, a.stream);
        return;
    }
    const uint64_t total = (uint64_t)a.batch * N;
    if (gid >= total)
        return;
    const uint32_t tb = (uint32_t)(gid >> a.logN), j = (uint32_t)gid & (N - 1u);
    const int64_t k   = a.kind == 1 ? sample_gaussian_int(gid, a.cdf, a.cdfLen, a.a, a.seed, a.stream) : sample_ternary_int(gid, a.seed, a.stream);
    for (uint32_t l = 0; l < a.nLimbs; ++l) {
        const uint64_t q = a.q[a.sel.idx[l]];
        a.out[(((uint64_t)tb * a.nLimbs + l) << a.logN) + j] = k < 0 ? q - (uint64_t)(-k) : (uint64_t)k;
    }
}

}  // namespace fhe
#endif
