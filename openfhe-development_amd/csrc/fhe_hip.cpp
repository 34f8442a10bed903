// fhe_hip.cpp — implementation of the C ABI in include/fhe_hip.h (compiled by hipcc for gfx950).
//
// Host logic only: contexts (device twiddle tables), pass planning for the NTT, conversion / key-switch
// plans (tables as CryptoParametersRNS::PrecomputeCRTTables builds them,
// src/pke/lib/schemerns/rns-cryptoparameters.cpp:80-350) and the launch sequences that replace
// KeySwitchHYBRID (src/pke/lib/keyswitch/keyswitch-hybrid.cpp:308-435), ApproxModDown
// (src/core/include/lattice/hal/default/dcrtpoly-impl.h:966-1005) and DropLastElementAndScale (:693-712).
// All arithmetic runs in the kernels of ntt_kernels.h / elemwise_kernels.h / basis_kernels.h.
#include "../../include/fhe_hip.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <vector>

#include "basis_kernels.h"
#include "bfv_kernels.h"
#include "elemwise_kernels.h"
#include "host_math.h"
#include "ntt_kernels.h"
#include "ntt_static.h"
#include "ntt_row8.h"
#include "rt.h"
#include "sampler_kernels.h"

using namespace fhe;

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_lastError;
static fhe_status fail(fhe_status code, const std::string& msg) {
    g_lastError = msg;
    return code;
}
#define RT_CHECK(expr)                                                                  \
    do {                                                                                \
        const char* _e = (expr);                                                        \
        if (_e)                                                                         \
            return fail(FHE_ERR_DEVICE, std::string(#expr) + ": " + _e);                \
    } while (0)
#define ARG_CHECK(cond, msg)                  \
    do {                                      \
        if (!(cond))                          \
            return fail(FHE_ERR_ARG, msg);    \
    } while (0)

extern "C" const char* fhe_last_error(void) { return g_lastError.c_str(); }
extern "C" const char* fhe_version(void) {
#ifdef FHE_EMU
    return "fhe_hip 0.1 (TEST lane emulator build — not a product build)";
#else
    return "fhe_hip 0.1 (gfx950)";
#endif
}
extern "C" int fhe_device_count(void) { return rt::device_count(); }

// ------------------------------------------------------------------------------------------------
// host math
// ------------------------------------------------------------------------------------------------
namespace fhe {
namespace host {
bool is_prime(uint64_t n) {
    static const uint64_t w[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
    if (n < 2)
        return false;
    for (uint64_t p : w) {
        if (n == p)
            return true;
        if (n % p == 0)
            return false;
    }
    uint64_t d = n - 1;
    int s      = 0;
    while (!(d & 1)) {
        d >>= 1;
        ++s;
    }
    for (uint64_t a : w) {
        uint64_t x = powmod(a, d, n);
        if (x == 1 || x == n - 1)
            continue;
        bool comp = true;
        for (int r = 1; r < s && comp; ++r) {
            x = mulmod(x, x, n);
            if (x == n - 1)
                comp = false;
        }
        if (comp)
            return false;
    }
    return true;
}
bool is_primitive_root_2n(uint64_t psi, uint64_t twoN, uint64_t q) {
    return psi != 0 && psi < q && powmod(psi, twoN / 2, q) == q - 1;
}
uint64_t first_prime(uint32_t bits, uint64_t m) {  // nbtheory-impl.h:329-347
    uint64_t q = (uint64_t)1 << bits, r = q % m, c = q + 1 - r;
    if (r > 0)
        c += m;
    while (!is_prime(c))
        c += m;
    return c;
}
uint64_t last_prime(uint32_t bits, uint64_t m) {  // nbtheory-impl.h:349-373
    uint64_t q = (uint64_t)1 << bits, r = q % m, c = q + 1 - r;
    if (r < 2)
        c -= m;
    while (!is_prime(c))
        c -= m;
    return c;
}
uint64_t previous_prime(uint64_t q, uint64_t m) {
    uint64_t c = q - m;
    while (!is_prime(c))
        c -= m;
    return c;
}
uint64_t next_prime(uint64_t q, uint64_t m) {
    uint64_t c = q + m;
    while (!is_prime(c))
        c += m;
    return c;
}
uint64_t min_root_of_unity(uint64_t m, uint64_t q) {  // nbtheory-impl.h:183-231
    uint64_t e = (q - 1) / m, root = 0;
    for (uint64_t g = 2; g < q; ++g) {
        uint64_t r = powmod(g, e, q);
        if (powmod(r, m / 2, q) == q - 1) {
            root = r;
            break;
        }
    }
    uint64_t sq = mulmod(root, root, q), x = root, best = root;
    for (uint64_t i = 1; i < m / 2; ++i) {
        x = mulmod(x, sq, q);
        if (x < best)
            best = x;
    }
    return best;
}
uint64_t prod_mod(const std::vector<uint64_t>& mods, int skip, uint64_t mod) {
    uint64_t v = 1 % mod;
    for (int k = 0; k < (int)mods.size(); ++k)
        if (k != skip)
            v = mulmod(v, mods[k] % mod, mod);
    return v;
}
}  // namespace host
}  // namespace fhe

// ------------------------------------------------------------------------------------------------
// host-side parameter helpers
// ------------------------------------------------------------------------------------------------
extern "C" uint64_t fhe_param_first_prime(uint32_t bits, uint64_t m) { return host::first_prime(bits, m); }
extern "C" uint64_t fhe_param_last_prime(uint32_t bits, uint64_t m) { return host::last_prime(bits, m); }
extern "C" uint64_t fhe_param_next_prime(uint64_t q, uint64_t m) { return host::next_prime(q, m); }
extern "C" uint64_t fhe_param_previous_prime(uint64_t q, uint64_t m) { return host::previous_prime(q, m); }
extern "C" uint64_t fhe_param_root_of_unity(uint64_t m, uint64_t q) {
    if (m == 0 || (m & (m - 1)) || (q - 1) % m != 0)
        return 0;
    return host::min_root_of_unity(m, q);
}
extern "C" fhe_status fhe_param_dcrt_chain(uint32_t order, uint32_t nLimbs, uint32_t bits, uint64_t* q, uint64_t* psi) {
    ARG_CHECK(q && psi && nLimbs >= 1, "fhe_param_dcrt_chain: null argument");
    ARG_CHECK(bits >= 4 && bits <= 60, "Invalid bits for ILDCRTParams");  // ildcrtparams.h:103-104 (MAX_MODULUS_SIZE 60)
    ARG_CHECK(order >= 2 && (order & (order - 1)) == 0, "fhe_param_dcrt_chain: order must be a power of two");
    uint64_t cur = host::last_prime(bits, order);
    for (uint32_t i = 0; i < nLimbs; ++i) {
        if (i)
            cur = host::previous_prime(cur, order);
        q[i]   = cur;
        psi[i] = host::min_root_of_unity(order, cur);
    }
    return FHE_OK;
}
extern "C" uint32_t fhe_param_find_automorphism_index_2n_complex(int32_t index, uint32_t m) {
    if (m < 4 || (m & (m - 1)))
        return 0;  // "m should be a power of two."
    if (index == 0)
        return 1;
    if (index == (int32_t)m - 1)
        return (uint32_t)index;
    const uint64_t mask = m - 1;
    uint64_t g0         = 5;
    if (index < 0) {  // 5^-1 mod 2^k by Hensel lifting (every step doubles the number of correct bits)
        uint64_t inv = 1;
        for (int it = 0; it < 6; ++it)
            inv = (inv * (2 - 5 * inv)) & mask;
        g0 = inv;
    }
    uint64_t g = g0;
    for (uint32_t j = 1, n = (uint32_t)(index < 0 ? -(int64_t)index : index); j < n; ++j)
        g = (g * g0) & mask;
    return (uint32_t)g;
}
extern "C" uint32_t fhe_param_select_p(uint32_t logN, uint32_t sizeQ, const uint64_t* q, uint32_t numPartQ,
                                       uint32_t auxBits, uint64_t* p, uint64_t* psiP) {
    if (!q || !p || !psiP || numPartQ == 0 || sizeQ == 0 || auxBits < 4 || auxBits > 60)
        return 0;
    const uint32_t a = (sizeQ + numPartQ - 1) / numPartQ;
    uint32_t maxBits = 0;
    for (uint32_t j = 0; j < numPartQ; ++j) {  // bit length of each composite digit (exact, multi-word product)
        std::vector<uint64_t> big(1, 1);
        for (uint32_t i = a * j; i < (j + 1) * a && i < sizeQ; ++i) {
            uint64_t carry = 0;
            for (auto& w : big) {
                host::u128 t = (host::u128)w * q[i] + carry;
                w            = (uint64_t)t;
                carry        = (uint64_t)(t >> 64);
            }
            if (carry)
                big.push_back(carry);
        }
        maxBits = std::max(maxBits, (uint32_t)(64 * (big.size() - 1) + host::bitlen(big.back())));
    }
    const uint32_t sizeP = (maxBits + auxBits - 1) / auxBits;
    if (sizeP > 64)
        return 0;
    const uint64_t step = 2ull << logN;
    uint64_t prev       = host::first_prime(auxBits, step);
    for (uint32_t i = 0; i < sizeP; ++i) {
        bool inQ;
        do {
            p[i] = host::previous_prime(prev, step);
            inQ  = false;
            for (uint32_t j = 0; j < sizeQ; ++j)
                inQ |= (p[i] == q[j]);
            prev = p[i];
        } while (inQ);
        psiP[i] = host::min_root_of_unity(step, p[i]);
    }
    return sizeP;
}

static uint32_t env_u32(const char* name, uint32_t dflt);

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
struct fhe_ctx {
    uint32_t logN = 0, N = 0, L = 0;
    int device    = 0;
    uint32_t cus  = 0;  // compute units of the device (grid of the persistent row pass, ntt_row8.h)
    std::vector<uint64_t> q, psi;
    // device tables
    TwPair* d_tw      = nullptr;  // [L][N] forward
    TwPair* d_twInv   = nullptr;  // [L][N] inverse
    TwPair* d_fin     = nullptr;  // [L][2] {N^-1, TableInv[1]*N^-1}
    std::vector<TwPair> h_fin;    // host copy (the key-switch plans derive tables with extra per-limb factors from it)
    TwPair* d_twRow   = nullptr;  // [L][N/4096][15][256] lane-major twiddles of the row pass's bit-0 step (N >= 4096), forward
    TwPair* d_twRowInv = nullptr; // ... inverse
    uint64_t* d_q     = nullptr;  // [L]
    LimbConst* d_lc   = nullptr;  // [L]
    uint64_t* d_red   = nullptr;  // [L] redM | redR << 32: quotient estimate of the static NTT kernels (ntt_static.h)
    uint64_t* d_mu128 = nullptr;  // [L][2]
    std::vector<void*> owned;     // every device allocation made for tables (freed in destroy)
    // discrete-Gaussian inversion tables by standard deviation (fhe_sample_gaussian): {device table, length, a}
    std::map<double, std::tuple<const double*, uint32_t, double>> dggTabs;
    // cached per-level rescale tables (ckksrns-cryptoparameters.cpp:60-81): sizeQl -> {A, B} device arrays
    std::map<uint32_t, std::pair<TwPair*, TwPair*>> rescaleTabs;
    std::map<std::vector<uint64_t>, TwPair*> constTabs;  // per-limb constants of callers' tables, by content (const_table); the map owns
                                                         // them and is emptied when it passes kMaxConstTabs entries (const_tables_trim)
    std::shared_mutex constTabsGate;  // shared: an entry point between asking for its tables and enqueuing the launches that read them
    // cached ModReduce tables per (sizeQl, t): [0..l) = A_i, [l..2l) = B_i, [2l] = negtInvModq   (fhe_mod_reduce)
    std::map<std::pair<uint32_t, uint64_t>, TwPair*> modReduceTabs;
    std::mutex cacheMutex;        // guards the lazily filled caches above (callers may be OpenMP threads)
};

static fhe_status upload(fhe_ctx* c, const void* host, size_t bytes, void** dev) {
    RT_CHECK(rt::dmalloc(dev, bytes));
    c->owned.push_back(*dev);
    RT_CHECK(rt::h2d(*dev, host, bytes, nullptr));
    RT_CHECK(rt::sync(nullptr));
    return FHE_OK;
}

extern "C" fhe_status fhe_ctx_create(uint32_t logN, uint32_t nLimbs, const uint64_t* q, const uint64_t* psi,
                                     int device, fhe_ctx** out) {
    ARG_CHECK(out != nullptr, "fhe_ctx_create: out is null");
    ARG_CHECK(logN >= 4 && logN <= 17, "fhe_ctx_create: logN must be in [4,17]");
    ARG_CHECK(nLimbs >= 1 && nLimbs <= (uint32_t)kMaxLimbs, "fhe_ctx_create: nLimbs must be in [1,256]");
    ARG_CHECK(q && psi, "fhe_ctx_create: null modulus/root array");
    if (rt::device_count() <= 0)
        return fail(FHE_ERR_DEVICE, "fhe_ctx_create: no HIP device available (there is no CPU fallback)");
    ARG_CHECK(device >= 0 && device < rt::device_count(), "fhe_ctx_create: bad device ordinal");
    const uint32_t N = 1u << logN;
    for (uint32_t i = 0; i < nLimbs; ++i) {
        ARG_CHECK(q[i] > 2 && q[i] < ((uint64_t)1 << 60), "fhe_ctx_create: modulus must be < 2^60");
        ARG_CHECK((q[i] - 1) % (2ull * N) == 0, "fhe_ctx_create: modulus must be 1 mod 2N");
        ARG_CHECK(host::is_primitive_root_2n(psi[i], 2ull * N, q[i]),
                  "fhe_ctx_create: psi is not a primitive 2N-th root of unity");
    }
    RT_CHECK(rt::set_device(device));
    fhe_ctx* c = new fhe_ctx;
    c->logN    = logN;
    c->N       = N;
    c->L       = nLimbs;
    c->device  = device;
    c->cus     = rt::cu_count(device);
    c->q.assign(q, q + nLimbs);
    c->psi.assign(psi, psi + nLimbs);

    // twiddle tables: Table[bitrev(i)] = psi^i, TableI[bitrev(i)] = psi^-i  (transformnat-impl.h:725-737)
    std::vector<TwPair> tw((size_t)nLimbs * N), twInv((size_t)nLimbs * N), fin((size_t)nLimbs * 2);
    const bool rowTables   = logN >= (uint32_t)kTileLog;
    const size_t rowPerLimb = rowTables ? (size_t)(N >> kTileLog) * kRowTwSlots * kThreads : 0;
    std::vector<TwPair> twRow(rowPerLimb * nLimbs), twRowInv(rowPerLimb * nLimbs);
    std::vector<LimbConst> lc(nLimbs);
    std::vector<uint64_t> mu(2 * (size_t)nLimbs), red(nLimbs);
    // one thread per limb at most: a team of every host core would keep spinning after the region (libgomp's default
    // wait policy) and slow down the caller's kernel launches for a while
#pragma omp parallel for schedule(dynamic) num_threads(std::max(1, std::min<int>((int)nLimbs, 32)))
    for (uint32_t l = 0; l < nLimbs; ++l) {
        const uint64_t ql = q[l], ps = psi[l], psInv = host::invmod(ps, ql);
        uint64_t x = 1, xi = 1;
        TwPair* t  = tw.data() + (size_t)l * N;
        TwPair* ti = twInv.data() + (size_t)l * N;
        for (uint32_t i = 0; i < N; ++i) {
            const uint32_t r = host::bitrev(i, logN);
            t[r]             = TwPair{x, host::shoup(x, ql)};
            ti[r]            = TwPair{xi, host::shoup(xi, ql)};
            x                = host::mulmod(x, ps, ql);
            xi               = host::mulmod(xi, psInv, ql);
        }
        // lane-major copy for the step whose register field is tile bit 0 (ntt_static.h, TwSrc): lane t of tile tr holds
        // coefficients j0 = tr*4096 + 16t .. +15; stage b, twiddle g: index 2^(logN-1-b) + ((j0 >> 4) << (3-b)) + g
        for (size_t tr = 0; rowTables && tr < (N >> kTileLog); ++tr)
            for (uint32_t b = 0; b < 4; ++b)
                for (uint32_t g = 0; g < (8u >> b); ++g) {
                    const size_t slot = (size_t)(1u << (3 - b)) - 1 + g;
                    TwPair* d  = twRow.data() + (size_t)l * rowPerLimb + (tr * kRowTwSlots + slot) * kThreads;
                    TwPair* di = twRowInv.data() + (size_t)l * rowPerLimb + (tr * kRowTwSlots + slot) * kThreads;
                    for (uint32_t lane = 0; lane < (uint32_t)kThreads; ++lane) {
                        const size_t idx = ((size_t)1 << (logN - 1 - b)) + (((tr << 8) + lane) << (3 - b)) + g;
                        d[lane]  = t[idx];
                        di[lane] = ti[idx];
                    }
                }
        const uint64_t nInv = host::invmod((uint64_t)N % ql, ql);
        const uint64_t w1n  = host::mulmod(ti[1].w, nInv, ql);  // transformnat-impl.h:533-534
        fin[2 * l]          = TwPair{nInv, host::shoup(nInv, ql)};
        fin[2 * l + 1]      = TwPair{w1n, host::shoup(w1n, ql)};
        lc[l]               = LimbConst{ql, host::barrett_mu(ql), host::bitlen(ql), 0};
        host::mu128(ql, &mu[2 * l]);
        // k = (x.hi * redM) >> (32 + redR) is floor(x / q) or one less for every 64-bit x once bitlen(q) >= 36:
        // redR = bitlen(q) - 33, redM = floor(2^(64 + redR) / q) < 2^32; smaller moduli take the ladder (redR = 255)
        const uint32_t bl = host::bitlen(ql);
        if (bl >= 36) {
            const uint32_t rr = bl - 33;
            red[l] = (uint64_t)((((unsigned __int128)1) << (64 + rr)) / ql) | ((uint64_t)rr << 32);
        }
        else
            red[l] = (uint64_t)255 << 32;
    }
    c->h_fin = fin;
    fhe_status s;
    if ((s = upload(c, tw.data(), tw.size() * sizeof(TwPair), (void**)&c->d_tw)) ||
        (s = upload(c, twInv.data(), twInv.size() * sizeof(TwPair), (void**)&c->d_twInv)) ||
        (s = upload(c, fin.data(), fin.size() * sizeof(TwPair), (void**)&c->d_fin)) ||
        (rowTables && (s = upload(c, twRow.data(), twRow.size() * sizeof(TwPair), (void**)&c->d_twRow))) ||
        (rowTables && (s = upload(c, twRowInv.data(), twRowInv.size() * sizeof(TwPair), (void**)&c->d_twRowInv))) ||
        (s = upload(c, c->q.data(), nLimbs * sizeof(uint64_t), (void**)&c->d_q)) ||
        (s = upload(c, lc.data(), nLimbs * sizeof(LimbConst), (void**)&c->d_lc)) ||
        (s = upload(c, red.data(), nLimbs * sizeof(uint64_t), (void**)&c->d_red)) ||
        (s = upload(c, mu.data(), mu.size() * sizeof(uint64_t), (void**)&c->d_mu128))) {
        fhe_ctx_destroy(c);
        return s;
    }
    *out = c;
    return FHE_OK;
}

extern "C" void fhe_ctx_destroy(fhe_ctx* c) {
    if (!c)
        return;
    rt::set_device(c->device);
    for (void* p : c->owned)
        rt::dfree(p);
    for (auto& kv : c->constTabs)
        rt::dfree(kv.second);
    delete c;
}
extern "C" uint32_t fhe_ctx_logn(const fhe_ctx* c) { return c ? c->logN : 0; }
extern "C" uint32_t fhe_ctx_limbs(const fhe_ctx* c) { return c ? c->L : 0; }
extern "C" int fhe_ctx_device(const fhe_ctx* c) { return c ? c->device : -1; }

// ------------------------------------------------------------------------------------------------
// memory / streams
// ------------------------------------------------------------------------------------------------
extern "C" fhe_status fhe_malloc(fhe_ctx* c, size_t bytes, void** p) {
    ARG_CHECK(c && p, "fhe_malloc: null argument");
    RT_CHECK(rt::set_device(c->device));
    if (const char* e = rt::dmalloc(p, bytes))
        return fail(FHE_ERR_ALLOC, std::string("fhe_malloc: ") + e);
    return FHE_OK;
}
extern "C" fhe_status fhe_mem_info(fhe_ctx* c, size_t* freeBytes, size_t* totalBytes) {
    ARG_CHECK(c && freeBytes && totalBytes, "fhe_mem_info: null argument");
    RT_CHECK(rt::set_device(c->device));
    RT_CHECK(rt::mem_info(freeBytes, totalBytes));
    return FHE_OK;
}
extern "C" fhe_status fhe_free(fhe_ctx* c, void* p) {
    ARG_CHECK(c, "fhe_free: null context");
    RT_CHECK(rt::dfree(p));
    return FHE_OK;
}
extern "C" fhe_status fhe_memcpy_h2d(fhe_ctx* c, void* d, const void* s, size_t n, void* st) {
    ARG_CHECK(c && d && s, "fhe_memcpy_h2d: null argument");
    RT_CHECK(rt::h2d(d, s, n, (rt::stream_t)st));
    return FHE_OK;
}
extern "C" fhe_status fhe_memcpy_d2h(fhe_ctx* c, void* d, const void* s, size_t n, void* st) {
    ARG_CHECK(c && d && s, "fhe_memcpy_d2h: null argument");
    RT_CHECK(rt::d2h(d, s, n, (rt::stream_t)st));
    return FHE_OK;
}
extern "C" fhe_status fhe_memcpy_d2d(fhe_ctx* c, void* d, const void* s, size_t n, void* st) {
    ARG_CHECK(c && d && s, "fhe_memcpy_d2d: null argument");
    RT_CHECK(rt::d2d(d, s, n, (rt::stream_t)st));
    return FHE_OK;
}
extern "C" fhe_status fhe_stream_sync(fhe_ctx* c, void* st) {
    ARG_CHECK(c, "fhe_stream_sync: null context");
    RT_CHECK(rt::sync((rt::stream_t)st));
    return FHE_OK;
}

// Streams and graphs.  Composite calls (EvalMult, key switch, BFV EvalMult ...) are sequences of 20-60 kernel launches of
// 50-1000 us each; on a busy host the launch path, not the GPU, sets their pace.  Every entry point only enqueues work on
// the caller's stream (tables are built on first use, so run the sequence once before capturing), hence a caller can
// record a sequence into a HIP graph once and replay it with a single launch.
extern "C" fhe_status fhe_stream_create(fhe_ctx* c, void** stream) {
    ARG_CHECK(c && stream, "fhe_stream_create: null argument");
    RT_CHECK(rt::set_device(c->device));
    rt::stream_t s;
    RT_CHECK(rt::stream_create(&s));
    *stream = (void*)s;
    return FHE_OK;
}
extern "C" fhe_status fhe_stream_destroy(fhe_ctx* c, void* stream) {
    ARG_CHECK(c, "fhe_stream_destroy: null context");
    RT_CHECK(rt::stream_destroy((rt::stream_t)stream));
    return FHE_OK;
}
// `stream` waits on the device (the host is not blocked) for everything enqueued so far on `other`: the one cross-stream
// primitive a multi-threaded host needs (one stream per host thread, hand-over of a tower between threads)
extern "C" fhe_status fhe_stream_wait(fhe_ctx* c, void* stream, void* other) {
    ARG_CHECK(c, "fhe_stream_wait: null context");
    if (stream == other)
        return FHE_OK;
    RT_CHECK(rt::set_device(c->device));
    RT_CHECK(rt::stream_wait((rt::stream_t)stream, (rt::stream_t)other));
    return FHE_OK;
}
extern "C" fhe_status fhe_event_create(fhe_ctx* c, void** event) {
    ARG_CHECK(c && event, "fhe_event_create: null argument");
    RT_CHECK(rt::set_device(c->device));
    rt::event_t e;
    RT_CHECK(rt::event_create(&e));
    *event = (void*)e;
    return FHE_OK;
}
extern "C" fhe_status fhe_event_record(fhe_ctx* c, void* event, void* stream) {
    ARG_CHECK(c && event, "fhe_event_record: null argument");
    RT_CHECK(rt::set_device(c->device));
    RT_CHECK(rt::event_record((rt::event_t)event, (rt::stream_t)stream));
    return FHE_OK;
}
extern "C" fhe_status fhe_stream_wait_event(fhe_ctx* c, void* stream, void* event) {
    ARG_CHECK(c && event, "fhe_stream_wait_event: null argument");
    RT_CHECK(rt::set_device(c->device));
    RT_CHECK(rt::stream_wait_event((rt::stream_t)stream, (rt::event_t)event));
    return FHE_OK;
}
extern "C" fhe_status fhe_event_destroy(fhe_ctx* c, void* event) {
    ARG_CHECK(c, "fhe_event_destroy: null context");
    if (event)
        RT_CHECK(rt::event_destroy((rt::event_t)event));
    return FHE_OK;
}
extern "C" fhe_status fhe_memset_zero(fhe_ctx* c, void* dst, size_t bytes, void* stream) {
    ARG_CHECK(c && dst, "fhe_memset_zero: null argument");
    RT_CHECK(rt::set_device(c->device));
    RT_CHECK(rt::dzero(dst, bytes, (rt::stream_t)stream));
    return FHE_OK;
}
extern "C" fhe_status fhe_graph_begin(fhe_ctx* c, void* stream) {
    ARG_CHECK(c && stream, "fhe_graph_begin: capture needs a stream created with fhe_stream_create");
    RT_CHECK(rt::set_device(c->device));
    RT_CHECK(rt::capture_begin((rt::stream_t)stream));
    return FHE_OK;
}
extern "C" fhe_status fhe_graph_end(fhe_ctx* c, void* stream, void** graph) {
    ARG_CHECK(c && stream && graph, "fhe_graph_end: null argument");
    rt::graph_t g;
    RT_CHECK(rt::capture_end((rt::stream_t)stream, &g));
    *graph = (void*)g;
    return FHE_OK;
}
extern "C" fhe_status fhe_graph_launch(fhe_ctx* c, void* graph, void* stream) {
    ARG_CHECK(c && graph, "fhe_graph_launch: null argument");
    RT_CHECK(rt::graph_launch((rt::graph_t)graph, (rt::stream_t)stream));
    return FHE_OK;
}
extern "C" void fhe_graph_destroy(void* graph) {
    if (graph)
        rt::graph_destroy((rt::graph_t)graph);
}

// ------------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------------
static fhe_status make_sel(const fhe_ctx* c, const uint32_t* limbIdx, uint32_t nLimbs, LimbSel* sel, const char* who) {
    if (nLimbs < 1 || nLimbs > (uint32_t)kMaxLimbs)
        return fail(FHE_ERR_ARG, std::string(who) + ": nLimbs out of range");
    for (uint32_t i = 0; i < nLimbs; ++i) {
        uint32_t l = limbIdx ? limbIdx[i] : i;
        if (l >= c->L)
            return fail(FHE_ERR_ARG, std::string(who) + ": limb index exceeds context size");
        sel->idx[i] = (uint8_t)l;
    }
    for (uint32_t i = nLimbs; i < (uint32_t)kMaxLimbs; ++i)
        sel->idx[i] = 0;
    return FHE_OK;
}
static inline uint32_t tiles_for(const fhe_ctx* c, uint64_t rows) {
    const uint64_t words = rows << c->logN;
    return (uint32_t)((words + kTile - 1) >> kTileLog);
}
#define LAUNCH_CHECK() RT_CHECK(rt::last_launch_error())

// ------------------------------------------------------------------------------------------------
// NTT planning
// ------------------------------------------------------------------------------------------------
struct PassPlan {
    bool layoutA;
    uint32_t T;
    uint32_t inBound  = 1;   // forward: bound (units of q) of the pass input ...
    uint32_t outBound = 16;  // ... and of the pass output under the lazy schedule
    uint32_t nSteps;
    NttStep steps[6];
};

// stages of one pass act on the T bits of the in-tile point index p (bit T-1 first for the forward
// transform, bit 0 first for the inverse); they are grouped into register-resident steps of <= 4 stages.
static void plan_pass(bool inverse, bool layoutA, uint32_t logN, uint32_t T, PassPlan* pp) {
    pp->layoutA     = layoutA;
    pp->T           = T;
    const int logC  = kTileLog - (int)T;
    const int jsh   = layoutA ? (int)(logN - T) : 0;  // p bit b  <->  j bit b + jsh
    const int ish   = layoutA ? logC : 0;             // p bit b  <->  tile-index bit b + ish
    const int nst   = (int)(T + 3) / 4;
    int sizes[4];
    for (int i = 0; i < nst; ++i)
        sizes[i] = (int)T / nst + (i < (int)T % nst ? 1 : 0);
    NttStep real[4] = {};
    if (!inverse) {
        int top = (int)T - 1;
        for (int i = 0; i < nst; ++i) {
            const int r = sizes[i], fp = std::max(top - 3, 0);
            real[i] = NttStep{(int8_t)(fp + ish), (int8_t)(fp + jsh), (int8_t)(top - fp), (int8_t)(top - fp - r + 1), 0, 0, 0, 0};
            top -= r;
        }
    }
    else {
        int bot = 0;
        for (int i = 0; i < nst; ++i) {
            const int r = sizes[i], fp = std::min(bot, (int)T - 4);
            real[i] = NttStep{(int8_t)(fp + ish), (int8_t)(fp + jsh), (int8_t)(bot - fp + r - 1), (int8_t)(bot - fp), 0, 0, 0, 0};
            bot += r;
        }
    }
    // staging steps: a first/last step whose register field sits below tile-index bit 4 would touch HBM in
    // < 128-byte pieces; route it through LDS with the fully coalesced mapping (field at bits 8..11) instead
    uint32_t n = 0;
    const NttStep stage{8, 0, -1, 0, 0, 0, 0, 0};
    if (!layoutA && real[0].fI < 4)
        pp->steps[n++] = stage;
    for (int i = 0; i < nst; ++i)
        pp->steps[n++] = real[i];
    if (!layoutA && real[nst - 1].fI < 4)
        pp->steps[n++] = stage;
    pp->nSteps = n;
}

// lazy-reduction schedule of a forward pass (the static kernels derive the same at compile time, ntt_static.h SPlan): a
// butterfly with the truncated quotient adds at most 3q to the bound of its operands and 16q < 2^64, so a step of r
// stages needs a correction (the `a` inputs of its first stage below 2q) only when bound + 3r would exceed 16.  A column
// pass of a two-pass ring ends by bringing its residues below 2q (it is HBM-bound: the reduction is free there).
// `bound` (in units of q) is the bound of the pass input on entry and of its output on return.
static void schedule_fwd(PassPlan& pp, uint32_t logN, uint32_t* bound) {
    pp.inBound = *bound;
    for (uint32_t i = 0; i < pp.nSteps; ++i) {
        NttStep& st = pp.steps[i];
        if (st.bHi < st.bLo)
            continue;
        const uint32_t r = (uint32_t)(st.bHi - st.bLo + 1);
        if (*bound + 3 * r <= 16) {
            st.mode = 0;
            *bound += 3 * r;
        }
        else {
            st.mode = 1;
            *bound  = 2 + 3 * r;
        }
    }
    if (pp.layoutA && logN >= (uint32_t)kTileLog)
        *bound = 2;
    (void)logN;
}
// twiddle index = 2^s + (j >> (Fj+4))...: lane-independent iff no lane-dependent bit of j lies above the field
static void mark_uniform(PassPlan& pp, uint32_t logN) {
    for (uint32_t i = 0; i < pp.nSteps; ++i) {
        NttStep& st = pp.steps[i];
        if (st.bHi < st.bLo)
            continue;
        st.uniformTw = pp.layoutA ? ((uint32_t)st.Fj + 4 == logN) : ((uint32_t)st.Fj + 4 >= (uint32_t)kTileLog);
    }
}

static uint32_t ntt_t1(uint32_t logN);
static int static_mode(const fhe_ctx* c, const struct PassPlan& pp, bool inverse);
// fused epilogue of a forward transform's last pass (NttPassArgs::epi*)
struct NttEpilogue {
    uint32_t mode = 0, split = 0, aStride = 0, aFirst = 0;
    int64_t aDelta = 0;  // != 0: towers of A are this many words apart (NttPassArgs::epiADelta)
    const uint64_t* A = nullptr;
    const TwPair* C   = nullptr;
    uint64_t *out0 = nullptr, *out1 = nullptr;
};
static uint32_t fill_pass_args(const fhe_ctx* c, const PassPlan& pp, bool inverse, const uint64_t* xin, uint64_t* xout,
                               const LimbSel& sel, uint32_t nLimbs, uint32_t batch, bool canonOut, uint32_t inStride,
                               uint32_t inFirst, uint32_t outStride, uint32_t outFirst, NttPassArgs& a) {
    a.outStride = outStride;
    a.outFirst  = outFirst;
    a.inStride = inStride;
    a.inFirst  = inFirst;
    a.xin      = xin;
    a.x        = xout;
    a.tw       = inverse ? c->d_twInv : c->d_tw;
    a.twRow    = inverse ? c->d_twRowInv : c->d_twRow;
    a.q        = c->d_q;
    a.red      = c->d_red;
    a.fin      = c->d_fin;
    a.logN     = c->logN;
    a.T        = pp.T;
    a.nLimbs   = nLimbs;
    a.rows     = batch * nLimbs;
    a.batch    = batch;
    a.nSteps   = pp.nSteps;
    // canonicalise right after the last step that has butterfly stages (a trailing staging step only moves data)
    a.canonStep = 0xffffffffu;
    a.canonLevels = 1;
    while ((1u << a.canonLevels) < pp.outBound)
        ++a.canonLevels;
    if (canonOut)
        for (uint32_t i = 0; i < pp.nSteps; ++i)
            if (pp.steps[i].bHi >= pp.steps[i].bLo)
                a.canonStep = i;
    for (uint32_t i = 0; i < 6; ++i)
        a.steps[i] = i < pp.nSteps ? pp.steps[i] : NttStep{0, 0, -1, 0, 0, 0, 0, 0};
    a.sel = sel;
    const uint32_t grid        = tiles_for(c, a.rows);
    const uint32_t tilesPerRow = c->N >= (uint32_t)kTile ? (c->N >> kTileLog) : 1u;
    a.xcdSwizzle = (c->N >= (uint32_t)kTile && ((nLimbs * tilesPerRow) % 8u == 0)) ? 1u : 0u;
    a.epiMode = 0, a.epiSplit = 0, a.epiAStride = 0, a.epiAFirst = 0;
    a.epiA = nullptr, a.epiC = nullptr, a.epiOut0 = a.epiOut1 = nullptr;
    a.proMode = 0, a.proSrcLimb = 0;
    a.inDelta = 0, a.epiADelta = 0;
    return grid;
}
// forward: bound class of a static pass's input; inverse: does the pass end the transform (ntt_static.h MODE)
static int static_mode(const fhe_ctx* c, const PassPlan& pp, bool inverse) {
    const bool twoPass = c->logN > (uint32_t)kTileLog;
    // (forward classes: 1 = canonical input, 9 = at most 9q, 16 = anything below 16q; a 4-stage first step sweeps
    // for every class above 1, so T = 12 has one instance for both)
    const int fclass = pp.inBound <= 1 ? 1 : 9;  // (9 = what a column pass leaves: below 2q since round 4)
    return inverse ? ((pp.layoutA || !twoPass) ? 1 : 0) : fclass;
}
// Which row pass runs (both are bit-exact; profiles/r06_sweeps.md section 4): ntt_row8.h for 9..11 stages (rings 2^13..2^15: 2-10 % faster),
// ntt_static.h's 16-residues-per-lane kernel for 12 stages (1-3 % faster there).  FHE_NTT_ROW8 = 0 / 1 forces one of them (measurements).
static bool row8_for(uint32_t T) {
    static const int forced = [] {
        const char* v = std::getenv("FHE_NTT_ROW8");
        return v && (v[0] == '0' || v[0] == '1') ? v[0] - '0' : -1;
    }();
    return forced >= 0 ? forced == 1 : T <= 11u;
}
static fhe_status launch_pass(const fhe_ctx* c, const PassPlan& pp, bool inverse, const uint64_t* xin, uint64_t* xout,
                              const LimbSel& sel, uint32_t nLimbs, uint32_t batch, bool canonOut, void* stream,
                              uint32_t inStride = 0, uint32_t inFirst = 0, uint32_t outStride = 0, uint32_t outFirst = 0,
                              const NttEpilogue* epi = nullptr, const uint32_t* proSrcLimb = nullptr, int64_t inDelta = 0) {
    NttPassArgs a;
    const uint32_t grid = fill_pass_args(c, pp, inverse, xin, xout, sel, nLimbs, batch, canonOut, inStride, inFirst, outStride,
                                         outFirst, a);
    if (inDelta) {  // (the static kernels only)
        if (c->logN < (uint32_t)kTileLog)
            return fail(FHE_ERR_UNSUPPORTED, "ntt: separately allocated towers need a ring of at least 4096");
        a.inDelta = inDelta;
    }
    if (proSrcLimb) {
        // the forward column pass that loads every limb from one row modulo q[*proSrcLimb] (ntt_prologue_supported)
        a.proMode = 1, a.proSrcLimb = *proSrcLimb;
        const int mode = static_mode(c, pp, inverse);
        if (pp.layoutA && !inverse && mode == 1 && pp.T == 4)
            FHE_LAUNCH_BARRIER((ntt_static_kernel<true, false, 4, 1, false, true>), grid, stream, a);
        else if (pp.layoutA && !inverse && mode == 1 && pp.T == 5)
            FHE_LAUNCH_BARRIER((ntt_static_kernel<true, false, 5, 1, false, true>), grid, stream, a);
        else
            return fail(FHE_ERR_UNSUPPORTED, "ntt: no prologue kernel for this pass shape");
        LAUNCH_CHECK();
        return FHE_OK;
    }
    if (epi && epi->mode) {
        // only the static forward row / single pass kernels carry the epilogue (ntt_epilogue_supported)
        a.epiMode = epi->mode, a.epiSplit = epi->split, a.epiAStride = epi->aStride, a.epiAFirst = epi->aFirst;
        a.epiA = epi->A, a.epiC = epi->C, a.epiOut0 = epi->out0, a.epiOut1 = epi->out1;
        a.epiADelta = epi->aDelta;
        const int mode = static_mode(c, pp, inverse);
        bool launched  = false;
#define FHE_EPI_CASE(TT, MODE) \
    if (!launched && !pp.layoutA && !inverse && pp.T == TT && mode == MODE) { \
        FHE_LAUNCH_BARRIER((ntt_static_kernel<false, false, TT, MODE, true>), grid, stream, a); \
        launched = true; \
    }
        FHE_EPI_CASE(12, 9) FHE_EPI_CASE(11, 9) FHE_EPI_CASE(10, 9) FHE_EPI_CASE(9, 9) FHE_EPI_CASE(12, 1)
#undef FHE_EPI_CASE
        if (!launched)
            return fail(FHE_ERR_UNSUPPORTED, "ntt: no epilogue kernel for this pass shape");
        LAUNCH_CHECK();
        return FHE_OK;
    }
    if (c->logN >= (uint32_t)kTileLog) {
        // compile-time pass plans (ntt_static.h): in-place pinned-register butterflies, immediate-offset LDS exchange
        const int mode = static_mode(c, pp, inverse);
        bool launched  = false;
#define FHE_STATIC_CASE(LA, INV, TT, MODE) \
    if (!launched && pp.layoutA == LA && inverse == INV && pp.T == TT && mode == MODE) { \
        FHE_LAUNCH_BARRIER((ntt_static_kernel<LA, INV, TT, MODE>), grid, stream, a); \
        launched = true; \
    }
        // row passes of two-pass rings at 8 residues per lane (ntt_row8.h: 8 waves per SIMD, one barrier per tile)
        if (!pp.layoutA && c->logN > (uint32_t)kTileLog && row8_for(pp.T)) {
            // tiles of 2^T words on 2^(T-9) waves: grid and XCD order for that tile size
            const uint32_t rgrid = a.rows << (c->logN - pp.T);
            a.xcdSwizzle         = ((nLimbs << (c->logN - pp.T)) % 8u == 0) ? 1u : 0u;
#define FHE_ROW8_CASE(INV, TT, MODE) \
    if (!launched && inverse == INV && pp.T == TT && mode == MODE) { \
        FHE_LAUNCH_BARRIER_N((r8::ntt_row8_kernel<INV, TT - 9, MODE>), rgrid, 64u << (TT - 9), stream, a); \
        launched = true; \
    }
            FHE_ROW8_CASE(false, 12, 9) FHE_ROW8_CASE(true, 12, 0) FHE_ROW8_CASE(false, 11, 9) FHE_ROW8_CASE(true, 11, 0)
            FHE_ROW8_CASE(false, 10, 9) FHE_ROW8_CASE(true, 10, 0) FHE_ROW8_CASE(false, 9, 9) FHE_ROW8_CASE(true, 9, 0)
#undef FHE_ROW8_CASE
        }
        // column passes of logN = 13..16 (T1 = 4) and 17 (T1 = 5); row passes T2 = logN - T1; the single pass of logN = 12
        FHE_STATIC_CASE(true, false, 4, 1) FHE_STATIC_CASE(true, true, 4, 1)
        FHE_STATIC_CASE(true, false, 5, 1) FHE_STATIC_CASE(true, true, 5, 1)
        FHE_STATIC_CASE(false, false, 12, 9) FHE_STATIC_CASE(false, true, 12, 0)
        FHE_STATIC_CASE(false, false, 11, 9) FHE_STATIC_CASE(false, true, 11, 0)
        FHE_STATIC_CASE(false, false, 10, 9) FHE_STATIC_CASE(false, true, 10, 0)
        FHE_STATIC_CASE(false, false, 9, 9) FHE_STATIC_CASE(false, true, 9, 0)
        FHE_STATIC_CASE(false, false, 12, 1) FHE_STATIC_CASE(false, true, 12, 1)
#undef FHE_STATIC_CASE
        if (!launched)
            return fail(FHE_ERR_UNSUPPORTED, "ntt: no kernel instance for this pass shape");
        LAUNCH_CHECK();
        return FHE_OK;
    }
    // rings below one tile (N < 4096): the generic small-ring kernel, several limbs per workgroup
    if (pp.layoutA) {
        if (inverse)
            FHE_LAUNCH_BARRIER((ntt_pass_kernel<true, true>), grid, stream, a);
        else
            FHE_LAUNCH_BARRIER((ntt_pass_kernel<true, false>), grid, stream, a);
    }
    else {
        if (inverse)
            FHE_LAUNCH_BARRIER((ntt_pass_kernel<false, true>), grid, stream, a);
        else
            FHE_LAUNCH_BARRIER((ntt_pass_kernel<false, false>), grid, stream, a);
    }
    LAUNCH_CHECK();
    return FHE_OK;
}

static uint32_t env_u32(const char* name, uint32_t dflt) {
    const char* v = std::getenv(name);
    return v ? (uint32_t)std::strtoul(v, nullptr, 10) : dflt;
}
// stages of the strided column pass of a two-pass ring: as few as possible (>= 4), so that the column pass reads rows of
// 2^(12-T1) consecutive words and, at T1 = 4, is a pure register radix-16 step; the other splits were measured slower
// (profiles/r01_sweeps.md) and their kernel instances are gone
static uint32_t ntt_t1(uint32_t logN) {
    const uint32_t dflt = std::max(4u, logN - (uint32_t)kTileLog);
    // FHE_NTT_T1 (measurements): another split of a two-pass ring, where both passes have kernel instances (column 4..5, row 9..12)
    static const uint32_t forced = env_u32("FHE_NTT_T1", 0);
    if (forced >= 4 && forced <= 5 && logN > (uint32_t)kTileLog && logN - forced >= 9 && logN - forced <= 12)
        return forced;
    return dflt;
}

// inStride != 0: xin is a [batch][inStride][N] view whose rows inFirst.. are transformed into the dense xout
// outStride != 0: xout is a [batch][outStride][N] view as well (rows outFirst..)
// does a forward transform of this ring end in a kernel that can carry the fused epilogue?
static bool ntt_epilogue_supported(const fhe_ctx* c) {
    if (c->logN < (uint32_t)kTileLog)
        return false;
    if (c->logN == (uint32_t)kTileLog)
        return true;  // the single pass
    // the epilogue instances of launch_pass: row passes of 9..12 stages whose input class is 9 (static_mode): any 12-stage
    // row pass (logN = 16 after 4 column stages, 17 after 5, ...), shorter ones only after a 4-stage column pass
    const uint32_t t1 = ntt_t1(c->logN), t2 = c->logN - t1;
    return t2 == 12u || (t1 == 4u && t2 >= 9u && t2 <= 12u);
}
// a two-pass forward transform whose column pass can carry the load prologue (NttPassArgs::proMode)
static bool ntt_prologue_supported(const fhe_ctx* c) {
    return c->logN > (uint32_t)kTileLog && ntt_t1(c->logN) <= 5u;
}
static fhe_status ntt_run(fhe_ctx* c, bool inverse, const uint64_t* xin, uint64_t* xout, const uint32_t* limbIdx,
                          uint32_t nLimbs, uint32_t batch, void* stream, uint32_t inStride = 0, uint32_t inFirst = 0,
                          uint32_t outStride = 0, uint32_t outFirst = 0, const NttEpilogue* epi = nullptr,
                          bool canonOut = true, const uint32_t* proSrcLimb = nullptr, int64_t inDelta = 0) {
    ARG_CHECK(c && xin && xout, "fhe_ntt: null argument");
    ARG_CHECK(batch >= 1, "fhe_ntt: batch must be >= 1");
    LimbSel sel;
    if (fhe_status s = make_sel(c, limbIdx, nLimbs, &sel, "fhe_ntt"))
        return s;
    RT_CHECK(rt::set_device(c->device));
    const uint32_t logN = c->logN;
    if (logN <= (uint32_t)kTileLog) {
        PassPlan p;
        plan_pass(inverse, false, logN, logN, &p);
        uint32_t bound = 1;
        mark_uniform(p, logN);
        if (!inverse)
            schedule_fwd(p, logN, &bound);
        p.outBound = bound;
        return launch_pass(c, p, inverse, xin, xout, sel, nLimbs, batch, canonOut, stream, inStride, inFirst, outStride, outFirst, epi,
                           nullptr, inDelta);
    }
    // two passes over HBM: a strided column pass of T1 stages (the coefficient index's top bits) and a
    // contiguous row pass of T2 = logN - T1 stages.  T1 is kept minimal (>= 4) so that the column pass reads
    // rows of 2^(12-T1) consecutive words (2 KiB at T1 = 4) and, at T1 = 4, needs no LDS at all.
    const uint32_t T1 = ntt_t1(logN), T2 = logN - T1;
    PassPlan pa, pb;
    plan_pass(inverse, true, logN, T1, &pa);
    plan_pass(inverse, false, logN, T2, &pb);
    mark_uniform(pa, logN);
    mark_uniform(pb, logN);
    if (!inverse) {
        uint32_t bound = 1;
        schedule_fwd(pa, logN, &bound);
        schedule_fwd(pb, logN, &bound);
        pb.outBound = bound;
    }
    const PassPlan& p1 = inverse ? pb : pa;
    const PassPlan& p2 = inverse ? pa : pb;
    // Both passes of the whole batch back to back on the caller's stream.  What else was built and measured, bit-exact and slower or
    // equal, and removed again (profiles/r02_sweeps.md session 3, profiles/r05_sweeps.md): both passes in one launch with the tower
    // kept in an XCD's L2 (no retention); launches per chunk sized to the Infinity Cache (its bandwidth is within 15 % of HBM's);
    // the column pass of one chunk and the row pass of another on two streams (the queues split a CU's four workgroup slots: the
    // pair takes the sum of its parts) or as one grid of alternating roles (6 % slower); persistent pass kernels (11 % slower: a
    // fresh workgroup's loads overlap the drain of its predecessor's stores); 5 + 11 stages instead of 4 + 12 (1 % slower).
    if (fhe_status s = launch_pass(c, p1, inverse, xin, xout, sel, nLimbs, batch, false, stream, inStride, inFirst, outStride, outFirst,
                                   nullptr, proSrcLimb, inDelta))
        return s;
    return launch_pass(c, p2, inverse, xout, xout, sel, nLimbs, batch, canonOut, stream, outStride, outFirst, outStride, outFirst, epi);
}

extern "C" fhe_status fhe_ntt_fwd(fhe_ctx* c, uint64_t* x, const uint32_t* li, uint32_t nl, uint32_t b, void* st) {
    return ntt_run(c, false, x, x, li, nl, b, st);
}
extern "C" fhe_status fhe_ntt_inv(fhe_ctx* c, uint64_t* x, const uint32_t* li, uint32_t nl, uint32_t b, void* st) {
    return ntt_run(c, true, x, x, li, nl, b, st);
}
extern "C" fhe_status fhe_ntt_fwd_oop(fhe_ctx* c, const uint64_t* xi, uint64_t* xo, const uint32_t* li, uint32_t nl,
                                      uint32_t b, void* st) {
    return ntt_run(c, false, xi, xo, li, nl, b, st);
}
extern "C" fhe_status fhe_ntt_inv_oop(fhe_ctx* c, const uint64_t* xi, uint64_t* xo, const uint32_t* li, uint32_t nl,
                                      uint32_t b, void* st) {
    return ntt_run(c, true, xi, xo, li, nl, b, st);
}

// ---- negacyclic polynomial product of COEFFICIENT-format towers: c = INTT(NTT(a) o NTT(b)) per limb ----
// What a caller of the reference writes as a.SetFormat(EVALUATION); b.SetFormat(EVALUATION); c = a * b; c.SetFormat(COEFFICIENT)
// (dcrtpoly-impl.h:1932-1940, dcrtpoly.h:174-189): three transforms and a Hadamard product.  Two-pass rings whose row pass has a
// static 9..12-stage instance run the fused kernels of ntt_static.h (poly_mul_row_*); every other ring runs the plain sequence.
extern "C" size_t fhe_poly_mul_workspace_bytes(const fhe_ctx* c, uint32_t nLimbs, uint32_t batch) {
    return c ? (((size_t)2 * batch * nLimbs) << c->logN) * 8 : 0;
}
static bool poly_mul_fused_supported(const fhe_ctx* c) {
    if (c->logN <= (uint32_t)kTileLog)
        return false;
    const uint32_t t1 = ntt_t1(c->logN), t2 = c->logN - t1;
    return (t1 == 4u && t2 >= 9u && t2 <= 12u) || (t1 == 5u && t2 == 12u);
}
extern "C" fhe_status fhe_poly_mul(fhe_ctx* c, const uint64_t* a, const uint64_t* b, uint64_t* out, const uint32_t* limbIdx,
                                   uint32_t nLimbs, uint32_t batch, void* wsv, size_t wsBytes, void* stream) {
    ARG_CHECK(c && a && b && out && wsv, "fhe_poly_mul: null argument");
    ARG_CHECK(batch >= 1, "fhe_poly_mul: batch must be >= 1");
    ARG_CHECK(wsBytes >= fhe_poly_mul_workspace_bytes(c, nLimbs, batch), "fhe_poly_mul: workspace too small");
    LimbSel sel;
    if (fhe_status s = make_sel(c, limbIdx, nLimbs, &sel, "fhe_poly_mul"))
        return s;
    RT_CHECK(rt::set_device(c->device));
    uint64_t* wa = (uint64_t*)wsv;
    uint64_t* wb = wa + (((size_t)batch * nLimbs) << c->logN);
    if (!poly_mul_fused_supported(c)) {
        if (fhe_status s = ntt_run(c, false, a, wa, limbIdx, nLimbs, batch, stream))
            return s;
        if (fhe_status s = ntt_run(c, false, b, wb, limbIdx, nLimbs, batch, stream))
            return s;
        if (fhe_status s = fhe_mul(c, out, wa, wb, limbIdx, nLimbs, batch, stream))
            return s;
        return ntt_run(c, true, out, out, limbIdx, nLimbs, batch, stream);
    }
    const uint32_t logN = c->logN, T1 = ntt_t1(logN), T2 = logN - T1;
    PassPlan fa, fb, ib, ia;  // forward column / row, inverse row / column
    plan_pass(false, true, logN, T1, &fa);
    plan_pass(false, false, logN, T2, &fb);
    plan_pass(true, false, logN, T2, &ib);
    plan_pass(true, true, logN, T1, &ia);
    for (PassPlan* p : {&fa, &fb, &ib, &ia})
        mark_uniform(*p, logN);
    uint32_t bound = 1;
    schedule_fwd(fa, logN, &bound);
    schedule_fwd(fb, logN, &bound);
    fb.outBound = bound;
    // column passes of a and b, out of place into the workspace
    if (fhe_status s = launch_pass(c, fa, false, a, wa, sel, nLimbs, batch, false, stream))
        return s;
    if (fhe_status s = launch_pass(c, fa, false, b, wb, sel, nLimbs, batch, false, stream))
        return s;
    PolyMulArgs g;
    const uint32_t grid = fill_pass_args(c, fb, false, wa, wa, sel, nLimbs, batch, true, 0, 0, 0, 0, g.fwd);
    g.fwd.xcdSwizzle = 0;  // block id = tile: the workspace tile of a block is addressed by its id
    g.inv = g.fwd;
    g.aEval = wa;
    g.lc = c->d_lc;
#define FHE_PM_A(TT) \
    if (T2 == TT)    \
        FHE_LAUNCH_BARRIER((poly_mul_row_a_kernel<TT>), grid, stream, g);
    FHE_PM_A(12) FHE_PM_A(11) FHE_PM_A(10) FHE_PM_A(9)
#undef FHE_PM_A
    LAUNCH_CHECK();
    fill_pass_args(c, fb, false, wb, wb, sel, nLimbs, batch, true, 0, 0, 0, 0, g.fwd);
    fill_pass_args(c, ib, true, out, out, sel, nLimbs, batch, false, 0, 0, 0, 0, g.inv);
    g.fwd.xcdSwizzle = g.inv.xcdSwizzle = 0;
#define FHE_PM_B(TT) \
    if (T2 == TT)    \
        FHE_LAUNCH_BARRIER((poly_mul_row_b_kernel<TT>), grid, stream, g);
    FHE_PM_B(12) FHE_PM_B(11) FHE_PM_B(10) FHE_PM_B(9)
#undef FHE_PM_B
    LAUNCH_CHECK();
    // inverse column pass: ends the transform (N^-1 folded in, canonical residues)
    return launch_pass(c, ia, true, out, out, sel, nLimbs, batch, true, stream);
}

// dir: 0 fwd, 1 inv, 2 fwd then inv; 10/11 = only the column / row pass of the forward transform,
// 12/13 = only the row / column pass of the inverse (two-pass rings; timing only, data is not meaningful)
extern "C" fhe_status fhe_time_ntt(fhe_ctx* c, uint64_t* x, const uint32_t* li, uint32_t nl, uint32_t b, int dir,
                                   int iters, void* st, float* ms) {
    ARG_CHECK(c && x && ms && iters > 0, "fhe_time_ntt: bad argument");
    LimbSel sel;
    if (fhe_status s = make_sel(c, li, nl, &sel, "fhe_time_ntt"))
        return s;
    RT_CHECK(rt::set_device(c->device));
    PassPlan pp;
    if (dir >= 10) {
        ARG_CHECK(c->logN > (uint32_t)kTileLog && dir <= 13, "fhe_time_ntt: single-pass timing needs a two-pass ring");
        const uint32_t T1 = ntt_t1(c->logN), T2 = c->logN - T1;
        const bool inv = dir >= 12, colPass = (dir == 10 || dir == 13);
        plan_pass(inv, colPass, c->logN, colPass ? T1 : T2, &pp);
        mark_uniform(pp, c->logN);
        if (!inv) {
            uint32_t bound = colPass ? 1u : 2u;  // the row pass sees what the column pass leaves
            schedule_fwd(pp, c->logN, &bound);
            pp.outBound = bound;
        }
    }
    rt::Timer tm;
    RT_CHECK(tm.start((rt::stream_t)st));
    for (int i = 0; i < iters; ++i) {
        if (dir >= 10) {
            if (fhe_status s = launch_pass(c, pp, dir >= 12, x, x, sel, nl, b, false, st))
                return s;
            continue;
        }
        if (dir == 0 || dir == 2)
            if (fhe_status s = fhe_ntt_fwd(c, x, li, nl, b, st))
                return s;
        if (dir == 1 || dir == 2)
            if (fhe_status s = fhe_ntt_inv(c, x, li, nl, b, st))
                return s;
    }
    float total = 0;
    RT_CHECK(tm.stop((rt::stream_t)st, &total));
    *ms = total / iters;
    return FHE_OK;
}

// ------------------------------------------------------------------------------------------------
// element-wise
// ------------------------------------------------------------------------------------------------
template <int OP>
static fhe_status elem_run(fhe_ctx* c, uint64_t* out, const uint64_t* a, const uint64_t* b, const TwPair* d_consts,
                           const uint32_t* limbIdx, uint32_t nLimbs, uint32_t batch, void* stream, const char* who,
                           uint32_t aStride = 0, uint32_t aFirst = 0, uint32_t bStride = 0, uint32_t bFirst = 0,
                           uint32_t oStride = 0, uint32_t oFirst = 0, const int64_t* deltas = nullptr) {
    ARG_CHECK(c && out && a, std::string(who) + ": null argument");
    ARG_CHECK(batch >= 1, std::string(who) + ": batch must be >= 1");
    ElemArgs g;
    if (fhe_status s = make_sel(c, limbIdx, nLimbs, &g.sel, who))
        return s;
    RT_CHECK(rt::set_device(c->device));
    g.out    = out;
    g.a      = a;
    g.b      = b;
    g.lc     = c->d_lc;
    g.consts = d_consts;
    g.logN   = c->logN;
    g.nLimbs = nLimbs;
    g.rows   = batch * nLimbs;
    g.aStride = aStride, g.aFirst = aFirst, g.bStride = bStride, g.bFirst = bFirst;
    g.oStride = oStride, g.oFirst = oFirst;
    if (deltas)  // towers allocated on their own: words between tower 0 and tower 1 of out / a / b
        g.oDelta = deltas[0], g.aDelta = deltas[1], g.bDelta = deltas[2];
    FHE_LAUNCH((elemwise_kernel<OP>), tiles_for(c, g.rows), stream, g);
    LAUNCH_CHECK();
    return FHE_OK;
}
extern "C" fhe_status fhe_add(fhe_ctx* c, uint64_t* o, const uint64_t* a, const uint64_t* b, const uint32_t* li,
                              uint32_t nl, uint32_t bt, void* st) {
    ARG_CHECK(b, "fhe_add: null argument");
    return elem_run<OP_ADD>(c, o, a, b, nullptr, li, nl, bt, st, "fhe_add");
}
extern "C" fhe_status fhe_sub(fhe_ctx* c, uint64_t* o, const uint64_t* a, const uint64_t* b, const uint32_t* li,
                              uint32_t nl, uint32_t bt, void* st) {
    ARG_CHECK(b, "fhe_sub: null argument");
    return elem_run<OP_SUB>(c, o, a, b, nullptr, li, nl, bt, st, "fhe_sub");
}
extern "C" fhe_status fhe_mul(fhe_ctx* c, uint64_t* o, const uint64_t* a, const uint64_t* b, const uint32_t* li,
                              uint32_t nl, uint32_t bt, void* st) {
    ARG_CHECK(b, "fhe_mul: null argument");
    return elem_run<OP_MUL>(c, o, a, b, nullptr, li, nl, bt, st, "fhe_mul");
}
// acc += a * b per limb (exact): the accumulation of KeySwitchHYBRID::EvalFastKeySwitchCoreExt, keyswitch-hybrid.cpp:419-430
// (`elements[k].SetElementAtIndex(i, elements[k].GetElementAtIndex(i) + cji * bji)`), on whole towers
extern "C" fhe_status fhe_mul_add(fhe_ctx* c, uint64_t* acc, const uint64_t* a, const uint64_t* b, const uint32_t* li,
                                  uint32_t nl, uint32_t bt, void* st) {
    ARG_CHECK(b, "fhe_mul_add: null argument");
    return elem_run<OP_MULT_ACC>(c, acc, a, b, nullptr, li, nl, bt, st, "fhe_mul_add");
}
extern "C" fhe_status fhe_neg(fhe_ctx* c, uint64_t* o, const uint64_t* a, const uint32_t* li, uint32_t nl, uint32_t bt,
                              void* st) {
    return elem_run<OP_NEG>(c, o, a, nullptr, nullptr, li, nl, bt, st, "fhe_neg");
}

// per-call constant vectors (Times(vector<NativeInteger>), MultAccEqNoCheck ...): the host constants become Shoup pairs
// and travel BY VALUE in the kernel arguments — no staging buffer, no synchronisation, capturable into a HIP graph
// (the whole vector on the host; a launch carries a window of kConstVecLimbs rows of it: launch_cv)
struct HostConstVec {
    TwPair c[kMaxLimbs];
};
template <int OP>
static void launch_cv(fhe_ctx* c, ElemArgs g, const HostConstVec& cv, void* stream) {
    for (uint32_t first = 0; first < g.nLimbs; first += (uint32_t)kConstVecLimbs) {
        ConstVec w;
        for (uint32_t i = 0; i < (uint32_t)kConstVecLimbs; ++i)
            w.c[i] = first + i < (uint32_t)kMaxLimbs ? cv.c[first + i] : TwPair{0, 0};
        g.cvFirst = first;
        FHE_LAUNCH((elemwise_cv_kernel<OP>), tiles_for(c, g.rows), stream, g, w);
    }
}
static fhe_status make_const_vec(const fhe_ctx* c, const uint64_t* consts, const uint32_t* limbIdx, uint32_t nLimbs,
                                 HostConstVec* cv, const char* who) {
    ARG_CHECK(c && consts, std::string(who) + ": null argument");
    ARG_CHECK(nLimbs >= 1 && nLimbs <= (uint32_t)kMaxLimbs, std::string(who) + ": nLimbs out of range");
    for (uint32_t i = 0; i < (uint32_t)kMaxLimbs; ++i)
        cv->c[i] = TwPair{0, 0};
    for (uint32_t i = 0; i < nLimbs; ++i) {
        const uint32_t l = limbIdx ? limbIdx[i] : i;
        ARG_CHECK(l < c->L, std::string(who) + ": limb index exceeds context size");
        const uint64_t ql = c->q[l], v = consts[i] % ql;
        cv->c[i]          = TwPair{v, host::shoup(v, ql)};
    }
    return FHE_OK;
}
template <int OP>
static fhe_status elem_cv_run(fhe_ctx* c, uint64_t* out, const uint64_t* a, const uint64_t* b, const HostConstVec& cv,
                              const uint32_t* limbIdx, uint32_t nLimbs, uint32_t batch, void* stream, const char* who,
                              uint32_t oStride = 0, uint32_t oFirst = 0, const int64_t* deltas = nullptr) {
    ARG_CHECK(c && out && a, std::string(who) + ": null argument");
    ARG_CHECK(batch >= 1, std::string(who) + ": batch must be >= 1");
    ElemArgs g;
    if (fhe_status s = make_sel(c, limbIdx, nLimbs, &g.sel, who))
        return s;
    RT_CHECK(rt::set_device(c->device));
    g.out = out, g.a = a, g.b = b, g.lc = c->d_lc, g.consts = nullptr;
    g.logN = c->logN, g.nLimbs = nLimbs, g.rows = batch * nLimbs;
    g.aStride = g.aFirst = g.bStride = g.bFirst = 0;
    g.oStride = oStride, g.oFirst = oFirst;
    if (deltas)
        g.oDelta = deltas[0], g.aDelta = deltas[1], g.bDelta = deltas[2];
    launch_cv<OP>(c, g, cv, stream);
    LAUNCH_CHECK();
    return FHE_OK;
}
extern "C" fhe_status fhe_mul_const(fhe_ctx* c, uint64_t* o, const uint64_t* a, const uint64_t* consts,
                                    const uint32_t* li, uint32_t nl, uint32_t bt, void* st) {
    HostConstVec cv;
    if (fhe_status s = make_const_vec(c, consts, li, nl, &cv, "fhe_mul_const"))
        return s;
    return elem_cv_run<OP_MUL_CONST>(c, o, a, nullptr, cv, li, nl, bt, st, "fhe_mul_const");
}
static fhe_status const_table(fhe_ctx* c, const uint32_t* limbIdx, const uint64_t* v, uint32_t n, const TwPair** out);
static fhe_status const_tables_trim(fhe_ctx* c);
// out = sum_i consts[i][.] (.) x[i]  (+ out when accumulate): the weighted sums of pke (ckksrns-advancedshe.cpp:97-136) as ONE launch per
// 16 terms.  consts: HOST array [nTerms][nLimbs], reduced modulo their limbs; their device table is cached by content (the Chebyshev
// coefficients of a bootstrap repeat), so only the first use of a table blocks for its upload.
extern "C" fhe_status fhe_lincomb(fhe_ctx* c, uint64_t* out, const uint64_t* const* x, const uint64_t* consts, uint32_t nTerms,
                                  const uint32_t* limbIdx, uint32_t nLimbs, uint32_t batch, int accumulate, void* stream) {
    ARG_CHECK(c && out && x && consts && nTerms >= 1 && batch >= 1, "fhe_lincomb: bad argument");
    LinCombArgs g;
    if (fhe_status s = make_sel(c, limbIdx, nLimbs, &g.sel, "fhe_lincomb"))
        return s;
    RT_CHECK(rt::set_device(c->device));
    // More than kMaxLinTerms terms take several launches and the first one overwrites `out`: terms that ARE `out` go into the first
    // launch (a launch reads every word before it writes it), so the sum is right wherever the caller put them.
    std::vector<uint32_t> order;
    order.reserve(nTerms);
    for (uint32_t i = 0; i < nTerms; ++i)
        if (x[i] == out)
            order.push_back(i);
    ARG_CHECK(order.size() <= (size_t)kMaxLinTerms || nTerms <= (uint32_t)kMaxLinTerms,
              "fhe_lincomb: more than 16 of more than 16 terms alias out");
    if (nTerms <= (uint32_t)kMaxLinTerms)
        order.clear();
    const size_t nAliased = order.size();
    for (uint32_t i = 0; i < nTerms; ++i)
        if (!(nAliased && x[i] == out))
            order.push_back(i);
    std::vector<uint32_t> li((size_t)nTerms * nLimbs);
    std::vector<uint64_t> cs((size_t)nTerms * nLimbs);
    for (uint32_t i = 0; i < nTerms; ++i)
        for (uint32_t r = 0; r < nLimbs; ++r) {
            li[(size_t)i * nLimbs + r] = limbIdx ? limbIdx[r] : r;
            cs[(size_t)i * nLimbs + r] = consts[(size_t)order[i] * nLimbs + r];
        }
    const TwPair* table = nullptr;
    if (fhe_status s = const_tables_trim(c))
        return s;
    std::shared_lock<std::shared_mutex> gate(c->constTabsGate);
    if (fhe_status s = const_table(c, li.data(), cs.data(), nTerms * nLimbs, &table))
        return s;
    g.out = out, g.q = c->d_q, g.logN = c->logN, g.nLimbs = nLimbs, g.rows = batch * nLimbs;
    for (uint32_t first = 0; first < nTerms; first += (uint32_t)kMaxLinTerms) {
        const uint32_t n = std::min<uint32_t>(kMaxLinTerms, nTerms - first);
        for (uint32_t i = 0; i < (uint32_t)kMaxLinTerms; ++i) {
            g.x[i] = i < n ? x[order[first + i]] : nullptr;
            ARG_CHECK(i >= n || g.x[i], "fhe_lincomb: null term");
        }
        g.consts = table + (size_t)first * nLimbs, g.nTerms = n, g.accumulate = (accumulate || first) ? 1u : 0u;
        FHE_LAUNCH(lincomb_kernel, tiles_for(c, g.rows), stream, g);
        LAUNCH_CHECK();
    }
    return FHE_OK;
}
// ---- the two elements of a ciphertext in ONE launch: towers 0 and 1 of every operand are separately allocated buffers ----
// (pke applies every operation element by element, base-leveledshe.cpp:562-606, ckksrns-leveledshe.cpp:748-759: at one ciphertext
// a launch per element leaves most of the chip idle — a tower of 14 limbs at N = 2^17 is 448 workgroups for 1024 resident slots)
static bool pair_deltas(const uint64_t* o0, const uint64_t* o1, const uint64_t* a0, const uint64_t* a1, const uint64_t* b0,
                        const uint64_t* b1, int64_t d[3]) {
    d[0] = o1 - o0, d[1] = a1 - a0, d[2] = b0 ? b1 - b0 : 1;
    return d[0] != 0 && d[1] != 0 && d[2] != 0;  // (0 means "dense" to the kernels: distinct towers are never 0 apart)
}
extern "C" fhe_status fhe_add_pair(fhe_ctx* c, uint64_t* o0, uint64_t* o1, const uint64_t* a0, const uint64_t* a1, const uint64_t* b0,
                                   const uint64_t* b1, const uint32_t* li, uint32_t nl, void* st) {
    int64_t d[3];
    ARG_CHECK(o0 && o1 && a0 && a1 && b0 && b1 && pair_deltas(o0, o1, a0, a1, b0, b1, d), "fhe_add_pair: bad argument");
    return elem_run<OP_ADD>(c, o0, a0, b0, nullptr, li, nl, 2, st, "fhe_add_pair", 0, 0, 0, 0, 0, 0, d);
}
extern "C" fhe_status fhe_sub_pair(fhe_ctx* c, uint64_t* o0, uint64_t* o1, const uint64_t* a0, const uint64_t* a1, const uint64_t* b0,
                                   const uint64_t* b1, const uint32_t* li, uint32_t nl, void* st) {
    int64_t d[3];
    ARG_CHECK(o0 && o1 && a0 && a1 && b0 && b1 && pair_deltas(o0, o1, a0, a1, b0, b1, d), "fhe_sub_pair: bad argument");
    return elem_run<OP_SUB>(c, o0, a0, b0, nullptr, li, nl, 2, st, "fhe_sub_pair", 0, 0, 0, 0, 0, 0, d);
}
extern "C" fhe_status fhe_mul_const_pair(fhe_ctx* c, uint64_t* o0, uint64_t* o1, const uint64_t* a0, const uint64_t* a1,
                                         const uint64_t* consts, const uint32_t* li, uint32_t nl, void* st) {
    int64_t d[3];
    ARG_CHECK(o0 && o1 && a0 && a1 && pair_deltas(o0, o1, a0, a1, nullptr, nullptr, d), "fhe_mul_const_pair: bad argument");
    HostConstVec cv;
    if (fhe_status s = make_const_vec(c, consts, li, nl, &cv, "fhe_mul_const_pair"))
        return s;
    return elem_cv_run<OP_MUL_CONST>(c, o0, a0, nullptr, cv, li, nl, 2, st, "fhe_mul_const_pair", 0, 0, d);
}
// DCRTPolyImpl::Plus(vector<Integer>) (dcrtpoly-impl.h:520-527 -> PolyImpl::Plus(Integer), poly-impl.h:211-218): limb i plus the
// constant polynomial consts[i] — every word in EVALUATION, coefficient 0 only in COEFFICIENT (coeff0Only)
extern "C" fhe_status fhe_add_const(fhe_ctx* c, uint64_t* o, const uint64_t* a, const uint64_t* consts, const uint32_t* li,
                                    uint32_t nl, uint32_t bt, int coeff0Only, void* st) {
    HostConstVec cv;
    if (fhe_status s = make_const_vec(c, consts, li, nl, &cv, "fhe_add_const"))
        return s;
    if (coeff0Only)
        return elem_cv_run<OP_ADD_CONST_AT0>(c, o, a, nullptr, cv, li, nl, bt, st, "fhe_add_const");
    return elem_cv_run<OP_ADD_CONST>(c, o, a, nullptr, cv, li, nl, bt, st, "fhe_add_const");
}
// DCRTPolyImpl::Minus(vector<Integer>) (dcrtpoly-impl.h:541-548 -> PolyImpl::Minus(Integer), poly-impl.h:221-225: ModSub on
// every word in both formats)
extern "C" fhe_status fhe_sub_const(fhe_ctx* c, uint64_t* o, const uint64_t* a, const uint64_t* consts, const uint32_t* li,
                                    uint32_t nl, uint32_t bt, void* st) {
    HostConstVec cv;
    if (fhe_status s = make_const_vec(c, consts, li, nl, &cv, "fhe_sub_const"))
        return s;
    for (uint32_t i = 0; i < nl; ++i) {  // a - k = a + (q - k)
        const uint64_t ql = c->q[li ? li[i] : i], k = cv.c[i].w;
        cv.c[i]           = TwPair{k ? ql - k : 0, 0};
    }
    return elem_cv_run<OP_ADD_CONST>(c, o, a, nullptr, cv, li, nl, bt, st, "fhe_sub_const");
}
// DCRTPolyImpl::TimesQovert (dcrtpoly-impl.h:868-885): every word x of limb i becomes ((x * NegQModt) mod t) * tInvModq[i] mod q_i
// (ModMulFastConst modulo t, then the generalized Barrett product modulo q_i); BFV encryption's scaling of the message by Q/t
extern "C" fhe_status fhe_times_q_over_t(fhe_ctx* c, uint64_t* o, const uint64_t* a, uint64_t t, uint64_t negQModt,
                                         const uint64_t* tInvModq, const uint32_t* li, uint32_t nl, uint32_t bt, void* st) {
    ARG_CHECK(c && o && a && tInvModq && t >= 2 && bt >= 1, "fhe_times_q_over_t: bad argument");
    HostConstVec cv;
    if (fhe_status s = make_const_vec(c, tInvModq, li, nl, &cv, "fhe_times_q_over_t"))
        return s;
    ElemArgs g;
    if (fhe_status s = make_sel(c, li, nl, &g.sel, "fhe_times_q_over_t"))
        return s;
    RT_CHECK(rt::set_device(c->device));
    g.out = o, g.a = a, g.b = nullptr, g.lc = c->d_lc, g.consts = nullptr;
    g.logN = c->logN, g.nLimbs = nl, g.rows = bt * nl;
    g.aStride = g.aFirst = g.bStride = g.bFirst = g.oStride = g.oFirst = 0;
    g.pre = TwPair{negQModt % t, host::shoup(negQModt % t, t)}, g.preMod = t;
    launch_cv<OP_TIMES_QOVERT>(c, g, cv, st);
    LAUNCH_CHECK();
    return FHE_OK;
}
// DCRTPolyImpl::SetValuesModSwitch (dcrtpoly-impl.h:630-647) on `words` COEFFICIENT words modulo qFrom -> residues modulo qTo
extern "C" fhe_status fhe_mod_switch_round(fhe_ctx* c, const uint64_t* x, uint64_t qFrom, uint64_t qTo, uint64_t* out, size_t words,
                                           void* st) {
    ARG_CHECK(c && x && out && qFrom >= 2 && qTo >= 2 && words >= 1, "fhe_mod_switch_round: bad argument");
    RT_CHECK(rt::set_device(c->device));
    ModSwitchRoundArgs g;
    g.x = x, g.out = out, g.qTo = qTo, g.words = words;
    g.ratio = static_cast<double>(qTo) / static_cast<double>(qFrom);  // :641
    FHE_LAUNCH(mod_switch_round_kernel, (words + kThreads - 1) / kThreads, st, g);
    LAUNCH_CHECK();
    return FHE_OK;
}
// NativeVectorT::MultAccEqNoCheck per limb (mubintvecnat.cpp:132-142): acc[r] += v[r] * I[r]  (I reduced first, Shoup
// product, ModAddFastEq)
extern "C" fhe_status fhe_mult_acc(fhe_ctx* c, uint64_t* acc, const uint64_t* v, const uint64_t* consts,
                                   const uint32_t* li, uint32_t nl, uint32_t bt, void* st) {
    HostConstVec cv;
    if (fhe_status s = make_const_vec(c, consts, li, nl, &cv, "fhe_mult_acc"))
        return s;
    return elem_cv_run<OP_MUL_CONST_ADD>(c, acc, v, acc, cv, li, nl, bt, st, "fhe_mult_acc");
}
// DCRTPolyImpl::ExpandCRTBasisQlHat (dcrtpoly-impl.h:1167-1187): limbs [0, sizeQl) times QlHatModq[i], limbs [sizeQl, sizeQ)
// zero; x [batch][sizeQl][N] -> out [batch][sizeQ][N] over context limbs limbIdx[0..sizeQ) (format unchanged)
// rows [first, first + nRows) of every tower of out[batch][stride][N] = 0, at copy speed (zero_rows_kernel, elemwise_kernels.h)
static fhe_status zero_rows(const fhe_ctx* c, uint64_t* out, uint32_t stride, uint32_t first, uint32_t nRows, uint32_t batch, void* st) {
    if (!nRows || !batch)
        return FHE_OK;
    ZeroRowsArgs g{out, c->logN, stride, first, nRows, batch};
    FHE_LAUNCH(zero_rows_kernel, tiles_for(c, (uint64_t)batch * nRows), st, g);
    LAUNCH_CHECK();
    return FHE_OK;
}
extern "C" fhe_status fhe_expand_crt_basis_ql_hat(fhe_ctx* c, const uint64_t* x, uint32_t sizeQl, const uint64_t* QlHatModq,
                                                  const uint32_t* li, uint32_t sizeQ, uint32_t bt, uint64_t* out, void* st) {
    ARG_CHECK(c && x && out && QlHatModq, "fhe_expand_crt_basis_ql_hat: null argument");
    ARG_CHECK(sizeQl >= 1 && sizeQl <= sizeQ && sizeQ <= (uint32_t)kMaxLimbs && bt >= 1, "fhe_expand_crt_basis_ql_hat: bad sizes");
    ARG_CHECK(out != x || sizeQl == sizeQ, "fhe_expand_crt_basis_ql_hat: in-place only when no limb is appended");
    HostConstVec cv;
    if (fhe_status s = make_const_vec(c, QlHatModq, li, sizeQl, &cv, "fhe_expand_crt_basis_ql_hat"))
        return s;
    for (uint32_t i = sizeQl; i < sizeQ; ++i)
        ARG_CHECK((li ? li[i] : i) < c->L, "fhe_expand_crt_basis_ql_hat: limb index exceeds context size");
    RT_CHECK(rt::set_device(c->device));
    if (fhe_status s = zero_rows(c, out, sizeQ, sizeQl, sizeQ - sizeQl, bt, st))
        return s;
    return elem_cv_run<OP_MUL_CONST>(c, out, x, nullptr, cv, li, sizeQl, bt, st, "fhe_expand_crt_basis_ql_hat", sizeQ, 0);
}

extern "C" fhe_status fhe_tensor(fhe_ctx* c, const uint64_t* a0, const uint64_t* a1, const uint64_t* b0,
                                 const uint64_t* b1, uint64_t* d0, uint64_t* d1, uint64_t* d2, const uint32_t* li,
                                 uint32_t nl, uint32_t bt, void* st) {
    ARG_CHECK(c && a0 && a1 && b0 && b1 && d0 && d1 && d2, "fhe_tensor: null argument");
    ARG_CHECK(bt >= 1, "fhe_tensor: batch must be >= 1");
    TensorArgs g;
    if (fhe_status s = make_sel(c, li, nl, &g.sel, "fhe_tensor"))
        return s;
    RT_CHECK(rt::set_device(c->device));
    g.a0 = a0, g.a1 = a1, g.b0 = b0, g.b1 = b1, g.d0 = d0, g.d1 = d1, g.d2 = d2;
    g.lc     = c->d_lc;
    g.logN   = c->logN;
    g.nLimbs = nl;
    g.rows   = bt * nl;
    FHE_LAUNCH(tensor_kernel, tiles_for(c, g.rows), st, g);
    LAUNCH_CHECK();
    return FHE_OK;
}

// LeveledSHEBase::EvalSquareCore for 2-element ciphertexts (base-leveledshe.cpp:646-664)
extern "C" fhe_status fhe_tensor_square(fhe_ctx* c, const uint64_t* a0, const uint64_t* a1, uint64_t* d0, uint64_t* d1,
                                        uint64_t* d2, const uint32_t* li, uint32_t nl, uint32_t bt, void* st) {
    ARG_CHECK(c && a0 && a1 && d0 && d1 && d2, "fhe_tensor_square: null argument");
    ARG_CHECK(bt >= 1, "fhe_tensor_square: batch must be >= 1");
    TensorSqArgs g;
    if (fhe_status s = make_sel(c, li, nl, &g.sel, "fhe_tensor_square"))
        return s;
    RT_CHECK(rt::set_device(c->device));
    g.a0 = a0, g.a1 = a1, g.d0 = d0, g.d1 = d1, g.d2 = d2;
    g.lc = c->d_lc, g.logN = c->logN, g.nLimbs = nl, g.rows = bt * nl;
    FHE_LAUNCH(tensor_square_kernel, tiles_for(c, g.rows), st, g);
    LAUNCH_CHECK();
    return FHE_OK;
}

extern "C" fhe_status fhe_automorph(fhe_ctx* c, uint64_t* out, const uint64_t* in, uint32_t k, int evalFormat,
                                    const uint32_t* li, uint32_t nl, uint32_t bt, void* st) {
    ARG_CHECK(c && out && in, "fhe_automorph: null argument");
    ARG_CHECK(out != in, "fhe_automorph: out must not alias in");
    ARG_CHECK(k % 2 == 1, "Automorphism index not odd");  // poly-impl.h:337-338
    ARG_CHECK(bt >= 1, "fhe_automorph: batch must be >= 1");
    AutoArgs g;
    if (fhe_status s = make_sel(c, li, nl, &g.sel, "fhe_automorph"))
        return s;
    RT_CHECK(rt::set_device(c->device));
    g.out = out, g.in = in, g.q = c->d_q;
    g.logN = c->logN, g.nLimbs = nl, g.rows = bt * nl, g.k = k, g.evalFormat = evalFormat ? 1u : 0u;
    FHE_LAUNCH(automorph_kernel, tiles_for(c, g.rows), st, g);
    LAUNCH_CHECK();
    return FHE_OK;
}
static fhe_status switch_modulus_run(fhe_ctx* c, uint64_t* out, const LimbSel& sel, uint32_t nl, const uint64_t* src,
                                     uint32_t srcLimbs, uint32_t srcPos, uint32_t srcCtxLimb, const TwPair* d_consts,
                                     uint32_t bt, void* st) {
    SwitchModArgs g;
    g.out = out, g.src = src, g.q = c->d_q, g.consts = d_consts;
    g.logN = c->logN, g.nLimbs = nl, g.rows = bt * nl;
    g.srcStrideLimbs = srcLimbs, g.srcLimbPos = srcPos, g.srcCtxLimb = srcCtxLimb;
    g.sel = sel;
    FHE_LAUNCH(switch_modulus_kernel, tiles_for(c, g.rows), st, g);
    LAUNCH_CHECK();
    return FHE_OK;
}
extern "C" fhe_status fhe_switch_modulus(fhe_ctx* c, uint64_t* out, const uint32_t* li, uint32_t nl, const uint64_t* src,
                                         uint32_t srcLimbs, uint32_t srcPos, uint32_t srcCtxLimb, uint32_t bt,
                                         void* st) {
    ARG_CHECK(c && out && src, "fhe_switch_modulus: null argument");
    ARG_CHECK(srcPos < srcLimbs && srcCtxLimb < c->L && bt >= 1, "fhe_switch_modulus: bad source limb");
    LimbSel sel;
    if (fhe_status s = make_sel(c, li, nl, &sel, "fhe_switch_modulus"))
        return s;
    RT_CHECK(rt::set_device(c->device));
    return switch_modulus_run(c, out, sel, nl, src, srcLimbs, srcPos, srcCtxLimb, nullptr, bt, st);
}

// DCRTPolyImpl::CRTDecompose(baseBits) (dcrtpoly-impl.h:230-285): x [nLimbs][N] COEFFICIENT -> out [towers][nLimbs][N] EVALUATION, towers in the
// reference's order (limb 0's digits, least significant first, then limb 1's ...).  One launch per source limb, ONE transform over all towers.
static uint32_t crt_windows(const fhe_ctx* c, uint32_t limb, uint32_t baseBits) {
    if (baseBits == 0)
        return 1;
    const uint32_t nBits = host::bitlen(c->q[limb]);
    return nBits / baseBits + (nBits % baseBits != 0);
}
extern "C" uint32_t fhe_crt_decompose_towers(const fhe_ctx* c, const uint32_t* li, uint32_t nl, uint32_t baseBits) {
    if (!c || nl < 1 || nl > (uint32_t)kMaxLimbs || baseBits > 31)  // (the reference's base is `1 << baseBits` in 32 bits)
        return 0;
    uint32_t towers = 0;
    for (uint32_t i = 0; i < nl; ++i) {
        const uint32_t l = li ? li[i] : i;
        if (l >= c->L)
            return 0;
        const uint32_t nW = crt_windows(c, l, baseBits);
        if ((uint64_t)nW * baseBits > 64)  // (the reference reads bits beyond the word there: undefined, left to the caller's host path)
            return 0;
        towers += nW;
    }
    return towers;
}
extern "C" fhe_status fhe_crt_decompose(fhe_ctx* c, const uint64_t* x, const uint32_t* li, uint32_t nl, uint32_t baseBits, uint64_t* out,
                                        void* st) {
    ARG_CHECK(c && x && out, "fhe_crt_decompose: null argument");
    const uint32_t towers = fhe_crt_decompose_towers(c, li, nl, baseBits);
    ARG_CHECK(towers >= 1, "fhe_crt_decompose: limb selection or digit size outside the device path (baseBits <= 31, every window inside the word)");
    CrtDigitsArgs g;
    if (fhe_status s = make_sel(c, li, nl, &g.sel, "fhe_crt_decompose"))
        return s;
    RT_CHECK(rt::set_device(c->device));
    g.q = c->d_q, g.logN = c->logN, g.nLimbs = nl, g.baseBits = baseBits;
    uint32_t t0 = 0;
    for (uint32_t i = 0; i < nl; ++i) {
        const uint32_t l = li ? li[i] : i;
        g.nW = crt_windows(c, l, baseBits), g.srcPos = i, g.srcCtxLimb = l;
        g.src = x + ((size_t)i << c->logN), g.out = out + (((size_t)t0 * nl) << c->logN);
        FHE_LAUNCH(crt_digits_kernel, tiles_for(c, (uint64_t)g.nW * nl), st, g);
        LAUNCH_CHECK();
        t0 += g.nW;
    }
    return fhe_ntt_fwd(c, out, li, nl, towers, st);
}

// ---- f3: sampled towers on the device (sampler_kernels.h) ----------------------------------------------------------------------
static fhe_status sample_run(fhe_ctx* c, uint64_t* out, const uint32_t* li, uint32_t nl, uint32_t bt, uint32_t kind, double sigma,
                             uint64_t seed, uint32_t streamId, void* st, const char* who) {
    ARG_CHECK(c && out && bt >= 1, "fhe_sample: null argument or empty batch");
    SampleArgs a;
    if (fhe_status s = make_sel(c, li, nl, &a.sel, who))
        return s;
    RT_CHECK(rt::set_device(c->device));
    a.out = out, a.q = c->d_q, a.logN = c->logN, a.nLimbs = nl, a.batch = bt, a.seed = seed, a.stream = streamId, a.kind = kind;
    a.cdf = nullptr, a.cdfLen = 0, a.a = 0.0;
    if (kind == 1) {
        // the reference's table (DiscreteGaussianGeneratorImpl::Initialize, discretegaussiangenerator-impl.h:75-89), built once per sigma
        ARG_CHECK(sigma > 1.000000001 && sigma < 300.0, "fhe_sample_gaussian: the inversion sampler covers 1 < sigma < 300 (the reference's Peikert range)");
        std::lock_guard<std::mutex> lk(c->cacheMutex);
        auto it = c->dggTabs.find(sigma);
        if (it == c->dggTabs.end()) {
            const double M = 12.00610553538285;
            const int64_t fin = (int64_t)std::ceil(sigma * M);
            std::vector<double> vals((size_t)fin);
            const double variance = 2 * sigma * sigma;
            double cusum = 0.0;
            for (int64_t x = 1; x <= fin; ++x)
                vals[(size_t)(x - 1)] = (cusum += std::exp(-((double)(x * x) / variance)));
            const double av = 1.0 / (2 * cusum + 1.0);
            for (auto& v : vals)
                v *= av;
            void* d = nullptr;
            if (fhe_status s = upload(c, vals.data(), vals.size() * sizeof(double), &d))
                return s;
            it = c->dggTabs.emplace(sigma, std::make_tuple((const double*)d, (uint32_t)fin, av)).first;
        }
        a.cdf = std::get<0>(it->second), a.cdfLen = std::get<1>(it->second), a.a = std::get<2>(it->second);
    }
    const uint64_t lanes = kind == 0 ? ((uint64_t)bt * nl) << c->logN : (uint64_t)bt << c->logN;
    FHE_LAUNCH(sample_kernel, (uint32_t)((lanes + kThreads - 1) / kThreads), st, a);
    LAUNCH_CHECK();
    return FHE_OK;
}
extern "C" fhe_status fhe_sample_uniform(fhe_ctx* c, uint64_t* out, const uint32_t* li, uint32_t nl, uint32_t bt, uint64_t seed,
                                         uint32_t streamId, void* st) {
    return sample_run(c, out, li, nl, bt, 0, 0.0, seed, streamId, st, "fhe_sample_uniform");
}
extern "C" fhe_status fhe_sample_gaussian(fhe_ctx* c, uint64_t* out, const uint32_t* li, uint32_t nl, uint32_t bt, double sigma,
                                          uint64_t seed, uint32_t streamId, void* st) {
    return sample_run(c, out, li, nl, bt, 1, sigma, seed, streamId, st, "fhe_sample_gaussian");
}
extern "C" fhe_status fhe_sample_ternary(fhe_ctx* c, uint64_t* out, const uint32_t* li, uint32_t nl, uint32_t bt, uint64_t seed,
                                         uint32_t streamId, void* st) {
    return sample_run(c, out, li, nl, bt, 2, 0.0, seed, streamId, st, "fhe_sample_ternary");
}

// ------------------------------------------------------------------------------------------------
// basis conversion plans
// ------------------------------------------------------------------------------------------------
struct fhe_conv {
    fhe_ctx* ctx;
    uint32_t nSrc, nDst;
    std::vector<uint32_t> srcIdx, dstIdx;  // context limbs of the two bases (empty for internal plans)
    ConvTables tb;                   // = chunks[0]
    std::vector<ConvTables> chunks;  // source limbs [32k, 32k+32): one launch each (more than one only beyond 32 source limbs)
    const TwPair* allHatInv  = nullptr;  // chunked exact plans: every source limb's multiplier / modulus / 1.0/q_i
    const uint64_t* allSrcQ  = nullptr;
    const double* allQInv    = nullptr;
    std::vector<void*> owned;
};

static fhe_status conv_upload(fhe_conv* cv, const void* h, size_t bytes, const void** d) {
    void* p = nullptr;
    RT_CHECK(rt::dmalloc(&p, bytes));
    cv->owned.push_back(p);
    RT_CHECK(rt::h2d(p, h, bytes, nullptr));
    RT_CHECK(rt::sync(nullptr));
    *d = p;
    return FHE_OK;
}

static uint32_t conv_nsrc_pad(uint32_t nSrc) { return nSrc <= 8 ? 8u : nSrc <= 16 ? 16u : 32u; }
// device plan from explicit tables: hatInv[i] (multiplier of source residue i, mod src_i), hatMod[i*nDst + j] (weight of
// y_i in target j, mod dst_j), alphaMod[a*nDst + j] / qInv[i] for the exact variant (may be null: approximate only).
// Table rows are padded to the NSRC of the kernel instantiation conv_run picks, so that the kernel's table reads are
// unconditional (basis_kernels.h).
static fhe_status conv_from_tables(fhe_ctx* c, const std::vector<uint64_t>& src, const std::vector<uint64_t>& dst,
                                   const uint64_t* hatInvIn, const uint64_t* hatModIn, const uint64_t* alphaIn,
                                   const double* qInvIn, fhe_conv** out) {
    const uint32_t nSrc = (uint32_t)src.size(), nDst = (uint32_t)dst.size();
    if (nSrc < 1 || nSrc > (uint32_t)kMaxLimbs || nDst < 1 || nDst > (uint32_t)kMaxLimbs)
        return fail(FHE_ERR_ARG, "basis conversion: 1..256 source and target limbs");
    fhe_conv* cv = new fhe_conv;
    cv->ctx      = c;
    cv->nSrc     = nSrc;
    cv->nDst     = nDst;
    // target-side tables, shared by all chunks
    std::vector<uint64_t> mu(2 * (size_t)nDst), alphaMod((size_t)(nSrc + 1) * nDst, 0), red(4 * (size_t)nDst);
    for (uint32_t j = 0; j < nDst; ++j) {
        host::mu128(dst[j], &mu[2 * j]);
        const uint64_t R = (uint64_t)((((unsigned __int128)1) << 64) % dst[j]);
        red[4 * j]       = dst[j];
        red[4 * j + 1]   = R;
        red[4 * j + 2]   = host::shoup(R, dst[j]);
        red[4 * j + 3]   = (uint64_t)((((unsigned __int128)1) << 64) / dst[j]);
    }
    if (alphaIn)
        for (size_t k = 0; k < alphaMod.size(); ++k)
            alphaMod[k] = alphaIn[k] % dst[k % nDst];
    ConvTables shared{};
    fhe_status s;
    if ((s = conv_upload(cv, dst.data(), dst.size() * 8, (const void**)&shared.dstQ)) ||
        (s = conv_upload(cv, mu.data(), mu.size() * 8, (const void**)&shared.dstMu)) ||
        (s = conv_upload(cv, red.data(), red.size() * 8, (const void**)&shared.dstRed)) ||
        (s = conv_upload(cv, alphaMod.data(), alphaMod.size() * 8, (const void**)&shared.alphaMod))) {
        fhe_conv_destroy(cv);
        return s;
    }
    // source-side tables per chunk of <= 32 source limbs (one chunk = one kernel launch; the usual case is one chunk)
    for (uint32_t c0 = 0; c0 < nSrc; c0 += 32u) {
        const uint32_t n   = std::min(32u, nSrc - c0);
        const uint32_t pad = conv_nsrc_pad(n);
        std::vector<TwPair> hatInv(32, TwPair{0, 0});
        std::vector<uint64_t> hatMod((size_t)pad * nDst, 0), srcPad(32, 1);
        std::vector<double> qInv(32, 0.0);
        for (uint32_t i = 0; i < n; ++i) {
            const uint64_t inv = hatInvIn[c0 + i] % src[c0 + i];
            hatInv[i]          = TwPair{inv, host::shoup(inv, src[c0 + i])};
            qInv[i]            = qInvIn ? qInvIn[c0 + i] : 1.0 / static_cast<double>(src[c0 + i]);
            srcPad[i]          = src[c0 + i];
            for (uint32_t j = 0; j < nDst; ++j)
                hatMod[(size_t)j * pad + i] = hatModIn[(size_t)(c0 + i) * nDst + j] % dst[j];
        }
        ConvTables tb = shared;
        if ((s = conv_upload(cv, hatInv.data(), hatInv.size() * sizeof(TwPair), (const void**)&tb.hatInv)) ||
            (s = conv_upload(cv, hatMod.data(), hatMod.size() * 8, (const void**)&tb.hatMod)) ||
            (s = conv_upload(cv, srcPad.data(), srcPad.size() * 8, (const void**)&tb.srcQ)) ||
            (s = conv_upload(cv, qInv.data(), qInv.size() * sizeof(double), (const void**)&tb.srcQInv))) {
            fhe_conv_destroy(cv);
            return s;
        }
        cv->chunks.push_back(tb);
    }
    cv->tb = cv->chunks[0];
    if (nSrc > 32u) {  // the last chunk of the exact variant counts the overflow over all source limbs
        std::vector<TwPair> allInv(nSrc);
        std::vector<double> allQInv(nSrc);
        for (uint32_t i = 0; i < nSrc; ++i) {
            const uint64_t inv = hatInvIn[i] % src[i];
            allInv[i]          = TwPair{inv, host::shoup(inv, src[i])};
            allQInv[i]         = qInvIn ? qInvIn[i] : 1.0 / static_cast<double>(src[i]);
        }
        if ((s = conv_upload(cv, allInv.data(), allInv.size() * sizeof(TwPair), (const void**)&cv->allHatInv)) ||
            (s = conv_upload(cv, src.data(), src.size() * 8, (const void**)&cv->allSrcQ)) ||
            (s = conv_upload(cv, allQInv.data(), allQInv.size() * sizeof(double), (const void**)&cv->allQInv))) {
            fhe_conv_destroy(cv);
            return s;
        }
    }
    *out = cv;
    return FHE_OK;
}
// tables for converting from moduli src[] to dst[]   (rns-cryptoparameters.cpp:214-246, 297-349); srcScale[i] (mod
// src_i) and dstScale[j] (mod dst_j), when given, are folded into the two constant sets: the plan then computes
// dstScale_j * SwitchBasis(srcScale_i * x_i) — exact as residues, used for the BGV t factors of ApproxModDown
static fhe_status conv_build(fhe_ctx* c, const std::vector<uint64_t>& src, const std::vector<uint64_t>& dst, fhe_conv** out,
                             const uint64_t* srcScale = nullptr, const uint64_t* dstScale = nullptr) {
    const uint32_t nSrc = (uint32_t)src.size(), nDst = (uint32_t)dst.size();
    std::vector<uint64_t> hatInv(nSrc), hatMod((size_t)nSrc * nDst), alphaMod((size_t)(nSrc + 1) * nDst);
    for (uint32_t i = 0; i < nSrc; ++i) {
        const uint64_t hat = host::prod_mod(src, (int)i, src[i]);
        hatInv[i]          = host::invmod(hat, src[i]);
        if (srcScale)
            hatInv[i] = host::mulmod(hatInv[i], srcScale[i] % src[i], src[i]);
        for (uint32_t j = 0; j < nDst; ++j) {
            uint64_t v = host::prod_mod(src, (int)i, dst[j]);
            if (dstScale)
                v = host::mulmod(v, dstScale[j] % dst[j], dst[j]);
            hatMod[(size_t)i * nDst + j] = v;
        }
    }
    for (uint32_t j = 0; j < nDst; ++j) {
        const uint64_t Qmod = host::prod_mod(src, -1, dst[j]);
        for (uint32_t a = 0; a <= nSrc; ++a)
            alphaMod[(size_t)a * nDst + j] = host::mulmod(a % dst[j], Qmod, dst[j]);
    }
    return conv_from_tables(c, src, dst, hatInv.data(), hatMod.data(), alphaMod.data(), nullptr, out);
}

extern "C" fhe_status fhe_conv_create(fhe_ctx* c, const uint32_t* srcIdx, uint32_t nSrc, const uint32_t* dstIdx,
                                      uint32_t nDst, fhe_conv** out) {
    ARG_CHECK(c && srcIdx && dstIdx && out, "fhe_conv_create: null argument");
    ARG_CHECK(nSrc >= 1 && nSrc <= (uint32_t)kMaxLimbs && nDst >= 1 && nDst <= (uint32_t)kMaxLimbs, "fhe_conv_create: bad basis size");
    std::vector<uint64_t> src(nSrc), dst(nDst);
    for (uint32_t i = 0; i < nSrc; ++i) {
        ARG_CHECK(srcIdx[i] < c->L, "fhe_conv_create: source limb exceeds context size");
        src[i] = c->q[srcIdx[i]];
    }
    for (uint32_t j = 0; j < nDst; ++j) {
        ARG_CHECK(dstIdx[j] < c->L, "fhe_conv_create: target limb exceeds context size");
        dst[j] = c->q[dstIdx[j]];
    }
    RT_CHECK(rt::set_device(c->device));
    if (fhe_status s = conv_build(c, src, dst, out))
        return s;
    (*out)->srcIdx.assign(srcIdx, srcIdx + nSrc);
    (*out)->dstIdx.assign(dstIdx, dstIdx + nDst);
    return FHE_OK;
}
// conversion plan with the CALLER's tables, laid out as the reference passes them to ApproxSwitchCRTBasis /
// SwitchCRTBasis (dcrtpoly-impl.h:888-932, 1008-1085): QHatInvModq[nSrc], QHatModp[nSrc][nDst]; for the exact variant
// alphaQModp[nSrc+1][nDst] and qInv[nSrc] (doubles), else null.  Needed where the tables are not the plain CRT ones,
// e.g. FastExpandCRTBasisPloverQ's mPlQHatInvModq / qInvModp (bfvrns-cryptoparameters.cpp).
extern "C" fhe_status fhe_conv_create_custom(fhe_ctx* c, const uint32_t* srcIdx, uint32_t nSrc, const uint32_t* dstIdx,
                                             uint32_t nDst, const uint64_t* hatInv, const uint64_t* hatMod,
                                             const uint64_t* alphaMod, const double* qInv, fhe_conv** out) {
    ARG_CHECK(c && srcIdx && dstIdx && hatInv && hatMod && out, "fhe_conv_create_custom: null argument");
    ARG_CHECK(nSrc >= 1 && nSrc <= (uint32_t)kMaxLimbs && nDst >= 1 && nDst <= (uint32_t)kMaxLimbs, "fhe_conv_create_custom: bad basis size");
    ARG_CHECK((alphaMod == nullptr) == (qInv == nullptr), "fhe_conv_create_custom: alphaMod and qInv go together");
    std::vector<uint64_t> src(nSrc), dst(nDst);
    for (uint32_t i = 0; i < nSrc; ++i) {
        ARG_CHECK(srcIdx[i] < c->L, "fhe_conv_create_custom: source limb exceeds context size");
        src[i] = c->q[srcIdx[i]];
    }
    for (uint32_t j = 0; j < nDst; ++j) {
        ARG_CHECK(dstIdx[j] < c->L, "fhe_conv_create_custom: target limb exceeds context size");
        dst[j] = c->q[dstIdx[j]];
    }
    RT_CHECK(rt::set_device(c->device));
    if (fhe_status s = conv_from_tables(c, src, dst, hatInv, hatMod, alphaMod, qInv, out))
        return s;
    (*out)->srcIdx.assign(srcIdx, srcIdx + nSrc);
    (*out)->dstIdx.assign(dstIdx, dstIdx + nDst);
    return FHE_OK;
}
extern "C" void fhe_conv_destroy(fhe_conv* cv) {
    if (!cv)
        return;
    for (void* p : cv->owned)
        rt::dfree(p);
    delete cv;
}

template <bool EXACT>
static fhe_status conv_run(fhe_conv* cv, const uint64_t* in, uint32_t inStride, uint32_t inFirst, uint64_t* out,
                           uint32_t outStride, uint32_t outFirst, uint32_t batch, void* st) {
    ARG_CHECK(cv && in && out, "fhe_switch_basis: null argument");
    ARG_CHECK(batch >= 1 && inFirst + cv->nSrc <= inStride && outFirst + cv->nDst <= outStride,
              "fhe_switch_basis: limb window exceeds tower stride");
    RT_CHECK(rt::set_device(cv->ctx->device));
    ConvArgs g{};
    g.in = in, g.out = out, g.tb = cv->tb;
    g.logN = cv->ctx->logN, g.batch = batch, g.nSrc = cv->nSrc, g.nDst = cv->nDst;
    g.inStride = inStride, g.inFirst = inFirst, g.outStride = outStride, g.outFirst = outFirst;
    const uint64_t coeffs = (uint64_t)batch << g.logN;
    const uint32_t grid   = (uint32_t)((coeffs + kThreads - 1) / kThreads);
    // column-sum form of the kernel: 2 = factors split at 30 bits (default), FHE_CONV_SUM8=1 = carry-counted 64-bit columns
    static const bool split30 = env_u32("FHE_CONV_SUM8", 2) != 1u;
    if (cv->chunks.size() > 1) {  // more than 32 source limbs: one launch per chunk, sums accumulated in `out`
        g.nSrcAll = cv->nSrc, g.inAll = in + ((size_t)inFirst << g.logN);
        g.allHatInv = cv->allHatInv, g.allSrcQ = cv->allSrcQ, g.allQInv = cv->allQInv;
        for (size_t k = 0; k < cv->chunks.size(); ++k) {
            g.tb      = cv->chunks[k];
            g.nSrc    = std::min(32u, cv->nSrc - 32u * (uint32_t)k);
            g.inFirst = inFirst + 32u * (uint32_t)k;
            g.acc     = k > 0;
            g.last    = k + 1 == cv->chunks.size();
            const uint32_t pad = conv_nsrc_pad(g.nSrc);
            if (pad == 8)
                FHE_LAUNCH((switch_basis_kernel<8, EXACT, 2, true>), grid, st, g);
            else if (pad == 16)
                FHE_LAUNCH((switch_basis_kernel<16, EXACT, 2, true>), grid, st, g);
            else
                FHE_LAUNCH((switch_basis_kernel<32, EXACT, 2, true>), grid, st, g);
            LAUNCH_CHECK();
        }
        return FHE_OK;
    }
    const uint32_t pad = conv_nsrc_pad(cv->nSrc);
    if (pad == 8 && split30)
        FHE_LAUNCH((switch_basis_kernel<8, EXACT, 2>), grid, st, g);
    else if (pad == 8)
        FHE_LAUNCH((switch_basis_kernel<8, EXACT, 1>), grid, st, g);
    else if (pad == 16 && split30)
        FHE_LAUNCH((switch_basis_kernel<16, EXACT, 2>), grid, st, g);
    else if (pad == 16)
        FHE_LAUNCH((switch_basis_kernel<16, EXACT, 1>), grid, st, g);
    else if (split30)
        FHE_LAUNCH((switch_basis_kernel<32, EXACT, 2>), grid, st, g);
    else
        FHE_LAUNCH((switch_basis_kernel<32, EXACT, 1>), grid, st, g);
    LAUNCH_CHECK();
    return FHE_OK;
}
extern "C" fhe_status fhe_approx_switch_basis(fhe_conv* cv, const uint64_t* in, uint32_t is, uint32_t ifst,
                                              uint64_t* out, uint32_t os, uint32_t ofst, uint32_t b, void* st) {
    return conv_run<false>(cv, in, is, ifst, out, os, ofst, b, st);
}
extern "C" fhe_status fhe_switch_basis_exact(fhe_conv* cv, const uint64_t* in, uint32_t is, uint32_t ifst,
                                             uint64_t* out, uint32_t os, uint32_t ofst, uint32_t b, void* st) {
    return conv_run<true>(cv, in, is, ifst, out, os, ofst, b, st);
}

// DCRTPolyImpl::ExpandCRTBasis / ExpandCRTBasisReverseOrder (dcrtpoly-impl.h:1088-1148): x over the plan's source basis Q
// (format inEval) -> out over Q u P in resultEval, P = SwitchCRTBasis of the coefficient form; reverse: P rows first.
extern "C" size_t fhe_expand_crt_basis_workspace_bytes(const fhe_conv* cv, uint32_t batch) {
    return cv ? (((size_t)batch * cv->nSrc) << cv->ctx->logN) * 8 : 0;
}
static fhe_status expand_run(fhe_conv* cv, const uint64_t* x, int inEval, uint64_t* out, int resultEval, int reverseOrder,
                             uint32_t batch, void* ws, size_t wsBytes, void* st, bool exact) {
    ARG_CHECK(cv && x && out && batch >= 1, "fhe_expand_crt_basis: bad argument");
    ARG_CHECK(!cv->srcIdx.empty(), "fhe_expand_crt_basis: the plan must come from fhe_conv_create[_custom]");
    fhe_ctx* c = cv->ctx;
    RT_CHECK(rt::set_device(c->device));
    const uint32_t nQ = cv->nSrc, nP = cv->nDst, tot = nQ + nP;
    const uint32_t qFirst = reverseOrder ? nP : 0, pFirst = reverseOrder ? 0 : nQ;
    const size_t rowB = (size_t)8 << c->logN;
    const uint64_t* coef = x;  // coefficient form of the Q part
    if (inEval) {              // :1096-1099
        ARG_CHECK(ws && wsBytes >= fhe_expand_crt_basis_workspace_bytes(cv, batch), "fhe_expand_crt_basis: workspace too small");
        if (fhe_status s = ntt_run(c, true, x, (uint64_t*)ws, cv->srcIdx.data(), nQ, batch, st))
            return s;
        coef = (const uint64_t*)ws;
    }
    if (fhe_status s = exact ? conv_run<true>(cv, coef, nQ, 0, out, tot, pFirst, batch, st)  // :1101-1102
                             : conv_run<false>(cv, coef, nQ, 0, out, tot, pFirst, batch, st))  // ApproxModUp :948
        return s;
    // Q rows of the result: the stored EVALUATION copy when it can be reused (:1104-1105), else the coefficient form
    const uint64_t* qsrc = (resultEval && inEval) ? x : coef;
    uint64_t* qdst       = out + ((size_t)qFirst << c->logN);
    RT_CHECK(rt::d2d_2d(qdst, tot * rowB, qsrc, nQ * rowB, nQ * rowB, batch, (rt::stream_t)st));
    if (resultEval) {  // :1112-1114
        if (!inEval)
            if (fhe_status s = ntt_run(c, false, out, out, cv->srcIdx.data(), nQ, batch, st, tot, qFirst, tot, qFirst))
                return s;
        return ntt_run(c, false, out, out, cv->dstIdx.data(), nP, batch, st, tot, pFirst, tot, pFirst);
    }
    return FHE_OK;
}
extern "C" fhe_status fhe_expand_crt_basis(fhe_conv* cv, const uint64_t* x, int inEval, uint64_t* out, int resultEval,
                                           int reverseOrder, uint32_t batch, void* ws, size_t wsBytes, void* st) {
    return expand_run(cv, x, inEval, out, resultEval, reverseOrder, batch, ws, wsBytes, st, true);
}
// DCRTPolyImpl::ApproxModUp (dcrtpoly-impl.h:935-963): x over the plan's source basis Q (format inEval) -> out over Q u P in
// EVALUATION: the EVALUATION copy of the Q limbs is kept when there is one (:943-946, 950-951), P = ApproxSwitchCRTBasis of the
// coefficient form (:948), every limb to EVALUATION (:958-960).  Workspace as fhe_expand_crt_basis (EVALUATION input only).
extern "C" fhe_status fhe_mod_up(fhe_conv* cv, const uint64_t* x, int inEval, uint64_t* out, uint32_t batch, void* ws,
                                 size_t wsBytes, void* st) {
    return expand_run(cv, x, inEval, out, 1, 0, batch, ws, wsBytes, st, false);
}
// DCRTPolyImpl::FastExpandCRTBasisPloverQ (dcrtpoly-impl.h:1151-1164), COEFFICIENT format: partPl =
// ApproxSwitchCRTBasis(x; toPl) with the caller's mPlQHatInvModq / qInvModp tables (fhe_conv_create_custom), partQl =
// SwitchCRTBasis(partPl; toQl); out = [Ql rows | Pl rows].
extern "C" fhe_status fhe_fast_expand_crt_basis_p_over_q(fhe_conv* toPl, fhe_conv* toQl, const uint64_t* x, uint64_t* out,
                                                         uint32_t batch, void* st) {
    ARG_CHECK(toPl && toQl && x && out && batch >= 1, "fhe_fast_expand_crt_basis_p_over_q: bad argument");
    ARG_CHECK(toQl->nSrc == toPl->nDst, "fhe_fast_expand_crt_basis_p_over_q: the second plan must start from the first plan's target basis");
    const uint32_t nQl = toQl->nDst, nPl = toPl->nDst, tot = nQl + nPl;
    if (fhe_status s = conv_run<false>(toPl, x, toPl->nSrc, 0, out, tot, nQl, batch, st))
        return s;
    return conv_run<true>(toQl, out, tot, nQl, out, tot, 0, batch, st);
}

// ------------------------------------------------------------------------------------------------
// HYBRID key switching
// ------------------------------------------------------------------------------------------------
struct fhe_ks_plan {
    fhe_ctx* ctx;
    uint32_t sizeQ, sizeP, numPartQ, alpha;
    // ModUp plans per (level sizeQl, digit): built lazily per level, cached
    struct Level {
        uint32_t sizeQl = 0, numParts = 0;
        std::vector<fhe_conv*> up;              // digit -> complement conversion
        std::vector<std::vector<uint32_t>> cidx;  // complement context-limb lists
        std::vector<uint32_t> partSize;
        fhe_conv* down = nullptr;               // P -> Q_l
        std::map<uint64_t, fhe_conv*> downT;    // BGV: P -> Q_l with t^-1 (mod p_j) and t (mod q_i) folded in, per t
        TwPair* d_PInv = nullptr;               // [sizeQl] Shoup pairs of [P^-1]_{q_i}
        TwPair* d_PModq = nullptr;              // [sizeQl] Shoup pairs of [P]_{q_i} (built on first use)
    };
    std::vector<Level*> levels;  // index sizeQl
    std::vector<void*> owned;
    std::map<std::vector<uint64_t>, void*> bsgsTables;  // device copies of the diagonal pointer tables, by content
    uint64_t* d_zeroRows = nullptr;                     // [sizeQ+sizeP][N] zeros: the "absent diagonal" of the BSGS transform
    std::mutex cacheMutex;                              // guards the lazily built levels / tables (callers may be OpenMP threads)
};
struct fhe_ks_key {
    fhe_ks_plan* plan;
    uint64_t *d_b = nullptr, *d_a = nullptr;
    size_t words = 0;
    bool owned = true;
};

extern "C" fhe_status fhe_ks_plan_create(fhe_ctx* c, uint32_t sizeQ, uint32_t sizeP, uint32_t numPartQ,
                                         fhe_ks_plan** out) {
    ARG_CHECK(c && out, "fhe_ks_plan_create: null argument");
    ARG_CHECK(sizeQ >= 1 && sizeP >= 1 && sizeQ + sizeP <= c->L, "fhe_ks_plan_create: context must hold Q then P limbs");
    ARG_CHECK(numPartQ >= 1, "numPartQ is zero");  // rns-cryptoparameters.cpp:82-83
    const uint32_t a = (sizeQ + numPartQ - 1) / numPartQ;
    // rns-cryptoparameters.cpp:87-91
    ARG_CHECK(sizeQ > a * (numPartQ - 1),
              "HYBRID key switching parameters: Can't appropriately distribute towers into digits");
    // (digits or P of more than 32 limbs — dnum = 1 or 2 on long chains — run as chunked conversions; more than 8 digits as
    // chunked inner products: the reference's loops have no such bounds, dcrtpoly-impl.h:895-915, keyswitch-hybrid.cpp:419-430)
    fhe_ks_plan* p = new fhe_ks_plan;
    p->ctx = c, p->sizeQ = sizeQ, p->sizeP = sizeP, p->numPartQ = numPartQ, p->alpha = a;
    p->levels.assign(sizeQ + 1, nullptr);
    *out = p;
    return FHE_OK;
}
static void ks_level_free(fhe_ks_plan::Level* lv) {
    if (!lv)
        return;
    for (auto* cv : lv->up)
        fhe_conv_destroy(cv);
    fhe_conv_destroy(lv->down);
    for (auto& kv : lv->downT)
        fhe_conv_destroy(kv.second);
    delete lv;
}
extern "C" void fhe_ks_plan_destroy(fhe_ks_plan* p) {
    if (!p)
        return;
    for (auto* lv : p->levels)
        ks_level_free(lv);
    for (void* q : p->owned)
        rt::dfree(q);
    for (auto& kv : p->bsgsTables)
        rt::dfree(kv.second);
    delete p;
}
extern "C" uint32_t fhe_ks_plan_alpha(const fhe_ks_plan* p) { return p ? p->alpha : 0; }

static fhe_status ks_level(fhe_ks_plan* p, uint32_t sizeQl, fhe_ks_plan::Level** out) {
    ARG_CHECK(sizeQl >= 1 && sizeQl <= p->sizeQ, "fhe_keyswitch: sizeQl out of range");
    std::lock_guard<std::mutex> lock(p->cacheMutex);
    if (p->levels[sizeQl]) {
        *out = p->levels[sizeQl];
        return FHE_OK;
    }
    fhe_ctx* c = p->ctx;
    auto* lv   = new fhe_ks_plan::Level;
    lv->sizeQl = sizeQl;
    // keyswitch-hybrid.cpp:329-333
    lv->numParts = std::min((sizeQl + p->alpha - 1) / p->alpha, p->numPartQ);
    for (uint32_t part = 0; part < lv->numParts; ++part) {
        const uint32_t start = p->alpha * part;
        const uint32_t sz    = (part == lv->numParts - 1) ? sizeQl - start : p->alpha;  // :341-352
        std::vector<uint64_t> src, dst;
        std::vector<uint32_t> cidx;
        for (uint32_t i = 0; i < sz; ++i)
            src.push_back(c->q[start + i]);
        // complementary basis (rns-cryptoparameters.cpp:253-283): Q_l limbs outside the digit, then P
        for (uint32_t i = 0; i < sizeQl; ++i)
            if (i < start || i >= start + sz)
                cidx.push_back(i);
        for (uint32_t j = 0; j < p->sizeP; ++j)
            cidx.push_back(p->sizeQ + j);
        for (uint32_t i : cidx)
            dst.push_back(c->q[i]);
        fhe_conv* cv = nullptr;
        if (fhe_status s = conv_build(c, src, dst, &cv)) {
            ks_level_free(lv);
            return s;
        }
        lv->up.push_back(cv);
        lv->cidx.push_back(cidx);
        lv->partSize.push_back(sz);
    }
    {
        std::vector<uint64_t> src, dst;
        for (uint32_t j = 0; j < p->sizeP; ++j)
            src.push_back(c->q[p->sizeQ + j]);
        for (uint32_t i = 0; i < sizeQl; ++i)
            dst.push_back(c->q[i]);
        if (fhe_status s = conv_build(c, src, dst, &lv->down)) {
            ks_level_free(lv);
            return s;
        }
        // [P^-1]_{q_i}  (rns-cryptoparameters.cpp:205-212)
        std::vector<TwPair> pinv(sizeQl);
        for (uint32_t i = 0; i < sizeQl; ++i) {
            const uint64_t v = host::invmod(host::prod_mod(src, -1, c->q[i]), c->q[i]);
            pinv[i]          = TwPair{v, host::shoup(v, c->q[i])};
        }
        void* d       = nullptr;
        const char* e = rt::dmalloc(&d, pinv.size() * sizeof(TwPair));
        if (!e) {
            p->owned.push_back(d);
            e = rt::h2d(d, pinv.data(), pinv.size() * sizeof(TwPair), nullptr);
        }
        if (!e)
            e = rt::sync(nullptr);
        if (e) {
            ks_level_free(lv);
            return fail(FHE_ERR_DEVICE, std::string("fhe_keyswitch: building the level tables: ") + e);
        }
        lv->d_PInv = (TwPair*)d;
    }
    p->levels[sizeQl] = lv;
    *out              = lv;
    return FHE_OK;
}

extern "C" fhe_status fhe_ks_key_alloc(fhe_ks_plan* p, fhe_ks_key** out) {
    ARG_CHECK(p && out, "fhe_ks_key_alloc: null argument");
    RT_CHECK(rt::set_device(p->ctx->device));
    fhe_ks_key* k = new fhe_ks_key;
    k->plan       = p;
    k->words      = (size_t)p->numPartQ * (p->sizeQ + p->sizeP) << p->ctx->logN;
    if (rt::dmalloc((void**)&k->d_b, k->words * 8) || rt::dmalloc((void**)&k->d_a, k->words * 8)) {
        fhe_ks_key_destroy(k);
        return fail(FHE_ERR_ALLOC, "fhe_ks_key_alloc: device allocation failed");
    }
    *out = k;
    return FHE_OK;
}
extern "C" fhe_status fhe_ks_key_upload(fhe_ks_plan* p, const uint64_t* keyB, const uint64_t* keyA, fhe_ks_key** out) {
    ARG_CHECK(keyB && keyA, "fhe_ks_key_upload: null key");
    fhe_ks_key* k = nullptr;
    if (fhe_status s = fhe_ks_key_alloc(p, &k))
        return s;
    RT_CHECK(rt::h2d(k->d_b, keyB, k->words * 8, nullptr));
    RT_CHECK(rt::h2d(k->d_a, keyA, k->words * 8, nullptr));
    RT_CHECK(rt::sync(nullptr));
    *out = k;
    return FHE_OK;
}
extern "C" fhe_status fhe_ks_key_wrap(fhe_ks_plan* p, uint64_t* devB, uint64_t* devA, fhe_ks_key** out) {
    ARG_CHECK(p && devB && devA && out, "fhe_ks_key_wrap: null argument");
    fhe_ks_key* k = new fhe_ks_key;
    k->plan       = p;
    k->words      = (size_t)p->numPartQ * (p->sizeQ + p->sizeP) << p->ctx->logN;
    k->d_b        = devB;
    k->d_a        = devA;
    k->owned      = false;
    *out          = k;
    return FHE_OK;
}
extern "C" void fhe_ks_key_destroy(fhe_ks_key* k) {
    if (!k)
        return;
    if (k->owned && k->d_b)
        rt::dfree(k->d_b);
    if (k->owned && k->d_a)
        rt::dfree(k->d_a);
    delete k;
}
extern "C" uint64_t* fhe_ks_key_devptr(fhe_ks_key* k, int which) { return k ? (which ? k->d_a : k->d_b) : nullptr; }
extern "C" size_t fhe_ks_key_words(const fhe_ks_key* k) { return k ? k->words : 0; }

// workspace layout (words), all sized for `batch` towers:
//   coef   [batch][sizeQl][N]              INTT of the input
//   dig_j  [batch][nc_j][N]  j < numParts  ModUp complements
//   e0,e1  [batch][sizeQl+sizeP][N]        inner products
//   md     [2*batch][sizeQl][N]            ModDown conversion output  (also used as [2*batch][sizeP][N] scratch: pcoef)
//   pcoef  [2*batch][sizeP][N]
//   d2     [batch][sizeQl][N], k0,k1 [batch][sizeQl][N]   (eval_mult only)
struct KsLayout {
    size_t coef = 0, e0 = 0, e1 = 0, md = 0, pcoef = 0, d2 = 0, k0 = 0, k1 = 0, total = 0;
    std::vector<size_t> dig;
};
static KsLayout ks_layout(const fhe_ks_plan* p, uint32_t sizeQl, uint32_t batch) {
    KsLayout w;
    const size_t N = (size_t)1 << p->ctx->logN;
    const uint32_t numParts = std::min((sizeQl + p->alpha - 1) / p->alpha, p->numPartQ);
    w.dig.assign(numParts, 0);
    size_t off = 0;
    w.coef = off, off += (size_t)batch * sizeQl * N;
    for (uint32_t j = 0; j < numParts; ++j) {
        const uint32_t start = p->alpha * j;
        const uint32_t sz    = (j == numParts - 1) ? sizeQl - start : p->alpha;
        w.dig[j] = off, off += (size_t)batch * (sizeQl - sz + p->sizeP) * N;
    }
    w.e0 = off, off += (size_t)batch * (sizeQl + p->sizeP) * N;
    w.e1 = off, off += (size_t)batch * (sizeQl + p->sizeP) * N;
    w.md = off, off += (size_t)2 * batch * sizeQl * N;
    w.pcoef = off, off += (size_t)2 * batch * p->sizeP * N;
    w.d2 = off, off += (size_t)batch * sizeQl * N;
    w.k0 = off, off += (size_t)batch * sizeQl * N;
    w.k1 = off, off += (size_t)batch * sizeQl * N;
    w.total = off;
    return w;
}
extern "C" size_t fhe_ks_workspace_bytes(const fhe_ks_plan* p, uint32_t sizeQl, uint32_t batch) {
    if (!p || sizeQl < 1 || sizeQl > p->sizeQ || batch < 1)
        return 0;
    return ks_layout(p, sizeQl, batch).total * 8;
}

// ApproxModDown (dcrtpoly-impl.h:966-1005) in two parts so that the two accumulators of a key switch share the
// INTT / conversion / NTT launches:
//   mod_down_core: x[nTow][sizeQl+sizeP][N] -> md[nTow][sizeQl][N] = NTT(ApproxSwitchCRTBasis(INTT(P part)))
//   mod_down_tail: out_i = (x_i - md_i) * [P^-1]_{q_i}     (or out_i += ... when `accumulate`)
static fhe_status mod_down_core(fhe_ks_plan* p, fhe_ks_plan::Level* lv, const uint64_t* x, uint32_t nTow, uint64_t* pcoef,
                                uint64_t* md, void* st, fhe_conv* down = nullptr, const NttEpilogue* epi = nullptr) {
    fhe_ctx* c            = p->ctx;
    const uint32_t sizeQl = lv->sizeQl, sizeP = p->sizeP, sizeQlP = sizeQl + sizeP;
    std::vector<uint32_t> pIdx(sizeP);
    for (uint32_t j = 0; j < sizeP; ++j)
        pIdx[j] = p->sizeQ + j;
    // P part to COEFFICIENT (:978-985): INTT of rows [sizeQl, sizeQl+sizeP) of every tower, written densely
    if (fhe_status s = ntt_run(c, true, x, pcoef, pIdx.data(), sizeP, nTow, st, sizeQlP, sizeQl))
        return s;
    // P -> Q_l (:987-988)
    if (fhe_status s = fhe_approx_switch_basis(down ? down : lv->down, pcoef, sizeP, 0, md, sizeQl, 0, nTow, st))
        return s;
    // back to EVALUATION (:1001); with an epilogue the last pass also applies (:1002) and stores the final result
    return ntt_run(c, false, md, md, nullptr, sizeQl, nTow, st, 0, 0, 0, 0, epi);
}
static fhe_status mod_down_tail(fhe_ks_plan* p, fhe_ks_plan::Level* lv, const uint64_t* x, const uint64_t* md, uint32_t nTow,
                                uint64_t* out, bool accumulate, void* st) {
    const uint32_t sizeQl = lv->sizeQl, sizeQlP = sizeQl + p->sizeP;
    // (:1002); x towers are sizeQlP rows apart
    if (accumulate)
        return elem_run<OP_SUB_MUL_CONST_ACC>(p->ctx, out, x, md, lv->d_PInv, nullptr, sizeQl, nTow, st, "fhe_approx_mod_down",
                                              sizeQlP, 0);
    return elem_run<OP_SUB_MUL_CONST>(p->ctx, out, x, md, lv->d_PInv, nullptr, sizeQl, nTow, st, "fhe_approx_mod_down", sizeQlP, 0);
}
static fhe_status mod_down_run(fhe_ks_plan* p, fhe_ks_plan::Level* lv, const uint64_t* x, uint32_t nTow, uint64_t* out,
                               uint64_t* pcoef, uint64_t* md, void* st) {
    if (fhe_status s = mod_down_core(p, lv, x, nTow, pcoef, md, st))
        return s;
    return mod_down_tail(p, lv, x, md, nTow, out, false, st);
}

// EvalKeySwitchPrecomputeCore (keyswitch-hybrid.cpp:314-379): digit decomposition + ModUp of every digit into the
// workspace's digit buffers (the digit's own limbs are NOT copied: the inner product reads them from `cin`)
static fhe_status ks_precompute_run(fhe_ks_plan* p, fhe_ks_plan::Level* lv, const uint64_t* cin, uint32_t batch,
                                    uint64_t* ws, const KsLayout& w, void* st) {
    fhe_ctx* c            = p->ctx;
    const uint32_t sizeQl = lv->sizeQl;
    if (fhe_status s = fhe_ntt_inv_oop(c, cin, ws + w.coef, nullptr, sizeQl, batch, st))
        return s;
    for (uint32_t j = 0; j < lv->numParts; ++j) {
        const uint32_t nc = (uint32_t)lv->cidx[j].size();
        uint64_t* dj      = ws + w.dig[j];
        if (fhe_status s = fhe_approx_switch_basis(lv->up[j], ws + w.coef, sizeQl, p->alpha * j, dj, nc, 0, batch, st))
            return s;
        // the digits are only read by the inner product, whose 128-bit accumulation takes any 64-bit operand: the
        // transform's outputs stay in the lazy range (< 16q), the 4-level canonicalisation is skipped
        if (fhe_status s = ntt_run(c, false, dj, dj, lv->cidx[j].data(), nc, batch, st, 0, 0, 0, 0, nullptr, false))
            return s;
    }
    return FHE_OK;
}
// EvalFastKeySwitchCore (keyswitch-hybrid.cpp:381-435) on digits already in the workspace.
// accumulate: out0/out1 += result (EvalMult's `cv[0] += ab[0]; cv[1] += ab[1]`, base-leveledshe.cpp:210-211)
// EvalFastKeySwitchCoreExt (keyswitch-hybrid.cpp:402-435) of ONE digit decomposition with nKeys keys, at most 8 digits: one launch per 16
// keys, every digit residue read once per launch; first != null: e0_t's Q_l rows += first * dP[i] (EvalFastRotationExt's addFirst,
// ckksrns-leveledshe.cpp:561-570) in the same store
static fhe_status ks_inner_multi_launch(fhe_ks_plan* p, fhe_ks_plan::Level* lv, const fhe_ks_key* const* keys, uint32_t nKeys,
                                        const uint64_t* cin, uint32_t batch, uint64_t* const* e0, uint64_t* const* e1,
                                        const uint64_t* first, const TwPair* dP, uint64_t* ws, const KsLayout& w, void* st,
                                        uint32_t towOff = 0) {
    fhe_ctx* c            = p->ctx;
    const uint32_t sizeQl = lv->sizeQl, sizeP = p->sizeP, sizeQlP = sizeQl + sizeP;
    const uint32_t tilesPerRow = c->N >= (uint32_t)kTile ? (c->N >> kTileLog) : 1u;
    for (uint32_t t0 = 0; t0 < nKeys; t0 += (uint32_t)kMaxMultiKeys) {
        KsInnerMultiArgs g;
        const uint32_t nt = std::min<uint32_t>(kMaxMultiKeys, nKeys - t0);
        for (uint32_t j = 0; j < (uint32_t)kMaxDigits; ++j) {
            g.nc[j]     = j < lv->numParts ? (uint32_t)lv->cidx[j].size() : 0u;
            g.digits[j] = j < lv->numParts ? ws + w.dig[j] + (((size_t)towOff * g.nc[j]) << c->logN) : nullptr;
        }
        for (uint32_t t = 0; t < (uint32_t)kMaxMultiKeys; ++t) {
            const uint32_t tt = t < nt ? t0 + t : t0;
            g.keyB[t] = keys[tt]->d_b, g.keyA[t] = keys[tt]->d_a, g.out0[t] = e0[tt], g.out1[t] = e1[tt];
        }
        g.c = cin + (((size_t)towOff * sizeQl) << c->logN), g.first = first, g.firstC = dP;
        g.lc = c->d_lc, g.mu128 = c->d_mu128, g.red = c->d_red;
        g.logN = c->logN, g.batch = batch, g.sizeQl = sizeQl, g.sizeQ = p->sizeQ, g.sizeP = sizeP;
        g.numDigits = lv->numParts, g.alpha = p->alpha, g.nKeys = nt;
        const uint64_t grid = (((uint64_t)tilesPerRow * sizeQlP + 7) / 8) * 8 * batch;
        // three digits (the usual dnum): two adjacent coefficients per lane, 16-byte accesses — round 6: 1341 -> 1090 us per launch in the
        // lockstep bootstrap, EvalMult composite +2-3 % (profiles/r06_sweeps.md 9).  FHE_KSM selects the other forms for A/B runs:
        //   0 one coefficient per lane, next key prefetched (rounds 4-5) | 1 two per lane, prefetched (default) | 2 two per lane, no prefetch |
        //   3 one per lane, no prefetch
        static const uint32_t ksm = env_u32("FHE_KSM", 1);
        uintptr_t bits = (uintptr_t)g.c | (uintptr_t)g.first;  // (16-byte accesses need 16-byte aligned towers: any device allocation is)
        for (uint32_t j = 0; j < lv->numParts && j < (uint32_t)kMaxDigits; ++j)
            bits |= (uintptr_t)g.digits[j];
        for (uint32_t t = 0; t < nt; ++t)
            bits |= (uintptr_t)g.keyB[t] | (uintptr_t)g.keyA[t] | (uintptr_t)g.out0[t] | (uintptr_t)g.out1[t];
        const bool wide = (bits & 15u) == 0 && c->N >= 2;
        if (lv->numParts <= 3 && ksm == 1 && wide)
            FHE_LAUNCH((ks_inner_multi_kernel<3, 2, true>), grid, st, g);
        else if (lv->numParts <= 3 && ksm == 2 && wide)
            FHE_LAUNCH((ks_inner_multi_kernel<3, 2, false>), grid, st, g);
        else if (lv->numParts <= 3 && ksm == 3)
            FHE_LAUNCH((ks_inner_multi_kernel<3, 1, false>), grid, st, g);
        else if (lv->numParts <= 3)
            FHE_LAUNCH((ks_inner_multi_kernel<3>), grid, st, g);
        else if (lv->numParts <= 4)
            FHE_LAUNCH((ks_inner_multi_kernel<4>), grid, st, g);
        else
            FHE_LAUNCH((ks_inner_multi_kernel<8>), grid, st, g);
        LAUNCH_CHECK();
    }
    return FHE_OK;
}
// EvalFastKeySwitchCoreExt (keyswitch-hybrid.cpp:402-435): inner product of the digits in the workspace with the key, both
// halves, result [batch][sizeQl+sizeP][N] in the extended basis
// towOff: the `batch` towers start at tower towOff of `cin` and of every digit buffer (a slice of a larger precompute)
static fhe_status ks_inner_run(fhe_ks_plan* p, fhe_ks_plan::Level* lv, const fhe_ks_key* key, const uint64_t* cin,
                               uint32_t batch, uint64_t* e0, uint64_t* e1, uint64_t* ws, const KsLayout& w, void* st,
                               uint32_t towOff = 0) {
    fhe_ctx* c            = p->ctx;
    const uint32_t sizeQl = lv->sizeQl, sizeP = p->sizeP, sizeQlP = sizeQl + sizeP;
    const uint32_t tilesPerRow = c->N >= (uint32_t)kTile ? (c->N >> kTileLog) : 1u;
    // at most 8 digits: the kernel that keeps the digits' residues in registers and issues every load of a coefficient up front (EvalMult
    // composite 8440 -> 8611 op/s against the chunked kernel below, session i); more digits: chunks of 8 whose exact sums add up
    if (lv->numParts <= (uint32_t)kMaxDigits)
        return ks_inner_multi_launch(p, lv, &key, 1, cin, batch, &e0, &e1, nullptr, nullptr, ws, w, st, towOff);
    for (uint32_t j0 = 0; j0 < lv->numParts; j0 += (uint32_t)kMaxDigits) {  // (one launch up to 8 digits)
        KsInnerArgs g;
        const uint32_t nd = std::min<uint32_t>(kMaxDigits, lv->numParts - j0);
        for (uint32_t j = 0; j < (uint32_t)kMaxDigits; ++j) {
            g.nc[j]     = j < nd ? (uint32_t)lv->cidx[j0 + j].size() : 0u;
            g.digits[j] = j < nd ? ws + w.dig[j0 + j] + (((size_t)towOff * g.nc[j]) << c->logN) : nullptr;
        }
        g.c = cin + (((size_t)towOff * sizeQl) << c->logN), g.keyB = key->d_b, g.keyA = key->d_a;
        g.out0 = e0, g.out1 = e1;
        g.lc = c->d_lc, g.mu128 = c->d_mu128, g.red = c->d_red;
        g.logN = c->logN, g.batch = batch, g.sizeQl = sizeQl, g.sizeQ = p->sizeQ, g.sizeP = sizeP;
        g.numDigits = nd, g.alpha = p->alpha, g.j0 = j0, g.acc = j0 > 0;
        FHE_LAUNCH(ks_inner_product_kernel, (((uint64_t)tilesPerRow * sizeQlP + 7) / 8) * 8 * batch, st, g);
        LAUNCH_CHECK();
    }
    return FHE_OK;
}
// The same for SEVERAL keys on one digit decomposition (the baby-step rotations of a BSGS transform), optionally with addFirst.  More
// than 8 digits: key by key through ks_inner_run + the element-wise pass.
static fhe_status ks_inner_multi_run(fhe_ks_plan* p, fhe_ks_plan::Level* lv, const fhe_ks_key* const* keys, uint32_t nKeys,
                                     const uint64_t* cin, uint32_t batch, uint64_t* const* e0, uint64_t* const* e1,
                                     const uint64_t* first, const TwPair* dP, uint64_t* ws, const KsLayout& w, void* st) {
    if (lv->numParts <= (uint32_t)kMaxDigits)
        return ks_inner_multi_launch(p, lv, keys, nKeys, cin, batch, e0, e1, first, dP, ws, w, st);
    const uint32_t sizeQl = lv->sizeQl, sizeQlP = sizeQl + p->sizeP;
    for (uint32_t t = 0; t < nKeys; ++t) {
        if (fhe_status s = ks_inner_run(p, lv, keys[t], cin, batch, e0[t], e1[t], ws, w, st))
            return s;
        if (first)
            if (fhe_status s = elem_run<OP_MUL_CONST_ADD>(p->ctx, e0[t], first, e0[t], dP, nullptr, sizeQl, batch, st,
                                                          "fhe_eval_fast_rotation_ext", 0, 0, sizeQlP, 0, sizeQlP, 0))
                return s;
    }
    return FHE_OK;
}
// the HYBRID inner product (keyswitch-hybrid.cpp:419-430) over towers in separate allocations: see inner_rows_kernel
extern "C" fhe_status fhe_inner_product(fhe_ctx* c, uint32_t nTerms, const uint64_t* const* x, const uint64_t* const* k0,
                                        const uint64_t* const* k1, const uint32_t* keyRow, const uint32_t* limbIdx,
                                        uint32_t rows, uint32_t batch, uint64_t* out0, uint64_t* out1, void* st) {
    ARG_CHECK(c && x && k0 && out0, "fhe_inner_product: null argument");
    ARG_CHECK((k1 != nullptr) == (out1 != nullptr), "fhe_inner_product: the second key and the second output go together");
    ARG_CHECK(batch >= 1, "fhe_inner_product: batch must be >= 1");
    ARG_CHECK(nTerms >= 1, "fhe_inner_product: no terms");
    InnerRowsArgs g;
    if (fhe_status s = make_sel(c, limbIdx, rows, &g.sel, "fhe_inner_product"))
        return s;
    for (uint32_t i = 0; i < (uint32_t)kMaxLimbs; ++i) {
        const uint32_t kr = i < rows ? (keyRow ? keyRow[i] : i) : 0u;
        ARG_CHECK(kr < 256u, "fhe_inner_product: key row out of range");
        g.keyRow[i] = (uint8_t)kr;
    }
    for (uint32_t j = 0; j < nTerms; ++j)
        ARG_CHECK(x[j] && k0[j] && (!k1 || k1[j]), "fhe_inner_product: null term");
    RT_CHECK(rt::set_device(c->device));
    g.out0 = out0, g.out1 = out1, g.lc = c->d_lc, g.mu128 = c->d_mu128;
    g.logN = c->logN, g.batch = batch, g.rows = rows;
    const uint32_t tilesPerRow = c->N >= (uint32_t)kTile ? (c->N >> kTileLog) : 1u;
    for (uint32_t t0 = 0; t0 < nTerms; t0 += (uint32_t)kMaxDigits) {  // (one launch up to 8 terms; exact sums, so chunks add up)
        const uint32_t nt = std::min<uint32_t>(kMaxDigits, nTerms - t0);
        for (uint32_t j = 0; j < (uint32_t)kMaxDigits; ++j) {
            g.x[j]  = j < nt ? x[t0 + j] : nullptr;
            g.k0[j] = j < nt ? k0[t0 + j] : nullptr;
            g.k1[j] = (j < nt && k1) ? k1[t0 + j] : nullptr;
        }
        g.nTerms = nt, g.acc = t0 > 0;
        FHE_LAUNCH(inner_rows_kernel, (uint64_t)tilesPerRow * rows * batch, st, g);
        LAUNCH_CHECK();
    }
    return FHE_OK;
}
static fhe_status ks_fast_run(fhe_ks_plan* p, fhe_ks_plan::Level* lv, const fhe_ks_key* key, const uint64_t* cin,
                              uint32_t batch, uint64_t* out0, uint64_t* out1, uint64_t* ws, const KsLayout& w, void* st,
                              bool accumulate) {
    fhe_ctx* c            = p->ctx;
    const uint32_t sizeQl = lv->sizeQl;
    if (fhe_status s = ks_inner_run(p, lv, key, cin, batch, ws + w.e0, ws + w.e1, ws, w, st))
        return s;
    // 2 x ApproxModDown (:381-400): e0 and e1 are adjacent in the workspace (w.e1 == w.e0 + batch*sizeQlP*N), so the
    // INTT / conversion / NTT run once over 2*batch towers; only the element-wise tails are per accumulator
    if (ntt_epilogue_supported(c)) {
        // (x_i - switched_i) * [P^-1]_{q_i} (+= into out for EvalMult) applied by the last NTT pass itself: the switched
        // tower is never written to HBM in EVALUATION form and the two tail kernels disappear
        NttEpilogue epi;
        epi.mode = accumulate ? 2u : 1u, epi.split = batch, epi.aStride = sizeQl + p->sizeP, epi.aFirst = 0;
        epi.A = ws + w.e0, epi.C = lv->d_PInv, epi.out0 = out0, epi.out1 = out1;
        return mod_down_core(p, lv, ws + w.e0, 2 * batch, ws + w.pcoef, ws + w.md, st, nullptr, &epi);
    }
    if (fhe_status s = mod_down_core(p, lv, ws + w.e0, 2 * batch, ws + w.pcoef, ws + w.md, st))
        return s;
    if (fhe_status s = mod_down_tail(p, lv, ws + w.e0, ws + w.md, batch, out0, accumulate, st))
        return s;
    return mod_down_tail(p, lv, ws + w.e1, ws + w.md + ((size_t)batch * sizeQl << c->logN), batch, out1, accumulate, st);
}
static fhe_status keyswitch_run(fhe_ks_plan* p, const fhe_ks_key* key, const uint64_t* cin, uint32_t sizeQl,
                                uint32_t batch, uint64_t* out0, uint64_t* out1, uint64_t* ws, const KsLayout& w,
                                void* st, bool accumulate = false) {
    fhe_ks_plan::Level* lv = nullptr;
    if (fhe_status s = ks_level(p, sizeQl, &lv))
        return s;
    if (fhe_status s = ks_precompute_run(p, lv, cin, batch, ws, w, st))
        return s;
    return ks_fast_run(p, lv, key, cin, batch, out0, out1, ws, w, st, accumulate);
}

extern "C" fhe_status fhe_keyswitch_hybrid(fhe_ks_plan* p, const fhe_ks_key* key, const uint64_t* cin, uint32_t sizeQl,
                                           uint32_t batch, uint64_t* out0, uint64_t* out1, void* ws, size_t wsBytes,
                                           void* st) {
    ARG_CHECK(p && key && cin && out0 && out1 && ws, "fhe_keyswitch_hybrid: null argument");
    ARG_CHECK(key->plan == p, "fhe_keyswitch_hybrid: key belongs to another plan");
    ARG_CHECK(sizeQl >= 1 && sizeQl <= p->sizeQ && batch >= 1, "fhe_keyswitch_hybrid: bad level or batch");
    const KsLayout w = ks_layout(p, sizeQl, batch);
    ARG_CHECK(wsBytes >= w.total * 8, "fhe_keyswitch_hybrid: workspace too small");
    RT_CHECK(rt::set_device(p->ctx->device));
    return keyswitch_run(p, key, cin, sizeQl, batch, out0, out1, (uint64_t*)ws, w, st);
}

// the same with `acc0 += ks0(c); acc1 += ks1(c)`: the tail of LeveledSHEBase::EvalMult(ct, ct, key) (base-leveledshe.cpp:207-211:
// KeySwitchCore on the third element, `cv[0] += ab[0]; cv[1] += ab[1]`), the additions fused into the ModDown epilogue
extern "C" fhe_status fhe_keyswitch_hybrid_acc(fhe_ks_plan* p, const fhe_ks_key* key, const uint64_t* cin, uint32_t sizeQl,
                                               uint32_t batch, uint64_t* acc0, uint64_t* acc1, void* ws, size_t wsBytes, void* st) {
    ARG_CHECK(p && key && cin && acc0 && acc1 && ws, "fhe_keyswitch_hybrid_acc: null argument");
    ARG_CHECK(key->plan == p, "fhe_keyswitch_hybrid_acc: key belongs to another plan");
    ARG_CHECK(sizeQl >= 1 && sizeQl <= p->sizeQ && batch >= 1, "fhe_keyswitch_hybrid_acc: bad level or batch");
    const KsLayout w = ks_layout(p, sizeQl, batch);
    ARG_CHECK(wsBytes >= w.total * 8, "fhe_keyswitch_hybrid_acc: workspace too small");
    RT_CHECK(rt::set_device(p->ctx->device));
    return keyswitch_run(p, key, cin, sizeQl, batch, acc0, acc1, (uint64_t*)ws, w, st, true);
}

extern "C" fhe_status fhe_ckks_eval_mult(fhe_ks_plan* p, const fhe_ks_key* key, const uint64_t* a0, const uint64_t* a1,
                                         const uint64_t* b0, const uint64_t* b1, uint32_t sizeQl, uint32_t batch,
                                         uint64_t* c0, uint64_t* c1, void* wsv, size_t wsBytes, void* st) {
    ARG_CHECK(p && key && a0 && a1 && b0 && b1 && c0 && c1 && wsv, "fhe_ckks_eval_mult: null argument");
    ARG_CHECK(key->plan == p, "fhe_ckks_eval_mult: key belongs to another plan");
    ARG_CHECK(sizeQl >= 1 && sizeQl <= p->sizeQ && batch >= 1, "fhe_ckks_eval_mult: bad level or batch");
    const KsLayout w = ks_layout(p, sizeQl, batch);
    ARG_CHECK(wsBytes >= w.total * 8, "fhe_ckks_eval_mult: workspace too small");
    fhe_ctx* c   = p->ctx;
    uint64_t* ws = (uint64_t*)wsv;
    RT_CHECK(rt::set_device(c->device));
    // EvalMultCore (base-leveledshe.cpp:607-644)
    if (fhe_status s = fhe_tensor(c, a0, a1, b0, b1, c0, c1, ws + w.d2, nullptr, sizeQl, batch, st))
        return s;
    // KeySwitchCore on d2 (:207) with `cv[0] += ab[0]; cv[1] += ab[1]` (:210-211) fused into the ModDown tails
    return keyswitch_run(p, key, ws + w.d2, sizeQl, batch, c0, c1, ws, w, st, true);
}

#define KS_COMMON_CHECKS(who)                                                                              \
    ARG_CHECK(p && ws, who ": null argument");                                                             \
    ARG_CHECK(sizeQl >= 1 && sizeQl <= p->sizeQ && batch >= 1, who ": bad level or batch");                \
    const KsLayout w = ks_layout(p, sizeQl, batch);                                                        \
    ARG_CHECK(wsBytes >= w.total * 8, who ": workspace too small");                                        \
    RT_CHECK(rt::set_device(p->ctx->device));                                                              \
    fhe_ks_plan::Level* lv = nullptr;                                                                      \
    if (fhe_status s_ = ks_level(p, sizeQl, &lv))                                                          \
        return s_;

extern "C" fhe_status fhe_ks_precompute(fhe_ks_plan* p, const uint64_t* c1, uint32_t sizeQl, uint32_t batch, void* ws,
                                        size_t wsBytes, void* st) {
    KS_COMMON_CHECKS("fhe_ks_precompute")
    ARG_CHECK(c1, "fhe_ks_precompute: null argument");
    return ks_precompute_run(p, lv, c1, batch, (uint64_t*)ws, w, st);
}
extern "C" fhe_status fhe_ks_fast_keyswitch(fhe_ks_plan* p, const fhe_ks_key* key, const uint64_t* c1, uint32_t sizeQl,
                                            uint32_t batch, uint64_t* out0, uint64_t* out1, void* ws, size_t wsBytes,
                                            void* st) {
    KS_COMMON_CHECKS("fhe_ks_fast_keyswitch")
    ARG_CHECK(key && c1 && out0 && out1 && key->plan == p, "fhe_ks_fast_keyswitch: bad key or null argument");
    return ks_fast_run(p, lv, key, c1, batch, out0, out1, (uint64_t*)ws, w, st, false);
}
// LeveledSHEBase::EvalFastRotation (base-leveledshe.cpp:432-463): ba = EvalFastKeySwitchCore(digits, key_k);
// ba[0] += cv[0]; both elements through AutomorphismTransform(k)
extern "C" fhe_status fhe_eval_fast_rotation(fhe_ks_plan* p, const fhe_ks_key* key, const uint64_t* c0, const uint64_t* c1,
                                             uint32_t k, uint32_t sizeQl, uint32_t batch, uint64_t* out0, uint64_t* out1,
                                             void* ws, size_t wsBytes, void* st) {
    KS_COMMON_CHECKS("fhe_eval_fast_rotation")
    ARG_CHECK(key && c0 && c1 && out0 && out1 && key->plan == p, "fhe_eval_fast_rotation: bad key or null argument");
    ARG_CHECK(k % 2 == 1, "Automorphism index not odd");
    uint64_t* wsp = (uint64_t*)ws;
    fhe_ctx* c    = p->ctx;
    if (fhe_status s = ks_fast_run(p, lv, key, c1, batch, wsp + w.k0, wsp + w.k1, wsp, w, st, false))
        return s;
    if (fhe_status s = fhe_add(c, wsp + w.k0, wsp + w.k0, c0, nullptr, sizeQl, batch, st))
        return s;
    if (fhe_status s = fhe_automorph(c, out0, wsp + w.k0, k, 1, nullptr, sizeQl, batch, st))
        return s;
    return fhe_automorph(c, out1, wsp + w.k1, k, 1, nullptr, sizeQl, batch, st);
}
// LeveledSHEBase::EvalAutomorphism (base-leveledshe.cpp:381-422) = KeySwitchInPlace + AutomorphismTransform on both
// elements; identical result to precompute + fast rotation
extern "C" fhe_status fhe_eval_automorphism(fhe_ks_plan* p, const fhe_ks_key* key, const uint64_t* c0, const uint64_t* c1,
                                            uint32_t k, uint32_t sizeQl, uint32_t batch, uint64_t* out0, uint64_t* out1,
                                            void* ws, size_t wsBytes, void* st) {
    if (fhe_status s = fhe_ks_precompute(p, c1, sizeQl, batch, ws, wsBytes, st))
        return s;
    return fhe_eval_fast_rotation(p, key, c0, c1, k, sizeQl, batch, out0, out1, ws, wsBytes, st);
}

// ---- double hoisting: work in the extended basis Q_l u P, one ModDown at the end (ckksrns-fhe.cpp:1830-2000) ----
static fhe_status ext_limbs(const fhe_ks_plan* p, uint32_t sizeQl, std::vector<uint32_t>& idx) {
    idx.resize(sizeQl + p->sizeP);
    for (uint32_t i = 0; i < sizeQl; ++i)
        idx[i] = i;
    for (uint32_t j = 0; j < p->sizeP; ++j)
        idx[sizeQl + j] = p->sizeQ + j;
    return FHE_OK;
}
// [P]_{q_i} as Shoup pairs for limbs [0, sizeQl)  (PModq, rns-cryptoparameters.cpp:200-203), cached per level
static fhe_status ks_pmodq(fhe_ks_plan* p, fhe_ks_plan::Level* lv, TwPair** d) {
    std::lock_guard<std::mutex> lock(p->cacheMutex);
    if (!lv->d_PModq) {
        fhe_ctx* c = p->ctx;
        std::vector<uint64_t> pm(p->sizeP);
        for (uint32_t j = 0; j < p->sizeP; ++j)
            pm[j] = c->q[p->sizeQ + j];
        std::vector<TwPair> h(lv->sizeQl);
        for (uint32_t i = 0; i < lv->sizeQl; ++i) {
            const uint64_t v = host::prod_mod(pm, -1, c->q[i]);
            h[i]             = TwPair{v, host::shoup(v, c->q[i])};
        }
        void* dp = nullptr;
        RT_CHECK(rt::dmalloc(&dp, h.size() * sizeof(TwPair)));
        p->owned.push_back(dp);
        RT_CHECK(rt::h2d(dp, h.data(), h.size() * sizeof(TwPair), nullptr));
        RT_CHECK(rt::sync(nullptr));
        lv->d_PModq = (TwPair*)dp;
    }
    *d = lv->d_PModq;
    return FHE_OK;
}
// KeySwitchHYBRID::KeySwitchExt for one element (keyswitch-hybrid.cpp:217-243): out [batch][sizeQl+sizeP][N], Q_l rows =
// c * [P]_{q_i}, P rows = 0
extern "C" fhe_status fhe_ks_ext(fhe_ks_plan* p, const uint64_t* cin, uint32_t sizeQl, uint32_t batch, uint64_t* out, void* st) {
    ARG_CHECK(p && cin && out, "fhe_ks_ext: null argument");
    ARG_CHECK(sizeQl >= 1 && sizeQl <= p->sizeQ && batch >= 1, "fhe_ks_ext: bad level or batch");
    fhe_ctx* c = p->ctx;
    RT_CHECK(rt::set_device(c->device));
    fhe_ks_plan::Level* lv = nullptr;
    if (fhe_status s = ks_level(p, sizeQl, &lv))
        return s;
    TwPair* dP = nullptr;
    if (fhe_status s = ks_pmodq(p, lv, &dP))
        return s;
    const uint32_t sizeQlP = sizeQl + p->sizeP;
    if (fhe_status s = zero_rows(c, out, sizeQlP, sizeQl, p->sizeP, batch, st))
        return s;
    return elem_run<OP_MUL_CONST>(c, out, cin, nullptr, dP, nullptr, sizeQl, batch, st, "fhe_ks_ext", 0, 0, 0, 0, sizeQlP, 0);
}
// EvalFastKeySwitchCoreExt on digits already in the workspace (fhe_ks_precompute): out0/out1 [batch][sizeQl+sizeP][N]
extern "C" fhe_status fhe_ks_fast_keyswitch_ext(fhe_ks_plan* p, const fhe_ks_key* key, const uint64_t* c1, uint32_t sizeQl,
                                                uint32_t batch, uint64_t* out0, uint64_t* out1, void* ws, size_t wsBytes,
                                                void* st) {
    KS_COMMON_CHECKS("fhe_ks_fast_keyswitch_ext")
    ARG_CHECK(key && c1 && out0 && out1 && key->plan == p, "fhe_ks_fast_keyswitch_ext: bad key or null argument");
    return ks_inner_run(p, lv, key, c1, batch, out0, out1, (uint64_t*)ws, w, st);
}
// LeveledSHECKKSRNS::EvalFastRotationExt (ckksrns-leveledshe.cpp:534-582): EvalFastKeySwitchCoreExt, optionally
// + c0 * [P]_{q_i} on the first element's Q_l limbs, then the automorphism on both extended elements
extern "C" fhe_status fhe_eval_fast_rotation_ext(fhe_ks_plan* p, const fhe_ks_key* key, const uint64_t* c0, const uint64_t* c1,
                                                 uint32_t k, int addFirst, uint32_t sizeQl, uint32_t batch, uint64_t* out0,
                                                 uint64_t* out1, void* ws, size_t wsBytes, void* st) {
    KS_COMMON_CHECKS("fhe_eval_fast_rotation_ext")
    ARG_CHECK(key && c0 && c1 && out0 && out1 && key->plan == p, "fhe_eval_fast_rotation_ext: bad key or null argument");
    ARG_CHECK(k % 2 == 1, "Automorphism index not odd");
    fhe_ctx* c    = p->ctx;
    uint64_t* wsp = (uint64_t*)ws;
    const uint32_t sizeQlP = sizeQl + p->sizeP;
    if (addFirst) {  // cTilda[0] += psiC0  (:561-570): e0 rows [0, sizeQl) += c0 * PModq, in the inner product's own store
        TwPair* dP = nullptr;
        if (fhe_status s = ks_pmodq(p, lv, &dP))
            return s;
        uint64_t *e0 = wsp + w.e0, *e1 = wsp + w.e1;
        if (fhe_status s = ks_inner_multi_run(p, lv, &key, 1, c1, batch, &e0, &e1, c0, dP, wsp, w, st))
            return s;
    }
    else if (fhe_status s = ks_inner_run(p, lv, key, c1, batch, wsp + w.e0, wsp + w.e1, wsp, w, st))
        return s;
    std::vector<uint32_t> idx;
    ext_limbs(p, sizeQl, idx);
    if (fhe_status s = fhe_automorph(c, out0, wsp + w.e0, k, 1, idx.data(), sizeQlP, batch, st))
        return s;
    return fhe_automorph(c, out1, wsp + w.e1, k, 1, idx.data(), sizeQlP, batch, st);
}
// KeySwitchHYBRID::KeySwitchDown (keyswitch-hybrid.cpp:245-278): ApproxModDown of both extended elements
extern "C" fhe_status fhe_ks_down(fhe_ks_plan* p, const uint64_t* x0, const uint64_t* x1, uint32_t sizeQl, uint32_t batch,
                                  uint64_t* out0, uint64_t* out1, void* ws, size_t wsBytes, void* st) {
    KS_COMMON_CHECKS("fhe_ks_down")
    ARG_CHECK(x0 && x1 && out0 && out1, "fhe_ks_down: null argument");
    uint64_t* wsp = (uint64_t*)ws;
    if (fhe_status s = mod_down_run(p, lv, x0, batch, out0, wsp + w.pcoef, wsp + w.md, st))
        return s;
    return mod_down_run(p, lv, x1, batch, out1, wsp + w.pcoef, wsp + w.md, st);
}

// ---- BSGS plaintext-matrix product with double hoisting ----
// FHECKKSRNS::EvalLinearTransform (ckksrns-fhe.cpp:1832-1882) and one level of EvalCoeffsToSlots / EvalSlotsToCoeffs
// (:1884-2198).  The reference walks the outer (giant) steps one after the other; here every stage runs ONCE over all
// outer steps (the accumulations `first += ...` and `outer += ...` are exact modular sums, so their order is free):
//   1. rot_j      inner rotations on one digit decomposition of c1, kept in the extended basis: the inner products with all
//                 rotation keys (+ c0 * P) in one launch, the digits read once; stored BEFORE the automorphism
//   2. inner_i    all outer steps' multiply-accumulate sums in one pass over the rot_j, each read through its rotation's
//                 index map (the automorphism is this kernel's gather)                               (1 kernel)
//   3. d_i        KeySwitchDown of every inner_i as one batch of 2*nOut*batch towers                (INTT, conversion, NTT)
//   4. first      sum_i Automorphism_{k_i}(d_i[0])                                                   (1 kernel)
//   5. digits     ModUp of every rotated step's d_i[1] as one batch                                  (INTT, conversions, NTTs)
//   6. e_i        inner product with the outer step's own key                                        (1 kernel per rotated step)
//   7. outer      sum_i Automorphism_{k_i}(e_i)  (+ inner_i[1] of the unrotated steps)               (2 kernels)
//   8. result     KeySwitchDown(outer), result[0] += first
// workspace = [key-switch layout for batch*nOut towers][rot: nIn x 2 x batch ext][inner: 2 x nOut x batch ext]
//             [d: 2 x nOut x batch Q_l][outer: 2 x batch ext][first: batch Q_l]
struct BsgsLayout {
    size_t ksTotal, rot, inner, d, outer, first, total;
};
static BsgsLayout bsgs_layout(const fhe_ks_plan* p, uint32_t sizeQl, uint32_t batch, uint32_t nIn, uint32_t nOut) {
    BsgsLayout b{};
    b.ksTotal        = ks_layout(p, sizeQl, batch * nOut).total;
    const size_t N   = (size_t)1 << p->ctx->logN;
    const size_t ext = (size_t)batch * (sizeQl + p->sizeP) * N, low = (size_t)batch * sizeQl * N;
    size_t off       = b.ksTotal;
    b.rot = off, off += (size_t)nIn * 2 * ext;
    b.inner = off, off += (size_t)nOut * 2 * ext;
    b.d = off, off += (size_t)nOut * 2 * low;
    b.outer = off, off += 2 * ext;
    b.first = off, off += low;
    b.total = off;
    return b;
}
extern "C" size_t fhe_ckks_bsgs_workspace_bytes(const fhe_ks_plan* p, uint32_t sizeQl, uint32_t batch, uint32_t nIn,
                                                uint32_t nOut) {
    if (!p || sizeQl < 1 || sizeQl > p->sizeQ || batch < 1 || nIn < 1 || nOut < 1)
        return 0;
    return bsgs_layout(p, sizeQl, batch, nIn, nOut).total * 8;
}
// KeySwitchDown of two adjacent extended towers x[2*batch] -> out0, out1 (keyswitch-hybrid.cpp:245-278); out1 == nullptr:
// all nTow towers go to out0 (one batch of ApproxModDown calls)
static fhe_status mod_down_many(fhe_ks_plan* p, fhe_ks_plan::Level* lv, const uint64_t* x, uint32_t nTow, uint32_t split,
                                uint64_t* out0, uint64_t* out1, uint64_t* pcoef, uint64_t* md, void* st) {
    fhe_ctx* c = p->ctx;
    if (ntt_epilogue_supported(c)) {
        NttEpilogue epi;
        epi.mode = 1u, epi.split = split, epi.aStride = lv->sizeQl + p->sizeP, epi.aFirst = 0;
        epi.A = x, epi.C = lv->d_PInv, epi.out0 = out0, epi.out1 = out1 ? out1 : out0;
        return mod_down_core(p, lv, x, nTow, pcoef, md, st, nullptr, &epi);
    }
    if (fhe_status s = mod_down_core(p, lv, x, nTow, pcoef, md, st))
        return s;
    if (fhe_status s = mod_down_tail(p, lv, x, md, split, out0, false, st))
        return s;
    if (split == nTow)
        return FHE_OK;
    const size_t ext = ((size_t)split * (lv->sizeQl + p->sizeP)) << c->logN, low = ((size_t)split * lv->sizeQl) << c->logN;
    return mod_down_tail(p, lv, x + ext, md + low, nTow - split, out1, false, st);
}
// out (+)= sum_s Automorphism_{k_s}(in_s) over towers of `nl` limbs
static fhe_status automorph_sum_run(fhe_ctx* c, uint64_t* out, const std::vector<const uint64_t*>& in, const std::vector<uint32_t>& k,
                                    const uint32_t* li, uint32_t nl, uint32_t bt, void* st) {
    for (size_t s0 = 0; s0 < in.size(); s0 += kMaxAutoSum) {
        AutoSumArgs g;
        if (fhe_status s = make_sel(c, li, nl, &g.sel, "automorphism"))
            return s;
        g.nSrc = (uint32_t)std::min<size_t>(kMaxAutoSum, in.size() - s0);
        for (uint32_t s = 0; s < (uint32_t)kMaxAutoSum; ++s) {
            g.in[s] = s < g.nSrc ? in[s0 + s] : nullptr;
            g.k[s]  = s < g.nSrc ? k[s0 + s] : 1u;
        }
        g.out = out, g.q = c->d_q, g.logN = c->logN, g.nLimbs = nl, g.rows = bt * nl, g.accumulate = s0 ? 1u : 0u;
        FHE_LAUNCH(automorph_sum_kernel, tiles_for(c, g.rows), st, g);
        LAUNCH_CHECK();
    }
    return FHE_OK;
}
extern "C" fhe_status fhe_ckks_bsgs_transform(fhe_ks_plan* p, const uint64_t* c0, const uint64_t* c1, uint32_t sizeQl,
                                              uint32_t batch, uint32_t nIn, const uint32_t* inK,
                                              const fhe_ks_key* const* inKeys, uint32_t nOut, const uint32_t* outK,
                                              const fhe_ks_key* const* outKeys, const uint64_t* const* diag, uint64_t* out0,
                                              uint64_t* out1, void* wsv, size_t wsBytes, void* st) {
    ARG_CHECK(p && c0 && c1 && inK && outK && diag && out0 && out1 && wsv, "fhe_ckks_bsgs_transform: null argument");
    ARG_CHECK(sizeQl >= 1 && sizeQl <= p->sizeQ && batch >= 1, "fhe_ckks_bsgs_transform: bad level or batch");
    ARG_CHECK(nIn >= 1 && nOut >= 1, "fhe_ckks_bsgs_transform: needs at least one inner and one outer step");
    const BsgsLayout L = bsgs_layout(p, sizeQl, batch, nIn, nOut);
    ARG_CHECK(wsBytes >= L.total * 8, "fhe_ckks_bsgs_transform: workspace too small");
    for (uint32_t j = 0; j < nIn; ++j) {
        ARG_CHECK(inK[j] == 0 || inK[j] % 2 == 1, "Automorphism index not odd");
        ARG_CHECK(inK[j] == 0 || (inKeys && inKeys[j] && inKeys[j]->plan == p), "fhe_ckks_bsgs_transform: missing inner rotation key");
    }
    for (uint32_t i = 0; i < nOut; ++i) {
        ARG_CHECK(outK[i] == 0 || outK[i] % 2 == 1, "Automorphism index not odd");
        ARG_CHECK(outK[i] == 0 || (outKeys && outKeys[i] && outKeys[i]->plan == p), "fhe_ckks_bsgs_transform: missing outer rotation key");
    }
    fhe_ctx* c = p->ctx;
    RT_CHECK(rt::set_device(c->device));
    fhe_ks_plan::Level* lv = nullptr;
    if (fhe_status s = ks_level(p, sizeQl, &lv))
        return s;
    uint64_t* ws           = (uint64_t*)wsv;
    const uint32_t sizeQlP = sizeQl + p->sizeP;
    const size_t ext = ((size_t)batch * sizeQlP) << c->logN, low = ((size_t)batch * sizeQl) << c->logN;
    uint64_t *rot = ws + L.rot, *inner = ws + L.inner, *outer = ws + L.outer, *d = ws + L.d, *first = ws + L.first;
    std::vector<uint32_t> extIdx;
    ext_limbs(p, sizeQl, extIdx);
    TwPair* dP = nullptr;
    if (fhe_status s = ks_pmodq(p, lv, &dP))
        return s;
    // outer steps without a rotation first, the rotated ones behind them (slices of one batch from stage 3 on)
    std::vector<uint32_t> perm;
    for (uint32_t i = 0; i < nOut; ++i)
        if (outK[i] == 0)
            perm.push_back(i);
    const uint32_t nZ = (uint32_t)perm.size(), nR = nOut - nZ;
    for (uint32_t i = 0; i < nOut; ++i)
        if (outK[i] != 0)
            perm.push_back(i);
    // the diagonals' pointer table on the device: outer steps permuted, rows padded to the kernel's unroll width, absent
    // terms and padding pointing at rows of zeros (cached by content: the same transform is applied many times)
    const uint32_t ninK   = nIn <= 4 ? 4u : nIn <= 8 ? 8u : (uint32_t)kMaxBsgsIn;  // kernel instance (NIN)
    const uint32_t nInPad = (nIn + ninK - 1) / ninK * ninK;
    std::unique_lock<std::mutex> bsgsLock(p->cacheMutex);
    if (!p->d_zeroRows) {
        void* dz = nullptr;
        const size_t bytes = ((size_t)(p->sizeQ + p->sizeP) << c->logN) * 8;
        RT_CHECK(rt::dmalloc(&dz, bytes));
        p->owned.push_back(dz);
        RT_CHECK(rt::dzero_2d(dz, bytes, bytes, 1, nullptr));
        RT_CHECK(rt::sync(nullptr));
        p->d_zeroRows = (uint64_t*)dz;
    }
    std::vector<uint64_t> tab((size_t)nOut * nInPad, (uint64_t)(uintptr_t)p->d_zeroRows);
    for (uint32_t i = 0; i < nOut; ++i)
        for (uint32_t j = 0; j < nIn; ++j)
            if (diag[(size_t)perm[i] * nIn + j])
                tab[(size_t)i * nInPad + j] = (uint64_t)(uintptr_t)diag[(size_t)perm[i] * nIn + j];
    auto it = p->bsgsTables.find(tab);
    if (it == p->bsgsTables.end()) {
        if (p->bsgsTables.size() >= 64) {  // a caller that keeps re-allocating its diagonals: drop the stale tables
            RT_CHECK(rt::device_sync());   // (a launch on ANY stream may still read one of them)
            for (auto& kv : p->bsgsTables)
                rt::dfree(kv.second);
            p->bsgsTables.clear();
        }
        void* dp = nullptr;
        RT_CHECK(rt::dmalloc(&dp, tab.size() * 8));
        RT_CHECK(rt::h2d(dp, tab.data(), tab.size() * 8, nullptr));
        RT_CHECK(rt::sync(nullptr));
        it = p->bsgsTables.emplace(tab, dp).first;
    }
    const uint64_t* const* dTab = (const uint64_t* const*)it->second;
    bsgsLock.unlock();

    // 1. inner (baby-step) rotations: one digit decomposition of c1 serves all of them (EvalFastRotationPrecompute, :1842), and ONE
    //    launch computes EvalFastKeySwitchCoreExt + `cTilda[0] += c0 * P` (EvalFastRotationExt(ct, index, digits, addFirst = true)) for all
    //    of them.  rot_j keeps the result BEFORE the rotation's AutomorphismTransform: stage 2 reads it through the index map
    const KsLayout wB = ks_layout(p, sizeQl, batch);
    {
        std::vector<const fhe_ks_key*> keys;
        std::vector<uint64_t*> e0, e1;
        for (uint32_t j = 0; j < nIn; ++j) {
            uint64_t* rj = rot + (size_t)j * 2 * ext;
            if (inK[j] == 0) {  // KeySwitchExt(ct, true)
                if (fhe_status s = fhe_ks_ext(p, c0, sizeQl, batch, rj, st))
                    return s;
                if (fhe_status s = fhe_ks_ext(p, c1, sizeQl, batch, rj + ext, st))
                    return s;
                continue;
            }
            keys.push_back(inKeys[j]), e0.push_back(rj), e1.push_back(rj + ext);
        }
        if (!keys.empty()) {
            if (fhe_status s = ks_precompute_run(p, lv, c1, batch, ws, wB, st))
                return s;
            if (fhe_status s = ks_inner_multi_run(p, lv, keys.data(), (uint32_t)keys.size(), c1, batch, e0.data(), e1.data(), c0, dP, ws,
                                                  wB, st))
                return s;
        }
    }
    // 2. inner_i = sum_j rot_j * diag[i][j] for every outer step: inner is [2][nOut][batch] extended towers
    for (uint32_t j0 = 0; j0 < nIn; j0 += ninK) {
        BsgsInnerArgs g;
        g.nIn = std::min<uint32_t>(ninK, nIn - j0), g.nInPad = nInPad, g.j0 = j0, g.nOut = nOut;
        g.rot = rot + (size_t)j0 * 2 * ext, g.diag = dTab, g.out = inner, g.lc = c->d_lc, g.mu128 = c->d_mu128;
        g.logN = c->logN, g.batch = batch, g.sizeQl = sizeQl, g.sizeQ = p->sizeQ, g.sizeP = p->sizeP;
        g.accumulate = j0 ? 1u : 0u;
        for (uint32_t j = 0; j < (uint32_t)kMaxBsgsIn; ++j)
            g.k[j] = (j < g.nIn && inK[j0 + j]) ? inK[j0 + j] : 1u;
        const uint32_t tilesPerRow = c->N >= (uint32_t)kTile ? (c->N >> kTileLog) : 1u;
        const uint64_t nGroups = (uint64_t)sizeQlP * tilesPerRow, grid = ((nGroups + 7) / 8) * 8 * 2 * batch;
        static const uint32_t cpl = env_u32("FHE_BSGS_CPL", 2);  // coefficients per lane of the 4- and 8-term instances
        if (ninK == 4 && cpl == 2 && c->N >= 2)
            FHE_LAUNCH((bsgs_inner_kernel<4, 2>), grid, st, g);
        else if (ninK == 4)
            FHE_LAUNCH((bsgs_inner_kernel<4, 1>), grid, st, g);
        else if (ninK == 8 && cpl == 2 && c->N >= 2)
            FHE_LAUNCH((bsgs_inner_kernel<8, 2>), grid, st, g);
        else if (ninK == 8)
            FHE_LAUNCH((bsgs_inner_kernel<8, 1>), grid, st, g);
        else
            FHE_LAUNCH((bsgs_inner_kernel<16, 1>), grid, st, g);
        LAUNCH_CHECK();
    }
    // 3. KeySwitchDown of every inner_i (KeySwitchDownFirstElement for the unrotated steps: their second element is only
    //    needed in the extended basis): element-0 towers of all steps, and element-1 towers when any step is rotated
    const KsLayout wM   = ks_layout(p, sizeQl, batch * nOut);
    const uint32_t nTow = (nR ? 2u : 1u) * nOut * batch;
    if (fhe_status s = mod_down_many(p, lv, inner, nTow, nTow, d, nullptr, ws + wM.pcoef, ws + wM.md, st))
        return s;
    // 4. first = sum_i Automorphism_{k_i}(d_i[0])      (:1861, 1873, 1976)
    {
        std::vector<const uint64_t*> src(nOut);
        std::vector<uint32_t> ks(nOut);
        for (uint32_t i = 0; i < nOut; ++i)
            src[i] = d + (size_t)i * low, ks[i] = outK[perm[i]] ? outK[perm[i]] : 1u;
        if (fhe_status s = automorph_sum_run(c, first, src, ks, nullptr, sizeQl, batch, st))
            return s;
    }
    // 5.-7. outer = sum over the rotated steps of EvalFastRotationExt(d_i, k_i, digits(d_i[1]), addFirst = false)
    //       (+ inner_i[1] of the unrotated steps in the second element)           (:1875-1876, 1863-1866, 1977-1979)
    std::vector<const uint64_t*> src0, src1;
    std::vector<uint32_t> k0, k1;
    for (uint32_t i = 0; i < nZ; ++i)
        src1.push_back(inner + (size_t)(nOut + i) * ext), k1.push_back(1u);
    if (nR) {
        const KsLayout wR  = ks_layout(p, sizeQl, batch * nR);
        const uint64_t* dR = d + (size_t)(nOut + nZ) * low;  // second elements of the rotated steps: nR*batch towers
        if (fhe_status s = ks_precompute_run(p, lv, dR, batch * nR, ws, wR, st))
            return s;
        for (uint32_t i = 0; i < nR; ++i) {
            uint64_t *e0 = ws + wR.e0 + (size_t)i * ext, *e1 = ws + wR.e1 + (size_t)i * ext;
            if (fhe_status s = ks_inner_run(p, lv, outKeys[perm[nZ + i]], dR, batch, e0, e1, ws, wR, st, i * batch))
                return s;
            src0.push_back(e0), src1.push_back(e1);
            k0.push_back(outK[perm[nZ + i]]), k1.push_back(outK[perm[nZ + i]]);
        }
        if (fhe_status s = automorph_sum_run(c, outer, src0, k0, extIdx.data(), sizeQlP, batch, st))
            return s;
    }
    else if (fhe_status s = zero_rows(c, outer, sizeQlP, 0, sizeQlP, batch, st))
        return s;
    if (fhe_status s = automorph_sum_run(c, outer + ext, src1, k1, extIdx.data(), sizeQlP, batch, st))
        return s;
    // 8. result = KeySwitchDown(outer); result[0] += first  (:1879-1880)
    if (fhe_status s = mod_down_many(p, lv, outer, 2 * batch, batch, out0, out1, ws + wB.pcoef, ws + wB.md, st))
        return s;
    return elem_run<OP_ADD>(c, out0, out0, first, nullptr, nullptr, sizeQl, batch, st, "fhe_ckks_bsgs_transform");
}

extern "C" fhe_status fhe_approx_mod_down(fhe_ks_plan* p, const uint64_t* x, uint32_t sizeQl, uint32_t batch,
                                          uint64_t* out, void* wsv, size_t wsBytes, void* st) {
    ARG_CHECK(p && x && out && wsv, "fhe_approx_mod_down: null argument");
    ARG_CHECK(sizeQl >= 1 && sizeQl <= p->sizeQ && batch >= 1, "fhe_approx_mod_down: bad level or batch");
    const KsLayout w = ks_layout(p, sizeQl, batch);
    ARG_CHECK(wsBytes >= w.total * 8, "fhe_approx_mod_down: workspace too small");
    RT_CHECK(rt::set_device(p->ctx->device));
    fhe_ks_plan::Level* lv = nullptr;
    if (fhe_status s = ks_level(p, sizeQl, &lv))
        return s;
    uint64_t* ws = (uint64_t*)wsv;
    return mod_down_run(p, lv, x, batch, out, ws + w.pcoef, ws + w.md, st);
}

// ApproxModDown with the BGV plaintext-modulus factors (dcrtpoly-impl.h:966-1005 with t > 0): the P part is multiplied
// by t^-1 mod p_j before the conversion (:981-983) and the converted part by t mod q_i after it (:996-998).  Both
// are constant multiplications of canonical residues, so they are folded into the conversion's two constant sets
// (y_j = x_j * [t^-1 * Phat_j^-1]_{p_j};  out_i = sum_j y_j * [t * Phat_j]_{q_i}): same words, no extra pass.
extern "C" fhe_status fhe_approx_mod_down_bgv(fhe_ks_plan* p, const uint64_t* x, uint32_t sizeQl, uint64_t t, uint32_t batch,
                                              uint64_t* out, void* wsv, size_t wsBytes, void* st) {
    ARG_CHECK(p && x && out && wsv, "fhe_approx_mod_down_bgv: null argument");
    ARG_CHECK(sizeQl >= 1 && sizeQl <= p->sizeQ && batch >= 1, "fhe_approx_mod_down_bgv: bad level or batch");
    ARG_CHECK(t >= 2, "fhe_approx_mod_down_bgv: t must be at least 2");
    const KsLayout w = ks_layout(p, sizeQl, batch);
    ARG_CHECK(wsBytes >= w.total * 8, "fhe_approx_mod_down_bgv: workspace too small");
    fhe_ctx* c = p->ctx;
    RT_CHECK(rt::set_device(c->device));
    fhe_ks_plan::Level* lv = nullptr;
    if (fhe_status s = ks_level(p, sizeQl, &lv))
        return s;
    std::unique_lock<std::mutex> lock(p->cacheMutex);
    auto it = lv->downT.find(t);
    if (it == lv->downT.end()) {
        std::vector<uint64_t> src(p->sizeP), dst(sizeQl), sScale(p->sizeP), dScale(sizeQl);
        for (uint32_t j = 0; j < p->sizeP; ++j) {
            src[j] = c->q[p->sizeQ + j];
            ARG_CHECK(t % src[j] != 0, "fhe_approx_mod_down_bgv: t must be invertible modulo every p_j");
            sScale[j] = host::invmod(t % src[j], src[j]);  // tInvModp (bgvrns-cryptoparameters.cpp:77-81)
        }
        for (uint32_t i = 0; i < sizeQl; ++i) {
            dst[i]    = c->q[i];
            dScale[i] = t % dst[i];
        }
        fhe_conv* cv = nullptr;
        if (fhe_status s = conv_build(c, src, dst, &cv, sScale.data(), dScale.data()))
            return s;
        it = lv->downT.emplace(t, cv).first;
    }
    fhe_conv* downT = it->second;
    lock.unlock();
    uint64_t* ws = (uint64_t*)wsv;
    if (fhe_status s = mod_down_core(p, lv, x, batch, ws + w.pcoef, ws + w.md, st, downT))
        return s;
    return mod_down_tail(p, lv, x, ws + w.md, batch, out, false, st);
}

// ------------------------------------------------------------------------------------------------
// CKKS rescale
// ------------------------------------------------------------------------------------------------
extern "C" size_t fhe_rescale_workspace_bytes(const fhe_ctx* c, uint32_t sizeQl, uint32_t batch) {
    if (!c || sizeQl < 2)
        return 0;
    // last[batch][1][N] + tmp[batch][sizeQl-1][N]
    return ((size_t)batch * sizeQl << c->logN) * 8;
}
// A caller that keeps inventing constants (weighted sums with ever new weights) must not grow the cache without bound: an entry point
// calls this BEFORE it asks for its tables; past the bound every table goes (after a device-wide wait: a launch on any stream may
// still read one) and the cache refills with what is in use.
constexpr size_t kMaxConstTabs = 8192;
static fhe_status const_tables_trim(fhe_ctx* c) {
    {
        std::lock_guard<std::mutex> lock(c->cacheMutex);
        if (c->constTabs.size() < kMaxConstTabs)
            return FHE_OK;
    }
    std::unique_lock<std::shared_mutex> gate(c->constTabsGate);  // (no entry point holds a table it has not launched with yet)
    std::lock_guard<std::mutex> lock(c->cacheMutex);
    RT_CHECK(rt::device_sync());
    for (auto& kv : c->constTabs)
        rt::dfree(kv.second);
    c->constTabs.clear();
    return FHE_OK;
}
// device copy of per-limb constants {v[i], Shoup(v[i], q[limb_i])}, cached by content (the caller's tables of one level)
static fhe_status const_table(fhe_ctx* c, const uint32_t* limbIdx, const uint64_t* v, uint32_t n, const TwPair** out) {
    std::vector<uint64_t> key(2 * (size_t)n);
    for (uint32_t i = 0; i < n; ++i)
        key[i] = limbIdx ? limbIdx[i] : i, key[n + i] = v[i];
    std::lock_guard<std::mutex> lock(c->cacheMutex);
    auto it = c->constTabs.find(key);
    if (it == c->constTabs.end()) {
        std::vector<TwPair> h(n);
        for (uint32_t i = 0; i < n; ++i) {
            const uint64_t qi = c->q[key[i]];
            if (v[i] >= qi)
                return fail(FHE_ERR_ARG, "const_table: constant is not reduced modulo its limb");
            h[i] = TwPair{v[i], host::shoup(v[i], qi)};
        }
        void* d = nullptr;
        RT_CHECK(rt::dmalloc(&d, n * sizeof(TwPair)));
        RT_CHECK(rt::h2d(d, h.data(), n * sizeof(TwPair), nullptr));
        RT_CHECK(rt::sync(nullptr));
        it = c->constTabs.emplace(std::move(key), (TwPair*)d).first;
    }
    *out = it->second;
    return FHE_OK;
}
// DropLastElementAndScale (dcrtpoly-impl.h:693-712) of towers x[batch][sizeQl][N] over context limbs limbIdx[0..sizeQl) (null: the
// leading ones) -> out[batch][sizeQl-1][N]; dA / dB: device constants QlQlInvModqlDivqlModq / qlInvModq of the kept limbs;
// negated: dA[i] == -dB[i] mod q_i (true for the reference's own tables).  ws: last[batch][N] (+ tmp[batch][l][N] on
// the unfused path).
//   fused (rings of two static passes, negated tables): x*B + NTT(SwitchModulus(last)*(-B)) == (x - NTT(SwitchModulus(last)))*B with
//   canonical residues on both sides, so the words are the reference's.  4 launches: INTT of the last limb (2), the column pass
//   that loads every limb from the one INTT row through SwitchModulus, the row pass whose store is (x - r)*B.  Neither the
//   switched tower nor its transform goes to HBM outside `out`.
// x1 / out1 != null: the towers are the two elements of one ciphertext, allocated on their own (x, x1 -> out, out1; batch must be 2):
// the same launches, each over both towers.
static fhe_status rescale_run(fhe_ctx* c, const uint64_t* x, const uint32_t* limbIdx, uint32_t sizeQl, const TwPair* dA,
                              const TwPair* dB, bool negated, uint32_t batch, uint64_t* out, uint64_t* ws, void* st,
                              const uint64_t* x1 = nullptr, uint64_t* out1 = nullptr) {
    const uint32_t l       = sizeQl - 1;
    const uint32_t lastIdx = limbIdx ? limbIdx[l] : l;
    uint64_t* last         = ws;                                 // [batch][N]
    uint64_t* tmp          = last + ((size_t)batch << c->logN);  // [batch][l][N]
    static const bool noFuse = env_u32("FHE_RESCALE_UNFUSED", 0) != 0;
    const bool fused = negated && !noFuse && ntt_epilogue_supported(c);
    if (x1 && !fused) {  // (small rings, foreign tables: element by element)
        if (fhe_status s = rescale_run(c, x, limbIdx, sizeQl, dA, dB, negated, 1, out, ws, st))
            return s;
        return rescale_run(c, x1, limbIdx, sizeQl, dA, dB, negated, 1, out1, ws, st);
    }
    const int64_t xDelta = x1 ? x1 - x : 0;
    // lastPoly.SetFormat(COEFFICIENT)  (:696-697): INTT of the last limb of every tower, written densely
    if (fhe_status s = ntt_run(c, true, x, last, &lastIdx, 1, batch, st, sizeQl, l, 0, 0, nullptr, true, nullptr, xDelta))
        return s;
    if (fused) {
        NttEpilogue epi;
        epi.mode = 1, epi.split = x1 ? 1 : batch, epi.aStride = sizeQl, epi.aFirst = 0, epi.aDelta = xDelta;
        epi.A = x, epi.C = dB, epi.out0 = out, epi.out1 = x1 ? out1 : out;
        // (two elements: the transform works in the workspace, its fused store goes to the two output towers)
        uint64_t* work = x1 ? tmp : out;
        if (ntt_prologue_supported(c))
            return ntt_run(c, false, last, work, limbIdx, l, batch, st, 1, 0, 0, 0, &epi, true, &lastIdx);
        // the single pass of N = 4096: SwitchModulus as a kernel of its own, then the transform with the fused store
        LimbSel sel;
        if (fhe_status s = make_sel(c, limbIdx, l, &sel, "fhe_rescale"))
            return s;
        if (fhe_status s = switch_modulus_run(c, work, sel, l, last, 1, 0, lastIdx, nullptr, batch, st))
            return s;
        return ntt_run(c, false, work, work, limbIdx, l, batch, st, 0, 0, 0, 0, &epi);
    }
    // tmp = SwitchModulus(last -> q_i) * QlQlInvModqlDivqlModq[i]   (:703-705)
    LimbSel sel;
    if (fhe_status s = make_sel(c, limbIdx, l, &sel, "fhe_rescale"))
        return s;
    if (fhe_status s = switch_modulus_run(c, tmp, sel, l, last, 1, 0, lastIdx, dA, batch, st))
        return s;
    // tmp.SwitchFormat()  (:706-707)
    if (fhe_status s = fhe_ntt_fwd(c, tmp, limbIdx, l, batch, st))
        return s;
    // m_vectors[i] = m_vectors[i] * qlInvModq[i] + tmp  (:708-709); x towers are sizeQl rows apart
    return elem_run<OP_MUL_CONST_ADD>(c, out, x, tmp, dB, limbIdx, l, batch, st, "fhe_rescale", sizeQl, 0);
}
extern "C" fhe_status fhe_rescale(fhe_ctx* c, const uint64_t* x, uint32_t sizeQl, uint32_t batch, uint64_t* out,
                                  void* wsv, size_t wsBytes, void* st) {
    ARG_CHECK(c && x && out && wsv, "fhe_rescale: null argument");
    ARG_CHECK(sizeQl >= 2 && sizeQl <= c->L, "Removing last element of DCRTPoly renders it invalid.");  // :672-673
    ARG_CHECK(batch >= 1 && wsBytes >= fhe_rescale_workspace_bytes(c, sizeQl, batch), "fhe_rescale: workspace too small");
    RT_CHECK(rt::set_device(c->device));
    const uint32_t l = sizeQl - 1;
    // tables (ckksrns-cryptoparameters.cpp:60-81): qlInvModq[i] = q_l^-1 mod q_i,
    // QlQlInvModqlDivqlModq[i] = floor(Q'*(Q'^-1 mod q_l)/q_l) mod q_i = -(q_l^-1) mod q_i   
    std::unique_lock<std::mutex> lock(c->cacheMutex);
    auto it = c->rescaleTabs.find(sizeQl);
    if (it == c->rescaleTabs.end()) {
        std::vector<TwPair> hA(l), hB(l);
        for (uint32_t i = 0; i < l; ++i) {
            const uint64_t qi = c->q[i];
            const uint64_t B  = host::invmod(c->q[l] % qi, qi);
            const uint64_t A  = (qi - B) % qi;
            hA[i]             = TwPair{A, host::shoup(A, qi)};
            hB[i]             = TwPair{B, host::shoup(B, qi)};
        }
        void *dA = nullptr, *dB = nullptr;
        RT_CHECK(rt::dmalloc(&dA, l * sizeof(TwPair)));
        c->owned.push_back(dA);
        RT_CHECK(rt::dmalloc(&dB, l * sizeof(TwPair)));
        c->owned.push_back(dB);
        RT_CHECK(rt::h2d(dA, hA.data(), l * sizeof(TwPair), nullptr));
        RT_CHECK(rt::h2d(dB, hB.data(), l * sizeof(TwPair), nullptr));
        RT_CHECK(rt::sync(nullptr));
        it = c->rescaleTabs.emplace(sizeQl, std::make_pair((TwPair*)dA, (TwPair*)dB)).first;
    }
    TwPair *dA = it->second.first, *dB = it->second.second;
    lock.unlock();
    return rescale_run(c, x, nullptr, sizeQl, dA, dB, true, batch, out, (uint64_t*)wsv, st);
}
// the same with the caller's tables (host arrays of sizeQl-1 residues: CryptoParametersRNS::GetQlQlInvModqlDivqlModq(l) /
// GetqlInvModq(l)) over any limbs of the context: what the DCRTPoly backend's DropLastElementAndScale calls
static fhe_status rescale_limbs_run(fhe_ctx* c, const uint64_t* x, const uint64_t* x1, const uint32_t* limbIdx, uint32_t sizeQl,
                                    const uint64_t* QlQlInvModqlDivqlModq, const uint64_t* qlInvModq, uint32_t batch, uint64_t* out,
                                    uint64_t* out1, void* wsv, size_t wsBytes, void* st, const char* who) {
    ARG_CHECK(c && x && out && wsv && QlQlInvModqlDivqlModq && qlInvModq, std::string(who) + ": null argument");
    ARG_CHECK(sizeQl >= 2 && sizeQl <= (uint32_t)kMaxLimbs, "Removing last element of DCRTPoly renders it invalid.");  // :672-673
    ARG_CHECK(batch >= 1 && wsBytes >= fhe_rescale_workspace_bytes(c, sizeQl, batch), std::string(who) + ": workspace too small");
    RT_CHECK(rt::set_device(c->device));
    const uint32_t l = sizeQl - 1;
    for (uint32_t i = 0; i < sizeQl; ++i)
        ARG_CHECK((limbIdx ? limbIdx[i] : i) < c->L, std::string(who) + ": limb index exceeds context size");
    bool negated = true;
    for (uint32_t i = 0; i < l; ++i) {
        const uint64_t qi = c->q[limbIdx ? limbIdx[i] : i];
        negated           = negated && qlInvModq[i] < qi && QlQlInvModqlDivqlModq[i] == (qi - qlInvModq[i]) % qi;
    }
    const TwPair *dA = nullptr, *dB = nullptr;
    if (fhe_status s = const_tables_trim(c))
        return s;
    std::shared_lock<std::shared_mutex> gate(c->constTabsGate);
    if (fhe_status s = const_table(c, limbIdx, QlQlInvModqlDivqlModq, l, &dA))
        return s;
    if (fhe_status s = const_table(c, limbIdx, qlInvModq, l, &dB))
        return s;
    return rescale_run(c, x, limbIdx, sizeQl, dA, dB, negated, batch, out, (uint64_t*)wsv, st, x1, out1);
}
extern "C" fhe_status fhe_rescale_limbs(fhe_ctx* c, const uint64_t* x, const uint32_t* limbIdx, uint32_t sizeQl,
                                        const uint64_t* QlQlInvModqlDivqlModq, const uint64_t* qlInvModq, uint32_t batch,
                                        uint64_t* out, void* wsv, size_t wsBytes, void* st) {
    return rescale_limbs_run(c, x, nullptr, limbIdx, sizeQl, QlQlInvModqlDivqlModq, qlInvModq, batch, out, nullptr, wsv, wsBytes, st,
                             "fhe_rescale_limbs");
}
// the two elements of one ciphertext (towers x0, x1 -> out0, out1, each allocated on its own) in the same four launches;
// ws of fhe_rescale_workspace_bytes(ctx, sizeQl, 2)
extern "C" fhe_status fhe_rescale_limbs_pair(fhe_ctx* c, const uint64_t* x0, const uint64_t* x1, const uint32_t* limbIdx, uint32_t sizeQl,
                                             const uint64_t* QlQlInvModqlDivqlModq, const uint64_t* qlInvModq, uint64_t* out0,
                                             uint64_t* out1, void* wsv, size_t wsBytes, void* st) {
    ARG_CHECK(x1 && out1 && x1 != x0 && out1 != out0, "fhe_rescale_limbs_pair: needs two distinct towers");
    return rescale_limbs_run(c, x0, x1, limbIdx, sizeQl, QlQlInvModqlDivqlModq, qlInvModq, 2, out0, out1, wsv, wsBytes, st,
                             "fhe_rescale_limbs_pair");
}

// DCRTPolyImpl::ModReduce (dcrtpoly-impl.h:736-755), the BGV modulus switch by the last limb with plaintext modulus t:
//   delta = [last limb]_COEFF * (-t^-1 mod q_l);  x_i = (x_i + t * SwitchModulus(delta -> q_i)) * q_l^-1,  i < l.
// Launched as x_i * B_i + NTT(SwitchModulus(delta) * A_i) with B_i = q_l^-1 and A_i = t * q_l^-1 mod q_i: the NTT is
// linear, every step yields canonical residues, so the words equal the reference's.  Same workspace as fhe_rescale.
extern "C" fhe_status fhe_mod_reduce(fhe_ctx* c, const uint64_t* x, uint32_t sizeQl, uint64_t t, int evalFormat,
                                     uint32_t batch, uint64_t* out, void* wsv, size_t wsBytes, void* st) {
    ARG_CHECK(c && x && out && wsv, "fhe_mod_reduce: null argument");
    ARG_CHECK(sizeQl >= 2 && sizeQl <= c->L, "Removing last element of DCRTPoly renders it invalid.");  // :672-673
    ARG_CHECK(batch >= 1 && wsBytes >= fhe_rescale_workspace_bytes(c, sizeQl, batch), "fhe_mod_reduce: workspace too small");
    RT_CHECK(rt::set_device(c->device));
    const uint32_t l  = sizeQl - 1;
    const uint64_t ql = c->q[l];
    ARG_CHECK(t >= 2 && t % ql != 0, "fhe_mod_reduce: t must be invertible modulo the dropped limb");
    uint64_t* last = (uint64_t*)wsv;                    // [batch][N]
    uint64_t* tmp  = last + ((size_t)batch << c->logN);  // [batch][l][N]
    std::unique_lock<std::mutex> lock(c->cacheMutex);
    auto it = c->modReduceTabs.find({sizeQl, t});
    if (it == c->modReduceTabs.end()) {
        std::vector<TwPair> h(2 * (size_t)l + 1);
        for (uint32_t i = 0; i < l; ++i) {
            const uint64_t qi = c->q[i];
            const uint64_t B  = host::invmod(ql % qi, qi);      // qlInvModq
            const uint64_t A  = host::mulmod(t % qi, B, qi);
            h[i]              = TwPair{A, host::shoup(A, qi)};
            h[l + i]          = TwPair{B, host::shoup(B, qi)};
        }
        const uint64_t negtInv = (ql - host::invmod(t % ql, ql)) % ql;  // negtInvModq
        h[2 * (size_t)l]       = TwPair{negtInv, host::shoup(negtInv, ql)};
        void* d = nullptr;
        RT_CHECK(rt::dmalloc(&d, h.size() * sizeof(TwPair)));
        c->owned.push_back(d);
        RT_CHECK(rt::h2d(d, h.data(), h.size() * sizeof(TwPair), nullptr));
        RT_CHECK(rt::sync(nullptr));
        it = c->modReduceTabs.emplace(std::make_pair(sizeQl, t), (TwPair*)d).first;
    }
    TwPair *dA = it->second, *dB = dA + l, *dN = dA + 2 * (size_t)l;
    lock.unlock();
    const uint32_t lastIdx = l;
    if (evalFormat) {  // delta.SetFormat(COEFFICIENT)  :741
        if (fhe_status s = ntt_run(c, true, x, last, &lastIdx, 1, batch, st, sizeQl, l))
            return s;
    }
    else
        RT_CHECK(rt::d2d_2d(last, (size_t)8 << c->logN, x + ((size_t)l << c->logN), ((size_t)sizeQl * 8) << c->logN,
                            (size_t)8 << c->logN, batch, (rt::stream_t)st));
    if (fhe_status s = elem_run<OP_MUL_CONST>(c, last, last, nullptr, dN, &lastIdx, 1, batch, st, "fhe_mod_reduce"))  // :742
        return s;
    LimbSel sel;
    if (fhe_status s = make_sel(c, nullptr, l, &sel, "fhe_mod_reduce"))
        return s;
    if (fhe_status s = switch_modulus_run(c, tmp, sel, l, last, 1, 0, l, dA, batch, st))  // :749, :752 (t folded into A_i)
        return s;
    if (evalFormat)  // :750-751
        if (fhe_status s = fhe_ntt_fwd(c, tmp, nullptr, l, batch, st))
            return s;
    return elem_run<OP_MUL_CONST_ADD>(c, out, x, tmp, dB, nullptr, l, batch, st, "fhe_mod_reduce", sizeQl, 0);  // :752-753
}

// ------------------------------------------------------------------------------------------------
// BFV side: ScaleAndRound family and BEHZ
// ------------------------------------------------------------------------------------------------
struct fhe_sr_plan {
    fhe_ctx* ctx;
    uint32_t sizeI, sizeO;
    bool exact;
    uint64_t *d_tab = nullptr, *d_o = nullptr, *d_mu = nullptr;
    double* d_frac  = nullptr;
    std::vector<void*> owned;
};
template <typename T>
static fhe_status dev_copy(std::vector<void*>& owned, const T* h, size_t n, T** d) {
    void* p = nullptr;
    RT_CHECK(rt::dmalloc(&p, n * sizeof(T)));
    owned.push_back(p);
    RT_CHECK(rt::h2d(p, h, n * sizeof(T), nullptr));
    RT_CHECK(rt::sync(nullptr));
    *d = (T*)p;
    return FHE_OK;
}
extern "C" fhe_status fhe_sr_plan_create(fhe_ctx* c, uint32_t sizeI, const uint32_t* outLimbIdx, uint32_t sizeO,
                                         const uint64_t* tab, const double* frac, fhe_sr_plan** out) {
    ARG_CHECK(c && outLimbIdx && tab && out, "fhe_sr_plan_create: null argument");
    // (the 128-bit sums of the kernel hold sizeI + 1 products of residues below 2^60: 2^8 terms — as the reference's DoubleNativeInt sums)
    ARG_CHECK(sizeI >= 1 && sizeO >= 1 && sizeI < (uint32_t)kMaxLimbs && sizeO <= (uint32_t)kMaxLimbs, "fhe_sr_plan_create: bad basis size");
    RT_CHECK(rt::set_device(c->device));
    std::vector<uint64_t> o(sizeO), mu(2 * (size_t)sizeO);
    for (uint32_t j = 0; j < sizeO; ++j) {
        ARG_CHECK(outLimbIdx[j] < c->L, "fhe_sr_plan_create: limb index exceeds context size");
        o[j] = c->q[outLimbIdx[j]];
        host::mu128(o[j], &mu[2 * j]);
    }
    fhe_sr_plan* p = new fhe_sr_plan;
    p->ctx = c, p->sizeI = sizeI, p->sizeO = sizeO, p->exact = frac != nullptr;
    fhe_status s;
    if ((s = dev_copy(p->owned, tab, (size_t)sizeO * (sizeI + 1), &p->d_tab)) || (s = dev_copy(p->owned, o.data(), o.size(), &p->d_o)) ||
        (s = dev_copy(p->owned, mu.data(), mu.size(), &p->d_mu)) || (frac && (s = dev_copy(p->owned, frac, sizeI, &p->d_frac)))) {
        fhe_sr_plan_destroy(p);
        return s;
    }
    *out = p;
    return FHE_OK;
}
extern "C" void fhe_sr_plan_destroy(fhe_sr_plan* p) {
    if (!p)
        return;
    for (void* q : p->owned)
        rt::dfree(q);
    delete p;
}
extern "C" fhe_status fhe_scale_and_round(fhe_sr_plan* p, const uint64_t* x, int outputFirst, uint64_t* out, uint32_t batch,
                                          void* st) {
    ARG_CHECK(p && x && out && batch >= 1, "fhe_scale_and_round: bad argument");
    RT_CHECK(rt::set_device(p->ctx->device));
    ScaleRoundArgs g;
    const uint32_t tot = p->sizeI + p->sizeO;
    uint64_t* xm       = const_cast<uint64_t*>(x);
    g.in  = TowerView{xm, tot, outputFirst ? p->sizeO : 0u};   // inputIndex  (:1527-1534)
    g.own = TowerView{xm, tot, outputFirst ? 0u : p->sizeI};   // outputIndex
    g.out = TowerView{out, p->sizeO, 0};
    g.tab = p->d_tab, g.frac = p->d_frac, g.o = p->d_o, g.mu = p->d_mu;
    g.logN = p->ctx->logN, g.batch = batch, g.sizeI = p->sizeI, g.sizeO = p->sizeO;
    const uint32_t grid = (uint32_t)((((uint64_t)batch << g.logN) + kThreads - 1) / kThreads);
    if (p->exact)
        FHE_LAUNCH((scale_round_kernel<true>), grid, st, g);
    else
        FHE_LAUNCH((scale_round_kernel<false>), grid, st, g);
    LAUNCH_CHECK();
    return FHE_OK;
}
extern "C" fhe_status fhe_scale_and_round_p_over_q(fhe_ctx* c, const uint64_t* x, const uint32_t* limbIdx, uint32_t sizeQ,
                                                   uint64_t* out, uint32_t batch, void* st) {
    ARG_CHECK(c && x && limbIdx && out && batch >= 1 && sizeQ >= 1 && sizeQ < (uint32_t)kMaxLimbs,
              "fhe_scale_and_round_p_over_q: bad argument");
    for (uint32_t i = 0; i <= sizeQ; ++i)
        ARG_CHECK(limbIdx[i] < c->L, "fhe_scale_and_round_p_over_q: limb index exceeds context size");
    RT_CHECK(rt::set_device(c->device));
    ARG_CHECK(sizeQ <= (uint32_t)kMaxPOverQ, "fhe_scale_and_round_p_over_q: at most 64 limbs supported");
    const uint64_t pLast = c->q[limbIdx[sizeQ]];
    POverQArgs g;
    for (uint32_t i = 0; i < (uint32_t)kMaxPOverQ; ++i)
        g.q[i] = 1, g.pInv[i] = TwPair{0, 0};
    for (uint32_t i = 0; i < sizeQ; ++i) {
        const uint64_t qi = c->q[limbIdx[i]], inv = host::invmod(pLast % qi, qi);  // pInvModq
        g.q[i]    = qi;
        g.pInv[i] = TwPair{inv, host::shoup(inv, qi)};
    }
    g.x = TowerView{const_cast<uint64_t*>(x), sizeQ + 1, 0}, g.out = TowerView{out, sizeQ, 0};
    g.pLast = pLast, g.logN = c->logN, g.batch = batch, g.sizeQ = sizeQ;
    FHE_LAUNCH(p_over_q_kernel, (uint32_t)((((uint64_t)batch << g.logN) + kThreads - 1) / kThreads), st, g);
    LAUNCH_CHECK();
    return FHE_OK;
}

// DCRTPolyImpl::ScaleAndRound -> NativePoly mod t (dcrtpoly-impl.h:1190-1467, BFV HPS decryption) with the caller's
// tables tQHatInvModqDivqModt / tQHatInvModqBDivqModt / tQHatInvModqDivqFrac / tQHatInvModqDivqBFrac; the branch the
// reference takes depends only on (max q_i, t, sizeQ) and is selected here with its own conditions.
static uint32_t msb64(uint64_t v) {
    uint32_t n = 0;
    while (v)
        ++n, v >>= 1;
    return n;
}
static fhe_status native_tables(fhe_ctx* c, uint32_t sizeQ, const uint64_t* a, const uint64_t* b, const uint64_t* modA,
                                const uint64_t* modB, std::vector<TwPair>& h) {
    h.assign(2 * (size_t)sizeQ, TwPair{0, 0});
    for (uint32_t i = 0; i < sizeQ; ++i) {
        h[i] = TwPair{a[i], host::shoup(a[i], modA[i])};  // PrepModMulConst: table values are residues of their modulus
        if (b)
            h[sizeQ + i] = TwPair{b[i], host::shoup(b[i], modB[i])};
    }
    (void)c;
    return FHE_OK;
}
extern "C" fhe_status fhe_scale_and_round_native(fhe_ctx* c, const uint64_t* x, const uint32_t* limbIdx, uint32_t sizeQ,
                                                 uint64_t t, const uint64_t* tabModt, const uint64_t* tabBModt,
                                                 const double* frac, const double* bfrac, uint32_t batch, uint64_t* out,
                                                 void* st) {
    ARG_CHECK(c && x && tabModt && frac && out && batch >= 1, "fhe_scale_and_round_native: bad argument");
    ARG_CHECK(sizeQ >= 1 && sizeQ <= (uint32_t)kMaxLimbs && t >= 2, "fhe_scale_and_round_native: bad basis size or t");
    RT_CHECK(rt::set_device(c->device));
    uint64_t qmax = 0;
    for (uint32_t i = 0; i < sizeQ; ++i) {
        const uint32_t l = limbIdx ? limbIdx[i] : i;
        ARG_CHECK(l < c->L, "fhe_scale_and_round_native: limb index exceeds context size");
        qmax = std::max(qmax, c->q[l]);
    }
    ScaleRoundNativeArgs g;
    const uint32_t qMSB = msb64(qmax), tMSB = msb64(t), sizeQMSB = msb64(sizeQ);
    g.qMSBHf = qMSB >> 1;
    g.pow2   = (t & (t - 1)) == 0 ? 1u : 0u;
    g.split  = (qMSB + sizeQMSB < 52) ? 0u : 1u;
    if (!g.split)
        g.nomod = g.pow2 ? (qMSB + sizeQMSB + tMSB < 63) : (qMSB + tMSB + sizeQMSB < 52);
    else
        g.nomod = g.pow2 ? (g.qMSBHf + tMSB + sizeQMSB < 62) : (g.qMSBHf + tMSB + sizeQMSB < 52);
    ARG_CHECK(!g.split || (tabBModt && bfrac), "fhe_scale_and_round_native: this (q, t, sizeQ) needs the B tables");
    std::vector<uint64_t> tv(sizeQ, t);
    std::vector<TwPair> h;
    native_tables(c, sizeQ, tabModt, g.split ? tabBModt : nullptr, tv.data(), tv.data(), h);
    std::vector<double> fr(2 * (size_t)sizeQ, 0.0);
    for (uint32_t i = 0; i < sizeQ; ++i) {
        fr[i] = frac[i];
        if (g.split)
            fr[sizeQ + i] = bfrac[i];
    }
    void *dT = nullptr, *dF = nullptr;
    RT_CHECK(rt::dmalloc(&dT, h.size() * sizeof(TwPair)));
    RT_CHECK(rt::dmalloc(&dF, fr.size() * sizeof(double)));
    RT_CHECK(rt::h2d(dT, h.data(), h.size() * sizeof(TwPair), (rt::stream_t)st));
    RT_CHECK(rt::h2d(dF, fr.data(), fr.size() * sizeof(double), (rt::stream_t)st));
    g.in = TowerView{const_cast<uint64_t*>(x), sizeQ, 0};
    g.out = out, g.q = nullptr;
    g.tabModt = (TwPair*)dT, g.tabBModt = (TwPair*)dT + sizeQ;
    g.frac = (double*)dF, g.bfrac = (double*)dF + sizeQ;
    g.t = t, g.logN = c->logN, g.batch = batch, g.sizeQ = sizeQ;
    FHE_LAUNCH(scale_round_native_kernel, (uint32_t)((((uint64_t)batch << g.logN) + kThreads - 1) / kThreads), st, g);
    const char* le = rt::last_launch_error();
    RT_CHECK(rt::sync((rt::stream_t)st));  // decryption is a once-per-result call: tables live for this call only
    rt::dfree(dT), rt::dfree(dF);
    RT_CHECK(le);
    return FHE_OK;
}
// DCRTPolyImpl::ScaleAndRound, BEHZ decryption overload (dcrtpoly-impl.h:1631-1671): tables tgammaQHatModq (mod q_i) and
// negInvqModtgamma (mod t*gamma, gamma = 2^26)
extern "C" fhe_status fhe_scale_and_round_behz_decrypt(fhe_ctx* c, const uint64_t* x, const uint32_t* limbIdx, uint32_t sizeQ,
                                                       uint64_t tgamma, const uint64_t* tgammaQHatModq,
                                                       const uint64_t* negInvqModtgamma, uint32_t batch, uint64_t* out,
                                                       void* st) {
    ARG_CHECK(c && x && tgammaQHatModq && negInvqModtgamma && out && batch >= 1, "fhe_scale_and_round_behz_decrypt: bad argument");
    ARG_CHECK(sizeQ >= 1 && sizeQ <= (uint32_t)kMaxLimbs && tgamma >= 2, "fhe_scale_and_round_behz_decrypt: bad basis size or t*gamma");
    RT_CHECK(rt::set_device(c->device));
    std::vector<uint64_t> qv(sizeQ), tg(sizeQ, tgamma);
    for (uint32_t i = 0; i < sizeQ; ++i) {
        const uint32_t l = limbIdx ? limbIdx[i] : i;
        ARG_CHECK(l < c->L, "fhe_scale_and_round_behz_decrypt: limb index exceeds context size");
        qv[i] = c->q[l];
    }
    std::vector<TwPair> h;
    native_tables(c, sizeQ, tgammaQHatModq, negInvqModtgamma, qv.data(), tg.data(), h);
    void *dT = nullptr, *dQ = nullptr;
    RT_CHECK(rt::dmalloc(&dT, h.size() * sizeof(TwPair)));
    RT_CHECK(rt::dmalloc(&dQ, qv.size() * 8));
    RT_CHECK(rt::h2d(dT, h.data(), h.size() * sizeof(TwPair), (rt::stream_t)st));
    RT_CHECK(rt::h2d(dQ, qv.data(), qv.size() * 8, (rt::stream_t)st));
    ScaleRoundNativeArgs g{};
    g.in = TowerView{const_cast<uint64_t*>(x), sizeQ, 0};
    g.out = out, g.q = (uint64_t*)dQ;
    g.tabModt = (TwPair*)dT, g.tabBModt = (TwPair*)dT + sizeQ;
    g.t = tgamma, g.logN = c->logN, g.batch = batch, g.sizeQ = sizeQ;
    FHE_LAUNCH(scale_round_behz_decrypt_kernel, (uint32_t)((((uint64_t)batch << g.logN) + kThreads - 1) / kThreads), st, g);
    const char* le = rt::last_launch_error();
    RT_CHECK(rt::sync((rt::stream_t)st));
    rt::dfree(dT), rt::dfree(dQ);
    RT_CHECK(le);
    return FHE_OK;
}

// ---- BEHZ ----
struct fhe_behz {
    fhe_ctx* ctx;
    uint32_t numQ, numBsk;
    std::vector<uint32_t> qIdx, bskIdx, allIdx;
    BehzTables tb;
    std::vector<void*> owned;
};
// Bsk = numQ primes below q.back() then m_sk  (bfvrns-cryptoparameters.cpp:682-711)
extern "C" uint32_t fhe_param_behz_bsk(uint32_t logN, uint32_t numQ, const uint64_t* q, uint64_t t, uint64_t* bsk,
                                       uint64_t* psiBsk) {
    if (!q || !bsk || !psiBsk || numQ < 1 || numQ > (uint32_t)kBehzWideLimbs - 1)
        return 0;
    const uint64_t M = 2ull << logN;
    uint64_t cur     = q[numQ - 1];
    for (uint32_t i = 0; i < numQ; ++i) {
        cur    = host::previous_prime(cur, M);
        bsk[i] = cur;
    }
    uint64_t msk = host::previous_prime(bsk[numQ - 1], M);
    uint32_t sb  = host::bitlen(msk);
    auto mulw = [](std::vector<uint64_t>& big, uint64_t m) {
        uint64_t carry = 0;
        for (auto& w : big) {
            host::u128 tt = (host::u128)w * m + carry;
            w             = (uint64_t)tt;
            carry         = (uint64_t)(tt >> 64);
        }
        if (carry)
            big.push_back(carry);
    };
    auto less = [](const std::vector<uint64_t>& a, const std::vector<uint64_t>& b) {
        if (a.size() != b.size())
            return a.size() < b.size();
        for (size_t k = a.size(); k-- > 0;)
            if (a[k] != b[k])
                return a[k] < b[k];
        return false;
    };
    std::vector<uint64_t> rhs(1, 1);
    mulw(rhs, M);
    mulw(rhs, t);
    for (uint32_t i = 0; i < numQ; ++i)
        mulw(rhs, q[i]);
    for (;;) {
        std::vector<uint64_t> lhs(1, 1);
        for (uint32_t i = 0; i < numQ; ++i)
            mulw(lhs, bsk[i]);
        mulw(lhs, msk);
        if (!less(lhs, rhs))
            break;
        if (++sb > 60)
            return 0;  // the reference throws: requested bit length exceeds MAX_MODULUS_SIZE
        msk = host::next_prime(host::first_prime(sb, M), M);
    }
    bsk[numQ] = msk;
    for (uint32_t i = 0; i <= numQ; ++i)
        psiBsk[i] = host::min_root_of_unity(M, bsk[i]);
    return numQ + 1;
}

extern "C" fhe_status fhe_behz_create(fhe_ctx* c, const uint32_t* qLimbIdx, uint32_t numQ, const uint32_t* bskLimbIdx,
                                      uint64_t t, fhe_behz** out) {
    ARG_CHECK(c && qLimbIdx && bskLimbIdx && out, "fhe_behz_create: null argument");
    ARG_CHECK(numQ >= 1 && numQ + 1 <= (uint32_t)kBehzWideLimbs, "fhe_behz_create: at most 127 Q limbs supported");
    const uint32_t numB = numQ, numBsk = numQ + 1;
    std::vector<uint64_t> q(numQ), bsk(numBsk);
    for (uint32_t i = 0; i < numQ; ++i) {
        ARG_CHECK(qLimbIdx[i] < c->L, "fhe_behz_create: limb index exceeds context size");
        q[i] = c->q[qLimbIdx[i]];
    }
    for (uint32_t j = 0; j < numBsk; ++j) {
        ARG_CHECK(bskLimbIdx[j] < c->L, "fhe_behz_create: limb index exceeds context size");
        bsk[j] = c->q[bskLimbIdx[j]];
    }
    RT_CHECK(rt::set_device(c->device));
    const uint64_t mtilde = 1ull << 16, msk = bsk[numB];
    std::vector<uint64_t> B(bsk.begin(), bsk.begin() + numB);
    // vectors padded to W, matrices stored [target][W]: 16 for the register-resident kernels, kBehzWideLimbs above 15 Q limbs (bfv_kernels.h)
    const size_t W = numBsk <= (uint32_t)kMaxBfvLimbs ? (size_t)kMaxBfvLimbs : (size_t)kBehzWideLimbs;
    std::vector<uint64_t> muQ(2 * W, 0), muBsk(2 * W, 0), QHatModbsk(W * numBsk, 0), QHatModmt(W, 0), qInvModbsk(W * numBsk, 0),
        BHatModmsk(W, 0), BHatModq(W * numQ, 0), qPad(W, 1), bskPad(W, 1);
    std::vector<TwPair> mtQHatInv(W, TwPair{0, 0}), tQHatInv(W, TwPair{0, 0}), BModq(W, TwPair{0, 0}), QModbsk(W, TwPair{0, 0}),
        mtInvModbsk(W, TwPair{0, 0}), tQInvModbsk(W, TwPair{0, 0}), BHatInv(W, TwPair{0, 0});
    std::copy(q.begin(), q.end(), qPad.begin());
    std::copy(bsk.begin(), bsk.end(), bskPad.begin());
    auto pair = [](uint64_t v, uint64_t m) { return TwPair{v, host::shoup(v, m)}; };
    for (uint32_t i = 0; i < numQ; ++i) {
        const uint64_t qi = q[i], hatInv = host::invmod(host::prod_mod(q, (int)i, qi), qi);
        tQHatInv[i]  = pair(host::mulmod(hatInv, t % qi, qi), qi);        // :722-733
        mtQHatInv[i] = pair(host::mulmod(hatInv, mtilde % qi, qi), qi);   // :755-768
        for (uint32_t j = 0; j < numBsk; ++j) {
            QHatModbsk[(size_t)j * W + i] = host::prod_mod(q, (int)i, bsk[j]);         // :735-747
            qInvModbsk[(size_t)j * W + i] = host::invmod(qi % bsk[j], bsk[j]);          // :749-755
        }
        uint64_t v = 1;
        for (uint32_t k = 0; k < numQ; ++k)
            if (k != i)
                v = (v * (q[k] & (mtilde - 1))) & (mtilde - 1);
        QHatModmt[i] = v;
        BModq[i]     = pair(host::prod_mod(B, -1, qi), qi);                // :838-845
        host::mu128(qi, &muQ[2 * i]);
    }
    uint64_t Qm = 1;
    for (uint32_t k = 0; k < numQ; ++k)
        Qm = (Qm * (q[k] & (mtilde - 1))) & (mtilde - 1);
    uint64_t inv = 1;
    for (int it = 0; it < 5; ++it)
        inv = (inv * (2 - Qm * inv)) & (mtilde - 1);
    for (uint32_t j = 0; j < numBsk; ++j) {
        const uint64_t bj = bsk[j], Qb = host::prod_mod(q, -1, bj);
        QModbsk[j]     = pair(Qb, bj);                                                        // :775-783
        mtInvModbsk[j] = pair(host::invmod(mtilde % bj, bj), bj);                              // :785-793
        tQInvModbsk[j] = pair(host::mulmod(host::invmod(Qb, bj), t % bj, bj), bj);             // :795-804
        host::mu128(bj, &muBsk[2 * j]);
    }
    for (uint32_t i = 0; i < numB; ++i) {
        BHatInv[i]    = pair(host::invmod(host::prod_mod(B, (int)i, B[i]), B[i]), B[i]);       // :806-817
        BHatModmsk[i] = host::prod_mod(B, (int)i, msk);                                      // :829-834
        for (uint32_t j = 0; j < numQ; ++j)
            BHatModq[(size_t)j * W + i] = host::prod_mod(B, (int)i, q[j]);                 // :819-827
    }
    fhe_behz* h = new fhe_behz;
    h->ctx = c, h->numQ = numQ, h->numBsk = numBsk;
    h->qIdx.assign(qLimbIdx, qLimbIdx + numQ);
    h->bskIdx.assign(bskLimbIdx, bskLimbIdx + numBsk);
    h->allIdx = h->qIdx;
    h->allIdx.insert(h->allIdx.end(), h->bskIdx.begin(), h->bskIdx.end());
    BehzTables& tb = h->tb;
    tb.numQ = numQ, tb.numBsk = numBsk, tb.W = (uint32_t)W;
    tb.negQInvModmt = ((mtilde - 1) * inv) & (mtilde - 1);                                    // :770-773
    tb.BInvModmsk   = pair(host::invmod(host::prod_mod(B, -1, msk), msk), msk);              // :836-837
    tb.mskMu        = host::barrett_mu(msk);
    tb.mskMsb       = host::bitlen(msk);
    fhe_status s;
    uint64_t *dq, *dbsk, *dmuQ, *dmuB, *d1, *d2, *d3, *d4, *d5;
    TwPair *p1, *p2, *p3, *p4, *p5, *p6, *p7;
    if ((s = dev_copy(h->owned, qPad.data(), qPad.size(), &dq)) || (s = dev_copy(h->owned, bskPad.data(), bskPad.size(), &dbsk)) ||
        (s = dev_copy(h->owned, muQ.data(), muQ.size(), &dmuQ)) || (s = dev_copy(h->owned, muBsk.data(), muBsk.size(), &dmuB)) ||
        (s = dev_copy(h->owned, QHatModbsk.data(), QHatModbsk.size(), &d1)) || (s = dev_copy(h->owned, QHatModmt.data(), QHatModmt.size(), &d2)) ||
        (s = dev_copy(h->owned, qInvModbsk.data(), qInvModbsk.size(), &d3)) || (s = dev_copy(h->owned, BHatModmsk.data(), BHatModmsk.size(), &d4)) ||
        (s = dev_copy(h->owned, BHatModq.data(), BHatModq.size(), &d5)) || (s = dev_copy(h->owned, mtQHatInv.data(), mtQHatInv.size(), &p1)) ||
        (s = dev_copy(h->owned, tQHatInv.data(), tQHatInv.size(), &p2)) || (s = dev_copy(h->owned, BModq.data(), BModq.size(), &p3)) ||
        (s = dev_copy(h->owned, QModbsk.data(), QModbsk.size(), &p4)) || (s = dev_copy(h->owned, mtInvModbsk.data(), mtInvModbsk.size(), &p5)) ||
        (s = dev_copy(h->owned, tQInvModbsk.data(), tQInvModbsk.size(), &p6)) || (s = dev_copy(h->owned, BHatInv.data(), BHatInv.size(), &p7))) {
        fhe_behz_destroy(h);
        return s;
    }
    tb.q = dq, tb.bsk = dbsk, tb.muQ = dmuQ, tb.muBsk = dmuB, tb.QHatModbsk = d1, tb.QHatModmt = d2, tb.qInvModbsk = d3;
    tb.BHatModmsk = d4, tb.BHatModq = d5, tb.mtQHatInv = p1, tb.tQHatInv = p2, tb.BModq = p3, tb.QModbsk = p4;
    tb.mtInvModbsk = p5, tb.tQInvModbsk = p6, tb.BHatInv = p7;
    *out = h;
    return FHE_OK;
}
extern "C" void fhe_behz_destroy(fhe_behz* h) {
    if (!h)
        return;
    for (void* p : h->owned)
        rt::dfree(p);
    delete h;
}
// The caller's tables in place of the derived ones, per member (the arguments the reference passes to that member, row-major as it
// indexes them).  A DCRTPoly backend receives exactly these vectors from pke (CryptoParametersBFVRNS getters): with an override the
// member computes with the CALLER's values, whatever they are, as the reference's member does.
static fhe_status behz_pairs(fhe_behz* h, const uint64_t* v, const uint64_t* mod, uint32_t n, const TwPair** slot) {
    std::vector<TwPair> t(h->tb.W, TwPair{0, 0});
    for (uint32_t i = 0; i < n; ++i) {
        const uint64_t w = v[i] % mod[i];
        t[i]             = TwPair{w, host::shoup(w, mod[i])};
    }
    TwPair* d = nullptr;
    if (fhe_status s = dev_copy(h->owned, t.data(), t.size(), &d))
        return s;
    *slot = d;
    return FHE_OK;
}
// matrix given as [nRow][nCol] (reference order) -> device [nCol][kMaxBfvLimbs] (one target's weights contiguous)
static fhe_status behz_matrix(fhe_behz* h, const uint64_t* m, uint32_t nRow, uint32_t nCol, const uint64_t* colMod, const uint64_t** slot) {
    const size_t W = h->tb.W;
    std::vector<uint64_t> t(W * nCol, 0);
    for (uint32_t i = 0; i < nRow; ++i)
        for (uint32_t j = 0; j < nCol; ++j)
            t[(size_t)j * W + i] = m[(size_t)i * nCol + j] % colMod[j];
    uint64_t* d = nullptr;
    if (fhe_status s = dev_copy(h->owned, t.data(), t.size(), &d))
        return s;
    *slot = d;
    return FHE_OK;
}
static void behz_moduli(const fhe_behz* h, std::vector<uint64_t>& q, std::vector<uint64_t>& bsk) {
    q.resize(h->numQ), bsk.resize(h->numBsk);
    for (uint32_t i = 0; i < h->numQ; ++i)
        q[i] = h->ctx->q[h->qIdx[i]];
    for (uint32_t j = 0; j < h->numBsk; ++j)
        bsk[j] = h->ctx->q[h->bskIdx[j]];
}
extern "C" fhe_status fhe_behz_override_q_to_bsk(fhe_behz* h, const uint64_t* mtildeQHatInvModq, const uint64_t* QHatModbsk,
                                                 const uint64_t* QHatModmtilde, const uint64_t* QModbsk, uint64_t negQInvModmtilde,
                                                 const uint64_t* mtildeInvModbsk) {
    ARG_CHECK(h && mtildeQHatInvModq && QHatModbsk && QHatModmtilde && QModbsk && mtildeInvModbsk, "fhe_behz_override_q_to_bsk: null argument");
    RT_CHECK(rt::set_device(h->ctx->device));
    std::vector<uint64_t> q, bsk;
    behz_moduli(h, q, bsk);
    fhe_status s;
    if ((s = behz_pairs(h, mtildeQHatInvModq, q.data(), h->numQ, &h->tb.mtQHatInv)) ||
        (s = behz_matrix(h, QHatModbsk, h->numQ, h->numBsk, bsk.data(), &h->tb.QHatModbsk)) ||
        (s = behz_pairs(h, QModbsk, bsk.data(), h->numBsk, &h->tb.QModbsk)) ||
        (s = behz_pairs(h, mtildeInvModbsk, bsk.data(), h->numBsk, &h->tb.mtInvModbsk)))
        return s;
    std::vector<uint64_t> mt(h->tb.W, 0);
    std::copy(QHatModmtilde, QHatModmtilde + h->numQ, mt.begin());
    uint64_t* d = nullptr;
    if ((s = dev_copy(h->owned, mt.data(), mt.size(), &d)))
        return s;
    h->tb.QHatModmt    = d;
    h->tb.negQInvModmt = negQInvModmtilde;
    return FHE_OK;
}
extern "C" fhe_status fhe_behz_override_floorq(fhe_behz* h, const uint64_t* tQHatInvModq, const uint64_t* QHatModbsk,
                                               const uint64_t* qInvModbsk, const uint64_t* tQInvModbsk) {
    ARG_CHECK(h && tQHatInvModq && QHatModbsk && qInvModbsk && tQInvModbsk, "fhe_behz_override_floorq: null argument");
    RT_CHECK(rt::set_device(h->ctx->device));
    std::vector<uint64_t> q, bsk;
    behz_moduli(h, q, bsk);
    fhe_status s;
    if ((s = behz_pairs(h, tQHatInvModq, q.data(), h->numQ, &h->tb.tQHatInv)) ||
        (s = behz_matrix(h, QHatModbsk, h->numQ, h->numBsk, bsk.data(), &h->tb.QHatModbsk)) ||
        (s = behz_matrix(h, qInvModbsk, h->numQ, h->numBsk, bsk.data(), &h->tb.qInvModbsk)) ||
        (s = behz_pairs(h, tQInvModbsk, bsk.data(), h->numBsk, &h->tb.tQInvModbsk)))
        return s;
    return FHE_OK;
}
extern "C" fhe_status fhe_behz_override_conv_sk(fhe_behz* h, const uint64_t* BHatInvModb, const uint64_t* BHatModmsk, uint64_t BInvModmsk,
                                                const uint64_t* BHatModq, const uint64_t* BModq) {
    ARG_CHECK(h && BHatInvModb && BHatModmsk && BHatModq && BModq, "fhe_behz_override_conv_sk: null argument");
    RT_CHECK(rt::set_device(h->ctx->device));
    std::vector<uint64_t> q, bsk;
    behz_moduli(h, q, bsk);
    const uint32_t numB = h->numQ;
    const uint64_t msk  = bsk[numB];
    fhe_status s;
    if ((s = behz_pairs(h, BHatInvModb, bsk.data(), numB, &h->tb.BHatInv)) ||
        (s = behz_matrix(h, BHatModq, numB, h->numQ, q.data(), &h->tb.BHatModq)) || (s = behz_pairs(h, BModq, q.data(), h->numQ, &h->tb.BModq)))
        return s;
    std::vector<uint64_t> bm(h->tb.W, 0);
    for (uint32_t i = 0; i < numB; ++i)
        bm[i] = BHatModmsk[i] % msk;
    uint64_t* d = nullptr;
    if ((s = dev_copy(h->owned, bm.data(), bm.size(), &d)))
        return s;
    h->tb.BHatModmsk = d;
    h->tb.BInvModmsk = TwPair{BInvModmsk % msk, host::shoup(BInvModmsk % msk, msk)};
    return FHE_OK;
}
static bool behz_split30() {  // the BEHZ dot products with 30-bit split factors (default; same knob as the conversion kernel)
    static const bool on = env_u32("FHE_CONV_SUM8", 2) != 1u;
    return on;
}
static uint32_t coeff_grid(const fhe_ctx* c, uint32_t batch) {
    return (uint32_t)((((uint64_t)batch << c->logN) + kThreads - 1) / kThreads);
}
extern "C" size_t fhe_behz_workspace_bytes(const fhe_behz* h, uint32_t batch) {
    return h ? ((size_t)batch * h->numQ << h->ctx->logN) * 8 : 0;
}
// DCRTPolyImpl::FastBaseConvqToBskMontgomery (dcrtpoly-impl.h:1694-1786): x is [batch][numQ+numBsk][N]; on entry its
// first numQ rows hold the input in `evalFormat`; on return ALL rows are in EVALUATION format
extern "C" fhe_status fhe_behz_q_to_bsk(fhe_behz* h, uint64_t* x, int evalFormat, uint32_t batch, void* ws, size_t wsBytes,
                                        void* st) {
    ARG_CHECK(h && x && batch >= 1, "fhe_behz_q_to_bsk: bad argument");
    fhe_ctx* c = h->ctx;
    RT_CHECK(rt::set_device(c->device));
    const uint32_t tot = h->numQ + h->numBsk;
    BehzArgs g;
    g.tb = h->tb, g.logN = c->logN, g.batch = batch;
    g.outBsk = TowerView{x, tot, h->numQ};
    g.inBsk = g.outBsk, g.outQ = TowerView{x, tot, 0};
    if (evalFormat) {
        ARG_CHECK(ws && wsBytes >= fhe_behz_workspace_bytes(h, batch), "fhe_behz_q_to_bsk: workspace too small");
        uint64_t* coef = (uint64_t*)ws;
        if (fhe_status s = ntt_run(c, true, x, coef, h->qIdx.data(), h->numQ, batch, st, tot, 0))  // :1708-1712
            return s;
        g.inQ = TowerView{coef, h->numQ, 0};
        if (h->tb.W > (uint32_t)kMaxBfvLimbs)
            FHE_LAUNCH(behz_q_to_bsk_wide_kernel, coeff_grid(c, batch), st, g);
        else if (behz_split30())
            FHE_LAUNCH((behz_q_to_bsk_kernel<true>), coeff_grid(c, batch), st, g);
        else
            FHE_LAUNCH((behz_q_to_bsk_kernel<false>), coeff_grid(c, batch), st, g);
        LAUNCH_CHECK();
        // only the new Bsk limbs go to EVALUATION; the Q limbs keep their original NTT form (:1776-1780)
        return ntt_run(c, false, x, x, h->bskIdx.data(), h->numBsk, batch, st, tot, h->numQ, tot, h->numQ);
    }
    g.inQ = TowerView{x, tot, 0};
    if (h->tb.W > (uint32_t)kMaxBfvLimbs)
        FHE_LAUNCH(behz_q_to_bsk_wide_kernel, coeff_grid(c, batch), st, g);
    else if (behz_split30())
        FHE_LAUNCH((behz_q_to_bsk_kernel<true>), coeff_grid(c, batch), st, g);
    else
        FHE_LAUNCH((behz_q_to_bsk_kernel<false>), coeff_grid(c, batch), st, g);
    LAUNCH_CHECK();
    return fhe_ntt_fwd(c, x, h->allIdx.data(), tot, batch, st);  // every limb to EVALUATION (:1774, :1781-1785)
}
// DCRTPolyImpl::FastRNSFloorq (:1791-1840), in place on x[batch][numQ+numBsk][N] COEFFICIENT
extern "C" fhe_status fhe_behz_floorq(fhe_behz* h, uint64_t* x, uint32_t batch, void* st) {
    ARG_CHECK(h && x && batch >= 1, "fhe_behz_floorq: bad argument");
    RT_CHECK(rt::set_device(h->ctx->device));
    const uint32_t tot = h->numQ + h->numBsk;
    BehzArgs g;
    g.tb = h->tb, g.logN = h->ctx->logN, g.batch = batch;
    g.inQ = g.outQ = TowerView{x, tot, 0};
    g.inBsk = g.outBsk = TowerView{x, tot, h->numQ};
    if (h->tb.W > (uint32_t)kMaxBfvLimbs)
        FHE_LAUNCH(behz_floorq_wide_kernel, coeff_grid(h->ctx, batch), st, g);
    else if (behz_split30())
        FHE_LAUNCH((behz_floorq_kernel<true>), coeff_grid(h->ctx, batch), st, g);
    else
        FHE_LAUNCH((behz_floorq_kernel<false>), coeff_grid(h->ctx, batch), st, g);
    LAUNCH_CHECK();
    return FHE_OK;
}
// DCRTPolyImpl::FastBaseConvSK (:1845-1929): x[batch][numQ+numBsk][N] COEFFICIENT -> out[batch][numQ][N]
extern "C" fhe_status fhe_behz_conv_sk(fhe_behz* h, const uint64_t* x, uint64_t* out, uint32_t batch, void* st) {
    ARG_CHECK(h && x && out && batch >= 1, "fhe_behz_conv_sk: bad argument");
    RT_CHECK(rt::set_device(h->ctx->device));
    const uint32_t tot = h->numQ + h->numBsk;
    BehzArgs g;
    g.tb = h->tb, g.logN = h->ctx->logN, g.batch = batch;
    uint64_t* xm = const_cast<uint64_t*>(x);
    g.inQ = TowerView{xm, tot, 0}, g.inBsk = TowerView{xm, tot, h->numQ};
    g.outQ = TowerView{out, h->numQ, 0}, g.outBsk = g.inBsk;
    if (h->tb.W > (uint32_t)kMaxBfvLimbs)
        FHE_LAUNCH(behz_conv_sk_wide_kernel, coeff_grid(h->ctx, batch), st, g);
    else if (behz_split30())
        FHE_LAUNCH((behz_conv_sk_kernel<true>), coeff_grid(h->ctx, batch), st, g);
    else
        FHE_LAUNCH((behz_conv_sk_kernel<false>), coeff_grid(h->ctx, batch), st, g);
    LAUNCH_CHECK();
    return FHE_OK;
}

// LeveledSHEBFVRNS::EvalMult, BEHZ branch, without relinearisation (bfvrns-leveledshe.cpp:198-445):
//   :302-325 FastBaseConvqToBskMontgomery + SetFormat(EVALUATION) on the four input elements,
//   :327-372 tensor product over Q u Bsk,  :414-437 SetFormat(COEFFICIENT); FastRNSFloorq; FastBaseConvSK per product.
// Inputs [batch][numQ][N] EVALUATION; outputs [batch][numQ][N], COEFFICIENT as the reference leaves them, or
// EVALUATION when outEval != 0 (what LeveledSHEBase::EvalMult(ct,ct,key) does next, base-leveledshe.cpp:204-205).
extern "C" size_t fhe_bfv_eval_mult_behz_workspace_bytes(const fhe_behz* h, uint32_t batch) {
    if (!h)
        return 0;
    const size_t ext = ((size_t)batch * (h->numQ + h->numBsk)) << h->ctx->logN;
    return (7 * ext) * 8 + fhe_behz_workspace_bytes(h, batch);
}
extern "C" fhe_status fhe_bfv_eval_mult_behz(fhe_behz* h, const uint64_t* a0, const uint64_t* a1, const uint64_t* b0,
                                             const uint64_t* b1, uint64_t* d0, uint64_t* d1, uint64_t* d2, int outEval,
                                             uint32_t batch, void* ws, size_t wsBytes, void* st) {
    ARG_CHECK(h && a0 && a1 && b0 && b1 && d0 && d1 && d2 && ws && batch >= 1, "fhe_bfv_eval_mult_behz: bad argument");
    ARG_CHECK(wsBytes >= fhe_bfv_eval_mult_behz_workspace_bytes(h, batch), "fhe_bfv_eval_mult_behz: workspace too small");
    fhe_ctx* c = h->ctx;
    RT_CHECK(rt::set_device(c->device));
    const uint32_t tot = h->numQ + h->numBsk;
    const size_t ext = ((size_t)batch * tot) << c->logN, rowB = (size_t)8 << c->logN;
    uint64_t* w        = (uint64_t*)ws;
    uint64_t* e[4]     = {w, w + ext, w + 2 * ext, w + 3 * ext};
    uint64_t* p[3]     = {w + 4 * ext, w + 5 * ext, w + 6 * ext};
    void* sub          = w + 7 * ext;
    const size_t subB  = fhe_behz_workspace_bytes(h, batch);
    const uint64_t* in[4] = {a0, a1, b0, b1};
    for (int k = 0; k < 4; ++k) {
        RT_CHECK(rt::d2d_2d(e[k], tot * rowB, in[k], h->numQ * rowB, h->numQ * rowB, batch, (rt::stream_t)st));
        if (fhe_status s = fhe_behz_q_to_bsk(h, e[k], 1, batch, sub, subB, st))
            return s;
    }
    if (fhe_status s = fhe_tensor(c, e[0], e[1], e[2], e[3], p[0], p[1], p[2], h->allIdx.data(), tot, batch, st))
        return s;
    uint64_t* out[3] = {d0, d1, d2};
    for (int k = 0; k < 3; ++k) {
        if (fhe_status s = fhe_ntt_inv(c, p[k], h->allIdx.data(), tot, batch, st))
            return s;
        if (fhe_status s = fhe_behz_floorq(h, p[k], batch, st))
            return s;
        if (fhe_status s = fhe_behz_conv_sk(h, p[k], out[k], batch, st))
            return s;
        if (outEval)
            if (fhe_status s = fhe_ntt_fwd(c, out[k], h->qIdx.data(), h->numQ, batch, st))
                return s;
    }
    return FHE_OK;
}

// LeveledSHEBase::EvalMult(ct, ct, key) for BFV/BEHZ (base-leveledshe.cpp:201-214 over bfvrns-leveledshe.cpp:198-445):
// EvalMultNoRelin (three elements, COEFFICIENT) -> SetFormat(EVALUATION) -> KeySwitchCore on the third element ->
// c0 += ks0, c1 += ks1.  The context holds Q (limbs 0..sizeQ-1, the key-switch plan's Q), P and the Bsk limbs.
extern "C" size_t fhe_bfv_eval_mult_relin_workspace_bytes(const fhe_behz* bz, const fhe_ks_plan* p, uint32_t batch) {
    if (!bz || !p)
        return 0;
    return fhe_bfv_eval_mult_behz_workspace_bytes(bz, batch) + fhe_ks_workspace_bytes(p, p->sizeQ, batch) +
           (((size_t)batch * p->sizeQ) << p->ctx->logN) * 8;
}
extern "C" fhe_status fhe_bfv_eval_mult_relin_behz(fhe_behz* bz, fhe_ks_plan* p, const fhe_ks_key* key, const uint64_t* a0,
                                                   const uint64_t* a1, const uint64_t* b0, const uint64_t* b1, uint64_t* c0,
                                                   uint64_t* c1, uint32_t batch, void* wsv, size_t wsBytes, void* st) {
    ARG_CHECK(bz && p && key && a0 && a1 && b0 && b1 && c0 && c1 && wsv && batch >= 1, "fhe_bfv_eval_mult_relin_behz: bad argument");
    ARG_CHECK(key->plan == p && bz->ctx == p->ctx, "fhe_bfv_eval_mult_relin_behz: plans / key do not belong together");
    ARG_CHECK(bz->numQ == p->sizeQ, "fhe_bfv_eval_mult_relin_behz: the BEHZ basis Q must be the key-switch plan's Q");
    for (uint32_t i = 0; i < bz->numQ; ++i)
        ARG_CHECK(bz->qIdx[i] == i, "fhe_bfv_eval_mult_relin_behz: Q must be the context's leading limbs");
    ARG_CHECK(wsBytes >= fhe_bfv_eval_mult_relin_workspace_bytes(bz, p, batch), "fhe_bfv_eval_mult_relin_behz: workspace too small");
    const size_t nrB = fhe_bfv_eval_mult_behz_workspace_bytes(bz, batch), ksB = fhe_ks_workspace_bytes(p, p->sizeQ, batch);
    char* w      = (char*)wsv;
    uint64_t* d2 = (uint64_t*)(w + nrB + ksB);
    if (fhe_status s = fhe_bfv_eval_mult_behz(bz, a0, a1, b0, b1, c0, c1, d2, 1, batch, w, nrB, st))
        return s;
    const KsLayout lay = ks_layout(p, p->sizeQ, batch);
    return keyswitch_run(p, key, d2, p->sizeQ, batch, c0, c1, (uint64_t*)(w + nrB), lay, st, true);
}

// whole-tower checksums (checksum_kernel): out[row] = {sum_i w_i, sum_i (2i + 1) w_i} mod 2^64 of every limb-row of x[rows][N]; out is DEVICE memory
extern "C" fhe_status fhe_checksum(fhe_ctx* c, const uint64_t* x, uint32_t rows, uint64_t* out, void* st) {
    ARG_CHECK(c && x && out && rows >= 1, "fhe_checksum: bad argument");
    RT_CHECK(rt::set_device(c->device));
    RT_CHECK(rt::dzero(out, (size_t)rows * 16, (rt::stream_t)st));
    ChecksumArgs g;
    g.x = x, g.out = out, g.logN = c->logN, g.rows = rows;
    const uint32_t tileLog = std::min<uint32_t>(c->logN, kTileLog);
    FHE_LAUNCH_BARRIER(checksum_kernel, ((uint64_t)rows << c->logN) >> tileLog, st, g);
    LAUNCH_CHECK();
    return FHE_OK;
}

extern "C" size_t fhe_launch_stats(char* buf, size_t cap, uint64_t* total) {
    std::map<std::string, uint64_t> byKernel;
    uint64_t sum = 0;
    for (auto* s = fhe::rt::LaunchSite::head().load(); s; s = s->next) {
        std::string k = s->kernel;
        k = k.substr(0, k.find('<'));
        k.erase(std::remove(k.begin(), k.end(), '('), k.end());
        byKernel[k] += s->n.load();
        sum += s->n.load();
    }
    std::vector<std::pair<uint64_t, std::string>> v;
    for (auto& kv : byKernel)
        if (kv.second)
            v.emplace_back(kv.second, kv.first);
    std::sort(v.rbegin(), v.rend());
    std::string text;
    for (auto& e : v)
        text += e.second + " " + std::to_string(e.first) + "\n";
    if (total)
        *total = sum;
    if (buf && cap) {
        const size_t n = std::min(cap - 1, text.size());
        std::memcpy(buf, text.data(), n);
        buf[n] = 0;
    }
    return text.size() + 1;
}
