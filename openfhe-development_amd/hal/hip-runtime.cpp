// hip-runtime.cpp — runtime of the HIP backend of lbcrypto::DCRTPoly (see lattice/hal/hip/hip-runtime.h).
// Compiled into libOPENFHEcore when OpenFHE is built with this repo's lattice/lat-hal.h in front of the reference's.
//
// The device library is loaded at first use with dlopen: $FHE_HIP_LIB, else the path compiled in as FHE_HIP_DEFAULT_LIB
// (libfhe_hip.so of this repo).  If it cannot be loaded or no device is visible the first DCRTPoly operation FAILS LOUDLY (message
// on stderr + exception): the backend has no silent CPU path.  FHE_HAL_ALLOW_HOST=1 opts into running every member on the host
// mirror instead (the class then behaves exactly like the default backend) — for machines without a GPU, never for measurements.
#include "lattice/hal/hip/hip-runtime.h"

#include <cxxabi.h>
#include <dlfcn.h>
#include <execinfo.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <vector>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>

#include "utils/exception.h"

namespace lbcrypto {
namespace hiprt {

namespace {
struct Runtime {
    Api api{};
    bool live = false;
    std::string why;
    // allocator
    std::mutex poolMutex;
    std::map<size_t, std::vector<uint64_t*>> freeLists;  // bucket (words) -> free buffers
    fhe_ctx* anyCtx = nullptr;                           // fhe_malloc wants a context handle (device selection only)
    // contexts
    struct Universe {
        uint32_t logN = 0;
        fhe_ctx* ctx  = nullptr;
        std::vector<uint64_t> q, psi;
        std::unordered_map<uint64_t, uint32_t> limbOf;  // modulus -> context limb
    };
    std::mutex ctxMutex;
    std::map<uint32_t, Universe> universes;  // by ring dimension
    // conversion plans
    std::mutex convMutex;
    std::map<std::vector<uint64_t>, fhe_conv*> convs;  // key = {ctx, nSrc, nDst, idx..., table words...}
    std::map<std::vector<uint64_t>, fhe_sr_plan*> srPlans;
    std::map<std::vector<uint64_t>, fhe_behz*> behzPlans;  // key = {ctx, numQ, qIdx..., bskIdx..., t}
    std::atomic<uint64_t> deviceOps{0}, hostFallbacks{0}, h2dBytes{0}, d2hBytes{0};
};

template <typename F>
bool sym(void* h, const char* name, F* out) {
    *out = reinterpret_cast<F>(dlsym(h, name));
    return *out != nullptr;
}

Runtime* build() {
    auto* r         = new Runtime;
    const char* env = std::getenv("FHE_HIP_LIB");
#ifdef FHE_HIP_DEFAULT_LIB
    const std::string path = env ? env : FHE_HIP_DEFAULT_LIB;
#else
    const std::string path = env ? env : "libfhe_hip.so";
#endif
    void* h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!h) {
        r->why = std::string("cannot load ") + path + ": " + dlerror();
    }
    else {
        Api& a  = r->api;
        bool ok = sym(h, "fhe_last_error", &a.last_error) && sym(h, "fhe_device_count", &a.device_count) &&
                  sym(h, "fhe_ctx_create", &a.ctx_create) && sym(h, "fhe_malloc", &a.malloc_) && sym(h, "fhe_free", &a.free_) &&
                  sym(h, "fhe_memcpy_h2d", &a.h2d) && sym(h, "fhe_memcpy_d2h", &a.d2h) && sym(h, "fhe_memcpy_d2d", &a.d2d) &&
                  sym(h, "fhe_stream_sync", &a.sync) && sym(h, "fhe_ntt_fwd", &a.ntt_fwd) && sym(h, "fhe_ntt_inv", &a.ntt_inv) &&
                  sym(h, "fhe_ntt_inv_oop", &a.ntt_inv_oop) && sym(h, "fhe_ntt_fwd_oop", &a.ntt_fwd_oop) &&
                  sym(h, "fhe_inner_product", &a.inner_product) && sym(h, "fhe_add", &a.add) && sym(h, "fhe_sub", &a.sub) &&
                  sym(h, "fhe_mul", &a.mul) && sym(h, "fhe_neg", &a.neg) && sym(h, "fhe_mul_add", &a.mul_add) && sym(h, "fhe_mul_const", &a.mul_const) &&
                  sym(h, "fhe_mult_acc", &a.mult_acc) && sym(h, "fhe_add_const", &a.add_const) && sym(h, "fhe_sub_const", &a.sub_const) && sym(h, "fhe_automorph", &a.automorph) &&
                  sym(h, "fhe_switch_modulus", &a.switch_modulus) && sym(h, "fhe_conv_create_custom", &a.conv_create_custom) &&
                  sym(h, "fhe_approx_switch_basis", &a.approx_switch_basis) &&
                  sym(h, "fhe_switch_basis_exact", &a.switch_basis_exact) && sym(h, "fhe_sr_plan_create", &a.sr_plan_create) &&
                  sym(h, "fhe_scale_and_round", &a.scale_and_round) && sym(h, "fhe_behz_create", &a.behz_create) &&
                  sym(h, "fhe_behz_workspace_bytes", &a.behz_workspace_bytes) && sym(h, "fhe_behz_q_to_bsk", &a.behz_q_to_bsk) &&
                  sym(h, "fhe_behz_floorq", &a.behz_floorq) && sym(h, "fhe_behz_conv_sk", &a.behz_conv_sk);
        if (!ok)
            r->why = path + " does not export the C ABI of include/fhe_hip.h";
        else if (a.device_count() < 1)
            r->why = path + ": no HIP device visible";
        else
            r->live = true;
    }
    if (!r->live && !std::getenv("FHE_HAL_ALLOW_HOST")) {
        std::fprintf(stderr, "HIP backend of DCRTPoly: %s (set FHE_HAL_ALLOW_HOST=1 to run on the host mirror instead)\n", r->why.c_str());
        OPENFHE_THROW("HIP backend of DCRTPoly: " + r->why);
    }
    return r;
}
Runtime& rt() {
    static Runtime* r = build();  // (never destroyed: device buffers of static objects may outlive main)
    return *r;
}
uint32_t log2u(uint32_t n) {
    uint32_t l = 0;
    while ((1u << l) < n)
        ++l;
    return l;
}
}  // namespace

bool Available() { return rt().live; }
const Api& api() { return rt().api; }
void Check(fhe_status s, const char* what) {
    if (s != FHE_OK)
        OPENFHE_THROW(std::string(what) + ": " + rt().api.last_error());
}
void CountDevice() { rt().deviceOps.fetch_add(1, std::memory_order_relaxed); }
static std::mutex g_traceMutex;
static std::map<std::string, uint64_t>* g_traceSites = nullptr;
// FHE_HAL_TRACE=1: which callers send work to the host mirror or move words over PCIe (a tuning aid: call sites by
// frequency / by bytes at exit)
static thread_local const char* t_member = nullptr;
void TraceMember(const char* member) { t_member = member; }
static void trace_site(const char* kind, const char* member, uint64_t amount) {
    static const bool trace = std::getenv("FHE_HAL_TRACE") != nullptr;
    if (!trace)
        return;
    void* bt[8];
    const int n     = backtrace(bt, 8);
    std::string key = std::string(kind) + " " + (member ? member : (t_member ? t_member : "")) + " <- ";
    for (int i = 3; i < n; ++i) {
        Dl_info info;
        if (dladdr(bt[i], &info) && info.dli_sname) {
            int st      = 0;
            char* dm    = abi::__cxa_demangle(info.dli_sname, nullptr, nullptr, &st);
            std::string nm = dm ? dm : info.dli_sname;
            free(dm);
            key += nm.substr(0, nm.find('(')).substr(0, 60) + " <- ";
        }
    }
    std::lock_guard<std::mutex> lk(g_traceMutex);
    if (!g_traceSites) {
        g_traceSites = new std::map<std::string, uint64_t>;
        std::atexit([] {
            std::vector<std::pair<uint64_t, std::string>> v;
            for (auto& kv : *g_traceSites)
                v.emplace_back(kv.second, kv.first);
            std::sort(v.rbegin(), v.rend());
            for (size_t i = 0; i < v.size() && i < 400; ++i)
                fprintf(stderr, "hal trace %12lu  %s\n", (unsigned long)v[i].first, v[i].second.c_str());
        });
    }
    (*g_traceSites)[key] += amount;
}
void CountHost(const char* member) {
    rt().hostFallbacks.fetch_add(1, std::memory_order_relaxed);
    trace_site("hostop", member, 1);
}
void D2D(fhe_ctx* c, uint64_t* dst, const uint64_t* src, size_t bytes, const char* what) {
    Check(rt().api.d2d(c, dst, src, bytes, nullptr), what);
    trace_site("d2dBytes", what, bytes);
}
void CountH2D(size_t b) {
    rt().h2dBytes.fetch_add(b, std::memory_order_relaxed);
    trace_site("h2dBytes", nullptr, b);
}
void CountD2H(size_t b) {
    rt().d2hBytes.fetch_add(b, std::memory_order_relaxed);
    trace_site("d2hBytes", nullptr, b);
}

// ---- allocator ----
static size_t bucket_of(size_t words) {
    size_t b = 1024;
    while (b < words)
        b <<= 1;
    if (b > (1u << 20) && words <= b - (b >> 2))  // above 8 MiB: 3/4 steps, so that odd tower heights do not waste 2x
        b -= b >> 2;
    return b;
}
DevBuf::~DevBuf() {
    if (!p)
        return;
    Runtime& r = rt();
    std::lock_guard<std::mutex> lk(r.poolMutex);
    r.freeLists[bucket_of(words)].push_back(p);
}
Buf Alloc(size_t words) {
    Runtime& r      = rt();
    const size_t bk = bucket_of(words);
    auto b          = std::make_shared<DevBuf>();
    b->words        = words;
    {
        std::lock_guard<std::mutex> lk(r.poolMutex);
        auto& fl = r.freeLists[bk];
        if (!fl.empty()) {
            b->p = fl.back();
            fl.pop_back();
            return b;
        }
    }
    void* d = nullptr;
    fhe_status s = r.api.malloc_(r.anyCtx, bk * 8, &d);
    if (s != FHE_OK) {  // memory pressure: give the cached buffers back to the device and retry once
        std::lock_guard<std::mutex> lk(r.poolMutex);
        for (auto& kv : r.freeLists) {
            for (uint64_t* q : kv.second)
                r.api.free_(r.anyCtx, q);
            kv.second.clear();
        }
        s = r.api.malloc_(r.anyCtx, bk * 8, &d);
    }
    Check(s, "HIP backend: device allocation");
    b->p = static_cast<uint64_t*>(d);
    return b;
}

// ---- contexts ----
bool Resolve(uint32_t ringDim, const std::vector<LimbSet>& sets, Resolved* out) {
    Runtime& r = rt();
    if (!r.live || ringDim < 16 || ringDim > (1u << 17) || (ringDim & (ringDim - 1)))
        return false;
    const uint64_t twoN = 2ull * ringDim;
    for (const auto& s : sets)
        for (uint32_t i = 0; i < s.n; ++i)
            if (s.q[i] < 3 || s.q[i] >= (1ull << 60) || (s.q[i] - 1) % twoN != 0 || s.psi[i] == 0)
                return false;
    std::lock_guard<std::mutex> lk(r.ctxMutex);
    auto& u = r.universes[ringDim];
    std::vector<uint64_t> q = u.q, psi = u.psi;
    bool grew = false;
    for (const auto& s : sets)
        for (uint32_t i = 0; i < s.n; ++i) {
            bool known = false;
            for (size_t k = 0; k < q.size() && !known; ++k)
                known = q[k] == s.q[i];
            if (!known) {
                q.push_back(s.q[i]);
                psi.push_back(s.psi[i]);
                grew = true;
            }
        }
    if (q.size() > 128) {
        // more distinct moduli than one device context holds (many CryptoContexts in one process): start over with the
        // moduli of this call; towers already on the device are plain words and resolve again at their next operation
        q.clear(), psi.clear();
        for (const auto& s : sets)
            for (uint32_t i = 0; i < s.n; ++i) {
                bool known = false;
                for (size_t k = 0; k < q.size() && !known; ++k)
                    known = q[k] == s.q[i];
                if (!known)
                    q.push_back(s.q[i]), psi.push_back(s.psi[i]);
            }
        if (q.size() > 128)
            return false;
        grew = true;
    }
    if (grew || !u.ctx) {
        fhe_ctx* c = nullptr;
        if (r.api.ctx_create(log2u(ringDim), (uint32_t)q.size(), q.data(), psi.data(), 0, &c) != FHE_OK)
            return false;  // (e.g. a root that is not primitive: leave the operation to the host mirror)
        // the previous context stays alive: operations of other threads may still be using its tables
        u.ctx = c, u.q = q, u.psi = psi, u.logN = log2u(ringDim);
        u.limbOf.clear();
        for (uint32_t k = 0; k < q.size(); ++k)
            u.limbOf[q[k]] = k;
        if (!r.anyCtx)
            r.anyCtx = c;
    }
    out->ctx = u.ctx;
    out->idx.assign(sets.size(), {});
    for (size_t si = 0; si < sets.size(); ++si) {
        out->idx[si].resize(sets[si].n);
        for (uint32_t i = 0; i < sets[si].n; ++i) {
            const uint32_t l = u.limbOf[sets[si].q[i]];
            if (u.psi[l] != sets[si].psi[i])
                return false;  // same modulus with another root of unity: another transform, not ours
            out->idx[si][i] = l;
        }
    }
    return true;
}

// ---- conversion plans ----
fhe_conv* ConvPlan(fhe_ctx* ctx, const std::vector<uint32_t>& srcIdx, const std::vector<uint32_t>& dstIdx, const uint64_t* hatInv,
                   const uint64_t* hatMod, const uint64_t* alphaMod, const double* qInv) {
    Runtime& r        = rt();
    const size_t nSrc = srcIdx.size(), nDst = dstIdx.size();
    std::vector<uint64_t> key;
    key.reserve(4 + nSrc + nDst + nSrc + nSrc * nDst + (alphaMod ? (nSrc + 1) * nDst + nSrc : 0));
    key.push_back(reinterpret_cast<uintptr_t>(ctx));
    key.push_back(nSrc);
    key.push_back(nDst);
    key.push_back(alphaMod ? 1 : 0);
    key.insert(key.end(), srcIdx.begin(), srcIdx.end());
    key.insert(key.end(), dstIdx.begin(), dstIdx.end());
    key.insert(key.end(), hatInv, hatInv + nSrc);
    key.insert(key.end(), hatMod, hatMod + nSrc * nDst);
    if (alphaMod) {
        key.insert(key.end(), alphaMod, alphaMod + (nSrc + 1) * nDst);
        for (size_t i = 0; i < nSrc; ++i) {
            uint64_t w;
            std::memcpy(&w, qInv + i, 8);
            key.push_back(w);
        }
    }
    std::lock_guard<std::mutex> lk(r.convMutex);
    auto it = r.convs.find(key);
    if (it != r.convs.end())
        return it->second;
    fhe_conv* cv = nullptr;
    Check(r.api.conv_create_custom(ctx, srcIdx.data(), (uint32_t)nSrc, dstIdx.data(), (uint32_t)nDst, hatInv, hatMod, alphaMod, qInv, &cv),
          "HIP backend: basis-conversion plan");
    r.convs.emplace(std::move(key), cv);
    return cv;
}

fhe_sr_plan* SrPlan(fhe_ctx* ctx, uint32_t sizeI, const std::vector<uint32_t>& outIdx, const uint64_t* tab, const double* frac) {
    Runtime& r         = rt();
    const size_t sizeO = outIdx.size();
    std::vector<uint64_t> key{reinterpret_cast<uintptr_t>(ctx), sizeI, sizeO, frac ? 1u : 0u};
    key.insert(key.end(), outIdx.begin(), outIdx.end());
    key.insert(key.end(), tab, tab + sizeO * (sizeI + 1));
    for (uint32_t i = 0; frac && i < sizeI; ++i) {
        uint64_t w;
        std::memcpy(&w, frac + i, 8);
        key.push_back(w);
    }
    std::lock_guard<std::mutex> lk(r.convMutex);
    auto it = r.srPlans.find(key);
    if (it != r.srPlans.end())
        return it->second;
    fhe_sr_plan* p = nullptr;
    Check(r.api.sr_plan_create(ctx, sizeI, outIdx.data(), (uint32_t)sizeO, tab, frac, &p), "HIP backend: ScaleAndRound plan");
    r.srPlans.emplace(std::move(key), p);
    return p;
}
fhe_behz* BehzPlan(fhe_ctx* ctx, const std::vector<uint32_t>& qIdx, const std::vector<uint32_t>& bskIdx, uint64_t t) {
    Runtime& r = rt();
    std::vector<uint64_t> key{reinterpret_cast<uintptr_t>(ctx), qIdx.size()};
    key.insert(key.end(), qIdx.begin(), qIdx.end());
    key.insert(key.end(), bskIdx.begin(), bskIdx.end());
    std::lock_guard<std::mutex> lk(r.convMutex);
    if (t == 0) {  // any plan over these bases
        auto lo = r.behzPlans.lower_bound(key);
        if (lo != r.behzPlans.end() && lo->first.size() == key.size() + 1 && std::equal(key.begin(), key.end(), lo->first.begin()))
            return lo->second;
        t = 65537;
    }
    key.push_back(t);
    auto it = r.behzPlans.find(key);
    if (it != r.behzPlans.end())
        return it->second;
    fhe_behz* p = nullptr;
    if (r.api.behz_create(ctx, qIdx.data(), (uint32_t)qIdx.size(), bskIdx.data(), t, &p) != FHE_OK)
        return nullptr;  // (bases the device kernels do not take: the member runs on the host mirror)
    r.behzPlans.emplace(std::move(key), p);
    return p;
}

}  // namespace hiprt
}  // namespace lbcrypto

// ---- PrecomputeAutoMap (nbtheory2.cpp:264-275), memoised: precomp[bitrev(j)] = bitrev(((2j+1)k mod 2n) >> 1).  pke calls it for
// every rotation and EvalFastRotation (ckksrns-leveledshe.cpp, ckksrns-fhe.cpp, base-leveledshe.cpp); at N = 2^17 that is 0.4 ms of
// host time 260 times per bootstrap.  The HIP build weakens the reference's definition (hal/Makefile) and links this one. ----
namespace lbcrypto {
void PrecomputeAutoMap(uint32_t n, uint32_t k, std::vector<uint32_t>* precomp) {
    static std::mutex mu;
    static std::map<std::pair<uint32_t, uint32_t>, std::shared_ptr<const std::vector<uint32_t>>> cache;
    std::shared_ptr<const std::vector<uint32_t>> tab;
    {
        std::lock_guard<std::mutex> lk(mu);
        auto it = cache.find({n, k});
        if (it != cache.end())
            tab = it->second;
    }
    if (!tab) {
        const uint32_t logn = (uint32_t)std::round(std::log2(n)), logm = (uint32_t)std::round(std::log2(2.0 * n));  // (:266-267)
        const uint64_t m    = 1ull << logm;
        auto rev = [logn](uint32_t x) {
            uint32_t y = 0;
            for (uint32_t b = 0; b < logn; ++b)
                y |= ((x >> b) & 1u) << (logn - 1u - b);
            return y;
        };
        auto t = std::make_shared<std::vector<uint32_t>>(n);
        for (uint32_t j = 0; j < n; ++j) {
            const uint32_t idx = (uint32_t)(((2ull * j + 1ull) * k) & (m - 1ull)) >> 1;
            (*t)[rev(j)]       = rev(idx);
        }
        tab = t;
        std::lock_guard<std::mutex> lk(mu);
        if (cache.size() < 1024)  // (a few hundred rotation indices per context at most; 4n bytes each)
            cache.emplace(std::make_pair(n, k), tab);
    }
    std::copy(tab->begin(), tab->end(), precomp->begin());
}
}  // namespace lbcrypto

extern "C" void fhe_hal_stats(uint64_t out[4]) {
    auto& r = lbcrypto::hiprt::rt();
    out[0] = r.deviceOps, out[1] = r.hostFallbacks, out[2] = r.h2dBytes, out[3] = r.d2hBytes;
}
extern "C" int fhe_hal_available(void) { return lbcrypto::hiprt::Available() ? 1 : 0; }
extern "C" void fhe_hal_trace_reset(void) {
    std::lock_guard<std::mutex> lk(lbcrypto::hiprt::g_traceMutex);
    if (lbcrypto::hiprt::g_traceSites)
        lbcrypto::hiprt::g_traceSites->clear();
}
