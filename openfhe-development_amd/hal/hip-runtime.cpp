// hip-runtime.cpp — runtime of the HIP backend of lbcrypto::DCRTPoly (see lattice/hal/hip/hip-runtime.h).
// Compiled into libOPENFHEcore when OpenFHE is built with this repo's lattice/lat-hal.h in front of the reference's.
//
// The device library is loaded at first use with dlopen: $FHE_HIP_LIB, else the path compiled in as FHE_HIP_DEFAULT_LIB
// (libfhe_hip.so of this repo).  If it cannot be loaded or no device is visible the first DCRTPoly operation FAILS LOUDLY (message
// on stderr + exception): the backend has no silent CPU path.  FHE_HAL_ALLOW_HOST=1 opts into running every member on the host
// mirror instead (the class then behaves exactly like the default backend) — for machines without a GPU, never for measurements.
// FHE_HAL_REQUIRE_DEVICE=1 is the opposite switch: a member that HAS a device path and nevertheless executes on the host mirror
// (a modulus outside the library's domain, a failed table build ...) throws instead of degrading silently.
#include "lattice/hal/hip/hip-runtime.h"
#include "hip-hooks.h"

#include <chrono>
#include <cxxabi.h>
#include <dlfcn.h>
#include <execinfo.h>
#include <sched.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "utils/exception.h"
#include "math/distributiongenerator.h"  // PseudoRandomNumberGenerator (the device sampler's seed is drawn from the reference's PRNG)

namespace lbcrypto {
namespace hiprt {

// a device context and every plan built on it.  The registry keeps the current one of a ring dimension; an operation keeps the one
// it resolved against (Resolved::hold); when a context has been replaced (new moduli appeared) and its last operation is over,
// the context, its twiddle tables and its plans go away (hipFree waits for the kernels still queued on them).
struct CtxHolder {
    fhe_ctx* ctx = nullptr;
    std::mutex mu;
    std::map<std::vector<uint64_t>, fhe_conv*> convs;      // key = {nSrc, nDst, exact, idx..., table words...}
    std::map<std::vector<uint64_t>, fhe_sr_plan*> srPlans;
    std::map<std::vector<uint64_t>, fhe_behz*> behzPlans;  // key = {which, t, numQ, qIdx..., bskIdx..., table words...}
    ~CtxHolder();
};

namespace {
// ---- streams: one per host thread ----
constexpr uint32_t kMaxStreams = 1024;
struct StreamState {
    void* s = nullptr;
    bool made = false;
    std::atomic<uint64_t> issued{0};    // operations started by the owning thread
    std::atomic<uint64_t> enqueued{0};  // ... whose launches have all been enqueued (published by the outermost Op)
    // buffers released by OTHER threads whose pending uses are all on this stream: the owning thread reuses them without any
    // cross-stream wait (its later launches follow its earlier ones anyway)
    std::atomic<bool> owned{false};
    std::atomic<uint32_t> inboxCount{0};
    std::mutex inboxMutex;
    std::vector<std::pair<size_t, uint64_t*>> inbox;  // {bucket, pointer}
    // the owning thread's cache of released buffers (ThreadState::freeLists) is guarded by flMutex: under memory pressure ANOTHER
    // thread may take it away (Alloc: a thread that cannot allocate returns every thread's cached buffers to the device — a batch
    // evaluated first over eight host threads and then in lockstep on one would otherwise keep eight caches of tower-sized buffers)
    std::mutex flMutex;
    struct ThreadState* ownerState = nullptr;  // the live thread that owns the stream (under flMutex)
};
struct Runtime {
    Api api{};
    size_t (*launch_stats)(char*, size_t, uint64_t*) = nullptr;  // (optional: the library's launch counters)
    const char* (*version)() = nullptr;                           // (optional: names the test-only lane emulator build)
    bool live = false;
    std::string why;
    int device       = 0;
    fhe_ctx* anyCtx  = nullptr;  // a minimal context of the device (allocation, copies and stream calls want a handle)
    // streams
    StreamState streams[kMaxStreams];
    std::mutex streamMutex;
    std::vector<uint32_t> freeStreamIds;
    uint32_t nextStreamId = 1;  // 0 = "no stream": host-complete
    // allocator: buffers nobody's stream has pending work on (handed over by exiting threads)
    std::mutex poolMutex;
    std::map<size_t, std::vector<uint64_t*>> orphanLists;  // bucket (words) -> free buffers
    // contexts
    struct Universe {
        uint32_t logN = 0;
        std::shared_ptr<CtxHolder> cur;
        std::vector<uint64_t> q, psi;
        std::unordered_map<uint64_t, uint32_t> limbOf;  // modulus -> context limb
    };
    std::mutex ctxMutex;
    std::map<uint32_t, Universe> universes;  // by ring dimension
    // the holder of every live context (plan look-ups go ctx -> holder; the caller of a look-up holds a reference)
    std::mutex holderMutex;
    std::map<fhe_ctx*, CtxHolder*> holders;
    std::atomic<uint64_t> deviceOps{0}, hostFallbacks{0}, h2dBytes{0}, d2hBytes{0}, hostSmallRing{0}, hostData{0};
    // why an operation left the device library's domain (Resolve): a ring outside [16, 2^17] or not a power of two; a modulus that is
    // not a prime-shaped NTT modulus below 2^60; more than 256 distinct moduli in one operation; a second root of unity for a modulus
    std::atomic<uint64_t> outRing{0}, outModulus{0}, outMoreThan128Moduli{0}, outOtherRoot{0};
    // operand bytes of the device operations: every tower (or key, or table held in a DevBuf) an operation reads / writes, counted once
    // per operation — what the sequence of fused operations has to move if every operand crossed HBM exactly once (the algorithmic
    // bytes of a composite's roofline)
    std::atomic<uint64_t> opReadBytes{0}, opWriteBytes{0};
    // bytes of released buffers the threads keep for reuse (free lists, inboxes, orphans).  A batch evaluated over eight host threads and
    // then in lockstep filled the 288 GB with caches until a kernel LAUNCH failed for want of memory (session g): an allocation that has
    // to go to the device first checks that a reserve stays free and otherwise takes every thread's cache back (Alloc)
    std::atomic<uint64_t> cachedBytes{0};
    std::atomic<uint64_t> stolen{0};          // allocations served from ANOTHER thread's free lists (Alloc)
    std::atomic<uint64_t> stolenExact{0};     // ... of which the taker waited for the buffer's own completion mark (not the owner's whole queue)
    std::atomic<uint64_t> deviceBytes{0};     // bytes obtained from the device and not given back (live towers + every cache)
    std::atomic<uint64_t> deviceBytesHigh{0}; // ... its high-water mark
    std::atomic<uint64_t> deviceMallocs{0}, cacheReleases{0};  // requests that reached the device; times the caches went back to it
    uint64_t cacheCap = ~0ull;               // FHE_HAL_CACHE_CAP_GB: a hard cap on the cached bytes (default: none)
    uint64_t reserveBytes = 8ull << 30;       // FHE_HAL_RESERVE_GB: free device memory kept for kernel launches (scratch, kernel arguments)
    bool requireDevice = false;
};

template <typename F>
bool sym(void* h, const char* name, F* out) {
    *out = reinterpret_cast<F>(dlsym(h, name));
    return *out != nullptr;
}
std::atomic<int> g_deviceOverride{-1};
// set by an atexit handler that is registered right after the device library (and with it the HIP runtime) was loaded, hence runs
// BEFORE the HIP runtime's own exit handlers: from then on no destructor of this backend calls into the device library any more
// (objects with static storage — crypto contexts, cached plaintexts — are destroyed after the HIP runtime is gone; what they hold is
// left to the process's end)
std::atomic<bool> g_exiting{false};

Runtime* build() {
    auto* r         = new Runtime;
    const char* env = std::getenv("FHE_HIP_LIB");
#ifdef FHE_HIP_DEFAULT_LIB
    const std::string path = env ? env : FHE_HIP_DEFAULT_LIB;
#else
    const std::string path = env ? env : "libfhe_hip.so";
#endif
    if (const char* cap = std::getenv("FHE_HAL_CACHE_CAP_GB"))
        r->cacheCap = (uint64_t)std::max(1, std::atoi(cap)) << 30;
    if (const char* res = std::getenv("FHE_HAL_RESERVE_GB"))
        r->reserveBytes = (uint64_t)std::max(0, std::atoi(res)) << 30;
    r->requireDevice = std::getenv("FHE_HAL_REQUIRE_DEVICE") != nullptr && std::string(std::getenv("FHE_HAL_REQUIRE_DEVICE")) != "0";
    r->device        = g_deviceOverride.load() >= 0 ? g_deviceOverride.load() : (std::getenv("FHE_HIP_DEVICE") ? std::atoi(std::getenv("FHE_HIP_DEVICE")) : 0);
    void* h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!h) {
        r->why = std::string("cannot load ") + path + ": " + dlerror();
    }
    else {
        Api& a = r->api;
        r->launch_stats = reinterpret_cast<size_t (*)(char*, size_t, uint64_t*)>(dlsym(h, "fhe_launch_stats"));
        r->version      = reinterpret_cast<const char* (*)()>(dlsym(h, "fhe_version"));
#define FHE_SYM(field, name) sym(h, #name, &a.field)
        bool ok = FHE_SYM(last_error, fhe_last_error) && FHE_SYM(device_count, fhe_device_count) && FHE_SYM(ctx_create, fhe_ctx_create) &&
                  FHE_SYM(ctx_destroy, fhe_ctx_destroy) && FHE_SYM(conv_destroy, fhe_conv_destroy) && FHE_SYM(sr_plan_destroy, fhe_sr_plan_destroy) &&
                  FHE_SYM(behz_destroy, fhe_behz_destroy) && FHE_SYM(ks_plan_destroy, fhe_ks_plan_destroy) &&
                  FHE_SYM(keyswitch_hybrid_acc, fhe_keyswitch_hybrid_acc) && FHE_SYM(ks_precompute, fhe_ks_precompute) &&
                  FHE_SYM(ks_fast_keyswitch, fhe_ks_fast_keyswitch) &&
                  FHE_SYM(malloc_, fhe_malloc) && FHE_SYM(free_, fhe_free) && FHE_SYM(h2d, fhe_memcpy_h2d) && FHE_SYM(d2h, fhe_memcpy_d2h) &&
                  FHE_SYM(d2d, fhe_memcpy_d2d) && FHE_SYM(memset_zero, fhe_memset_zero) && FHE_SYM(sync, fhe_stream_sync) &&
                  FHE_SYM(stream_create, fhe_stream_create) && FHE_SYM(stream_wait, fhe_stream_wait) && FHE_SYM(ntt_fwd, fhe_ntt_fwd) &&
                  FHE_SYM(ntt_inv, fhe_ntt_inv) && FHE_SYM(ntt_inv_oop, fhe_ntt_inv_oop) && FHE_SYM(ntt_fwd_oop, fhe_ntt_fwd_oop) &&
                  FHE_SYM(inner_product, fhe_inner_product) && FHE_SYM(add, fhe_add) && FHE_SYM(sub, fhe_sub) && FHE_SYM(mul, fhe_mul) &&
                  FHE_SYM(neg, fhe_neg) && FHE_SYM(mul_add, fhe_mul_add) && FHE_SYM(mul_const, fhe_mul_const) && FHE_SYM(mult_acc, fhe_mult_acc) &&
                  FHE_SYM(add_const, fhe_add_const) && FHE_SYM(sub_const, fhe_sub_const) && FHE_SYM(times_q_over_t, fhe_times_q_over_t) &&
                  FHE_SYM(mod_switch_round, fhe_mod_switch_round) && FHE_SYM(automorph, fhe_automorph) &&
                  FHE_SYM(switch_modulus, fhe_switch_modulus) && FHE_SYM(crt_decompose_towers, fhe_crt_decompose_towers) &&
                  FHE_SYM(crt_decompose, fhe_crt_decompose) && FHE_SYM(rescale_limbs, fhe_rescale_limbs) &&
                  FHE_SYM(event_create, fhe_event_create) && FHE_SYM(event_record, fhe_event_record) && FHE_SYM(stream_wait_event, fhe_stream_wait_event) &&
                  FHE_SYM(event_destroy, fhe_event_destroy) && FHE_SYM(sample_uniform, fhe_sample_uniform) && FHE_SYM(sample_gaussian, fhe_sample_gaussian) && FHE_SYM(sample_ternary, fhe_sample_ternary) &&
                  FHE_SYM(rescale_limbs_pair, fhe_rescale_limbs_pair) && FHE_SYM(add_pair, fhe_add_pair) && FHE_SYM(sub_pair, fhe_sub_pair) &&
                  FHE_SYM(mul_const_pair, fhe_mul_const_pair) && FHE_SYM(lincomb, fhe_lincomb) && FHE_SYM(mem_info, fhe_mem_info) &&
                  FHE_SYM(rescale_workspace_bytes, fhe_rescale_workspace_bytes) && FHE_SYM(conv_create_custom, fhe_conv_create_custom) &&
                  FHE_SYM(approx_switch_basis, fhe_approx_switch_basis) && FHE_SYM(switch_basis_exact, fhe_switch_basis_exact) &&
                  FHE_SYM(sr_plan_create, fhe_sr_plan_create) && FHE_SYM(scale_and_round, fhe_scale_and_round) &&
                  FHE_SYM(scale_and_round_p_over_q, fhe_scale_and_round_p_over_q) && FHE_SYM(scale_and_round_native, fhe_scale_and_round_native) &&
                  FHE_SYM(scale_and_round_behz_decrypt, fhe_scale_and_round_behz_decrypt) && FHE_SYM(behz_create, fhe_behz_create) &&
                  FHE_SYM(behz_override_q_to_bsk, fhe_behz_override_q_to_bsk) && FHE_SYM(behz_override_floorq, fhe_behz_override_floorq) &&
                  FHE_SYM(behz_override_conv_sk, fhe_behz_override_conv_sk) && FHE_SYM(behz_workspace_bytes, fhe_behz_workspace_bytes) &&
                  FHE_SYM(behz_q_to_bsk, fhe_behz_q_to_bsk) && FHE_SYM(behz_floorq, fhe_behz_floorq) && FHE_SYM(behz_conv_sk, fhe_behz_conv_sk) &&
                  FHE_SYM(tensor, fhe_tensor) && FHE_SYM(tensor_square, fhe_tensor_square) && FHE_SYM(ks_plan_create, fhe_ks_plan_create) &&
                  FHE_SYM(ks_key_wrap, fhe_ks_key_wrap) && FHE_SYM(ks_key_destroy, fhe_ks_key_destroy) &&
                  FHE_SYM(ks_workspace_bytes, fhe_ks_workspace_bytes) && FHE_SYM(keyswitch_hybrid, fhe_keyswitch_hybrid) &&
                  FHE_SYM(ckks_eval_mult, fhe_ckks_eval_mult) && FHE_SYM(bsgs_workspace_bytes, fhe_ckks_bsgs_workspace_bytes) &&
                  FHE_SYM(bsgs_transform, fhe_ckks_bsgs_transform) && FHE_SYM(checksum, fhe_checksum);
#undef FHE_SYM
        if (!ok)
            r->why = path + " does not export the C ABI of include/fhe_hip.h";
        else if (a.device_count() < 1)
            r->why = path + ": no HIP device visible";
        else if (r->device < 0 || r->device >= a.device_count())
            r->why = path + ": device " + std::to_string(r->device) + " requested (FHE_HIP_DEVICE / fhe_hal_set_device), " +
                     std::to_string(a.device_count()) + " visible";
        else {
            // the smallest context the library builds (N = 16, q = 97 = 1 mod 32, psi = a primitive 32nd root of unity mod 97): its
            // only purpose is to name the device in calls that take a context for that (allocation, copies, streams)
            const uint64_t q = 97;
            uint64_t psi     = 0;
            for (uint64_t g = 2; g < q && !psi; ++g) {
                uint64_t p16 = 1;
                for (int i = 0; i < 16; ++i)
                    p16 = p16 * g % q;
                if (p16 == q - 1)  // g^16 = -1: order exactly 32
                    psi = g;
            }
            if (a.ctx_create(4, 1, &q, &psi, r->device, &r->anyCtx) != FHE_OK)
                r->why = path + ": " + a.last_error();
            else {
                r->live = true;
                std::atexit([] { g_exiting.store(true); });
            }
        }
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
CtxHolder& holder_of(fhe_ctx* ctx) {
    Runtime& r = rt();
    std::lock_guard<std::mutex> lk(r.holderMutex);
    auto it = r.holders.find(ctx);
    if (it == r.holders.end())
        OPENFHE_THROW("HIP backend: plan requested on a context the registry does not know");
    return *it->second;
}
uint32_t log2u(uint32_t n) {
    uint32_t l = 0;
    while ((1u << l) < n)
        ++l;
    return l;
}

// ---- per-thread state: the thread's stream, what it has already waited for, its free lists ----
size_t bucket_of(size_t words) {
    size_t b = 1024;
    while (b < words)
        b <<= 1;
    if (b > (1u << 27)) {  // above 1 GiB: sixteenths of the power of two (a 34.9 GiB workspace takes 36 GiB, not 48: round 5)
        const size_t g = b >> 4;
        return (words + g - 1) / g * g;
    }
    if (b > (1u << 20) && words <= b - (b >> 2))  // above 8 MiB: 3/4 steps, so that odd tower heights do not waste 2x
        b -= b >> 2;
    return b;
}
struct ThreadState {
    uint32_t id   = 0;
    uint32_t depth = 0;                 // nesting of Op objects on this thread
    uint64_t outerSeq = 0;
    std::vector<uint64_t> waited;       // per stream id: everything enqueued there up to this seq precedes this thread's later work
    std::map<size_t, std::vector<uint64_t*>> freeLists;
    // completion mark of a cached buffer (round 6): an event recorded on THIS thread's stream when the buffer joined the free lists, i.e.
    // behind every pending use of it.  Another thread that takes the buffer (Alloc's last resort before the device) waits for that event —
    // not for everything this stream has enqueued by then, which made two busy lockstep groups run one behind the other (round 5: 16x4
    // groups 37-40 bootstraps/s against 51).  Guarded by the stream's flMutex, like the lists.
    std::unordered_map<uint64_t*, void*> marks;
    std::vector<void*> eventPool;
    void* NewEvent() {
        if (!eventPool.empty()) {
            void* e = eventPool.back();
            eventPool.pop_back();
            return e;
        }
        void* e = nullptr;
        Runtime& r = rt();
        return r.api.event_create(r.anyCtx, &e) == FHE_OK ? e : nullptr;
    }
    ThreadState() {
        Runtime& r = rt();
        std::lock_guard<std::mutex> lk(r.streamMutex);
        if (!r.freeStreamIds.empty()) {
            id = r.freeStreamIds.back();
            r.freeStreamIds.pop_back();
        }
        else {
            if (r.nextStreamId >= kMaxStreams)
                OPENFHE_THROW("HIP backend: more than 1023 host threads with device work alive at once");
            id = r.nextStreamId++;
        }
        StreamState& st = r.streams[id];
        if (!st.made) {
            Check(r.api.stream_create(r.anyCtx, &st.s), "HIP backend: stream for a host thread");
            st.made = true;
        }
        waited.assign(kMaxStreams, 0);
        st.owned.store(true);
        std::lock_guard<std::mutex> fl(st.flMutex);
        st.ownerState = this;
    }
    void TakeInbox() {
        StreamState& st = rt().streams[id];
        if (st.inboxCount.load(std::memory_order_acquire) == 0)
            return;
        std::vector<std::pair<size_t, uint64_t*>> in;
        {
            std::lock_guard<std::mutex> lk(st.inboxMutex);
            in.swap(st.inbox);
            st.inboxCount.store(0, std::memory_order_release);
        }
        std::lock_guard<std::mutex> fl(st.flMutex);
        for (auto& e : in)
            freeLists[e.first].push_back(e.second);
    }
    ~ThreadState() {
        if (g_exiting.load())
            return;
        // The thread ends.  No call into the HIP runtime from here (thread-local destructors run when tools layered under the runtime —
        // rocprofv3's per-thread state — may already be gone): the cached buffers stay with the STREAM, parked in its inbox, and the
        // next thread that takes the stream over (ids are reused last-in first-out; the sequence counters keep counting, so old
        // stamps stay "enqueued") finds them there — whatever is still pending on them precedes that thread's launches on the same stream.
        Runtime& r      = rt();
        StreamState& st = r.streams[id];
        st.owned.store(false);
        {
            std::lock_guard<std::mutex> fl(st.flMutex);
            st.ownerState = nullptr;
            std::lock_guard<std::mutex> lk(st.inboxMutex);
            for (auto& kv : freeLists)
                for (uint64_t* p : kv.second)
                    st.inbox.emplace_back(kv.first, p);
            st.inboxCount.store((uint32_t)st.inbox.size(), std::memory_order_release);
        }
        std::lock_guard<std::mutex> lk(r.streamMutex);
        r.freeStreamIds.push_back(id);
    }
};
thread_local bool t_dead = false;
struct ThreadHolder {
    ThreadState* ts = nullptr;
    ~ThreadHolder() {
        delete ts;
        ts     = nullptr;
        t_dead = true;
    }
};
thread_local ThreadHolder t_holder;
ThreadState* thread_state() {  // nullptr once the thread's state has been destroyed (static destruction at exit)
    if (t_dead)
        return nullptr;
    if (!t_holder.ts)
        t_holder.ts = new ThreadState;
    return t_holder.ts;
}
// the calling thread's later work must come after use `u` of a buffer.  Never called with a buffer's mutex held: the wait below is for
// ANOTHER thread to close the operation that made the use (its launch is enqueued when its outermost Op closes), and that thread may
// need the same mutex on the way.  The use is recorded as ordered only up to the mark actually observed; two threads that each wait for
// a use the other has stamped but not yet enqueued (host code racing on the same towers) end in the exception below, not in words
// computed before their producer ran.
void order_after(ThreadState* ts, const DevBuf::Use& u) {
    Runtime& r = rt();
    if (u.stream == 0 || u.stream == ts->id || u.seq <= ts->waited[u.stream])
        return;
    StreamState& other = r.streams[u.stream];
    uint64_t mark      = other.enqueued.load(std::memory_order_acquire);
    const auto t0      = std::chrono::steady_clock::now();
    // A slow producer is not a race: the lane emulator runs launches synchronously under one mutex, an Op may sit in Alloc while the
    // caches go back to the device, a table upload blocks.  The limit is generous (FHE_HAL_ORDER_TIMEOUT_S, default 120 s on a device,
    // 1800 s on the emulator) and only says "somebody stamped a use and never enqueued it".
    static const int64_t limitS = [] {
        if (const char* e = std::getenv("FHE_HAL_ORDER_TIMEOUT_S"))
            return (int64_t)std::max(1, std::atoi(e));
        const char* v = rt().version ? rt().version() : nullptr;
        return (int64_t)((v && std::strstr(v, "emulator")) ? 1800 : 120);
    }();
    for (uint32_t spin = 0; mark < u.seq; ++spin) {  // (the other thread is still inside the operation that made this use: it is about to finish)
        if ((spin & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(limitS))
            OPENFHE_THROW("HIP backend: a use of a tower stamped by another host thread was not enqueued within " + std::to_string(limitS) +
                          " s (host threads racing on the same DCRTPoly, or a thread blocked inside a device operation)");
        sched_yield();
        mark = other.enqueued.load(std::memory_order_acquire);
    }
    Check(r.api.stream_wait(r.anyCtx, r.streams[ts->id].s, other.s), "HIP backend: ordering two host threads' streams");
    ts->waited[u.stream] = mark;  // everything the other stream had enqueued when observed (>= u.seq) precedes this thread's later work
}
DevBuf* root_of(DevBuf* b) {
    while (b->parent)
        b = b->parent.get();
    return b;
}
}  // namespace

CtxHolder::~CtxHolder() {
    if (g_exiting.load())
        return;
    Runtime& r = rt();
    {
        std::lock_guard<std::mutex> lk(r.holderMutex);
        r.holders.erase(ctx);
    }
    for (auto& kv : convs)
        r.api.conv_destroy(kv.second);
    for (auto& kv : srPlans)
        r.api.sr_plan_destroy(kv.second);
    for (auto& kv : behzPlans)
        r.api.behz_destroy(kv.second);
    r.api.ctx_destroy(ctx);
}
bool Available() { return rt().live; }
bool DeviceSamplerEnabled() {
    static const bool on = [] {
        const char* v = std::getenv("FHE_HAL_DEVICE_SAMPLER");
        return v && v[0] == '1';
    }();
    return on && Available();
}
void DeviceSamplerStream(uint64_t* seed, uint32_t* streamId) {
    static std::once_flag once;
    static uint64_t s = 0;
    static std::atomic<uint32_t> next{1};
    std::call_once(once, [] {
        auto& g = lbcrypto::PseudoRandomNumberGenerator::GetPRNG();
        s       = ((uint64_t)g() << 32) | (uint64_t)g();
    });
    *seed     = s;
    *streamId = next.fetch_add(1, std::memory_order_relaxed);
}
const Api& api() { return rt().api; }
int Device() { return rt().device; }
fhe_ctx* AnyCtx() { return rt().anyCtx; }
void Check(fhe_status s, const char* what) {
    if (s != FHE_OK)
        OPENFHE_THROW(std::string(what) + ": " + rt().api.last_error());
}

// ---- operations and buffers ----
Op::Op() {
    ThreadState* ts = thread_state();
    if (!ts)
        OPENFHE_THROW("HIP backend: device operation on a thread that is shutting down");
    Runtime& r = rt();
    s          = r.streams[ts->id].s;
    m_seq      = r.streams[ts->id].issued.fetch_add(1, std::memory_order_relaxed) + 1;
    if (ts->depth++ == 0)
        ts->outerSeq = m_seq;
}
Op::~Op() {
    ThreadState* ts = thread_state();
    if (ts && --ts->depth == 0) {
        StreamState& st = rt().streams[ts->id];
        st.enqueued.store(st.issued.load(std::memory_order_relaxed), std::memory_order_release);
    }
}
static void count_operand_bytes(uint64_t bytes, bool write);  // (per member scope, below)
// (both stamp the buffer under its mutex and order the calling stream behind the earlier uses AFTER releasing it: the stamp only says
// "this stream, this sequence number will touch the words"; the launch follows when the caller returns from R / W)
const uint64_t* Op::R(const Buf& b) {
    ThreadState* ts = thread_state();
    DevBuf* root    = root_of(b.get());
    DevBuf::Use writer;
    {
        std::lock_guard<std::mutex> lk(root->mu);
        writer     = root->writer;
        bool found = false;
        for (auto& u : root->readers)
            if (u.stream == ts->id) {
                u.seq = m_seq;
                found = true;
                break;
            }
        if (!found)
            root->readers.push_back(DevBuf::Use{ts->id, m_seq});
    }
    order_after(ts, writer);
    count_operand_bytes((uint64_t)b->words * 8, false);
    return b->p;
}
uint64_t* Op::W(const Buf& b, bool operand) {
    ThreadState* ts = thread_state();
    DevBuf* root    = root_of(b.get());
    std::vector<DevBuf::Memo> history;  // (released after the lock)
    std::vector<DevBuf::Use> before;
    {
        std::lock_guard<std::mutex> lk(root->mu);
        before.push_back(root->writer);
        before.insert(before.end(), root->readers.begin(), root->readers.end());
        root->readers.clear();
        root->writer = DevBuf::Use{ts->id, m_seq};
        history.swap(root->memo);  // the words change: what was derived from them is history
    }
    try {
        for (const auto& u : before)
            order_after(ts, u);
    }
    catch (...) {  // (the buffer keeps what it knew about its pending uses: whoever touches it next still orders behind them)
        std::lock_guard<std::mutex> lk(root->mu);
        root->readers.insert(root->readers.end(), before.begin(), before.end());
        throw;
    }
    if (operand)
        count_operand_bytes((uint64_t)b->words * 8, true);
    return b->p;
}
std::atomic<uint64_t> g_memoHits{0};
static thread_local uint32_t t_width = 1;
uint32_t ThreadWidth() { return t_width; }
WidthScope::WidthScope(uint32_t k) : saved(t_width) { t_width = k ? k : 1; }
WidthScope::~WidthScope() { t_width = saved; }
Buf MemoFind(const Buf& src, const std::vector<uint64_t>& key) {
    if (!src || src->parent)
        return nullptr;
    std::lock_guard<std::mutex> lk(src->mu);
    for (const auto& m : src->memo)
        if (m.key == key) {
            g_memoHits.fetch_add(1, std::memory_order_relaxed);
            return m.result;
        }
    return nullptr;
}
// the buffers that hold remembered results: under memory pressure Alloc drops every one of them (DropMemos) — a remembered result pins a
// tower-sized buffer for as long as its source lives
static std::mutex g_memoOwnersMutex;
static std::vector<std::weak_ptr<DevBuf>> g_memoOwners;
void MemoStore(const Buf& src, std::vector<uint64_t> key, const Buf& result) {
    if (!src || src->parent || !result)
        return;
    DevBuf::Memo evicted;
    bool first = false;
    {
        std::lock_guard<std::mutex> lk(src->mu);
        first = src->memo.empty();
        if (src->memo.size() >= 4) {
            evicted = std::move(src->memo.front());
            src->memo.erase(src->memo.begin());
        }
        src->memo.push_back(DevBuf::Memo{std::move(key), result});
    }
    if (first) {
        std::lock_guard<std::mutex> lk(g_memoOwnersMutex);
        if (g_memoOwners.size() >= 4096 && (g_memoOwners.size() & (g_memoOwners.size() - 1)) == 0)  // (at every doubling: forget the dead)
            g_memoOwners.erase(std::remove_if(g_memoOwners.begin(), g_memoOwners.end(), [](const std::weak_ptr<DevBuf>& w) { return w.expired(); }),
                               g_memoOwners.end());
        g_memoOwners.push_back(src);
    }
}
// every remembered result goes (their buffers join the calling thread's free lists, which the caller returns to the device next)
static void DropMemos() {
    std::vector<std::weak_ptr<DevBuf>> owners;
    {
        std::lock_guard<std::mutex> lk(g_memoOwnersMutex);
        owners.swap(g_memoOwners);
    }
    for (auto& w : owners)
        if (Buf b = w.lock()) {
            std::vector<DevBuf::Memo> gone;  // (released after the lock)
            std::lock_guard<std::mutex> lk(b->mu);
            gone.swap(b->memo);
        }
}
void Op::HostSync() {
    Runtime& r = rt();
    Check(r.api.sync(r.anyCtx, s), "HIP backend: stream synchronisation");
}

DevBuf::~DevBuf() {
    if (parent)
        parent->views.fetch_sub(1, std::memory_order_relaxed);
    if (!p || parent || external || g_exiting.load())
        return;
    Runtime& r      = rt();
    ThreadState* ts = thread_state();
    const size_t bucket  = cap ? cap : bucket_of(words);
    const uint64_t bytes = (uint64_t)bucket * 8;
    if (!ts || r.cachedBytes.load(std::memory_order_relaxed) + bytes > r.cacheCap) {
        // static destruction at process exit (no thread state), or the caches are full: wait for the pending uses on the host, then park
        // the buffer with the orphans / give it back to the device
        if (writer.stream)
            r.api.sync(r.anyCtx, r.streams[writer.stream].s);
        for (const auto& u : readers)
            r.api.sync(r.anyCtx, r.streams[u.stream].s);
        if (ts) {
            r.api.free_(r.anyCtx, p);
            r.deviceBytes.fetch_sub(std::min<uint64_t>(bytes, r.deviceBytes.load()), std::memory_order_relaxed);
            return;
        }
        std::lock_guard<std::mutex> lk(r.poolMutex);
        r.orphanLists[bucket].push_back(p);
        r.cachedBytes.fetch_add(bytes, std::memory_order_relaxed);
        return;
    }
    // Pending uses all on ONE other thread's stream (a result computed by thread A and dropped by thread B: pke's loops over
    // ciphertexts with dynamic schedules): the buffer goes back to that thread.  Ordering this thread's stream behind them instead
    // would make it wait for everything the other stream has enqueued so far — streams of a batch then run in lock step.
    static const bool toOwner = !(std::getenv("FHE_HAL_FREE_TO_OWNER") && std::string(std::getenv("FHE_HAL_FREE_TO_OWNER")) == "0");
    uint32_t owner = writer.stream;
    bool single    = owner != 0;
    for (const auto& u : readers) {
        if (owner == 0)
            owner = u.stream, single = owner != 0;
        else if (u.stream != owner)
            single = false;
    }
    if (toOwner && single && owner != ts->id) {
        const uint64_t last = std::max(writer.stream == owner ? writer.seq : 0, [&] {
            uint64_t m = 0;
            for (const auto& u : readers)
                m = std::max(m, u.seq);
            return m;
        }());
        StreamState& st = r.streams[owner];
        if (last > ts->waited[owner] && st.owned.load()) {
            std::lock_guard<std::mutex> lk(st.inboxMutex);
            st.inbox.emplace_back(bucket, p);
            st.inboxCount.fetch_add(1, std::memory_order_release);
            r.cachedBytes.fetch_add(bytes, std::memory_order_relaxed);
            return;
        }
    }
    // the buffer joins THIS thread's free lists: whatever this thread launches later is ordered behind the buffer's pending uses
    order_after(ts, writer);
    for (const auto& u : readers)
        order_after(ts, u);
    std::lock_guard<std::mutex> fl(r.streams[ts->id].flMutex);
    ts->freeLists[bucket].push_back(p);
    static const bool marksOn = !(std::getenv("FHE_HAL_RELEASE_MARKS") && std::string(std::getenv("FHE_HAL_RELEASE_MARKS")) == "0");
    if (marksOn && bytes >= (1u << 20)) {  // (small buffers are not worth an event: a taker orders behind the stream's tail, as before)
        if (void* e = ts->NewEvent()) {
            if (r.api.event_record(r.anyCtx, e, r.streams[ts->id].s) == FHE_OK)
                ts->marks[p] = e;
            else
                ts->eventPool.push_back(e);
        }
    }
    r.cachedBytes.fetch_add(bytes, std::memory_order_relaxed);
}
// Everything the backend holds beyond live towers goes back to the device: the remembered results, every live thread's free lists (under
// its stream's mutex, after synchronising that stream), the orphans and what exited threads left parked.  Alloc calls it under memory
// pressure; fhe_hal_release_caches() lets a process that is done with a batch hand the device to the next one (bench.py between legs).
static void ReleaseCaches(bool dropMemos = true) {
    Runtime& r = rt();
    r.cacheReleases.fetch_add(1, std::memory_order_relaxed);
    if (dropMemos)
        DropMemos();
    uint64_t freed = 0;
    uint32_t nStreams;
    {
        std::lock_guard<std::mutex> lk(r.streamMutex);
        nStreams = r.nextStreamId;
    }
    for (uint32_t i = 1; i < nStreams; ++i) {  // every live thread's cache (this thread's included)
        StreamState& st = r.streams[i];
        std::vector<uint64_t*> taken;
        {
            std::lock_guard<std::mutex> flk(st.flMutex);
            if (!st.ownerState)
                continue;
            for (auto& kv : st.ownerState->freeLists) {
                taken.insert(taken.end(), kv.second.begin(), kv.second.end());
                freed += (uint64_t)kv.first * 8 * kv.second.size();
                kv.second.clear();
            }
            for (auto& mk : st.ownerState->marks)  // (the buffers go back to the device behind a host-side wait: their marks are spent)
                st.ownerState->eventPool.push_back(mk.second);
            st.ownerState->marks.clear();
        }
        if (taken.empty())
            continue;
        r.api.sync(r.anyCtx, st.s);  // (what is still pending on the buffers is on their owner's stream)
        for (uint64_t* q : taken)
            r.api.free_(r.anyCtx, q);
    }
    {
        std::lock_guard<std::mutex> lk(r.poolMutex);
        for (auto& kv : r.orphanLists) {
            for (uint64_t* q : kv.second)
                r.api.free_(r.anyCtx, q);
            freed += (uint64_t)kv.first * 8 * kv.second.size();
            kv.second.clear();
        }
    }
    for (uint32_t i = 1; i < nStreams; ++i) {  // what other threads sent back to a stream and its thread has not taken yet (or never will: it exited)
        StreamState& st = r.streams[i];
        if (st.inboxCount.load() == 0)
            continue;
        std::vector<std::pair<size_t, uint64_t*>> in;
        {
            std::lock_guard<std::mutex> lk(st.inboxMutex);
            in.swap(st.inbox);
            st.inboxCount.store(0);
        }
        r.api.sync(r.anyCtx, st.s);
        for (auto& e : in) {
            r.api.free_(r.anyCtx, e.second);
            freed += (uint64_t)e.first * 8;
        }
    }
    r.cachedBytes.fetch_sub(std::min<uint64_t>(freed, r.cachedBytes.load()), std::memory_order_relaxed);
    r.deviceBytes.fetch_sub(std::min<uint64_t>(freed, r.deviceBytes.load()), std::memory_order_relaxed);
}
void ReleaseAllCaches() { ReleaseCaches(); }
uint64_t CachedBytes() { return rt().cachedBytes.load(std::memory_order_relaxed); }
void SyncAllStreams() {
    Runtime& r = rt();
    for (uint32_t i = 1; i < r.nextStreamId; ++i)
        Check(r.api.sync(r.anyCtx, r.streams[i].s), "HIP backend: stream synchronisation");
}
Buf Alloc(size_t words) {
    Runtime& r      = rt();
    ThreadState* ts = thread_state();
    const size_t bk = bucket_of(words);
    auto b          = std::make_shared<DevBuf>();
    b->words        = words;
    // A released buffer of the request's size class, else of the next larger classes (up to 4x): the towers of an evaluation shrink level by
    // level, so what the high levels released serves the low ones and the caches hold the evaluation's high-water mark in BYTES — filed
    // by exact class only, every class keeps its own high-water mark (several times the live bytes: 32 ciphertexts in flight filled 288 GB)
    auto take = [&](std::map<size_t, std::vector<uint64_t*>>& lists) -> bool {
        for (auto it = lists.lower_bound(bk); it != lists.end() && it->first <= 4 * bk; ++it)
            if (!it->second.empty()) {
                b->p = it->second.back();
                it->second.pop_back();
                b->cap = it->first;
                r.cachedBytes.fetch_sub((uint64_t)it->first * 8, std::memory_order_relaxed);
                return true;
            }
        return false;
    };
    if (ts) {
        ts->TakeInbox();
        std::lock_guard<std::mutex> flk(r.streams[ts->id].flMutex);
        if (take(ts->freeLists)) {
            auto mk = ts->marks.find(b->p);
            if (mk != ts->marks.end()) {  // (this thread's own launches follow the buffer's pending uses in stream order: the mark is not needed)
                ts->eventPool.push_back(mk->second);
                ts->marks.erase(mk);
            }
            // Kernels of the buffer's previous life may still be pending on THIS thread's stream (free lists and inboxes hold such
            // buffers on purpose: ~DevBuf only orders the stream).  This thread's own launches follow them in stream order; a first use
            // by ANOTHER thread (a tower allocated here and filled by an OpenMP worker) must be ordered behind them: the new buffer
            // starts its life written "by this stream, at everything issued so far".
            b->writer = DevBuf::Use{ts->id, r.streams[ts->id].issued.load(std::memory_order_relaxed)};
            return b;
        }
    }
    {
        std::lock_guard<std::mutex> lk(r.poolMutex);
        if (take(r.orphanLists))  // (orphans were parked after a host-side wait for their pending uses: nothing to order behind)
            return b;
    }
    // Another thread's cache before the device (round 5): free lists are per thread, so N host threads each kept the high-water mark of
    // their own evaluation — four lockstep groups in flight held four times the workspace of one and the device ran out while tens of
    // GB sat cached next door (16x4 / 8x4 groups: 7-20 bootstraps/s against 50).
    static const bool steal = !(std::getenv("FHE_HAL_STEAL") && std::string(std::getenv("FHE_HAL_STEAL")) == "0");
    if (steal && ts) {
        uint32_t nStreams;
        {
            std::lock_guard<std::mutex> lk(r.streamMutex);
            nStreams = r.nextStreamId;
        }
        for (uint32_t i = 1; i < nStreams; ++i) {
            if (i == ts->id)
                continue;
            StreamState& st = r.streams[i];
            std::lock_guard<std::mutex> flk(st.flMutex);
            if (!st.ownerState)
                continue;
            if (take(st.ownerState->freeLists)) {
                // The buffer's completion mark (recorded on T's stream when T cached it, behind every pending use): this thread's stream
                // waits for exactly that.  Without a mark (small buffers, buffers that came through T's inbox) it waits for everything
                // submitted to T's stream so far, as in round 5.  Either way a device-side wait, recorded now: nobody spins on the host.
                void* mark = nullptr;
                auto mk = st.ownerState->marks.find(b->p);
                if (mk != st.ownerState->marks.end()) {
                    mark = mk->second;
                    st.ownerState->marks.erase(mk);
                }
                const fhe_status ws = mark ? r.api.stream_wait_event(r.anyCtx, r.streams[ts->id].s, mark)
                                           : r.api.stream_wait(r.anyCtx, r.streams[ts->id].s, st.s);
                if (mark)
                    st.ownerState->eventPool.push_back(mark);  // (a recorded wait keeps the event's state of that moment: the object is reusable)
                if (ws != FHE_OK) {
                    // the wait could not be recorded: the buffer goes back where it was, with its pending uses unordered for nobody
                    // (round-5 advisor: `b` used to die into the taker's lists with an empty writer)
                    st.ownerState->freeLists[b->cap].push_back(b->p);
                    r.cachedBytes.fetch_add((uint64_t)b->cap * 8, std::memory_order_relaxed);
                    b->p = nullptr;
                    Check(ws, "HIP backend: ordering behind the last use of a cached buffer");
                }
                b->writer = DevBuf::Use{ts->id, r.streams[ts->id].issued.load(std::memory_order_relaxed)};
                r.stolen.fetch_add(1, std::memory_order_relaxed);
                if (mark)
                    r.stolenExact.fetch_add(1, std::memory_order_relaxed);
                return b;
            }
        }
    }
    void* d      = nullptr;
    // (the allocation goes to the device: if it would eat into the reserve kept for kernel launches while buffers sit in caches, the
    // caches go back first — a launch that fails for want of scratch memory cannot be retried from here)
    // Taking the caches back is a global stall (every stream is synchronised, every cached buffer freed): it happens only when the
    // caches hold enough to matter for THIS request (at least its size, or 1 GiB) — a device that is legitimately near-full with a
    // few KB cached would otherwise pay it on every allocation — and the remembered results go only if the buffers were not enough.
    bool pressed = false;
    const uint64_t cached = r.cachedBytes.load(std::memory_order_relaxed);
    if (cached >= std::min<uint64_t>((uint64_t)bk * 8, 1ull << 30)) {
        size_t freeB = 0, totalB = 0;
        if (r.api.mem_info(r.anyCtx, &freeB, &totalB) == FHE_OK && freeB < (uint64_t)bk * 8 + r.reserveBytes)
            pressed = true;
    }
    r.deviceMallocs.fetch_add(1, std::memory_order_relaxed);
    static const bool traceAlloc = std::getenv("FHE_HAL_TRACE_ALLOC") != nullptr;  // (tuning aid: every request that reaches the device)
    if (traceAlloc)
        fprintf(stderr, "halalloc: %zu MiB from the device (class of %zu words), %llu MiB cached\n", bk * 8 >> 20, words,
                (unsigned long long)(cached >> 20));
    fhe_status s = pressed ? FHE_ERR_ALLOC : r.api.malloc_(r.anyCtx, bk * 8, &d);
    if (s != FHE_OK) {  // memory pressure: every thread's and the shared cached buffers go back to the device, then the remembered results
        ReleaseCaches(false);
        s = r.api.malloc_(r.anyCtx, bk * 8, &d);
        if (s != FHE_OK) {
            ReleaseCaches(true);
            s = r.api.malloc_(r.anyCtx, bk * 8, &d);
        }
    }
    Check(s, "HIP backend: device allocation");
    b->p   = static_cast<uint64_t*>(d);
    b->cap = bk;
    const uint64_t now = r.deviceBytes.fetch_add((uint64_t)bk * 8, std::memory_order_relaxed) + (uint64_t)bk * 8;
    uint64_t hw = r.deviceBytesHigh.load(std::memory_order_relaxed);
    while (now > hw && !r.deviceBytesHigh.compare_exchange_weak(hw, now, std::memory_order_relaxed)) {
    }
    return b;
}
Buf WrapExternal(uint64_t* devPtr, size_t words) {
    auto b      = std::make_shared<DevBuf>();
    b->p        = devPtr;
    b->words    = words;
    b->external = true;
    return b;
}
Buf View(const Buf& parent, size_t offsetWords, size_t words) {
    auto b    = std::make_shared<DevBuf>();
    b->p      = parent->p + offsetWords;
    b->words  = words;
    b->parent = parent;
    parent->views.fetch_add(1, std::memory_order_relaxed);
    return b;
}

// ---- counters ----
namespace {
struct MemberCounters {
    std::atomic<const char*> name{nullptr};
    std::atomic<uint64_t> device{0}, host{0}, reads{0}, bytes{0};  // bytes: operand bytes of the member's device operations
};
constexpr size_t kMemberSlots = 1024;
MemberCounters g_members[kMemberSlots];
MemberCounters& member_slot(const char* name) {  // keyed by the literal's address (merged by content when reported)
    size_t h = (reinterpret_cast<uintptr_t>(name) >> 3) * 0x9E3779B97F4A7C15ull >> 54;
    for (size_t probe = 0; probe < kMemberSlots; ++probe, h = (h + 1) % kMemberSlots) {
        const char* cur = g_members[h].name.load(std::memory_order_acquire);
        if (cur == name)
            return g_members[h];
        if (!cur) {
            const char* expected = nullptr;
            if (g_members[h].name.compare_exchange_strong(expected, name) || expected == name)
                return g_members[h];
        }
    }
    return g_members[0];
}
thread_local const char* t_scope  = nullptr;
thread_local const char* t_member = nullptr;
}  // namespace
static void count_operand_bytes(uint64_t bytes, bool write) {
    (write ? rt().opWriteBytes : rt().opReadBytes).fetch_add(bytes, std::memory_order_relaxed);
    static const char* const kNoScope = "(outside a member)";
    member_slot(t_scope ? t_scope : kNoScope).bytes.fetch_add(bytes, std::memory_order_relaxed);
}
namespace {
// members of the backend class that have a device path: with FHE_HAL_REQUIRE_DEVICE they may not run on the host mirror
const char* const kDeviceMembers[] = {"SwitchFormat", "operator+=", "operator-=", "operator*=", "Plus", "Minus", "Times", "TimesNoCheck", "Negate",
                                      "operator-", "AutomorphismTransform", "ApproxSwitchCRTBasis", "ApproxModUp", "ApproxModDown", "SwitchCRTBasis",
                                      "ExpandCRTBasis", "ExpandCRTBasisReverseOrder", "FastExpandCRTBasisPloverQ", "ExpandCRTBasisQlHat",
                                      "ScaleAndRound", "ApproxScaleAndRound", "ScaleAndRoundPOverQ", "FastBaseConvqToBskMontgomery", "FastRNSFloorq",
                                      "FastBaseConvSK", "DropLastElementAndScale", "ModReduce", "CloneTowers", "TimesQovert", "AssembleRows",
                                      "InnerProduct", "MultAccRows", "DropLastElement", "DropLastElements", "SetValuesToZero"};
bool has_device_path(const char* member) {
    for (const char* m : kDeviceMembers)
        if (std::strcmp(m, member) == 0)
            return true;
    return false;
}
}  // namespace
MemberScope::MemberScope(const char* member) : outer(t_scope == nullptr) {
    if (outer)
        t_scope = member;
}
MemberScope::~MemberScope() {
    if (outer)
        t_scope = nullptr;
}
static void trace_site(const char* kind, const char* member, uint64_t amount);
void CountDevice(const char* member) {
    rt().deviceOps.fetch_add(1, std::memory_order_relaxed);
    member_slot(t_scope ? t_scope : member).device.fetch_add(1, std::memory_order_relaxed);
    trace_site("devop", t_scope ? t_scope : member, 1);
}
static std::mutex g_traceMutex;
static std::map<std::string, uint64_t>* g_traceSites = nullptr;
// FHE_HAL_TRACE=1: which callers send work to the host mirror or move words over PCIe (a tuning aid: call sites by
// frequency / by bytes at exit)
void TraceMember(const char* member) { t_member = member; }
static void trace_site(const char* kind, const char* member, uint64_t amount) {
    static const bool trace = std::getenv("FHE_HAL_TRACE") != nullptr;
    if (!trace)
        return;
    void* bt[10];
    const int n     = backtrace(bt, 10);
    std::string key = std::string(kind) + " " + (member ? member : (t_scope ? t_scope : (t_member ? t_member : ""))) + " <- ";
    for (int i = 3; i < n; ++i) {
        Dl_info info;
        if (dladdr(bt[i], &info) && info.dli_sname) {
            int st         = 0;
            char* dm       = abi::__cxa_demangle(info.dli_sname, nullptr, nullptr, &st);
            std::string nm = dm ? dm : info.dli_sname;
            free(dm);
            key += nm.substr(0, nm.find('(')).substr(0, 60) + " <- ";
        }
    }
    std::lock_guard<std::mutex> lk(g_traceMutex);
    if (!g_traceSites) {
        g_traceSites = new std::map<std::string, uint64_t>;
        std::atexit([] {
            std::map<std::string, std::vector<std::pair<uint64_t, std::string>>> byKind;  // hostop / hostread / h2dBytes / d2hBytes / d2dBytes
            for (auto& kv : *g_traceSites)
                byKind[kv.first.substr(0, kv.first.find(' '))].emplace_back(kv.second, kv.first);
            for (auto& kind : byKind) {
                auto& v = kind.second;
                std::sort(v.rbegin(), v.rend());
                for (size_t i = 0; i < v.size() && i < 120; ++i)
                    fprintf(stderr, "hal trace %12lu  %s\n", (unsigned long)v[i].first, v[i].second.c_str());
            }
        });
    }
    (*g_traceSites)[key] += amount;
}
void OtherHostCounts(uint64_t out[2]) { out[0] = rt().hostSmallRing, out[1] = rt().hostData; }
// why a device plan or a context could not be built (the member then runs on the host mirror): "<what>: <the library's message>" -> count
static std::mutex g_declineMutex;
static std::map<std::string, uint64_t> g_declines;
void Declined(const char* what, const std::string& why) {
    std::lock_guard<std::mutex> lk(g_declineMutex);
    g_declines[std::string(what) + ": " + why] += 1;
}
void CountHost(const char* member, uint32_t ringDim, bool hostData) {
    Runtime& r       = rt();
    const char* name = t_scope ? t_scope : member;
    if (hostData) {
        r.hostData.fetch_add(1, std::memory_order_relaxed);
        trace_site("hostdata", name, 1);
        return;
    }
    if (ringDim != 0 && ringDim < 16) {
        r.hostSmallRing.fetch_add(1, std::memory_order_relaxed);
        trace_site("hostop-ring<16", name, 1);
        return;
    }
    r.hostFallbacks.fetch_add(1, std::memory_order_relaxed);
    member_slot(name).host.fetch_add(1, std::memory_order_relaxed);
    if (t_scope && std::strcmp(t_scope, member) != 0) {  // (the trace names the member inside the scope it is attributed to)
        const std::string both = std::string(t_scope) + "/" + member;
        trace_site("hostop", both.c_str(), 1);
    }
    else
        trace_site("hostop", name, 1);
    if (r.requireDevice && r.live && has_device_path(name))
        OPENFHE_THROW(std::string("HIP backend: DCRTPoly::") + name + " ran on the host mirror although FHE_HAL_REQUIRE_DEVICE is set");
}
void CountHostRead(const char* member) {
    const char* name = t_scope ? t_scope : member;
    member_slot(name).reads.fetch_add(1, std::memory_order_relaxed);
    trace_site("hostread", name, 1);
}
void D2D(Op& op, uint64_t* dst, const uint64_t* src, size_t bytes, const char* what) {
    Check(rt().api.d2d(rt().anyCtx, dst, src, bytes, op.s), what);
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

// ---- contexts ----
bool Resolve(uint32_t ringDim, const std::vector<LimbSet>& sets, Resolved* out) {
    Runtime& r = rt();
    if (!r.live)
        return false;
    if (ringDim < 16 || ringDim > (1u << 17) || (ringDim & (ringDim - 1))) {
        r.outRing.fetch_add(1, std::memory_order_relaxed);
        return false;
    }
    const uint64_t twoN = 2ull * ringDim;
    for (const auto& s : sets)
        for (uint32_t i = 0; i < s.n; ++i)
            if (s.q[i] < 3 || s.q[i] >= (1ull << 60) || (s.q[i] - 1) % twoN != 0 || s.psi[i] == 0) {
                r.outModulus.fetch_add(1, std::memory_order_relaxed);
                return false;
            }
    std::lock_guard<std::mutex> lk(r.ctxMutex);
    auto& u = r.universes[ringDim];
    // fast path: every modulus is known (the usual case after the first few operations of a CryptoContext)
    bool known = u.cur != nullptr;
    for (size_t si = 0; known && si < sets.size(); ++si)
        for (uint32_t i = 0; known && i < sets[si].n; ++i)
            known = u.limbOf.count(sets[si].q[i]) != 0;
    if (!known) {
        std::vector<uint64_t> q = u.q, psi = u.psi;
        auto add = [&](std::vector<uint64_t>& qq, std::vector<uint64_t>& pp) {
            for (const auto& s : sets)
                for (uint32_t i = 0; i < s.n; ++i)
                    if (std::find(qq.begin(), qq.end(), s.q[i]) == qq.end())
                        qq.push_back(s.q[i]), pp.push_back(s.psi[i]);
        };
        add(q, psi);
        if (q.size() > kMaxDeviceLimbs) {
            // more distinct moduli than one device context holds (many CryptoContexts in one process): start over with the
            // moduli of this call; towers already on the device are plain words and resolve again at their next operation
            q.clear(), psi.clear();
            add(q, psi);
            if (q.size() > kMaxDeviceLimbs) {
                r.outMoreThan128Moduli.fetch_add(1, std::memory_order_relaxed);
                Declined("device context", "more than 256 distinct moduli in one operation (" + std::to_string(q.size()) + ")");
                return false;
            }
        }
        fhe_ctx* c = nullptr;
        if (r.api.ctx_create(log2u(ringDim), (uint32_t)q.size(), q.data(), psi.data(), r.device, &c) != FHE_OK) {
            Declined("device context", r.api.last_error());
            return false;  // (e.g. a root that is not primitive: leave the operation to the host mirror)
        }
        // the previous context lives on while operations of other threads hold it (Resolved::hold), then it is destroyed
        auto h = std::make_shared<CtxHolder>();
        h->ctx = c;
        {
            std::lock_guard<std::mutex> hl(r.holderMutex);
            r.holders[c] = h.get();
        }
        u.cur = std::move(h), u.q = q, u.psi = psi, u.logN = log2u(ringDim);
        u.limbOf.clear();
        for (uint32_t k = 0; k < q.size(); ++k)
            u.limbOf[q[k]] = k;
    }
    out->ctx  = u.cur->ctx;
    out->hold = u.cur;
    out->idx.assign(sets.size(), {});
    for (size_t si = 0; si < sets.size(); ++si) {
        out->idx[si].resize(sets[si].n);
        for (uint32_t i = 0; i < sets[si].n; ++i) {
            const uint32_t l = u.limbOf[sets[si].q[i]];
            if (u.psi[l] != sets[si].psi[i]) {
                r.outOtherRoot.fetch_add(1, std::memory_order_relaxed);
                return false;  // same modulus with another root of unity: another transform, not ours
            }
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
    CtxHolder& H = holder_of(ctx);
    std::lock_guard<std::mutex> lk(H.mu);
    auto it = H.convs.find(key);
    if (it != H.convs.end())
        return it->second;
    fhe_conv* cv = nullptr;
    Check(r.api.conv_create_custom(ctx, srcIdx.data(), (uint32_t)nSrc, dstIdx.data(), (uint32_t)nDst, hatInv, hatMod, alphaMod, qInv, &cv),
          "HIP backend: basis-conversion plan");
    H.convs.emplace(std::move(key), cv);
    return cv;
}

fhe_sr_plan* SrPlan(fhe_ctx* ctx, uint32_t sizeI, const std::vector<uint32_t>& outIdx, const uint64_t* tab, const double* frac) {
    Runtime& r         = rt();
    const size_t sizeO = outIdx.size();
    std::vector<uint64_t> key{sizeI, sizeO, frac ? 1u : 0u};
    key.insert(key.end(), outIdx.begin(), outIdx.end());
    key.insert(key.end(), tab, tab + sizeO * (sizeI + 1));
    for (uint32_t i = 0; frac && i < sizeI; ++i) {
        uint64_t w;
        std::memcpy(&w, frac + i, 8);
        key.push_back(w);
    }
    CtxHolder& H = holder_of(ctx);
    std::lock_guard<std::mutex> lk(H.mu);
    auto it = H.srPlans.find(key);
    if (it != H.srPlans.end())
        return it->second;
    fhe_sr_plan* p = nullptr;
    Check(r.api.sr_plan_create(ctx, sizeI, outIdx.data(), (uint32_t)sizeO, tab, frac, &p), "HIP backend: ScaleAndRound plan");
    H.srPlans.emplace(std::move(key), p);
    return p;
}
// One plan per (bases, member, content of the member's table arguments): the plan is created with derived tables (which also fixes
// the Barrett constants, functions of the moduli alone) and the member's tables are then REPLACED by the caller's values, so the
// kernel computes with exactly what the reference's member would read.
// tables layout (numQ = |Q|, numBsk = |Bsk| = numQ + 1, numB = numQ):
//   which 0: mtildeQHatInvModq[numQ] QHatModbsk[numQ][numBsk] QHatModmtilde[numQ] QModbsk[numBsk] negQInvModmtilde mtildeInvModbsk[numBsk]
//   which 1: tQHatInvModq[numQ] QHatModbsk[numQ][numBsk] qInvModbsk[numQ][numBsk] tQInvModbsk[numBsk]
//   which 2: BHatInvModb[numB] BHatModmsk[numB] BInvModmsk BHatModq[numB][numQ] BModq[numQ]
fhe_behz* BehzPlan(fhe_ctx* ctx, const std::vector<uint32_t>& qIdx, const std::vector<uint32_t>& bskIdx, int which, uint64_t t,
                   const std::vector<uint64_t>& tables) {
    Runtime& r = rt();
    const size_t numQ = qIdx.size(), numBsk = bskIdx.size();
    const size_t need = which == 0 ? numQ + numQ * numBsk + numQ + numBsk + 1 + numBsk
                                   : which == 1 ? numQ + 2 * numQ * numBsk + numBsk : numQ + numQ + 1 + numQ * numQ + numQ;
    if (tables.size() != need)
        return nullptr;
    std::vector<uint64_t> key{(uint64_t)which, t, numQ};
    key.insert(key.end(), qIdx.begin(), qIdx.end());
    key.insert(key.end(), bskIdx.begin(), bskIdx.end());
    key.insert(key.end(), tables.begin(), tables.end());
    CtxHolder& H = holder_of(ctx);
    std::lock_guard<std::mutex> lk(H.mu);
    auto it = H.behzPlans.find(key);
    if (it != H.behzPlans.end())
        return it->second;
    fhe_behz* p = nullptr;
    if (r.api.behz_create(ctx, qIdx.data(), (uint32_t)numQ, bskIdx.data(), t ? t : 65537, &p) != FHE_OK) {
        Declined("BEHZ plan", r.api.last_error());
        return nullptr;  // (bases the device kernels do not take: the member runs on the host mirror)
    }
    const uint64_t* T = tables.data();
    fhe_status s;
    if (which == 0)
        s = r.api.behz_override_q_to_bsk(p, T, T + numQ, T + numQ + numQ * numBsk, T + 2 * numQ + numQ * numBsk,
                                         T[2 * numQ + numQ * numBsk + numBsk], T + 2 * numQ + numQ * numBsk + numBsk + 1);
    else if (which == 1)
        s = r.api.behz_override_floorq(p, T, T + numQ, T + numQ + numQ * numBsk, T + numQ + 2 * numQ * numBsk);
    else
        s = r.api.behz_override_conv_sk(p, T, T + numQ, T[2 * numQ], T + 2 * numQ + 1, T + 2 * numQ + 1 + numQ * numQ);
    Check(s, "HIP backend: BEHZ plan from the caller's tables");
    H.behzPlans.emplace(std::move(key), p);
    return p;
}

// ---- PrecomputeAutoMap (nbtheory2.cpp:264-275), memoised: precomp[bitrev(j)] = bitrev(((2j+1)k mod 2n) >> 1) ----
namespace {
std::shared_ptr<const std::vector<uint32_t>> auto_map(uint32_t n, uint32_t k) {
    static std::mutex mu;
    static std::map<std::pair<uint32_t, uint32_t>, std::shared_ptr<const std::vector<uint32_t>>> cache;
    {
        std::lock_guard<std::mutex> lk(mu);
        auto it = cache.find({n, k});
        if (it != cache.end())
            return it->second;
    }
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
    std::lock_guard<std::mutex> lk(mu);
    if (cache.size() < 4096)  // (a few hundred rotation indices per context at most; 4n bytes each)
        cache.emplace(std::make_pair(n, k), t);
    return t;
}
}  // namespace
// the map a caller hands to AutomorphismTransform(i, vec), compared word for word with PrecomputeAutoMap(n, i): the device kernel
// computes that permutation on the fly, so any other map must take the reference's code
bool IsAutoMap(uint32_t n, uint32_t k, const std::vector<uint32_t>& vec) {
    if (vec.size() != n || !(k & 1u) || n == 0 || (n & (n - 1)))
        return false;
    const auto tab = auto_map(n, k);
    return std::memcmp(tab->data(), vec.data(), (size_t)n * 4) == 0;
}

// ---- key-switching domains for the batched composites (hip-hooks.h) ----
struct KsDomain {
    fhe_ctx* ctx      = nullptr;
    fhe_ks_plan* plan = nullptr;
    uint32_t sizeQ = 0, sizeP = 0, numPartQ = 0;
    size_t N = 0;
    std::mutex mu;
    struct Entry {
        PackedKey pk;
        std::vector<Buf> sources;  // kept alive: their identity IS the cache key
        uint64_t lastUse = 0;
    };
    std::map<std::vector<const DevBuf*>, Entry> keys;
    uint64_t tick = 0;
    std::array<std::vector<uint8_t>, kCompositeKinds> checked;
    ~KsDomain() {
        if (g_exiting.load())
            return;
        Runtime& r = rt();
        keys.clear();
        if (plan)
            r.api.ks_plan_destroy(plan);
        if (ctx)
            r.api.ctx_destroy(ctx);
    }
};
namespace {
std::mutex g_domainMutex;
std::vector<std::pair<std::vector<uint64_t>, std::shared_ptr<KsDomain>>> g_domains;  // most recently used first
std::atomic<uint64_t> g_compositeCalls{0}, g_checksOk{0}, g_checksBad{0};
}  // namespace
std::shared_ptr<KsDomain> GetKsDomain(uint32_t ringDim, const LimbSet& Q, const LimbSet& P, uint32_t numPartQ) {
    Runtime& r = rt();
    if (!r.live || ringDim < 16 || ringDim > (1u << 17) || (ringDim & (ringDim - 1)) || Q.n == 0 || P.n == 0 || Q.n + P.n > kMaxDeviceLimbs || numPartQ == 0)
        return nullptr;
    std::vector<uint64_t> key{ringDim, numPartQ, Q.n, P.n};
    key.insert(key.end(), Q.q, Q.q + Q.n);
    key.insert(key.end(), Q.psi, Q.psi + Q.n);
    key.insert(key.end(), P.q, P.q + P.n);
    key.insert(key.end(), P.psi, P.psi + P.n);
    std::lock_guard<std::mutex> lk(g_domainMutex);
    for (size_t i = 0; i < g_domains.size(); ++i)
        if (g_domains[i].first == key) {
            auto hit = g_domains[i];
            g_domains.erase(g_domains.begin() + i);
            g_domains.insert(g_domains.begin(), hit);
            return hit.second;
        }
    const uint64_t twoN = 2ull * ringDim;
    std::vector<uint64_t> q(Q.q, Q.q + Q.n), psi(Q.psi, Q.psi + Q.n);
    q.insert(q.end(), P.q, P.q + P.n);
    psi.insert(psi.end(), P.psi, P.psi + P.n);
    for (size_t i = 0; i < q.size(); ++i)
        if (q[i] < 3 || q[i] >= (1ull << 60) || (q[i] - 1) % twoN != 0 || psi[i] == 0)
            return nullptr;
    auto d = std::make_shared<KsDomain>();
    if (r.api.ctx_create(log2u(ringDim), (uint32_t)q.size(), q.data(), psi.data(), r.device, &d->ctx) != FHE_OK)
        return nullptr;
    if (r.api.ks_plan_create(d->ctx, Q.n, P.n, numPartQ, &d->plan) != FHE_OK)
        return nullptr;
    d->sizeQ = Q.n, d->sizeP = P.n, d->numPartQ = numPartQ, d->N = ringDim;
    for (auto& c : d->checked)
        c.assign(Q.n + 1, 0);
    g_domains.insert(g_domains.begin(), {key, d});
    if (g_domains.size() > 4)  // (a domain holds twiddle tables for Q u P and packed copies of its keys)
        g_domains.pop_back();
    return d;
}
fhe_ctx* DomainCtx(const KsDomain& d) { return d.ctx; }
fhe_ks_plan* DomainPlan(const KsDomain& d) { return d.plan; }
PackedKey DomainKey(KsDomain& d, const std::vector<Buf>& b, const std::vector<Buf>& a, Op& op) {
    Runtime& r = rt();
    PackedKey none;
    if (b.size() != d.numPartQ || a.size() != d.numPartQ)
        return none;
    const size_t towerWords = (size_t)(d.sizeQ + d.sizeP) * d.N;
    std::vector<const DevBuf*> id;
    for (const auto& x : b)
        id.push_back(x.get());
    for (const auto& x : a)
        id.push_back(x.get());
    for (const auto* x : id)
        if (!x || x->words < towerWords)
            return none;
    std::lock_guard<std::mutex> lk(d.mu);
    auto it = d.keys.find(id);
    if (it != d.keys.end()) {
        it->second.lastUse = ++d.tick;
        return it->second.pk;
    }
    if (d.keys.size() >= 160) {  // (a bootstrapping key set is 60-70 keys) drop the least recently used one
        auto victim = d.keys.begin();
        for (auto jt = d.keys.begin(); jt != d.keys.end(); ++jt)
            if (jt->second.lastUse < victim->second.lastUse)
                victim = jt;
        d.keys.erase(victim);
    }
    KsDomain::Entry e;
    // towers that are consecutive windows of one buffer (a key set adopted from a replicated buffer, fbb_adopt_keys) ARE the packed
    // layout: no copy; anything else is packed once
    auto packed = [&](const std::vector<Buf>& v) -> Buf {
        for (uint32_t j = 0; j < d.numPartQ; ++j)
            if (!v[j]->parent || v[j]->parent != v[0]->parent || v[j]->p != v[0]->p + j * towerWords)
                return nullptr;
        return View(v[0]->parent, (size_t)(v[0]->p - v[0]->parent->p), towerWords * d.numPartQ);
    };
    uint64_t *pb, *pa;
    if ((e.pk.b = packed(b)) && (e.pk.a = packed(a))) {
        pb = e.pk.b->p, pa = e.pk.a->p;
        op.R(e.pk.b), op.R(e.pk.a);
    }
    else {
        e.pk.b = Alloc(towerWords * d.numPartQ);
        e.pk.a = Alloc(towerWords * d.numPartQ);
        pb = op.W(e.pk.b), pa = op.W(e.pk.a);
        for (uint32_t j = 0; j < d.numPartQ; ++j) {
            D2D(op, pb + j * towerWords, op.R(b[j]), towerWords * 8, "evaluation key packed for the key-switching plan");
            D2D(op, pa + j * towerWords, op.R(a[j]), towerWords * 8, "evaluation key packed for the key-switching plan");
        }
    }
    fhe_ks_key* raw = nullptr;
    Check(r.api.ks_key_wrap(d.plan, pb, pa, &raw), "HIP backend: evaluation key for the key-switching plan");
    e.pk.key = std::shared_ptr<fhe_ks_key>(raw, [](fhe_ks_key* k) {
        if (!g_exiting.load())
            rt().api.ks_key_destroy(k);
    });
    e.sources = b;
    e.sources.insert(e.sources.end(), a.begin(), a.end());
    e.lastUse = ++d.tick;
    PackedKey out = e.pk;
    d.keys.emplace(std::move(id), std::move(e));
    return out;
}
int DomainChecked(const KsDomain& d, CompositeKind kind, uint32_t sizeQl) {
    std::lock_guard<std::mutex> lk(const_cast<KsDomain&>(d).mu);  // (the verdicts are written by other host threads under the same mutex)
    return sizeQl < d.checked[kind].size() ? d.checked[kind][sizeQl] : 2;
}
void DomainSetChecked(KsDomain& d, CompositeKind kind, uint32_t sizeQl, bool identical) {
    std::lock_guard<std::mutex> lk(d.mu);
    if (sizeQl < d.checked[kind].size())
        d.checked[kind][sizeQl] = identical ? 1 : 2;
    (identical ? g_checksOk : g_checksBad).fetch_add(1);
    if (!identical)
        std::fprintf(stderr, "HIP backend: the batched composite %u differs from the member-by-member path at level %u: composite disabled for it\n",
                     (unsigned)kind, sizeQl);
}
std::vector<uint64_t> Checksums(fhe_ctx* ctx, const Buf& words, uint32_t rows) {
    Runtime& r = rt();
    std::vector<uint64_t> out((size_t)rows * 2);
    Op op;
    auto d = Alloc(out.size());
    Check(r.api.checksum(ctx, op.R(words), rows, op.W(d), op.s), "HIP backend: checksums");
    Check(r.api.d2h(r.anyCtx, out.data(), d->p, out.size() * 8, op.s), "HIP backend: checksums");
    op.HostSync();
    return out;
}
void CountComposite() { g_compositeCalls.fetch_add(1, std::memory_order_relaxed); }

}  // namespace hiprt
}  // namespace lbcrypto

// pke calls PrecomputeAutoMap for every rotation and EvalFastRotation (ckksrns-leveledshe.cpp, ckksrns-fhe.cpp, base-leveledshe.cpp); at
// N = 2^17 that is 0.4 ms of host time 260 times per bootstrap.  The HIP build weakens the reference's definition (hal/Makefile) and
// links this memoising one (same table).
namespace lbcrypto {
void PrecomputeAutoMap(uint32_t n, uint32_t k, std::vector<uint32_t>* precomp) {
    const auto tab = hiprt::auto_map(n, k);
    std::copy(tab->begin(), tab->end(), precomp->begin());
}
}  // namespace lbcrypto

// hands every cached buffer and remembered result back to the device (a process that is done with a batch and shares the GPU with another
// process: bench.py between its legs)
extern "C" void fhe_hal_release_caches() {
    if (lbcrypto::hiprt::Available())
        lbcrypto::hiprt::ReleaseAllCaches();
}
// bytes of released buffers the backend holds for reuse right now (free lists, inboxes, orphans)
extern "C" uint64_t fhe_hal_cached_bytes() { return lbcrypto::hiprt::Available() ? lbcrypto::hiprt::CachedBytes() : 0; }
// {bytes cached, allocations served from another thread's cache, requests that reached the device, times the caches went back to the
// device, free bytes of the device, total bytes of the device}
extern "C" void fhe_hal_alloc_stats(uint64_t out[6]) {
    for (int i = 0; i < 6; ++i)
        out[i] = 0;
    if (!lbcrypto::hiprt::Available())
        return;
    auto& r = lbcrypto::hiprt::rt();
    out[0] = r.cachedBytes, out[1] = r.stolen, out[2] = r.deviceMallocs, out[3] = r.cacheReleases;
    size_t f = 0, t = 0;
    if (r.api.mem_info(r.anyCtx, &f, &t) == FHE_OK)
        out[4] = f, out[5] = t;
}
// {bytes the backend holds from the device now (live towers + caches), the high-water mark of that, takes from another thread's cache that
// waited for the buffer's own completion mark}
extern "C" void fhe_hal_alloc_stats2(uint64_t out[3]) {
    out[0] = out[1] = out[2] = 0;
    if (!lbcrypto::hiprt::Available())
        return;
    auto& r = lbcrypto::hiprt::rt();
    out[0] = r.deviceBytes, out[1] = r.deviceBytesHigh, out[2] = r.stolenExact;
}
// Pre-sizes the caches: one buffer of `bytes` is obtained from the device NOW and released into the calling thread's cache, where the first
// large request of an evaluation (a lockstep group's 35-70 GiB BSGS workspace) finds it — instead of growing it against a device that the
// caches of other size classes have filled (round 5: a cold 32x2 run spent its first passes at 13-17 bootstraps/s doing that).
extern "C" int fhe_hal_reserve(uint64_t bytes) {
    if (!lbcrypto::hiprt::Available() || bytes == 0)
        return 1;
    try {
        auto b = lbcrypto::hiprt::Alloc((size_t)((bytes + 7) / 8));
        (void)b;
    }
    catch (...) {
        return 2;
    }
    return 0;
}
// the host waits until every stream of the backend has run dry (the end of a timed pass of a harness)
extern "C" void fhe_hal_device_sync() {
    if (lbcrypto::hiprt::Available())
        lbcrypto::hiprt::SyncAllStreams();
}
extern "C" void fhe_hal_stats(uint64_t out[4]) {
    auto& r = lbcrypto::hiprt::rt();
    out[0] = r.deviceOps, out[1] = r.hostFallbacks, out[2] = r.h2dBytes, out[3] = r.d2hBytes;
}
extern "C" size_t fhe_hal_member_stats(char* buf, size_t cap) {
    using namespace lbcrypto::hiprt;
    std::map<std::string, std::array<uint64_t, 4>> merged;
    for (auto& m : g_members) {
        const char* n = m.name.load();
        if (!n)
            continue;
        auto& e = merged[n];
        e[0] += m.device.load(), e[1] += m.host.load(), e[2] += m.reads.load(), e[3] += m.bytes.load();
    }
    std::string s;
    for (auto& kv : merged)
        s += kv.first + " " + std::to_string(kv.second[0]) + " " + std::to_string(kv.second[1]) + " " + std::to_string(kv.second[2]) + " " +
             std::to_string(kv.second[3]) + "\n";  // (<member> <device ops> <host-mirror executions> <host reads> <operand bytes>)
    if (buf && cap) {
        const size_t n = std::min(cap - 1, s.size());
        std::memcpy(buf, s.data(), n);
        buf[n] = 0;
    }
    return s.size() + 1;
}
extern "C" void fhe_hal_stats_reset(void) {
    using namespace lbcrypto::hiprt;
    auto& r = rt();
    r.deviceOps = 0, r.hostFallbacks = 0, r.h2dBytes = 0, r.d2hBytes = 0, r.hostSmallRing = 0, r.hostData = 0;
    r.outRing = 0, r.outModulus = 0, r.outMoreThan128Moduli = 0, r.outOtherRoot = 0;
    r.opReadBytes = 0, r.opWriteBytes = 0;
    for (auto& m : g_members)
        m.device = 0, m.host = 0, m.reads = 0, m.bytes = 0;
}
extern "C" int fhe_hal_available(void) { return lbcrypto::hiprt::Available() ? 1 : 0; }
extern "C" void fhe_hal_set_device(int device) { lbcrypto::hiprt::g_deviceOverride.store(device); }
extern "C" int fhe_hal_device(void) { return lbcrypto::hiprt::Device(); }
extern "C" void fhe_hal_trace_reset(void) {
    std::lock_guard<std::mutex> lk(lbcrypto::hiprt::g_traceMutex);
    if (lbcrypto::hiprt::g_traceSites)
        lbcrypto::hiprt::g_traceSites->clear();
}
// the device library's kernel launches since it was loaded ("<kernel> <launches>" lines, see fhe_launch_stats), *total = their sum
extern "C" size_t fhe_hal_launch_stats(char* buf, size_t cap, uint64_t* total) {
    auto& r = lbcrypto::hiprt::rt();
    if (total)
        *total = 0;
    return r.launch_stats ? r.launch_stats(buf, cap, total) : 0;
}
// results of pure members (DropLastElementAndScale, Times by constants) taken from the memo of a shared buffer instead of recomputed
extern "C" uint64_t fhe_hal_memo_hits() { return lbcrypto::hiprt::g_memoHits.load(); }
extern "C" void fhe_hal_composite_stats(uint64_t out[3]) {
    out[0] = lbcrypto::hiprt::g_compositeCalls, out[1] = lbcrypto::hiprt::g_checksOk, out[2] = lbcrypto::hiprt::g_checksBad;
}
extern "C" void fhe_hal_other_host_counts(uint64_t out[2]) { lbcrypto::hiprt::OtherHostCounts(out); }
// operations that left the device library's domain, by reason: {ring outside [16, 2^17], modulus outside the domain, more than 256
// distinct moduli, another root of unity for a known modulus}
// operand bytes of the device operations so far: {read, written} (see Runtime::opReadBytes)
extern "C" void fhe_hal_operand_bytes(uint64_t out[2]) {
    auto& r = lbcrypto::hiprt::rt();
    out[0] = r.opReadBytes, out[1] = r.opWriteBytes;
}
// "<what>: <why> <count>" lines: device plans / contexts that could not be built (Declined)
extern "C" size_t fhe_hal_decline_stats(char* buf, size_t cap) {
    std::string s;
    {
        std::lock_guard<std::mutex> lk(lbcrypto::hiprt::g_declineMutex);
        for (auto& kv : lbcrypto::hiprt::g_declines)
            s += kv.first + " x" + std::to_string(kv.second) + "\n";
    }
    if (buf && cap) {
        const size_t n = std::min(cap - 1, s.size());
        std::memcpy(buf, s.data(), n);
        buf[n] = 0;
    }
    return s.size() + 1;
}
extern "C" void fhe_hal_out_of_domain(uint64_t out[4]) {
    auto& r = lbcrypto::hiprt::rt();
    out[0] = r.outRing, out[1] = r.outModulus, out[2] = r.outMoreThan128Moduli, out[3] = r.outOtherRoot;
}

// ---- ChineseRemainderTransformFTT<NativeVector> (math/math-hal.h:60-106, math/hal/transform.h:60-163): the four transform members, declared
// as explicit specialisations by this backend's math/hal/intnat/transformnat-impl.h.  One limb of a ring N >= 2^12 (FHE_HAL_FTT_MIN_LOGN) whose
// modulus is in the device library's domain goes to the device with the CALLER's root of unity (the context is built from it: identical
// twiddles); everything else runs the reference's wrapper lines (transformnat-impl.h:646-712) on its NumberTheoreticTransformNat. ----
#include "math/math-hal.h"
#include "math/nbtheory.h"
namespace intnat {
namespace {
// element (N words modulo q) through fhe_ntt_fwd / fhe_ntt_inv; false: not the device's
bool FttOnDevice(bool inverse, const HipFttInteger& rootOfUnity, uint32_t cycloOrder, const HipFttVector& in, HipFttVector* out) {
    using namespace lbcrypto::hiprt;
    static const uint32_t minLogN = std::getenv("FHE_HAL_FTT_MIN_LOGN") ? std::atoi(std::getenv("FHE_HAL_FTT_MIN_LOGN")) : 12;
    const uint32_t N = cycloOrder >> 1;
    if (!Available() || N < (1u << minLogN) || in.GetLength() != N || (N & (N - 1)))
        return false;
    const uint64_t q = in.GetModulus().ConvertToInt<uint64_t>(), psi = rootOfUnity.ConvertToInt<uint64_t>();
    Resolved r;
    if (!Resolve(N, {LimbSet{&q, &psi, 1}}, &r))
        return false;
    if (out != &in) {
        if (out->GetLength() != N)
            *out = HipFttVector(N, in.GetModulus());
        else
            out->SetModulus(in.GetModulus());
    }
    Op op;
    auto d       = Alloc(N);
    uint64_t* dp = op.W(d);
    Check(api().h2d(r.ctx, dp, &in[0], (size_t)N * 8, op.s), "ChineseRemainderTransformFTT host -> device");
    Check((inverse ? api().ntt_inv : api().ntt_fwd)(r.ctx, dp, r.idx[0].data(), 1, 1, op.s), "ChineseRemainderTransformFTT");
    Check(api().d2h(r.ctx, &(*out)[0], dp, (size_t)N * 8, op.s), "ChineseRemainderTransformFTT device -> host");
    op.HostSync();
    CountH2D((size_t)N * 8);
    CountD2H((size_t)N * 8);
    CountDevice("ChineseRemainderTransformFTT");
    return true;
}
}  // namespace

template <>
void ChineseRemainderTransformFTTNat<HipFttVector>::ForwardTransformToBitReverseInPlace(const HipFttInteger& rootOfUnity, const uint32_t cycloOrder,
                                                                                        HipFttVector* element) {
    if (rootOfUnity == HipFttInteger(1) || rootOfUnity == HipFttInteger(0))
        return;
    if (FttOnDevice(false, rootOfUnity, cycloOrder, *element, element))
        return;
    auto modulus = element->GetModulus();
    PreCompute(rootOfUnity, cycloOrder, modulus);
    NumberTheoreticTransformNat<HipFttVector>().ForwardTransformToBitReverseInPlace(m_rootOfUnityReverseTableByModulus[modulus],
                                                                                   m_rootOfUnityPreconReverseTableByModulus[modulus], element);
}
template <>
void ChineseRemainderTransformFTTNat<HipFttVector>::ForwardTransformToBitReverse(const HipFttVector& element, const HipFttInteger& rootOfUnity,
                                                                                 const uint32_t cycloOrder, HipFttVector* result) {
    if (rootOfUnity == HipFttInteger(1) || rootOfUnity == HipFttInteger(0)) {
        *result = element;
        return;
    }
    if (FttOnDevice(false, rootOfUnity, cycloOrder, element, result))
        return;
    auto modulus = element.GetModulus();
    PreCompute(rootOfUnity, cycloOrder, modulus);
    NumberTheoreticTransformNat<HipFttVector>().ForwardTransformToBitReverse(element, m_rootOfUnityReverseTableByModulus[modulus],
                                                                            m_rootOfUnityPreconReverseTableByModulus[modulus], result);
}
template <>
void ChineseRemainderTransformFTTNat<HipFttVector>::InverseTransformFromBitReverseInPlace(const HipFttInteger& rootOfUnity, const uint32_t cycloOrder,
                                                                                          HipFttVector* element) {
    if (rootOfUnity == HipFttInteger(1) || rootOfUnity == HipFttInteger(0))
        return;
    if (FttOnDevice(true, rootOfUnity, cycloOrder, *element, element))
        return;
    auto modulus = element->GetModulus();
    PreCompute(rootOfUnity, cycloOrder, modulus);
    const uint32_t msb = lbcrypto::GetMSB((cycloOrder >> 1) - 1);
    NumberTheoreticTransformNat<HipFttVector>().InverseTransformFromBitReverseInPlace(
        m_rootOfUnityInverseReverseTableByModulus[modulus], m_rootOfUnityInversePreconReverseTableByModulus[modulus],
        m_cycloOrderInverseTableByModulus[modulus][msb], m_cycloOrderInversePreconTableByModulus[modulus][msb], element);
}
template <>
void ChineseRemainderTransformFTTNat<HipFttVector>::InverseTransformFromBitReverse(const HipFttVector& element, const HipFttInteger& rootOfUnity,
                                                                                   const uint32_t cycloOrder, HipFttVector* result) {
    if (rootOfUnity == HipFttInteger(1) || rootOfUnity == HipFttInteger(0)) {
        *result = element;
        return;
    }
    if (FttOnDevice(true, rootOfUnity, cycloOrder, element, result))
        return;
    auto modulus = element.GetModulus();
    result->SetModulus(modulus);
    PreCompute(rootOfUnity, cycloOrder, modulus);
    const uint32_t n = element.GetLength();
    for (uint32_t i = 0; i < n; ++i)
        (*result)[i] = element[i];
    const uint32_t msb = lbcrypto::GetMSB(n - 1);
    NumberTheoreticTransformNat<HipFttVector>().InverseTransformFromBitReverseInPlace(
        m_rootOfUnityInverseReverseTableByModulus[modulus], m_rootOfUnityInversePreconReverseTableByModulus[modulus],
        m_cycloOrderInverseTableByModulus[modulus][msb], m_cycloOrderInversePreconTableByModulus[modulus][msb], result);
}
}  // namespace intnat
