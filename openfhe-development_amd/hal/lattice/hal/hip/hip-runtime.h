// hip-runtime.h — the non-template part of the HIP backend of lbcrypto::DCRTPoly: binding of the C ABI (include/fhe_hip.h,
// libfhe_hip.so loaded with dlopen), the registry that maps (ring dimension, modulus) to a limb of a device context, one HIP stream
// per host thread with the cross-thread ordering of device buffers, a caching device allocator, the caches of plans built from
// the reference's own CRT tables, and the per-member device / host-mirror counters the tests assert on.
// Implemented in openfhe-development_amd/hal/hip-runtime.cpp (one translation unit added to libOPENFHEcore).
#ifndef LBCRYPTO_INC_LATTICE_HAL_HIP_RUNTIME_H
#define LBCRYPTO_INC_LATTICE_HAL_HIP_RUNTIME_H

#include <cstddef>
#include <cstdint>
#include <atomic>
#include <memory>
#include <mutex>
#include <vector>

#include "fhe_hip.h"  // the C ABI (repo root include/)

namespace lbcrypto {
namespace hiprt {
// rows of one tower / distinct moduli of one device context (csrc/ntt_kernels.h kMaxLimbs: the limb map of a launch is one byte per row)
constexpr size_t kMaxDeviceLimbs = 256;

// entry points of the C ABI the backend uses (resolved once with dlsym)
struct Api {
    decltype(&fhe_last_error) last_error;
    decltype(&fhe_device_count) device_count;
    decltype(&fhe_ctx_create) ctx_create;
    decltype(&fhe_ctx_destroy) ctx_destroy;
    decltype(&fhe_malloc) malloc_;
    decltype(&fhe_free) free_;
    decltype(&fhe_memcpy_h2d) h2d;
    decltype(&fhe_memcpy_d2h) d2h;
    decltype(&fhe_memcpy_d2d) d2d;
    decltype(&fhe_memset_zero) memset_zero;
    decltype(&fhe_stream_sync) sync;
    decltype(&fhe_stream_create) stream_create;
    decltype(&fhe_stream_wait) stream_wait;
    decltype(&fhe_ntt_fwd) ntt_fwd;
    decltype(&fhe_ntt_inv) ntt_inv;
    decltype(&fhe_ntt_inv_oop) ntt_inv_oop;
    decltype(&fhe_ntt_fwd_oop) ntt_fwd_oop;
    decltype(&fhe_inner_product) inner_product;
    decltype(&fhe_add) add;
    decltype(&fhe_sub) sub;
    decltype(&fhe_mul) mul;
    decltype(&fhe_neg) neg;
    decltype(&fhe_mul_add) mul_add;
    decltype(&fhe_mul_const) mul_const;
    decltype(&fhe_mult_acc) mult_acc;
    decltype(&fhe_add_const) add_const;
    decltype(&fhe_sub_const) sub_const;
    decltype(&fhe_times_q_over_t) times_q_over_t;
    decltype(&fhe_mod_switch_round) mod_switch_round;
    decltype(&fhe_automorph) automorph;
    decltype(&fhe_switch_modulus) switch_modulus;
    decltype(&fhe_crt_decompose_towers) crt_decompose_towers;
    decltype(&fhe_crt_decompose) crt_decompose;
    decltype(&fhe_event_create) event_create;        // completion marks of cached buffers (Alloc)
    decltype(&fhe_event_record) event_record;
    decltype(&fhe_stream_wait_event) stream_wait_event;
    decltype(&fhe_event_destroy) event_destroy;
    decltype(&fhe_sample_uniform) sample_uniform;    // (optional sampling on the device: FHE_HAL_DEVICE_SAMPLER=1)
    decltype(&fhe_sample_gaussian) sample_gaussian;
    decltype(&fhe_sample_ternary) sample_ternary;
    decltype(&fhe_rescale_limbs) rescale_limbs;
    decltype(&fhe_rescale_limbs_pair) rescale_limbs_pair;
    decltype(&fhe_add_pair) add_pair;
    decltype(&fhe_sub_pair) sub_pair;
    decltype(&fhe_mul_const_pair) mul_const_pair;
    decltype(&fhe_lincomb) lincomb;
    decltype(&fhe_mem_info) mem_info;
    decltype(&fhe_rescale_workspace_bytes) rescale_workspace_bytes;
    decltype(&fhe_conv_create_custom) conv_create_custom;
    decltype(&fhe_conv_destroy) conv_destroy;
    decltype(&fhe_approx_switch_basis) approx_switch_basis;
    decltype(&fhe_switch_basis_exact) switch_basis_exact;
    decltype(&fhe_sr_plan_create) sr_plan_create;
    decltype(&fhe_sr_plan_destroy) sr_plan_destroy;
    decltype(&fhe_scale_and_round) scale_and_round;
    decltype(&fhe_scale_and_round_p_over_q) scale_and_round_p_over_q;
    decltype(&fhe_scale_and_round_native) scale_and_round_native;
    decltype(&fhe_scale_and_round_behz_decrypt) scale_and_round_behz_decrypt;
    decltype(&fhe_behz_create) behz_create;
    decltype(&fhe_behz_destroy) behz_destroy;
    decltype(&fhe_behz_override_q_to_bsk) behz_override_q_to_bsk;
    decltype(&fhe_behz_override_floorq) behz_override_floorq;
    decltype(&fhe_behz_override_conv_sk) behz_override_conv_sk;
    decltype(&fhe_behz_workspace_bytes) behz_workspace_bytes;
    decltype(&fhe_behz_q_to_bsk) behz_q_to_bsk;
    decltype(&fhe_behz_floorq) behz_floorq;
    decltype(&fhe_behz_conv_sk) behz_conv_sk;
    decltype(&fhe_tensor) tensor;
    decltype(&fhe_tensor_square) tensor_square;
    decltype(&fhe_ks_plan_create) ks_plan_create;
    decltype(&fhe_ks_plan_destroy) ks_plan_destroy;
    decltype(&fhe_keyswitch_hybrid_acc) keyswitch_hybrid_acc;
    decltype(&fhe_ks_precompute) ks_precompute;
    decltype(&fhe_ks_fast_keyswitch) ks_fast_keyswitch;
    decltype(&fhe_ks_key_wrap) ks_key_wrap;
    decltype(&fhe_ks_key_destroy) ks_key_destroy;
    decltype(&fhe_ks_workspace_bytes) ks_workspace_bytes;
    decltype(&fhe_keyswitch_hybrid) keyswitch_hybrid;
    decltype(&fhe_ckks_eval_mult) ckks_eval_mult;
    decltype(&fhe_ckks_bsgs_workspace_bytes) bsgs_workspace_bytes;
    decltype(&fhe_ckks_bsgs_transform) bsgs_transform;
    decltype(&fhe_checksum) checksum;
};

// true when the library is loaded and a device is usable; otherwise every DCRTPoly member runs on its host mirror
bool Available();
// FHE_HAL_DEVICE_SAMPLER=1: the sampling constructors of DCRTPoly (uniform / Gaussian / ternary) run as device kernels on a counter-based
// generator (csrc/sampler_kernels.h): the distributions are the reference's, the words are not its Blake2 stream's.  Off by default: with
// the reference's PRNG seeded the default backend and this one then produce the same keys and ciphertexts, word for word.
bool DeviceSamplerEnabled();
// the process's sampler seed (drawn once from the reference's PRNG, so a seeded PRNG gives reproducible device streams) and a fresh stream id
void DeviceSamplerStream(uint64_t* seed, uint32_t* streamId);
const Api& api();
// throws (OPENFHE_THROW) with the library's message when a call failed
void Check(fhe_status s, const char* what);
// the device this process computes on: fhe_hal_set_device(), else $FHE_HIP_DEVICE, else 0 (one process per GPU: a rank of a
// multi-GPU job passes its LOCAL_RANK)
int Device();
// a context of that device (for calls that need one only to name the device: allocation, copies, stream primitives)
fhe_ctx* AnyCtx();

// ---- device memory.  A buffer remembers which stream wrote it last and which streams read it since (every host thread has a
// stream of its own): Op::R / Op::W below order a thread's kernels after the foreign ones they depend on with device-side
// waits (fhe_stream_wait: the host never blocks), and a buffer returns to the free lists of the thread that drops its last
// reference only after that thread's stream has been ordered behind the buffer's pending uses. ----
struct DevBuf {
    uint64_t* p  = nullptr;
    size_t words = 0;
    size_t cap   = 0;  // words of the allocation behind p (its size class; a recycled buffer may be larger than `words` asks for)
    std::shared_ptr<DevBuf> parent;  // a window of another buffer (packed evaluation keys): uses are recorded on the parent
    bool external = false;           // memory the backend does not own (e.g. a tensor an RCCL collective filled): never pooled or freed
    struct Use {
        uint32_t stream;
        uint64_t seq;
    };
    std::mutex mu;
    Use writer{0, 0};
    std::vector<Use> readers;
    // windows (View) of this buffer alive right now: each holds a reference through its `parent`, so use_count() alone does not tell how
    // many TOWERS own these words (a packed wide buffer with K windows is not a clone of anything: round-5 advisor)
    std::atomic<uint32_t> views{0};
    // towers (and locals) holding this buffer itself, given the caller's own extra references
    long Owners(const std::shared_ptr<DevBuf>& self, long callersRefs) const { return self.use_count() - (long)views.load() - callersRefs; }
    // Results of pure members applied to these words (same member, same tables, same words -> same words), kept while several towers
    // share the buffer (clones): pke's weighted sums rescale a fresh clone of the same power T_i in every sum they form
    // (ckksrns-advancedshe.cpp:143-193).  Dropped when the buffer is written (Op::W).
    struct Memo {
        std::vector<uint64_t> key;  // {member, limbs, context limbs..., table words...}
        std::shared_ptr<DevBuf> result;
    };
    std::vector<Memo> memo;
    ~DevBuf();
};
using Buf = std::shared_ptr<DevBuf>;
// the remembered result of `key` on src's words (nullptr: none); MemoStore keeps at most a few per buffer (whole buffers only, no windows)
Buf MemoFind(const Buf& src, const std::vector<uint64_t>& key);
void MemoStore(const Buf& src, std::vector<uint64_t> key, const Buf& result);
Buf Alloc(size_t words);
// every cached released buffer and remembered result goes back to the device (what Alloc does under memory pressure)
void ReleaseAllCaches();
uint64_t CachedBytes();
// the calling host thread waits for every stream of the backend
void SyncAllStreams();
Buf View(const Buf& parent, size_t offsetWords, size_t words);
// device memory owned by the caller (it must outlive every tower that adopts a window of it)
Buf WrapExternal(uint64_t* devPtr, size_t words);

// WIDE towers.  pke's control flow depends on parameters and metadata only, never on ciphertext words: K ciphertexts with equal metadata
// can be evaluated in lockstep as ONE ciphertext whose towers hold K towers each ([K][limbs][N], the C ABI's batch dimension) — every
// launch then works on K towers and every evaluation key is read once for all of them.  The width of the towers that pke creates on
// the way (accumulators) is the calling thread's current width.
uint32_t ThreadWidth();
struct WidthScope {
    explicit WidthScope(uint32_t k);
    ~WidthScope();
    uint32_t saved;
};

// One device operation (a group of launches) of the calling thread, on that thread's stream.
class Op {
public:
    Op();
    ~Op();
    Op(const Op&)            = delete;
    Op& operator=(const Op&) = delete;
    void* s;                              // the stream the launches of this operation go to
    const uint64_t* R(const Buf& b);      // this operation reads b   (ordered after b's foreign writer)
    // this operation writes b  (ordered after b's foreign writer and readers); operand = false: a workspace of the operation, not one
    // of its operands (left out of the operand-byte count of fhe_hal_operand_bytes)
    uint64_t* W(const Buf& b, bool operand = true);
    void HostSync();                      // the host waits for everything this thread has enqueued so far
private:
    uint64_t m_seq;
};

// ---- contexts: one device context per ring dimension holding every modulus seen so far ----
struct LimbSet {  // the moduli / roots of one tower (an ILDCRTParams), in tower order
    const uint64_t* q;
    const uint64_t* psi;
    uint32_t n;
};
struct CtxHolder;  // a device context with the plans built on it; destroyed when the registry has replaced it and the last
                   // operation that resolved against it is over
struct Resolved {
    fhe_ctx* ctx = nullptr;
    std::vector<std::vector<uint32_t>> idx;  // context limbs of every requested set
    std::shared_ptr<CtxHolder> hold;         // keeps ctx (and its plans) alive for the duration of the operation
};
// registers the moduli of all sets (growing the context when new ones appear) and returns their context limbs; false when
// the ring or a modulus is outside the device library's domain (N not 2^4..2^17, q >= 2^60, q != 1 mod 2N, > 256 limbs)
bool Resolve(uint32_t ringDim, const std::vector<LimbSet>& sets, Resolved* out);

// ---- basis-conversion plans from the caller's (= the reference's CryptoParameters') tables, cached by content ----
// hatInv[nSrc], hatMod[nSrc][nDst] row-major; alphaMod[(nSrc+1)][nDst] + qInv[nSrc] for the exact variant or both null
fhe_conv* ConvPlan(fhe_ctx* ctx, const std::vector<uint32_t>& srcIdx, const std::vector<uint32_t>& dstIdx, const uint64_t* hatInv,
                   const uint64_t* hatMod, const uint64_t* alphaMod, const double* qInv);

// ---- ScaleAndRound plans (the caller's tables: tab [sizeO][sizeI+1], frac [sizeI] or null for ApproxScaleAndRound) ----
fhe_sr_plan* SrPlan(fhe_ctx* ctx, uint32_t sizeI, const std::vector<uint32_t>& outIdx, const uint64_t* tab, const double* frac);
// ---- BEHZ plans, one per member (which: 0 = FastBaseConvqToBskMontgomery, 1 = FastRNSFloorq, 2 = FastBaseConvSK) and table
// content: `tables` is the concatenation of the member's table arguments as the reference passes them (flattened row-major), the
// plan computes with exactly these values (fhe_behz_override_*); nullptr when the library does not take the bases ----
fhe_behz* BehzPlan(fhe_ctx* ctx, const std::vector<uint32_t>& qIdx, const std::vector<uint32_t>& bskIdx, int which, uint64_t t,
                   const std::vector<uint64_t>& tables);

// ---- PrecomputeAutoMap(n, k) memoised (the table pke passes to AutomorphismTransform(i, vec)); `vec` is compared with it in full ----
bool IsAutoMap(uint32_t n, uint32_t k, const std::vector<uint32_t>& vec);

// ---- counters.  Every public member of the backend class opens a MemberScope; device operations, host-mirror executions
// (the member ran the reference's code on the mirror and produced / modified words there) and host reads (a const member was
// served by the mirror after a device -> host copy) are attributed to the OUTERMOST member of the calling thread.  With
// FHE_HAL_REQUIRE_DEVICE=1 a host-mirror execution of a member that has a device path throws instead of degrading silently. ----
struct MemberScope {
    explicit MemberScope(const char* member);
    ~MemberScope();
    bool outer;
};
struct Stats {
    uint64_t deviceOps, hostFallbacks, h2dBytes, d2hBytes;
};
// {host-mirror executions on rings below 16, host-side productions of words} since process start / the last reset
void OtherHostCounts(uint64_t out[2]);
void TraceMember(const char* member);  // the member about to touch words (FHE_HAL_TRACE attributes PCIe bytes to it)
void D2D(Op& op, uint64_t* dst, const uint64_t* src, size_t bytes, const char* what);  // device copy (+ trace)
void CountDevice(const char* member = __builtin_FUNCTION());
// member = the DCRTPoly member that went to the host mirror.  ringDim < 16 (outside the device library's domain by specification) and
// hostData (the host PRODUCES the words: encoders and samplers filling limbs with SetElementAtIndex / operator=, no device copy involved)
// are counted apart from the fall-backs of arithmetic
void CountHost(const char* member, uint32_t ringDim = 0, bool hostData = false);
void CountHostRead(const char* member);  // a const member read the mirror after a device -> host copy
void CountH2D(size_t bytes);
void CountD2H(size_t bytes);

}  // namespace hiprt
}  // namespace lbcrypto

extern "C" {
// {device operations, host fallbacks, bytes host->device, bytes device->host} since process start
void fhe_hal_stats(uint64_t out[4]);
// {host-mirror executions on rings of dimension < 16, words produced on the host (encoders, samplers)}: not fall-backs, counted apart
void fhe_hal_other_host_counts(uint64_t out[2]);
// per member: "name deviceOps hostOps hostReads\n" for every member seen so far, into buf (returns the length needed)
size_t fhe_hal_member_stats(char* buf, size_t cap);
// forgets all counters (a test program calls it after its set-up phase)
void fhe_hal_stats_reset(void);
// 1 when the HIP backend is live (library loaded, device present), 0 when every operation runs on the host mirror
int fhe_hal_available(void);
// the device of this process; call before the first DCRTPoly operation (a rank of a multi-GPU job: its LOCAL_RANK)
void fhe_hal_set_device(int device);
int fhe_hal_device(void);
// forgets the call sites FHE_HAL_TRACE has collected so far (a test program calls it after its set-up phase)
void fhe_hal_trace_reset(void);
}

#endif
