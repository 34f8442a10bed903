// hip-runtime.h — the non-template part of the HIP backend of lbcrypto::DCRTPoly: binding of the C ABI (include/fhe_hip.h,
// libfhe_hip.so loaded with dlopen), the registry that maps (ring dimension, modulus) to a limb of a device context, a
// caching device allocator and the cache of basis-conversion plans built from the reference's own CRT tables.
// Implemented in openfhe-development_amd/hal/hip-runtime.cpp (one translation unit added to libOPENFHEcore).
#ifndef LBCRYPTO_INC_LATTICE_HAL_HIP_RUNTIME_H
#define LBCRYPTO_INC_LATTICE_HAL_HIP_RUNTIME_H

#include <cstddef>
#include <cstdint>
#include <memory>
#include <vector>

#include "fhe_hip.h"  // the C ABI (repo root include/)

namespace lbcrypto {
namespace hiprt {

// entry points of the C ABI the shim uses (resolved once with dlsym)
struct Api {
    decltype(&fhe_last_error) last_error;
    decltype(&fhe_device_count) device_count;
    decltype(&fhe_ctx_create) ctx_create;
    decltype(&fhe_malloc) malloc_;
    decltype(&fhe_free) free_;
    decltype(&fhe_memcpy_h2d) h2d;
    decltype(&fhe_memcpy_d2h) d2h;
    decltype(&fhe_memcpy_d2d) d2d;
    decltype(&fhe_stream_sync) sync;
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
    decltype(&fhe_automorph) automorph;
    decltype(&fhe_switch_modulus) switch_modulus;
    decltype(&fhe_conv_create_custom) conv_create_custom;
    decltype(&fhe_approx_switch_basis) approx_switch_basis;
    decltype(&fhe_switch_basis_exact) switch_basis_exact;
    decltype(&fhe_sr_plan_create) sr_plan_create;
    decltype(&fhe_scale_and_round) scale_and_round;
    decltype(&fhe_behz_create) behz_create;
    decltype(&fhe_behz_workspace_bytes) behz_workspace_bytes;
    decltype(&fhe_behz_q_to_bsk) behz_q_to_bsk;
    decltype(&fhe_behz_floorq) behz_floorq;
    decltype(&fhe_behz_conv_sk) behz_conv_sk;
};

// true when the library is loaded and a device is usable; otherwise every DCRTPoly member runs on its host mirror
bool Available();
const Api& api();
// throws (OPENFHE_THROW) with the library's message when a call failed
void Check(fhe_status s, const char* what);

// ---- device memory: size-bucketed free lists over fhe_malloc (hipMalloc / hipFree are far too slow per operation) ----
struct DevBuf {
    uint64_t* p  = nullptr;
    size_t words = 0;
    ~DevBuf();
};
using Buf = std::shared_ptr<DevBuf>;
Buf Alloc(size_t words);

// ---- contexts: one device context per ring dimension holding every modulus seen so far ----
struct LimbSet {  // the moduli / roots of one tower (an ILDCRTParams), in tower order
    const uint64_t* q;
    const uint64_t* psi;
    uint32_t n;
};
struct Resolved {
    fhe_ctx* ctx = nullptr;
    std::vector<std::vector<uint32_t>> idx;  // context limbs of every requested set
};
// registers the moduli of all sets (growing the context when new ones appear) and returns their context limbs; false when
// the ring or a modulus is outside the device library's domain (N not 2^4..2^17, q >= 2^60, q != 1 mod 2N, > 128 limbs)
bool Resolve(uint32_t ringDim, const std::vector<LimbSet>& sets, Resolved* out);

// ---- basis-conversion plans from the caller's (= the reference's CryptoParameters') tables, cached by content ----
// hatInv[nSrc], hatMod[nSrc][nDst] row-major; alphaMod[(nSrc+1)][nDst] + qInv[nSrc] for the exact variant or both null
fhe_conv* ConvPlan(fhe_ctx* ctx, const std::vector<uint32_t>& srcIdx, const std::vector<uint32_t>& dstIdx, const uint64_t* hatInv,
                   const uint64_t* hatMod, const uint64_t* alphaMod, const double* qInv);

// ---- ScaleAndRound plans (the caller's tables: tab [sizeO][sizeI+1], frac [sizeI] or null for ApproxScaleAndRound) and BEHZ
// plans (tables derived from the moduli and t exactly as CryptoParametersBFVRNS derives them: bfvrns-cryptoparameters.cpp:673-850;
// t = 0: any plan over these bases will do — FastBaseConvqToBskMontgomery and FastBaseConvSK do not depend on t) ----
fhe_sr_plan* SrPlan(fhe_ctx* ctx, uint32_t sizeI, const std::vector<uint32_t>& outIdx, const uint64_t* tab, const double* frac);
fhe_behz* BehzPlan(fhe_ctx* ctx, const std::vector<uint32_t>& qIdx, const std::vector<uint32_t>& bskIdx, uint64_t t);

// ---- counters (tests assert that the device path really ran) ----
struct Stats {
    uint64_t deviceOps, hostFallbacks, h2dBytes, d2hBytes;
};
void TraceMember(const char* member);  // the member about to touch words (FHE_HAL_TRACE attributes PCIe bytes to it)
void D2D(fhe_ctx* c, uint64_t* dst, const uint64_t* src, size_t bytes, const char* what);  // device copy (+ trace)
void CountDevice();
void CountHost(const char* member);  // member = the DCRTPoly member that went to the host mirror (FHE_HAL_TRACE)
void CountH2D(size_t bytes);
void CountD2H(size_t bytes);

}  // namespace hiprt
}  // namespace lbcrypto

extern "C" {
// {device operations, host fallbacks, bytes host->device, bytes device->host} since process start
void fhe_hal_stats(uint64_t out[4]);
// 1 when the HIP backend is live (library loaded, device present), 0 when every operation runs on the host mirror
int fhe_hal_available(void);
// forgets the call sites FHE_HAL_TRACE has collected so far (a test program calls it after its set-up phase)
void fhe_hal_trace_reset(void);
}

#endif
