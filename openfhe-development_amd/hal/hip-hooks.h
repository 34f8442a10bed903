// hip-hooks.h — what the backend's pke-level hooks (hal/keyswitch-hybrid-hip.cpp) need from the runtime beyond the DCRTPoly class:
// the library's batched COMPOSITES (a whole key switch, a whole BSGS linear transform as ONE call) run on a key-switching plan over
// a context that holds Q then P in the reference's order, with every evaluation key packed [numPartQ][sizeQ+sizeP][N].
// Implemented in hal/hip-runtime.cpp.  (Not part of lattice/hal/hip/: only the hooks include it.)
#ifndef LBCRYPTO_HAL_HIP_HOOKS_H
#define LBCRYPTO_HAL_HIP_HOOKS_H

#include <memory>
#include <vector>

#include "lattice/hal/hip/hip-runtime.h"

namespace lbcrypto {
namespace hiprt {

// The HYBRID key-switching domain of one parameter set (ring, Q, P, number of digits).
struct KsDomain;
// nullptr when the parameters are outside the library's domain.  Domains are cached (a few most recently used ones).
std::shared_ptr<KsDomain> GetKsDomain(uint32_t ringDim, const LimbSet& Q, const LimbSet& P, uint32_t numPartQ);
fhe_ctx* DomainCtx(const KsDomain& d);
fhe_ks_plan* DomainPlan(const KsDomain& d);
// the evaluation key whose digits are the device towers b[j], a[j] (each [sizeQ+sizeP][N], EVALUATION), packed for the plan on first
// use (cached by the identity of the source buffers, which the cache keeps alive).  The packed words are valid for operations
// declared through `op` (the packing copies are ordered on op's stream, later users are ordered through packedB / packedA).
struct PackedKey {
    std::shared_ptr<fhe_ks_key> key;  // (shared: the cache may drop its entry while a call of another thread still uses the key)
    Buf b, a;                         // the packed words: declare op.R() on them before a call that reads the key
};
PackedKey DomainKey(KsDomain& d, const std::vector<Buf>& b, const std::vector<Buf>& a, Op& op);
// Composites are checked once per (domain, kind, level) against the member-by-member path, which computes with the tables the
// caller passes (the reference's CryptoParameters): 0 = not yet checked, 1 = identical (use the composite), 2 = differed (never use).
enum CompositeKind : uint32_t { kKeySwitchAcc = 0, kBsgs = 1, kKeySwitch = 2, kCompositeKinds = 3 };
int DomainChecked(const KsDomain& d, CompositeKind kind, uint32_t sizeQl);
void DomainSetChecked(KsDomain& d, CompositeKind kind, uint32_t sizeQl, bool identical);
// {sum, position-weighted sum} of every row of a device buffer [rows][N] brought to the host (fhe_checksum: the second word depends on the
// ORDER of the words): the comparison of two results on the device
std::vector<uint64_t> Checksums(fhe_ctx* ctx, const Buf& words, uint32_t rows);
void CountComposite();

}  // namespace hiprt
}  // namespace lbcrypto

extern "C" {
// {composite calls, first-use checks that matched, first-use checks that differed} since process start
void fhe_hal_composite_stats(uint64_t out[3]);
// results of pure members taken from the memo of a shared buffer (DevBuf::memo) instead of recomputed
uint64_t fhe_hal_memo_hits();
// the device library's kernel launches since it was loaded: "<kernel> <launches>\n" lines into buf, *total = their sum
size_t fhe_hal_launch_stats(char* buf, size_t cap, uint64_t* total);
}
#endif
