// bootstrap_batch.cpp — BASELINE configs[3] as north_star states it: a BATCH of independent ciphertexts bootstrapped through the
// reference's own API (cc->EvalBootstrap, benchmark/src/ckks-bootstrapping.cpp:59-79), the batch sharded over one process per GPU, the
// evaluation keys generated on ONE rank and replicated over xGMI (scatter + all-gather, shard.py).  This is a small C ABI over the
// reference's CryptoContext so that the Python driver (bench.py, one rank per GPU with torch.distributed) can
//   * build the context and the bootstrapping precomputations on every rank (cc->EvalBootstrapSetup),
//   * generate the key set on rank 0 (cc->EvalMultKeyGen + cc->EvalBootstrapKeyGen) and EXPORT its device words into one packed buffer,
//   * create the same key OBJECTS without any words on the other ranks and ADOPT windows of the buffer the collective filled as their
//     device words (DCRTPoly::AdoptDeviceWords: no copy, no PCIe),
//   * bootstrap the rank's slice of the batch, the ciphertexts spread over host threads (one HIP stream per thread).
// Compiled twice from this one source: against the HIP backend of DCRTPoly (openfhe-development_amd/hal/_build/libfhe_boot_batch_hip.so, the
// measured path) and against the stock libraries (tests/hal/_build/libfhe_boot_batch_stock.so, TEST ONLY: the byte-for-byte reference).
#include <omp.h>

#include <chrono>
#include <cstdio>
#include <fstream>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "math/distributiongenerator.h"
#include "openfhe.h"

using namespace lbcrypto;

#ifdef WITH_HIP
#include "lattice/hal/hip/hip-runtime.h"
extern "C" void fhe_hal_set_device(int device);
extern "C" void fhe_hal_operand_bytes(uint64_t out[2]);
extern "C" void fhe_hal_stats(uint64_t out[4]);
extern "C" void fhe_hal_release_caches();
extern "C" void fhe_hal_device_sync();
extern "C" size_t fhe_hal_launch_stats(char* buf, size_t cap, uint64_t* total);
extern "C" size_t fhe_hal_member_stats(char* buf, size_t cap);
extern "C" void fhe_hal_alloc_stats(uint64_t out[6]);
extern "C" void fhe_hal_alloc_stats2(uint64_t out[3]);
extern "C" int fhe_hal_reserve(uint64_t bytes);
#endif

namespace {
struct Batch {
    CryptoContext<DCRTPoly> cc;
    KeyPair<DCRTPoly> kp;
    uint32_t slots = 0, depth = 0, logN = 0;
    std::vector<uint32_t> levelBudget;
    std::vector<Ciphertext<DCRTPoly>> in, out, saved;
    uint32_t first = 0;
    std::string error;
#ifdef WITH_HIP
    hiprt::Buf external;  // the packed key set the towers have adopted windows of
#endif
};
// waits until every stream of the backend has run dry (no word crosses PCIe for it); nothing to wait for on the stock backend
static void Drain() {
#ifdef WITH_HIP
    fhe_hal_device_sync();
#endif
}
// the key set in a fixed order: the relinearisation key, then the automorphism keys by ascending index; per key the b vector's
// digits, then the a vector's
std::vector<EvalKey<DCRTPoly>> KeyList(Batch& b, std::vector<uint32_t>* indices = nullptr) {
    std::vector<EvalKey<DCRTPoly>> keys;
    keys.push_back(b.cc->GetEvalMultKeyVector(b.kp.secretKey->GetKeyTag())[0]);
    for (const auto& kv : b.cc->GetEvalAutomorphismKeyMap(b.kp.secretKey->GetKeyTag())) {
        keys.push_back(kv.second);
        if (indices)
            indices->push_back(kv.first);
    }
    return keys;
}
std::vector<double> Message(uint32_t index) {  // the plaintext of ciphertext `index` (8 values, the rest of the slots zero)
    std::vector<double> x(8);
    for (uint32_t i = 0; i < 8; ++i)
        x[i] = 0.25 * (i + 1) + 0.03125 * (index % 16);
    return x;
}
template <typename F>
int Guard(Batch* b, F&& f) {
    try {
        f();
        return 0;
    }
    catch (const std::exception& e) {
        b->error = e.what();
        return 1;
    }
}
}  // namespace

extern "C" {
// context (benchmark/src/ckks-bootstrapping.cpp:70: 2^17, 2^16 slots, 59/60-bit moduli, HYBRID, {4,4}, SPARSE_TERNARY, FLEXIBLEAUTO;
// ring, slots and level budget overridable), bootstrapping precomputations, key pair.  prngLib: the deterministic test PRNG (every
// rank then draws the same secret key and the same encryption randomness) or NULL for the library's own generator.
void* fbb_create(uint32_t logN, uint32_t slots, uint32_t budgetEnc, uint32_t budgetDec, uint32_t levelsAfter, const char* prngLib, int device) {
    auto* b = new Batch;
    if (Guard(b, [&] {
#ifdef WITH_HIP
            fhe_hal_set_device(device);
#endif
            if (prngLib && *prngLib)
                PseudoRandomNumberGenerator::InitPRNGEngine(prngLib);
            CCParams<CryptoContextCKKSRNS> p;
            const SecretKeyDist skd = SPARSE_TERNARY;
            p.SetSecretKeyDist(skd);
            p.SetSecurityLevel(HEStd_NotSet);
            p.SetRingDim(1u << logN);
            p.SetScalingTechnique(FLEXIBLEAUTO);
            p.SetScalingModSize(59);
            p.SetFirstModSize(60);
            p.SetKeySwitchTechnique(HYBRID);
            b->levelBudget = {budgetEnc, budgetDec};
            b->depth       = levelsAfter + FHECKKSRNS::GetBootstrapDepth(b->levelBudget, skd);
            p.SetMultiplicativeDepth(b->depth);
            b->cc = GenCryptoContext(p);
            b->cc->Enable(PKE);
            b->cc->Enable(KEYSWITCH);
            b->cc->Enable(LEVELEDSHE);
            b->cc->Enable(ADVANCEDSHE);
            b->cc->Enable(FHE);
            b->slots = slots, b->logN = logN;
            b->cc->EvalBootstrapSetup(b->levelBudget, {0, 0}, slots);
            b->kp = b->cc->KeyGen();
        })) {
        std::fprintf(stderr, "fbb_create: %s\n", b->error.c_str());
    }
    return b;
}
const char* fbb_error(void* h) { return static_cast<Batch*>(h)->error.c_str(); }
// the OpenMP team of pke's own loops on the calling thread (key generation draws from thread-local PRNGs: two runs produce the same
// keys only with the same team)
void fbb_set_omp_threads(int n) { omp_set_num_threads(n); }
// 0: every OpenMP region of the process runs on its calling thread alone (the latency of ONE bootstrap on ONE host thread / stream:
// pke's inner loops would otherwise fork inside a one-thread batch loop); 1: the default (one level of parallelism)
void fbb_set_active_levels(int n) { omp_set_max_active_levels(n); }
void fbb_destroy(void* h) {
    // pke keeps evaluation keys and contexts in static registries: the harness owns the process's only context, so they go with the batch
    // (the key towers may be windows of `external`, which the batch releases after them)
    CryptoContextImpl<DCRTPoly>::ClearEvalMultKeys();
    CryptoContextImpl<DCRTPoly>::ClearEvalAutomorphismKeys();
    delete static_cast<Batch*>(h);
    CryptoContextFactory<DCRTPoly>::ReleaseAllContexts();
#ifdef WITH_HIP
    fhe_hal_release_caches();  // (the batch's keys and towers went to the buffer caches: the next user of the GPU may be another process)
#endif
}
// {ring dimension, Q limbs, P limbs, digits, depth}
void fbb_shape(void* h, uint32_t out[5]) {
    auto* b       = static_cast<Batch*>(h);
    const auto cp = std::dynamic_pointer_cast<CryptoParametersRNS>(b->cc->GetCryptoParameters());
    out[0] = b->cc->GetRingDimension(), out[1] = cp->GetElementParams()->GetParams().size(), out[2] = cp->GetParamsP()->GetParams().size();
    out[3] = cp->GetNumPartQ(), out[4] = b->depth;
}
// encrypts the ciphertexts [0, total) of the batch in order (the same randomness on every rank with the deterministic PRNG) and keeps
// [first, first + count): the rank's slice
int fbb_encrypt(void* h, uint32_t total, uint32_t first, uint32_t count) {
    auto* b = static_cast<Batch*>(h);
    return Guard(b, [&] {
        b->in.clear();
        b->first = first;
        for (uint32_t i = 0; i < total; ++i) {
            auto pt = b->cc->MakeCKKSPackedPlaintext(Message(i), 1, b->depth - 1, nullptr, b->slots);
            auto ct = b->cc->Encrypt(b->kp.publicKey, pt);
            if (i >= first && i < first + count)
                b->in.push_back(ct);
        }
    });
}
// rank 0: the relinearisation key and the bootstrapping rotation keys (ckksrns-fhe.cpp:264-300)
int fbb_keygen(void* h) {
    auto* b = static_cast<Batch*>(h);
    return Guard(b, [&] {
        b->cc->EvalMultKeyGen(b->kp.secretKey);
        b->cc->EvalBootstrapKeyGen(b->kp.secretKey, b->slots);
    });
}
// number of keys in the set (relinearisation key first); indices[1..) = the automorphism indices of the rotation keys (indices[0] = 0)
uint32_t fbb_key_count(void* h, uint32_t* indices, uint32_t cap) {
    auto* b = static_cast<Batch*>(h);
    std::vector<uint32_t> idx;
    const auto keys = KeyList(*b, &idx);
    if (indices && cap >= keys.size()) {
        indices[0] = 0;
        for (size_t i = 0; i < idx.size(); ++i)
            indices[i + 1] = idx[i];
    }
    return (uint32_t)keys.size();
}
// words of one key tower ((Q + P limbs) * N) and towers per key (2 * digits)
void fbb_key_layout(void* h, uint64_t* towerWords, uint32_t* towersPerKey) {
    auto* b       = static_cast<Batch*>(h);
    const auto cp = std::dynamic_pointer_cast<CryptoParametersRNS>(b->cc->GetCryptoParameters());
    *towerWords   = (uint64_t)cp->GetParamsQP()->GetParams().size() * b->cc->GetRingDimension();
    *towersPerKey = 2 * cp->GetNumPartQ();
}
// the other ranks: the same key OBJECTS (tags, automorphism indices, tower shapes) with no words at all
int fbb_make_key_shells(void* h, const uint32_t* indices, uint32_t count) {
    auto* b = static_cast<Batch*>(h);
    return Guard(b, [&] {
        const auto cp       = std::dynamic_pointer_cast<CryptoParametersRNS>(b->cc->GetCryptoParameters());
        const auto paramsQP = cp->GetParamsQP();
        auto shell = [&] {
            auto ek = std::make_shared<EvalKeyRelinImpl<DCRTPoly>>(b->cc);
            std::vector<DCRTPoly> av, bv;
            for (uint32_t j = 0; j < cp->GetNumPartQ(); ++j) {
                av.emplace_back(paramsQP, Format::EVALUATION, false);
                bv.emplace_back(paramsQP, Format::EVALUATION, false);
            }
            ek->SetAVector(std::move(av));
            ek->SetBVector(std::move(bv));
            ek->SetKeyTag(b->kp.secretKey->GetKeyTag());
            return ek;
        };
        b->cc->InsertEvalMultKey({shell()}, b->kp.secretKey->GetKeyTag());
        auto m = std::make_shared<std::map<uint32_t, EvalKey<DCRTPoly>>>();
        for (uint32_t i = 1; i < count; ++i)
            (*m)[indices[i]] = shell();
        b->cc->InsertEvalAutomorphismKey(m, b->kp.secretKey->GetKeyTag());
    });
}
#ifdef WITH_HIP
// rank 0: every key tower's device words copied into dst[key][tower][towerWords] (device memory of the caller, e.g. a torch tensor)
int fbb_export_keys(void* h, uint64_t* devDst) {
    auto* b = static_cast<Batch*>(h);
    return Guard(b, [&] {
        uint64_t words;
        uint32_t per;
        fbb_key_layout(h, &words, &per);
        const auto keys = KeyList(*b);
        auto ext        = hiprt::WrapExternal(devDst, words * per * keys.size());
        hiprt::Op op;
        uint64_t* dst = op.W(ext);
        size_t at     = 0;
        for (const auto& k : keys)
            for (const auto* vec : {&k->GetBVector(), &k->GetAVector()})
                for (const auto& tower : *vec) {
                    auto src = tower.DeviceWords();
                    if (!src)
                        OPENFHE_THROW("fbb_export_keys: a key tower has no device words");
                    hiprt::D2D(op, dst + at, op.R(src), words * 8, "evaluation keys exported for replication");
                    at += words;
                }
        op.HostSync();
    });
}
// every rank: the key towers take windows of src[key][tower][towerWords] as their device words (src must stay alive)
int fbb_adopt_keys(void* h, uint64_t* devSrc) {
    auto* b = static_cast<Batch*>(h);
    return Guard(b, [&] {
        uint64_t words;
        uint32_t per;
        fbb_key_layout(h, &words, &per);
        const auto keys = KeyList(*b);
        b->external     = hiprt::WrapExternal(devSrc, words * per * keys.size());
        size_t at       = 0;
        for (const auto& k : keys)
            for (const auto* vec : {&k->GetBVector(), &k->GetAVector()})
                for (const auto& tower : *vec) {
                    tower.AdoptDeviceWords(hiprt::View(b->external, at, words));
                    at += words;
                }
    });
}
#else
int fbb_export_keys(void*, uint64_t*) { return 1; }
int fbb_adopt_keys(void*, uint64_t*) { return 1; }
#endif
// bootstraps the rank's ciphertexts, spread over `threads` host threads, `reps` passes after one warm-up pass (the first use of every
// level builds tables and checks the composites): seconds per pass
double fbb_bootstrap_all(void* h, int threads, int reps, int warmup) {
    auto* b = static_cast<Batch*>(h);
    double sec = -1;
    Guard(b, [&] {
        const int n = (int)b->in.size();
        b->out.assign(n, nullptr);
        std::string err;
        auto pass = [&] {
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
            for (int i = 0; i < n; ++i) {
                try {
                    b->out[i] = b->cc->EvalBootstrap(b->in[i]);
                }
                catch (const std::exception& e) {
#pragma omp critical
                    err = e.what();
                }
            }
            Drain();  // (the pass ends when every stream has run dry)
            if (!err.empty())
                OPENFHE_THROW(err);
        };
        for (int w = 0; w < warmup; ++w)
            pass();
        const auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < reps; ++r)
            pass();
        sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / std::max(1, reps);
    });
    return sec;
}
// The rank's ciphertexts bootstrapped in LOCKSTEP: pke's control flow depends on parameters and metadata only, so the K ciphertexts
// (equal metadata: fresh encryptions at one level) are packed into ONE ciphertext whose towers hold K towers each and cc->EvalBootstrap
// runs once — every launch works on K towers, every evaluation key is read once for all of them.  `group` ciphertexts per wide
// evaluation (0 = all of the rank's); one narrow bootstrap must have run before (fbb_bootstrap_all with a warm-up pass: the composites
// are checked against the member-by-member path at their first use, which is a narrow evaluation).  Seconds per pass over all ciphertexts.
// `threads` > 1: the groups are spread over that many host threads (one HIP stream each): while one group's thread is in pke's host code
// between two launches, the other groups' kernels keep the device busy (a lockstep pass on ONE thread leaves the GPU idle for a fifth of
// its span: profiles/r04_bootstrap_wide_kernels.txt).
double fbb_bootstrap_wide_mt(void* h, uint32_t group, int reps, int threads) {
    auto* b    = static_cast<Batch*>(h);
    double sec = -1;
#ifdef WITH_HIP
    Guard(b, [&] {
        const uint32_t n = (uint32_t)b->in.size();
        if (group == 0 || group > n)
            group = n;
        b->out.assign(n, nullptr);
        const int groups = (int)((n + group - 1) / group);
        std::string err;
        auto pass = [&] {
#pragma omp parallel for schedule(dynamic, 1) num_threads(std::max(1, std::min(threads, groups)))
            for (int g = 0; g < groups; ++g) {
              try {
                const uint32_t first = (uint32_t)g * group;
                const uint32_t k = std::min(group, n - first);
                const auto& c0   = b->in[first];
                auto wide        = c0->CloneEmpty();
                std::vector<DCRTPoly> elements;
                for (size_t e = 0; e < c0->GetElements().size(); ++e) {
                    std::vector<const DCRTPoly*> towers;
                    for (uint32_t i = 0; i < k; ++i) {
                        const auto& ci = b->in[first + i];
                        if (ci->GetLevel() != c0->GetLevel() || ci->GetNoiseScaleDeg() != c0->GetNoiseScaleDeg() ||
                            ci->GetScalingFactor() != c0->GetScalingFactor() || ci->GetSlots() != c0->GetSlots() ||
                            ci->GetElements().size() != c0->GetElements().size())
                            OPENFHE_THROW("fbb_bootstrap_wide: the ciphertexts of a group must have equal metadata");
                        towers.push_back(&ci->GetElements()[e]);
                    }
                    elements.push_back(DCRTPoly::PackWide(towers, /*adopt: this driver owns the group's ciphertexts*/ true));
                }
                wide->SetElements(std::move(elements));
                Ciphertext<DCRTPoly> res;
                {
                    hiprt::WidthScope scope(k);  // (towers pke creates on the way — accumulators — are k wide)
                    res = b->cc->EvalBootstrap(wide);
                }
                for (uint32_t i = 0; i < k; ++i) {
                    auto one = res->CloneEmpty();
                    std::vector<DCRTPoly> el;
                    for (const auto& t : res->GetElements())
                        el.push_back(t.UnpackTower(i));
                    one->SetElements(std::move(el));
                    b->out[first + i] = one;
                }
              }
              catch (const std::exception& e) {
#pragma omp critical
                err = e.what();
              }
            }
            if (!err.empty())
                OPENFHE_THROW(err);
            Drain();  // (the pass ends when every stream has run dry)
        };
        pass();
        const auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < reps; ++r)
            pass();
        sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / std::max(1, reps);
    });
#else
    (void)group, (void)reps, (void)threads;
    b->error = "fbb_bootstrap_wide: the stock backend has no wide towers";
#endif
    return sec;
}
double fbb_bootstrap_wide(void* h, uint32_t group, int reps) { return fbb_bootstrap_wide_mt(h, group, reps, 1); }
// keeps the current outputs (of a narrow pass) for fbb_compare_saved: the objects stay alive, the next pass produces new ones
int fbb_save_outputs(void* h) {
    auto* b = static_cast<Batch*>(h);
    return Guard(b, [&] { b->saved = b->out; });
}
// the current outputs (of a wide pass) against the saved ones, limb by limb and word for word on the host: the number of ciphertexts
// that differ in any word, element count, limb count or format; -1 when nothing was saved or the counts differ
long fbb_compare_saved(void* h) {
    auto* b   = static_cast<Batch*>(h);
    long diff = -1;
    Guard(b, [&] {
        if (b->saved.empty() || b->saved.size() != b->out.size())
            return;
        diff = 0;
        for (size_t i = 0; i < b->out.size(); ++i) {
            bool same = b->out[i] && b->saved[i] && b->out[i]->GetElements().size() == b->saved[i]->GetElements().size();
            for (size_t e = 0; same && e < b->out[i]->GetElements().size(); ++e) {
                const auto& x = b->out[i]->GetElements()[e];
                const auto& y = b->saved[i]->GetElements()[e];
                same = x.GetFormat() == y.GetFormat() && x.GetNumOfElements() == y.GetNumOfElements();
                if (!same)
                    break;
                const auto& lx = x.GetAllElements();
                const auto& ly = y.GetAllElements();
                for (size_t l = 0; same && l < lx.size(); ++l) {
                    same = lx[l].GetModulus() == ly[l].GetModulus() && lx[l].GetLength() == ly[l].GetLength();
                    for (uint32_t j = 0; same && j < lx[l].GetLength(); ++j)
                        same = lx[l][j] == ly[l][j];
                }
            }
            diff += same ? 0 : 1;
        }
    });
    return diff;
}
// counters of the backend since the process started: {operand bytes read, operand bytes written (every tower / key an operation touches,
// once per operation), kernel launches, host->device bytes, device->host bytes}; all zero on the stock backend
void fbb_counters(uint64_t out[5]) {
    for (int i = 0; i < 5; ++i)
        out[i] = 0;
#ifdef WITH_HIP
    fhe_hal_operand_bytes(out);
    uint64_t total = 0;
    fhe_hal_launch_stats(nullptr, 0, &total);
    out[2] = total;
    uint64_t st[4];
    fhe_hal_stats(st);
    out[3] = st[2], out[4] = st[3];
#endif
}
// the backend's buffer caches: {cached bytes, takes from another thread's cache, requests that reached the device, cache releases, free
// device bytes, total device bytes, bytes held from the device, high-water mark of that, takes that waited for the buffer's own mark}
void fbb_alloc_stats(uint64_t out[9]) {
    for (int i = 0; i < 9; ++i)
        out[i] = 0;
#ifdef WITH_HIP
    fhe_hal_alloc_stats(out);
    fhe_hal_alloc_stats2(out + 6);
#endif
}
// one buffer of `bytes` into the calling thread's cache before the evaluation (fhe_hal_reserve); 0 = done
int fbb_reserve(uint64_t bytes) {
#ifdef WITH_HIP
    return fhe_hal_reserve(bytes);
#else
    (void)bytes;
    return 1;
#endif
}
// "<member> <device ops> <host-mirror executions> <host reads> <operand bytes>" lines of the backend (empty on the stock backend)
size_t fbb_member_stats(char* buf, size_t cap) {
#ifdef WITH_HIP
    return fhe_hal_member_stats(buf, cap);
#else
    if (buf && cap)
        buf[0] = 0;
    return 1;
#endif
}
// decrypts output i of the rank's slice: the first 8 slots into vals; returns the largest absolute error against the message
double fbb_check(void* h, uint32_t i, double* vals) {
    auto* b = static_cast<Batch*>(h);
    double worst = 1e300;
    Guard(b, [&] {
        Plaintext pt;
        b->cc->Decrypt(b->kp.secretKey, b->out.at(i), &pt);
        pt->SetLength(8);
        const auto v = pt->GetRealPackedValue();
        const auto m = Message(b->first + i);
        worst        = 0;
        for (int k = 0; k < 8; ++k) {
            if (vals)
                vals[k] = v[k];
            worst = std::max(worst, std::abs(v[k] - m[k]));
        }
    });
    return worst;
}
// the limbs of outputs [from, to) of the rank's slice, appended to a file (the byte comparison with the stock backend)
int fbb_dump(void* h, const char* path, uint32_t from, uint32_t to) {
    auto* b = static_cast<Batch*>(h);
    return Guard(b, [&] {
        std::ofstream f(path, std::ios::binary | std::ios::app);
        for (uint32_t i = from; i < to && i < b->out.size(); ++i)
            for (const auto& e : b->out[i]->GetElements()) {
                const auto& limbs = e.GetAllElements();
                uint64_t hdr[3]   = {limbs.size(), e.GetRingDimension(), static_cast<uint64_t>(e.GetFormat())};
                f.write(reinterpret_cast<const char*>(hdr), 24);
                for (const auto& l : limbs)
                    for (uint32_t j = 0; j < l.GetLength(); ++j) {
                        const uint64_t v = l[j].ConvertToInt<uint64_t>();
                        f.write(reinterpret_cast<const char*>(&v), 8);
                    }
            }
    });
}
}
