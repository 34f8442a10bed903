// TEST PROGRAM — the reference's own CryptoContext API (openfhe.h), compiled twice from this one source:
//   * against the stock libraries of oracle/_ref (default DCRTPoly backend), and
//   * against openfhe-development_amd/hal/_build (the same reference sources built with the HIP backend of DCRTPoly).
// Both runs use the deterministic test PRNG, so every key and ciphertext is the same: the program dumps the limbs of every
// ciphertext it produces; tests/test_hal_shim.py compares the two dumps byte for byte and checks the decryptions.
//
//   shim_ckks <out.bin> <prng.so> <mode> [logN]      mode: leveled | bootstrap
#include <atomic>
#include <chrono>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "openfhe.h"
#include "math/distributiongenerator.h"

using namespace lbcrypto;

extern "C" void fhe_hal_stats(uint64_t out[4]) __attribute__((weak));
extern "C" int fhe_hal_available(void) __attribute__((weak));
extern "C" void fhe_hal_trace_reset(void) __attribute__((weak));
extern "C" size_t fhe_hal_member_stats(char* buf, size_t cap) __attribute__((weak));
extern "C" size_t fhe_hal_launch_stats(char* buf, size_t cap, uint64_t* total) __attribute__((weak));
extern "C" uint64_t fhe_hal_memo_hits() __attribute__((weak));
extern "C" void fhe_hal_device_sync(void) __attribute__((weak));
extern "C" uint64_t fhe_hal_cached_bytes(void) __attribute__((weak));
extern "C" void fhe_hal_alloc_stats(uint64_t out[6]) __attribute__((weak));
extern "C" void fhe_hal_alloc_stats2(uint64_t out[3]) __attribute__((weak));
extern "C" int fhe_hal_reserve(uint64_t bytes) __attribute__((weak));
static std::atomic<bool> g_release_thread{false};
extern "C" uint64_t fhe_hal_cached_bytes(void) __attribute__((weak));
extern "C" void fhe_hal_release_caches(void) __attribute__((weak));
static std::map<std::string, uint64_t> launches_by_kernel(uint64_t* total) {
    std::map<std::string, uint64_t> m;
    *total = 0;
    if (!fhe_hal_launch_stats)
        return m;
    std::string buf(fhe_hal_launch_stats(nullptr, 0, total), '\0');
    fhe_hal_launch_stats(&buf[0], buf.size(), total);
    std::istringstream in(buf);
    std::string k;
    uint64_t n;
    while (in >> k >> n)
        m[k] = n;
    return m;
}
extern "C" void fhe_hal_stats_reset(void) __attribute__((weak));
extern "C" void fhe_hal_composite_stats(uint64_t out[3]) __attribute__((weak));
extern "C" void fhe_hal_other_host_counts(uint64_t out[2]) __attribute__((weak));
// the set-up phase (context, keys, encryption: samplers and encoders produce their words on the host) ends here: the counters the
// tests assert on cover the EVALUATION phase (and the decryptions at the end) only
static void print_member_stats(const char* tag) {  // "<tag> <member> <device ops> <host-mirror executions> <host reads after a device->host copy>"
    if (!fhe_hal_member_stats)
        return;
    std::string buf(fhe_hal_member_stats(nullptr, 0), '\0');
    fhe_hal_member_stats(&buf[0], buf.size());
    size_t at = 0;
    while (at < buf.size() && buf[at]) {
        const size_t nl = buf.find('\n', at);
        std::cout << tag << " " << buf.substr(at, nl - at) << std::endl;
        at = nl + 1;
    }
}
static void evaluation_phase_begins() {
    // the set-up window (key generation and encryption) is reported on its own: "halsetup <member> ..." — the tests require the
    // backend's key generation (KeySwitchGenInternal on whole device towers) and the encryption arithmetic to have run on the device
    print_member_stats("halsetup");
    if (fhe_hal_stats) {
        uint64_t st[4];
        fhe_hal_stats(st);
        std::cout << "halsetup: deviceOps " << st[0] << " hostOps " << st[1] << " h2dBytes " << st[2] << " d2hBytes " << st[3] << std::endl;
    }
    if (fhe_hal_stats_reset)
        fhe_hal_stats_reset();
    if (fhe_hal_trace_reset)
        fhe_hal_trace_reset();
}

static std::ofstream g_out;
static void dump(const char* name, const Ciphertext<DCRTPoly>& ct) {
    const auto& els = ct->GetElements();
    uint64_t hdr[2] = {els.size(), 0};
    g_out.write(reinterpret_cast<const char*>(hdr), 16);
    for (const auto& e : els) {
        const auto& limbs = e.GetAllElements();
        uint64_t h2[3]    = {limbs.size(), e.GetRingDimension(), static_cast<uint64_t>(e.GetFormat())};
        g_out.write(reinterpret_cast<const char*>(h2), 24);
        for (const auto& l : limbs) {
            uint64_t q = l.GetModulus().ConvertToInt<uint64_t>();
            g_out.write(reinterpret_cast<const char*>(&q), 8);
            for (uint32_t j = 0; j < l.GetLength(); ++j) {
                uint64_t v = l[j].ConvertToInt<uint64_t>();
                g_out.write(reinterpret_cast<const char*>(&v), 8);
            }
        }
    }
    std::cout << "dumped " << name << ": " << els.size() << " elements x " << els[0].GetNumOfElements() << " limbs" << std::endl;
}
// every word of every ciphertext of a batch, as one 128-bit digest per ciphertext (two independent FNV-1a-style sums over the element /
// limb headers and the residues in order), appended to the output file: byte comparison of the two backends' files then covers ALL
// products of a batch without writing gigabytes
static void dumpDigests(const char* name, const std::vector<Ciphertext<DCRTPoly>>& cts) {
    for (const auto& ct : cts) {
        uint64_t h1 = 0xcbf29ce484222325ull, h2 = 0x9e3779b97f4a7c15ull;
        auto mix = [&](uint64_t v) {
            h1 = (h1 ^ v) * 0x100000001b3ull;
            h2 = (h2 + v) * 0xff51afd7ed558ccdull;
            h2 ^= h2 >> 29;
        };
        const auto& els = ct->GetElements();
        mix(els.size());
        for (const auto& e : els) {
            const auto& limbs = e.GetAllElements();
            mix(limbs.size()), mix(e.GetRingDimension()), mix(static_cast<uint64_t>(e.GetFormat()));
            for (const auto& l : limbs) {
                mix(l.GetModulus().ConvertToInt<uint64_t>());
                for (uint32_t j = 0; j < l.GetLength(); ++j)
                    mix(l[j].ConvertToInt<uint64_t>());
            }
        }
        uint64_t d[2] = {h1, h2};
        g_out.write(reinterpret_cast<const char*>(d), 16);
    }
    std::cout << "dumped digests of " << name << ": " << cts.size() << " ciphertexts" << std::endl;
}
static void show(const char* name, CryptoContext<DCRTPoly>& cc, const PrivateKey<DCRTPoly>& sk, const Ciphertext<DCRTPoly>& ct,
                 size_t n) {
    Plaintext pt;
    cc->Decrypt(sk, ct, &pt);
    pt->SetLength(n);
    auto v = pt->GetRealPackedValue();
    std::cout << "value " << name << ":";
    for (size_t i = 0; i < n; ++i)
        std::printf(" %.6f", v[i]);
    std::cout << std::endl;
}

int main(int argc, char** argv) {
    if (argc < 4) {
        std::cerr << "usage: shim_ckks out.bin prng.so leveled|bootstrap [logN]" << std::endl;
        return 2;
    }
    g_out.open(argv[1], std::ios::binary);
    PseudoRandomNumberGenerator::InitPRNGEngine(argv[2]);
    const std::string mode = argv[3];
    const uint32_t logN    = argc > 4 ? std::atoi(argv[4]) : 11;

#ifdef WITH_HIP
    if (mode == "buffers") {
        // the backend's cache of released device buffers (hip-runtime.cpp Alloc): a released buffer serves a later request of its own or of
        // a smaller size class (best fit up to 4x, the allocation keeps its class), never a larger one; fhe_hal_release_caches() hands
        // everything back to the device
        const size_t M = (size_t)1 << 20;
        auto a           = hiprt::Alloc(3 * M);
        const auto* pa   = a->p;
        const size_t cap = a->cap;
        a.reset();
        const uint64_t cached1 = fhe_hal_cached_bytes();
        auto b = hiprt::Alloc(M);  // a smaller class: takes the released 3 Mi-word allocation
        std::cout << "buffers smaller request reuses the released allocation: " << (b->p == pa && b->cap == cap) << " cached " << cached1
                  << " -> " << fhe_hal_cached_bytes() << std::endl;
        auto c = hiprt::Alloc(3 * M);  // the same class while b holds the allocation: a fresh one
        std::cout << "buffers same class while in use gets a fresh allocation: " << (c->p != pa) << std::endl;
        b.reset();
        auto d = hiprt::Alloc(16 * M);  // a larger class: the released 3 Mi-word allocation cannot serve it
        std::cout << "buffers larger request does not take a smaller allocation: " << (d->p != pa && d->cap >= 16 * M) << std::endl;
        auto e = hiprt::Alloc(M / 8);  // more than 4x smaller than what is cached: a fresh small allocation
        std::cout << "buffers much smaller request leaves the large allocation alone: " << (e->p != pa) << std::endl;
        c.reset(), d.reset(), e.reset();
        {
            // round 5: a buffer released by ANOTHER host thread (its free list) serves this thread's request before the device does, ordered
            // behind that thread's stream; above 1 GiB a size class is a sixteenth of the power of two (34.9 GiB -> 36 GiB, not 48)
            uint64_t before[6], after[6];
            const uint64_t* other = nullptr;
            size_t otherCap       = 0;
            std::thread t([&] {
                auto x   = hiprt::Alloc(40 * M);
                other    = x->p;
                otherCap = x->cap;
                x.reset();  // -> that thread's free list; the thread stays alive until the main thread has taken the buffer
                while (!g_release_thread.load())
                    std::this_thread::yield();
            });
            while (!other || fhe_hal_cached_bytes() < otherCap * 8)
                std::this_thread::yield();
            uint64_t b2[3], a2[3];
            fhe_hal_alloc_stats(before);
            fhe_hal_alloc_stats2(b2);
            auto y = hiprt::Alloc(40 * M);
            fhe_hal_alloc_stats(after);
            fhe_hal_alloc_stats2(a2);
            std::cout << "buffers request served from another thread's cache: " << (y->p == other && after[1] == before[1] + 1 && after[2] == before[2])
                      << std::endl;
            // round 6: the taker waited for the buffer's own completion mark (an event the releasing thread recorded), not for that thread's queue
            std::cout << "buffers the taker waited for the buffer's own completion mark: " << (a2[2] == b2[2] + 1) << std::endl;
            g_release_thread.store(true);
            t.join();
            y.reset();
            auto big = hiprt::Alloc((size_t)150 * M);  // 1.17 GiB: classes of 1/16 x 2 GiB = 128 MiB above 1 GiB
            std::cout << "buffers size class above 1 GiB is a sixteenth step: " << (big->cap == (size_t)160 * M) << std::endl;
        }
        {
            // round 6: fhe_hal_reserve pre-sizes the cache — the next request of that class does not reach the device; the backend's own
            // account of what it holds from the device follows allocations and releases, its high-water mark stays
            uint64_t s0[6], s1[6], h0[3], h1[3], h2[3];
            fhe_hal_release_caches();
            fhe_hal_alloc_stats2(h0);
            const int rc = fhe_hal_reserve((uint64_t)200 * M * 8);
            fhe_hal_alloc_stats(s0);
            fhe_hal_alloc_stats2(h1);
            auto w = hiprt::Alloc((size_t)200 * M);
            fhe_hal_alloc_stats(s1);
            std::cout << "buffers a reserved buffer serves the first large request: "
                      << (rc == 0 && s1[2] == s0[2] && h1[0] >= h0[0] + (uint64_t)200 * M * 8 && h1[1] >= h1[0]) << std::endl;
            w.reset();
            fhe_hal_release_caches();
            fhe_hal_alloc_stats2(h2);
            std::cout << "buffers the held bytes go down on release and the high-water mark stays: " << (h2[0] + (uint64_t)200 * M * 8 <= h1[0] && h2[1] == h1[1])
                      << std::endl;
        }
        auto keep = hiprt::Alloc(M);
        keep.reset();
        const uint64_t cached2 = fhe_hal_cached_bytes();
        fhe_hal_release_caches();
        std::cout << "buffers released caches: " << cached2 << " -> " << fhe_hal_cached_bytes() << std::endl;
        return 0;
    }
#endif
    if (mode == "leveled") {
        CCParams<CryptoContextCKKSRNS> p;
        p.SetSecurityLevel(HEStd_NotSet);
        p.SetRingDim(1u << logN);
        p.SetMultiplicativeDepth(4);
        p.SetScalingModSize(50);
        p.SetFirstModSize(60);
        p.SetKeySwitchTechnique(HYBRID);
        p.SetNumLargeDigits(2);
        // argv[5]: FIXEDMANUAL (default) | FIXEDAUTO | FLEXIBLEAUTO | FLEXIBLEAUTOEXT (the library default): the automatic
        // techniques rescale inside EvalMult / adjust levels inside EvalAdd, the explicit Rescale calls below are then no-ops
        const std::string st = argc > 5 ? argv[5] : "FIXEDMANUAL";
        p.SetScalingTechnique(st == "FIXEDAUTO" ? FIXEDAUTO : st == "FLEXIBLEAUTO" ? FLEXIBLEAUTO : st == "FLEXIBLEAUTOEXT" ? FLEXIBLEAUTOEXT : FIXEDMANUAL);
        auto cc = GenCryptoContext(p);
        cc->Enable(PKE);
        cc->Enable(KEYSWITCH);
        cc->Enable(LEVELEDSHE);
        auto kp = cc->KeyGen();
        cc->EvalMultKeyGen(kp.secretKey);
        cc->EvalRotateKeyGen(kp.secretKey, {1, -2});
        std::vector<double> x = {0.25, 0.5, 0.75, 1.0, 2.0, 3.0, 0.4, 0.5}, y = {1.0, 2.0, 0.5, 0.25, -1.0, 0.125, 0.3, -0.5};
        auto cx = cc->Encrypt(kp.publicKey, cc->MakeCKKSPackedPlaintext(x));
        auto cy = cc->Encrypt(kp.publicKey, cc->MakeCKKSPackedPlaintext(y));
        dump("x", cx);
        dump("y", cy);
        evaluation_phase_begins();
        auto m = cc->EvalMult(cx, cy);  // EvalMultCore + HYBRID KeySwitchCore (base-leveledshe.cpp:201-214)
        dump("x*y", m);
        auto r = cc->Rescale(m);  // DropLastElementAndScale
        dump("rescale", r);
        auto rot = cc->EvalRotate(r, 1);  // EvalAutomorphism: key switch + AutomorphismTransform
        dump("rotate1", rot);
        auto rot2 = cc->EvalRotate(rot, -2);
        dump("rotate-2", rot2);
        auto s = cc->EvalAdd(rot2, r);
        auto sq = cc->Rescale(cc->EvalMult(s, s));  // a second multiplication one level down
        dump("square", sq);
        auto d = cc->Rescale(cc->EvalMult(sq, 0.5));  // plaintext-constant path, a third level
        dump("final", d);
        show("x*y", cc, kp.secretKey, r, 8);
        show("rot", cc, kp.secretKey, rot2, 8);
        show("final", cc, kp.secretKey, d, 8);
    }
    else if (mode == "memo") {
        // Results of DropLastElementAndScale / Times(constants) are remembered per device buffer while clones share it (DevBuf::memo):
        // a second clone takes the remembered result, and a write to the source's words must drop it.
        CCParams<CryptoContextCKKSRNS> p;
        p.SetSecurityLevel(HEStd_NotSet);
        p.SetRingDim(1u << logN);
        p.SetMultiplicativeDepth(3);
        p.SetScalingModSize(50);
        p.SetFirstModSize(60);
        p.SetKeySwitchTechnique(HYBRID);
        p.SetScalingTechnique(FIXEDMANUAL);
        auto cc = GenCryptoContext(p);
        cc->Enable(PKE);
        cc->Enable(KEYSWITCH);
        cc->Enable(LEVELEDSHE);
        auto kp = cc->KeyGen();
        cc->EvalMultKeyGen(kp.secretKey);
        std::vector<double> x = {0.25, 0.5, 0.75, 1.0, 2.0, 3.0, 0.4, 0.5};
        auto cx = cc->Encrypt(kp.publicKey, cc->MakeCKKSPackedPlaintext(x));
        evaluation_phase_begins();
        auto t = cc->EvalMult(cx, cx);
        auto a = t->Clone();
        cc->RescaleInPlace(a);  // computed, remembered on t's buffers (shared with the clone)
        auto b = t->Clone();
        cc->RescaleInPlace(b);  // the remembered result
        const uint64_t hits1 = fhe_hal_memo_hits ? fhe_hal_memo_hits() : 0;
        cc->EvalAddInPlace(t, t);  // t's words change in place (no clone shares them any more): what was remembered is history
        auto c = t->Clone();
        cc->RescaleInPlace(c);  // must be rescale(2 x*x), not the result remembered for x*x
        auto e = cc->EvalMult(t->Clone(), 0.5);  // Times(constants) on a clone ...
        auto f = cc->EvalMult(t->Clone(), 0.5);  // ... and its remembered result
        dump("a", a);
        dump("b", b);
        dump("c", c);
        dump("e", e);
        dump("f", f);
        show("a", cc, kp.secretKey, a, 8);
        show("c", cc, kp.secretKey, c, 8);
        std::cout << "memo hits after the second clone " << hits1 << " at the end " << (fhe_hal_memo_hits ? fhe_hal_memo_hits() : 0) << std::endl;
    }
    else if (mode == "bfv") {
        // BFV (BASELINE configs[4]): EvalMult in each of the reference's multiplication techniques (bfvrns-leveledshe.cpp:198-445:
        // BEHZ = FastBaseConvqToBskMontgomery / FastRNSFloorq / FastBaseConvSK; HPS* = ExpandCRTBasis, FastExpandCRTBasisPloverQ,
        // ScaleAndRound, SwitchCRTBasis, ExpandCRTBasisQlHat), HYBRID relinearisation, a rotation, a second multiplication
        const std::string tech = argc > 5 ? argv[5] : "BEHZ";
        CCParams<CryptoContextBFVRNS> p;
        p.SetSecurityLevel(HEStd_NotSet);
        p.SetRingDim(1u << logN);
        p.SetPlaintextModulus(65537);
        p.SetMultiplicativeDepth(argc > 6 ? std::atoi(argv[6]) : 2);
        p.SetScalingModSize(60);
        p.SetKeySwitchTechnique(HYBRID);
        p.SetMultiplicationTechnique(tech == "BEHZ" ? BEHZ : tech == "HPS" ? HPS : tech == "HPSPOVERQ" ? HPSPOVERQ : HPSPOVERQLEVELED);
        auto cc = GenCryptoContext(p);
        cc->Enable(PKE);
        cc->Enable(KEYSWITCH);
        cc->Enable(LEVELEDSHE);
        auto kp = cc->KeyGen();
        cc->EvalMultKeyGen(kp.secretKey);
        cc->EvalRotateKeyGen(kp.secretKey, {1});
        std::vector<int64_t> x = {1, 2, 3, 4, 5, 6, 7, 8}, y = {3, -2, 5, 1, -4, 2, 9, -7};
        auto cx = cc->Encrypt(kp.publicKey, cc->MakePackedPlaintext(x));
        auto cy = cc->Encrypt(kp.publicKey, cc->MakePackedPlaintext(y));
        dump("x", cx);
        dump("y", cy);
        evaluation_phase_begins();
        auto t0 = std::chrono::steady_clock::now();
        auto m3 = cc->EvalMultNoRelin(cx, cy);
        dump("x*y (3 elements)", m3);
        auto m = cc->EvalMult(cx, cy);
        dump("x*y", m);
        auto r = cc->EvalRotate(m, 1);
        dump("rotate1", r);
        auto m2 = cc->EvalMult(cc->EvalAdd(r, cx), m);  // second level
        dump("second", m2);
        auto sq = cc->EvalSquare(cx);
        dump("square", sq);
        std::cout << "bfv " << tech << " ring 2^" << logN << " limbs " << cx->GetElements()[0].GetNumOfElements() << " eval seconds "
                  << std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() << std::endl;
        for (auto& pr : std::vector<std::pair<const char*, Ciphertext<DCRTPoly>>>{{"x*y", m}, {"second", m2}, {"square", sq}}) {
            Plaintext pt;
            cc->Decrypt(kp.secretKey, pr.second, &pt);
            pt->SetLength(8);
            std::cout << "value " << pr.first << ": " << pt->GetPackedValue() << std::endl;
        }
        if (argc > 7) {  // timing: EvalMult (with relinearisation) repeated
            const int reps = std::atoi(argv[7]);
            auto w         = cc->EvalMult(cx, cy);
            (void)w->GetElements()[0].GetElementAtIndex(0);
            t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < reps; ++i)
                w = cc->EvalMult(cx, cy);
            (void)w->GetElements()[0].GetElementAtIndex(0);
            std::cout << "bfv EvalMult seconds " << std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / reps << " ("
                      << reps << " reps)" << std::endl;
        }
    }
    else if (mode == "behztables") {
        // The BEHZ members with tables that are NOT CryptoParametersBFVRNS's: a backend that derived its own tables from the moduli
        // instead of using the arguments would compute something else than the reference here.  Run 0 passes the parameters' own
        // tables, run 1 a perturbed copy of every table set (entries kept below their moduli, Shoup companions recomputed).
        CCParams<CryptoContextBFVRNS> p;
        p.SetSecurityLevel(HEStd_NotSet);
        p.SetRingDim(1u << logN);
        p.SetPlaintextModulus(65537);
        p.SetMultiplicativeDepth(2);
        p.SetScalingModSize(60);
        p.SetKeySwitchTechnique(HYBRID);
        p.SetMultiplicationTechnique(BEHZ);
        auto cc = GenCryptoContext(p);
        cc->Enable(PKE);
        cc->Enable(KEYSWITCH);
        cc->Enable(LEVELEDSHE);
        auto kp = cc->KeyGen();
        std::vector<int64_t> x = {1, 2, 3, 4, 5, 6, 7, 8};
        auto cx       = cc->Encrypt(kp.publicKey, cc->MakePackedPlaintext(x));
        const auto cp = std::dynamic_pointer_cast<CryptoParametersBFVRNS>(cc->GetCryptoParameters());
        auto dumpTower = [&](const char* name, const DCRTPoly& e) {
            const auto& limbs = e.GetAllElements();
            uint64_t h2[3]    = {limbs.size(), e.GetRingDimension(), static_cast<uint64_t>(e.GetFormat())};
            g_out.write(reinterpret_cast<const char*>(h2), 24);
            for (const auto& l : limbs)
                for (uint32_t j = 0; j < l.GetLength(); ++j) {
                    uint64_t v = l[j].ConvertToInt<uint64_t>();
                    g_out.write(reinterpret_cast<const char*>(&v), 8);
                }
            std::cout << "dumped " << name << ": " << limbs.size() << " limbs" << std::endl;
        };
        auto precon = [](const std::vector<NativeInteger>& v, const std::vector<NativeInteger>& mod) {
            std::vector<NativeInteger> r(v.size());
            for (size_t i = 0; i < v.size(); ++i)
                r[i] = v[i].PrepModMulConst(mod[i]);
            return r;
        };
        auto bump = [](NativeInteger& v, const NativeInteger& mod, uint64_t by) { v = (v + NativeInteger(by)).Mod(mod); };
        evaluation_phase_begins();
        for (int run = 0; run < 2; ++run) {
            const auto& Q   = cp->GetModuliQ();
            const auto& Bsk = cp->GetModuliBsk();
            auto mtQHatInv = cp->GetmtildeQHatInvModq();
            auto QHatModbsk = cp->GetQHatModbsk();
            auto QHatModmt  = cp->GetQHatModmtilde();
            auto QModbsk    = cp->GetQModbsk();
            uint64_t negQInv = cp->GetNegQInvModmtilde();
            auto mtInvModbsk = cp->GetmtildeInvModbsk();
            auto tQHatInv    = cp->GettQHatInvModq();
            auto qInvModbsk  = cp->GetqInvModbsk();
            auto tQInvModbsk = cp->GettQInvModbsk();
            auto BHatInv     = cp->GetBHatInvModb();
            auto BHatModmsk  = cp->GetBHatModmsk();
            auto BInvModmsk  = cp->GetBInvModmsk();
            auto BHatModq    = cp->GetBHatModq();
            auto BModq       = cp->GetBModq();
            const NativeInteger msk = Bsk.back();
            std::vector<NativeInteger> B(Bsk.begin(), Bsk.end() - 1);
            if (run == 1) {
                bump(mtQHatInv[0], Q[0], 1);
                bump(QHatModbsk[0][1], Bsk[1], 3);
                QHatModmt[0] = (QHatModmt[0] + 7) & 0xffff;
                bump(QModbsk[0], Bsk[0], 11);
                negQInv = (negQInv ^ 0x55) & 0xffff;
                bump(mtInvModbsk[1], Bsk[1], 5);
                bump(tQHatInv[0], Q[0], 1);
                bump(qInvModbsk[1][0], Bsk[0], 2);
                bump(tQInvModbsk[0], Bsk[0], 9);
                bump(BHatInv[0], B[0], 1);
                bump(BHatModmsk[0], msk, 4);
                bump(BInvModmsk, msk, 1);
                bump(BHatModq[0][0], Q[0], 1);
                bump(BModq[0], Q[0], 5);
            }
            DCRTPoly a = cx->GetElements()[0];
            a.FastBaseConvqToBskMontgomery(cp->GetParamsQBsk(), Q, Bsk, cp->GetModbskBarrettMu(), mtQHatInv, precon(mtQHatInv, Q), QHatModbsk,
                                           QHatModmt, QModbsk, precon(QModbsk, Bsk), negQInv, mtInvModbsk, precon(mtInvModbsk, Bsk));
            dumpTower(run ? "q->Bsk, perturbed tables" : "q->Bsk", a);
            a.SetFormat(Format::COEFFICIENT);
            a.FastRNSFloorq(NativeInteger(65537), Q, Bsk, cp->GetModbskBarrettMu(), tQHatInv, precon(tQHatInv, Q), QHatModbsk, qInvModbsk,
                            tQInvModbsk, precon(tQInvModbsk, Bsk));
            dumpTower(run ? "floor, perturbed tables" : "floor", a);
            a.FastBaseConvSK(cp->GetElementParams(), cp->GetModqBarrettMu(), Bsk, cp->GetModbskBarrettMu(), BHatInv, precon(BHatInv, B), BHatModmsk,
                             BInvModmsk, BInvModmsk.PrepModMulConst(msk), BHatModq, BModq, precon(BModq, Q));
            dumpTower(run ? "SK, perturbed tables" : "SK", a);
        }
    }
    else if (mode == "ftt") {
        // ChineseRemainderTransformFTT<NativeVector> (math/math-hal.h:60-106) directly and through NativePoly::SwitchFormat (poly-impl.h:420-440):
        // rings 2^lo .. 2^logN, a 59-bit NTT prime each, deterministic coefficients; every transform's output is dumped
        const uint32_t lo = argc > 5 ? std::atoi(argv[5]) : 12;
        evaluation_phase_begins();
        for (uint32_t ln = lo; ln <= logN; ++ln) {
            const uint32_t N = 1u << ln, M = 2 * N;
            const NativeInteger q = LastPrime<NativeInteger>(59, M), root = RootOfUnity<NativeInteger>(M, q);
            auto params = std::make_shared<ILNativeParams>(M, q, root);
            NativeVector v(N, q);
            uint64_t x = 0x9E3779B97F4A7C15ull * (ln + 1);
            for (uint32_t i = 0; i < N; ++i) {
                x ^= x << 13, x ^= x >> 7, x ^= x << 17;
                v[i] = NativeInteger(x).Mod(q);
            }
            v[0] = NativeInteger(0), v[1] = q - NativeInteger(1);
            auto dumpVec = [&](const NativeVector& w) {
                for (uint32_t i = 0; i < w.GetLength(); ++i) {
                    const uint64_t u = w[i].ConvertToInt<uint64_t>();
                    g_out.write(reinterpret_cast<const char*>(&u), 8);
                }
            };
            NativePoly p(params, Format::COEFFICIENT, true);
            p.SetValues(v, Format::COEFFICIENT);
            p.SwitchFormat();  // forward, in place
            dumpVec(p.GetValues());
            p.SwitchFormat();  // inverse, in place: the round trip
            dumpVec(p.GetValues());
            if (p.GetValues() != v) {
                std::cerr << "ftt: round trip failed at 2^" << ln << std::endl;
                return 1;
            }
            NativeVector f(N, q), b(N, q);
            ChineseRemainderTransformFTT<NativeVector>().ForwardTransformToBitReverse(v, root, M, &f);  // out of place
            dumpVec(f);
            ChineseRemainderTransformFTT<NativeVector>().InverseTransformFromBitReverse(f, root, M, &b);
            dumpVec(b);
            std::cout << "ftt ring 2^" << ln << " modulus " << q << " done" << std::endl;
        }
    }
    else if (mode == "multbatch") {
        // BASELINE configs[2] through the reference's CryptoContext: B ciphertexts at N = 2^logN, depth 20 (l = 21, dnum = 3),
        // cc->EvalMult (tensor + HYBRID key switch) on each, the ciphertexts spread over host threads (one OpenMP loop over the
        // batch: pke's own inner loops then run inside each thread); the products are dumped for the byte comparison
        const uint32_t depth = argc > 5 ? std::atoi(argv[5]) : 20;
        const int B          = argc > 6 ? std::atoi(argv[6]) : 64;
        const int reps       = argc > 7 ? std::atoi(argv[7]) : 3;
        CCParams<CryptoContextCKKSRNS> p;
        p.SetSecurityLevel(HEStd_NotSet);
        p.SetRingDim(1u << logN);
        p.SetMultiplicativeDepth(depth);
        p.SetScalingModSize(59);
        p.SetFirstModSize(60);
        p.SetKeySwitchTechnique(HYBRID);
        p.SetScalingTechnique(FLEXIBLEAUTO);
        auto cc = GenCryptoContext(p);
        cc->Enable(PKE);
        cc->Enable(KEYSWITCH);
        cc->Enable(LEVELEDSHE);
        auto kp = cc->KeyGen();
        cc->EvalMultKeyGen(kp.secretKey);
        std::vector<Ciphertext<DCRTPoly>> a(B), b(B), c(B);
        for (int i = 0; i < B; ++i) {
            std::vector<double> x = {0.25 + i, 0.5, -1.0}, y = {2.0, 0.5 * i, 3.0};
            a[i] = cc->Encrypt(kp.publicKey, cc->MakeCKKSPackedPlaintext(x));
            b[i] = cc->Encrypt(kp.publicKey, cc->MakeCKKSPackedPlaintext(y));
        }
        const auto cp = std::dynamic_pointer_cast<CryptoParametersRNS>(cc->GetCryptoParameters());
        std::cout << "multbatch ring 2^" << logN << " sizeQ " << a[0]->GetElements()[0].GetNumOfElements() << " sizeP "
                  << cp->GetParamsP()->GetParams().size() << " dnum " << cp->GetNumPartQ() << " ciphertexts " << B << std::endl;
        evaluation_phase_begins();
        // argv[8] = G > 0 (HIP backend only): the ciphertexts in LOCKSTEP, G at a time as one ciphertext whose towers hold G towers each
        // (wide towers): cc->EvalMult runs once per group, every launch works on G towers
        const int group = argc > 8 ? std::atoi(argv[8]) : 0;
        double tPack = 0, tMult = 0, tUnpack = 0, tDrain = 0;  // host seconds of the lockstep pass's phases (printed with FHE_SHIM_TIMING)
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double>(now() - t).count(); };
        auto pass = [&] {
#ifdef WITH_HIP
            if (group > 0) {
                static const bool marks = std::getenv("FHE_SHIM_TIMING") != nullptr;
                for (int first = 0; first < B; first += group) {
                    if (marks)
                        std::cerr << "shim: group " << first / group << " cached MiB " << (fhe_hal_cached_bytes ? fhe_hal_cached_bytes() >> 20 : 0) << std::endl;
                    auto tp = now();
                    const int k = std::min(group, B - first);
                    auto packed = [&](const std::vector<Ciphertext<DCRTPoly>>& v) {
                        auto w = v[first]->CloneEmpty();
                        std::vector<DCRTPoly> el;
                        for (size_t e = 0; e < v[first]->GetElements().size(); ++e) {
                            std::vector<const DCRTPoly*> towers;
                            for (int i = 0; i < k; ++i)
                                towers.push_back(&v[first + i]->GetElements()[e]);
                            el.push_back(DCRTPoly::PackWide(towers, /*adopt: this driver owns the group's ciphertexts*/ true));
                        }
                        w->SetElements(std::move(el));
                        return w;
                    };
                    auto wa = packed(a), wb = packed(b);
                    tPack += since(tp), tp = now();
                    Ciphertext<DCRTPoly> wc;
                    {
                        hiprt::WidthScope scope(k);
                        wc = cc->EvalMult(wa, wb);
                    }
                    tMult += since(tp), tp = now();
                    for (int i = 0; i < k; ++i) {
                        auto one = wc->CloneEmpty();
                        std::vector<DCRTPoly> el;
                        for (const auto& t : wc->GetElements())
                            el.push_back(t.UnpackTower(i));
                        // (the first and the last product of the batch are the ones the program dumps: they get buffers of their own —
                        // DCRTPoly::Detach, the call for outputs that outlive their group — and must still be the stock backend's bytes)
                        if (first + i == 0 || first + i + 1 == (int)c.size())
                            for (auto& e : el)
                                e.Detach();
                        one->SetElements(std::move(el));
                        c[first + i] = one;
                    }
                    tUnpack += since(tp);
                }
                return;  // (the caller drains the device queue once, behind the last timed pass: a pipeline's throughput)
            }
#endif
#pragma omp parallel for schedule(dynamic, 1)
            for (int i = 0; i < B; ++i)
                c[i] = cc->EvalMult(a[i], b[i]);
            for (int i = 0; i < B; i += std::max(1, B / 4))
                (void)c[i]->GetElements()[0].GetElementAtIndex(0);  // drain the device queue
        };
        if (group > 0) {  // (the composites' first use at this level is a narrow evaluation)
            auto warm = cc->EvalMult(a[0], b[0]);
            (void)warm->GetElements()[0].GetElementAtIndex(0);
        }
        // Lockstep passes are timed as a pipeline: the passes are enqueued back to back and the device queue is drained once, behind the
        // last one (a wait after every pass lets the idle device drop into a power state it needs ~3 ms to leave: the packed pass's few
        // hundred microseconds of host work between the wait and the first launch cost 2.9 ms per pass, profiles/r05_sweeps.md section 2)
        auto drainAll = [&] {
#ifdef WITH_HIP
            if (group > 0) {
                auto td = now();
                fhe_hal_device_sync();
                tDrain += since(td);
            }
#endif
        };
        pass();
        drainAll();
        auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < reps; ++r)
            pass();
        drainAll();
        const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / reps;
        std::cout << "multbatch seconds per pass " << sec << " EvalMult per second " << B / sec << std::endl;
        if (group > 0 && std::getenv("FHE_SHIM_TIMING"))
            std::cout << "multbatch host seconds per pass: pack " << tPack / (reps + 1) << " EvalMult " << tMult / (reps + 1) << " unpack "
                      << tUnpack / (reps + 1) << " drain " << tDrain / (reps + 1) << std::endl;
#ifdef WITH_HIP
        if (group > 0) {
            // the same with the operands RESIDENT as wide ciphertexts (packed before, unpacked after the timed region): what a caller that
            // keeps its batch wide between operations gets; the unpacked products replace c[] so that the dumps below cover this path
            std::vector<Ciphertext<DCRTPoly>> WA, WB, WC;
            for (int first = 0; first < B; first += group) {
                const int k = std::min(group, B - first);
                auto packed = [&](const std::vector<Ciphertext<DCRTPoly>>& v) {
                    auto w = v[first]->CloneEmpty();
                    std::vector<DCRTPoly> el;
                    for (size_t e = 0; e < v[first]->GetElements().size(); ++e) {
                        std::vector<const DCRTPoly*> towers;
                        for (int i = 0; i < k; ++i)
                            towers.push_back(&v[first + i]->GetElements()[e]);
                        el.push_back(DCRTPoly::PackWide(towers, /*adopt: this driver owns the group's ciphertexts*/ true));
                    }
                    w->SetElements(std::move(el));
                    return w;
                };
                WA.push_back(packed(a)), WB.push_back(packed(b));
            }
            WC.resize(WA.size());
            auto resident = [&] {
                for (size_t g = 0; g < WA.size(); ++g) {
                    hiprt::WidthScope scope((uint32_t)std::min(group, B - (int)g * group));
                    WC[g] = cc->EvalMult(WA[g], WB[g]);
                }
            };
            resident();
            fhe_hal_device_sync();
            t0 = std::chrono::steady_clock::now();
            for (int r = 0; r < reps; ++r)
                resident();
            fhe_hal_device_sync();
            const double rsec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / reps;
            std::cout << "multbatch resident seconds per pass " << rsec << " EvalMult per second " << B / rsec << std::endl;
            const auto packedPass = c;  // (the products of the timed passes above: pack, multiply, unpack)
            for (size_t g = 0; g < WC.size(); ++g)
                for (int i = 0; i < std::min(group, B - (int)g * group); ++i) {
                    auto one = WC[g]->CloneEmpty();
                    std::vector<DCRTPoly> el;
                    for (const auto& t : WC[g]->GetElements())
                        el.push_back(t.UnpackTower(i));
                    one->SetElements(std::move(el));
                    c[g * group + i] = one;
                }
            int differing = 0;
            for (int i = 0; i < B; ++i) {
                const auto& ex = c[i]->GetElements();  // (const views: reading must not count as a host-side write access)
                const auto& ey = packedPass[i]->GetElements();
                bool same = ex.size() == ey.size();
                for (size_t e = 0; same && e < ex.size(); ++e) {
                    const auto& x = ex[e].GetAllElements();
                    const auto& y = ey[e].GetAllElements();
                    same = x.size() == y.size();
                    for (size_t l = 0; same && l < x.size(); ++l)
                        same = x[l].GetValues() == y[l].GetValues();
                }
                differing += same ? 0 : 1;
            }
            std::cout << "multbatch resident products differing from the packed pass's: " << differing << " of " << B << std::endl;
        }
#endif
        dump("product 0", c[0]);
        dump("product last", c[B - 1]);
        dumpDigests("all products", c);
        show("product 0", cc, kp.secretKey, c[0], 3);
    }
    else if (mode == "bgv") {
        // BGV: EvalMult + HYBRID key switch (ApproxModDown with t > 0), ModReduce, a rotation, a second level
        CCParams<CryptoContextBGVRNS> p;
        p.SetSecurityLevel(HEStd_NotSet);
        p.SetRingDim(1u << logN);
        p.SetPlaintextModulus(65537);
        p.SetMultiplicativeDepth(3);
        // argv[6] = "BV<digit size>": the BV key switch (DCRTPoly::CRTDecompose cuts the digits: dcrtpoly-impl.h:230-285, keyswitch-bv.cpp:254)
        const std::string ks = argc > 6 ? argv[6] : "HYBRID";
        if (ks.rfind("BV", 0) == 0) {
            p.SetKeySwitchTechnique(BV);
            p.SetDigitSize((uint32_t)std::stoul(ks.substr(2)));
        }
        else
            p.SetKeySwitchTechnique(HYBRID);
        const std::string st = argc > 5 ? argv[5] : "FIXEDMANUAL";
        p.SetScalingTechnique(st == "FIXEDAUTO" ? FIXEDAUTO : st == "FLEXIBLEAUTO" ? FLEXIBLEAUTO : st == "FLEXIBLEAUTOEXT" ? FLEXIBLEAUTOEXT : FIXEDMANUAL);
        auto cc = GenCryptoContext(p);
        cc->Enable(PKE);
        cc->Enable(KEYSWITCH);
        cc->Enable(LEVELEDSHE);
        auto kp = cc->KeyGen();
        cc->EvalMultKeyGen(kp.secretKey);
        cc->EvalRotateKeyGen(kp.secretKey, {1});
        std::vector<int64_t> x = {1, 2, 3, 4, 5, 6, 7, 8}, y = {3, -2, 5, 1, -4, 2, 9, -7};
        auto cx = cc->Encrypt(kp.publicKey, cc->MakePackedPlaintext(x));
        auto cy = cc->Encrypt(kp.publicKey, cc->MakePackedPlaintext(y));
        dump("x", cx);
        evaluation_phase_begins();
        auto m = cc->EvalMult(cx, cy);
        dump("x*y", m);
        auto r = cc->ModReduce(m);
        dump("modreduce", r);
        auto rot = cc->EvalRotate(r, 1);
        dump("rotate1", rot);
        auto m2 = cc->ModReduce(cc->EvalMult(cc->EvalAdd(rot, r), r));
        dump("second", m2);
        for (auto& pr : std::vector<std::pair<const char*, Ciphertext<DCRTPoly>>>{{"x*y", r}, {"second", m2}}) {
            Plaintext pt;
            cc->Decrypt(kp.secretKey, pr.second, &pt);
            pt->SetLength(8);
            std::cout << "value " << pr.first << ": " << pt->GetPackedValue() << std::endl;
        }
    }
    else if (mode == "opbench") {
        // host cost per DCRTPoly operation: small towers (the kernels take a few microseconds), many repetitions, one thread
        const uint32_t L = argc > 5 ? std::atoi(argv[5]) : 8;
        CCParams<CryptoContextCKKSRNS> p;
        p.SetSecurityLevel(HEStd_NotSet);
        p.SetRingDim(1u << logN);
        p.SetMultiplicativeDepth(L - 1);
        p.SetScalingModSize(50);
        p.SetFirstModSize(60);
        p.SetScalingTechnique(FIXEDMANUAL);
        auto cc = GenCryptoContext(p);
        cc->Enable(PKE);
        cc->Enable(LEVELEDSHE);
        auto kp = cc->KeyGen();
        std::vector<double> x = {0.25, 0.5};
        auto ct               = cc->Encrypt(kp.publicKey, cc->MakeCKKSPackedPlaintext(x));
        DCRTPoly a = ct->GetElements()[0], b = ct->GetElements()[1];
        std::vector<NativeInteger> consts(a.GetNumOfElements(), NativeInteger(12345));
        auto time = [&](const char* what, int reps, auto&& body) {
            body();
            (void)a.GetElementAtIndex(0);  // drain the device queue
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < reps; ++i)
                body();
            auto t1 = std::chrono::steady_clock::now();
            (void)a.GetElementAtIndex(0);
            auto t2 = std::chrono::steady_clock::now();
            std::printf("%-28s %8.2f us/op issue   %8.2f us/op incl. drain\n", what, std::chrono::duration<double, std::micro>(t1 - t0).count() / reps,
                        std::chrono::duration<double, std::micro>(t2 - t0).count() / reps);
        };
        const int R = 2000;
        time("a += b", R, [&] { a += b; });
        time("c = a + b", R, [&] { DCRTPoly c = a + b; });
        time("c = a * b", R, [&] { DCRTPoly c = a * b; });
        time("copy (shares words)", R, [&] { DCRTPoly c(a); });
        time("copy then += (COW)", R, [&] { DCRTPoly c(a); c += b; });
        time("Times(vector<NativeInt>)", R, [&] { DCRTPoly c = a.Times(consts); });
        time("SwitchFormat x2", R / 2, [&] { a.SwitchFormat(); a.SwitchFormat(); });
        time("Negate", R, [&] { DCRTPoly c = a.Negate(); });
        time("GetNumOfElements/GetParams", R, [&] { volatile auto n = a.GetNumOfElements() + a.GetParams()->GetParams().size(); (void)n; });
    }
    else if (mode == "bootkeys" || mode == "boottime") {
        // BASELINE configs[3]: benchmark/src/ckks-bootstrapping.cpp:70 {2^17, 2^16 slots, 59, 60, auto digits, 5 levels after,
        // {4,4}, SPARSE_TERNARY, FLEXIBLEAUTO} (ring overridable).  bootkeys: size of the rotation-key set; boottime: + one timed EvalBootstrap
        CCParams<CryptoContextCKKSRNS> p;
        SecretKeyDist skd = SPARSE_TERNARY;
        p.SetSecretKeyDist(skd);
        p.SetSecurityLevel(HEStd_NotSet);
        p.SetRingDim(1u << logN);
        p.SetScalingTechnique(FLEXIBLEAUTO);
        p.SetScalingModSize(59);
        p.SetFirstModSize(60);
        p.SetKeySwitchTechnique(HYBRID);
        std::vector<uint32_t> levelBudget = {4, 4};
        const usint depth = 5 + FHECKKSRNS::GetBootstrapDepth(levelBudget, skd);
        p.SetMultiplicativeDepth(depth);
        auto cc = GenCryptoContext(p);
        cc->Enable(PKE);
        cc->Enable(KEYSWITCH);
        cc->Enable(LEVELEDSHE);
        cc->Enable(ADVANCEDSHE);
        cc->Enable(FHE);
        const usint slots = argc > 5 ? std::atoi(argv[5]) : (1u << (logN - 1));
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
            return std::chrono::duration<double>(b - a).count();
        };
        auto t0 = now();
        cc->EvalBootstrapSetup(levelBudget, {0, 0}, slots, 0, mode == "boottime");
        std::cout << "setup seconds " << secs(t0, now()) << std::endl;
        auto kp = cc->KeyGen();
        cc->EvalMultKeyGen(kp.secretKey);
        t0 = now();
        cc->EvalBootstrapKeyGen(kp.secretKey, slots);
        std::cout << "keygen seconds " << secs(t0, now()) << std::endl;
        const auto cp  = std::dynamic_pointer_cast<CryptoParametersRNS>(cc->GetCryptoParameters());
        const auto& km = cc->GetEvalAutomorphismKeyMap(kp.secretKey->GetKeyTag());
        size_t bytes = 0;
        for (const auto& kv : km)
            for (const auto* vec : {&kv.second->GetAVector(), &kv.second->GetBVector()})
                for (const auto& e : *vec)
                    bytes += (size_t)e.GetNumOfElements() * e.GetRingDimension() * 8;
        std::cout << "config4 ring 2^" << logN << " slots " << slots << " depth " << depth << " sizeQ " << cp->GetElementParams()->GetParams().size()
                  << " sizeP " << cp->GetParamsP()->GetParams().size() << " dnum " << cp->GetNumPartQ() << " rotation keys " << km.size()
                  << " key set GB " << bytes / 1e9 << std::endl;
        if (mode == "boottime") {
            std::vector<double> x = {0.25, 0.5, 0.75, 1.0, 2.0, 3.0, 4.0, 5.0};
            auto pt = cc->MakeCKKSPackedPlaintext(x, 1, depth - 1, nullptr, slots);
            auto c  = cc->Encrypt(kp.publicKey, pt);
            // argv[7]: iterations of EvalBootstrap (2 = the two-iteration "Meta-BTS" variant, ckksrns-fhe.cpp:464-519), argv[8]: its precision
            const uint32_t iters = argc > 7 ? std::atoi(argv[7]) : 1, prec = argc > 8 ? std::atoi(argv[8]) : 0;
            auto b  = cc->EvalBootstrap(c, iters, prec);  // warm-up (tables, plans)
            const int reps = argc > 6 ? std::atoi(argv[6]) : 1;
            if (fhe_hal_trace_reset)
                fhe_hal_trace_reset();
            uint64_t s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0}, l0 = 0, l1 = 0;
            if (fhe_hal_stats)
                fhe_hal_stats(s0);
            auto k0 = launches_by_kernel(&l0);
            t0 = now();
            for (int i = 0; i < reps; ++i) {
                auto t1 = now();
                b       = cc->EvalBootstrap(c, iters, prec);
                std::cout << "  rep " << i << " seconds " << secs(t1, now()) << std::endl;
            }
            std::cout << "bootstrap seconds " << secs(t0, now()) / reps << " (" << reps << " reps)" << std::endl;
            if (fhe_hal_stats) {
                fhe_hal_stats(s1);
                std::cout << "per bootstrap: deviceOps " << (s1[0] - s0[0]) / reps << " hostOps " << (s1[1] - s0[1]) / reps << " h2dMB "
                          << (s1[2] - s0[2]) / reps / 1e6 << " d2hMB " << (s1[3] - s0[3]) / reps / 1e6 << std::endl;
            }
            auto k1 = launches_by_kernel(&l1);
            if (l1 > l0) {
                std::cout << "per bootstrap: launches " << (l1 - l0) / reps << std::endl;
                for (auto& kv : k1)
                    if (kv.second > k0[kv.first])
                        std::cout << "  launches " << kv.first << " " << (kv.second - k0[kv.first]) / reps << std::endl;
            }
            dump("bootstrapped", b);
            show("bootstrapped", cc, kp.secretKey, b, 8);
        }
    }
    else {  // CKKS bootstrapping (ckksrns-fhe.cpp:429-760): ModRaise, CoeffsToSlots, Chebyshev sine, SlotsToCoeffs
        CCParams<CryptoContextCKKSRNS> p;
        SecretKeyDist skd = SPARSE_TERNARY;
        p.SetSecretKeyDist(skd);
        p.SetSecurityLevel(HEStd_NotSet);
        p.SetRingDim(1u << logN);
        p.SetScalingTechnique(FLEXIBLEAUTO);
        p.SetScalingModSize(59);
        p.SetFirstModSize(60);
        p.SetNumLargeDigits(3);
        p.SetKeySwitchTechnique(HYBRID);
        std::vector<uint32_t> levelBudget = {2, 2};
        const uint32_t levelsAfter        = 2;
        const usint depth                 = levelsAfter + FHECKKSRNS::GetBootstrapDepth(levelBudget, skd);
        p.SetMultiplicativeDepth(depth);
        auto cc = GenCryptoContext(p);
        cc->Enable(PKE);
        cc->Enable(KEYSWITCH);
        cc->Enable(LEVELEDSHE);
        cc->Enable(ADVANCEDSHE);
        cc->Enable(FHE);
        const usint slots = 8;
        cc->EvalBootstrapSetup(levelBudget, {0, 0}, slots);
        auto kp = cc->KeyGen();
        cc->EvalMultKeyGen(kp.secretKey);
        cc->EvalBootstrapKeyGen(kp.secretKey, slots);
        std::vector<double> x = {0.25, 0.5, 0.75, 1.0, 2.0, 3.0, 4.0, 5.0};
        auto pt = cc->MakeCKKSPackedPlaintext(x, 1, depth - 1, nullptr, slots);
        auto c  = cc->Encrypt(kp.publicKey, pt);
        dump("in", c);
        evaluation_phase_begins();
        auto b = cc->EvalBootstrap(c);
        dump("bootstrapped", b);
        std::cout << "levels remaining " << depth - b->GetLevel() - (b->GetNoiseScaleDeg() - 1) << std::endl;
        show("bootstrapped", cc, kp.secretKey, b, slots);
    }
    g_out.close();
    if (fhe_hal_stats) {
        uint64_t st[4];
        fhe_hal_stats(st);
        print_member_stats("halmember");
        if (fhe_hal_composite_stats) {
            uint64_t cs[3];
            fhe_hal_composite_stats(cs);
            std::cout << "halcomposite calls " << cs[0] << " checksIdentical " << cs[1] << " checksDiffered " << cs[2] << std::endl;
            if (fhe_hal_memo_hits)
                std::cout << "halmemo hits " << fhe_hal_memo_hits() << std::endl;
        }
        std::cout << "hal: available " << (fhe_hal_available ? fhe_hal_available() : -1) << " deviceOps " << st[0] << " hostOps " << st[1]
                  << " h2dBytes " << st[2] << " d2hBytes " << st[3] << std::endl;
    }
    else
        std::cout << "hal: stock backend" << std::endl;
    return 0;
}
