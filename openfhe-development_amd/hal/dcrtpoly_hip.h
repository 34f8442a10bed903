// dcrtpoly_hip.h — header-only C++ host side over the C ABI (include/fhe_hip.h).
//
// Mirrors the part of the reference class surface that lies on the hot path, with the reference's names,
// argument meaning and error behaviour (errors are thrown as exceptions carrying the library's message, the way
// OPENFHE_THROW does in src/core/include/utils/exception.h):
//   lbcrypto::DCRTPolyImpl  (src/core/include/lattice/hal/default/dcrtpoly.h:59-398, dcrtpoly-impl.h)
//   ILDCRTParams             (src/core/include/lattice/hal/default/ildcrtparams.h:70-372)
//   KeySwitchHYBRID          (src/pke/lib/keyswitch/keyswitch-hybrid.cpp:308-435)
// A tower object owns ONE device allocation uint64_t[batch][limbs][N]; `batch` > 1 is the extension over the
// reference (a DCRTPoly is batch == 1): every method applies to all towers of the batch in one launch.
// This header is a convenience binding for stand-alone C++ programs that want the BATCHED composites (tests/hal_smoke.cpp,
// tests/hal_parity.cpp).  It is NOT what the drop-in backend uses: lbcrypto::DCRTPolyHipImpl (lattice/hal/hip/dcrtpoly-hip.h)
// binds the C ABI itself through hip-runtime.cpp.
#ifndef FHE_HAL_DCRTPOLY_HIP_H
#define FHE_HAL_DCRTPOLY_HIP_H
#include <cstdint>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/fhe_hip.h"

namespace fhehip {

enum Format { EVALUATION = 0, COEFFICIENT = 1 };  // lbcrypto::Format, utils/inttypes.h:65

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};
inline void check(fhe_status s) {
    if (s != FHE_OK)
        throw Error(std::string(fhe_last_error()));
}

// ILDCRTParams + device twiddle tables (shared, immutable)
class Params {
public:
    // ILDCRTParams(corder, moduli, rootsOfUnity)  ildcrtparams.h:130-145
    Params(uint32_t cyclotomicOrder, const std::vector<uint64_t>& moduli, const std::vector<uint64_t>& roots, int device = 0)
        : m_moduli(moduli), m_roots(roots) {
        if (moduli.size() != roots.size())
            throw Error("sizes of moduli and roots of unity do not match 1");
        uint32_t logN = 0;
        while ((2u << logN) < cyclotomicOrder)
            ++logN;
        check(fhe_ctx_create(logN, (uint32_t)moduli.size(), moduli.data(), roots.data(), device, &m_ctx));
    }
    // ILDCRTParams(corder, depth, bits)  ildcrtparams.h:100-117
    static std::shared_ptr<Params> Generate(uint32_t cyclotomicOrder, uint32_t depth, uint32_t bits, int device = 0) {
        std::vector<uint64_t> q(depth), psi(depth);
        check(fhe_param_dcrt_chain(cyclotomicOrder, depth, bits, q.data(), psi.data()));
        return std::make_shared<Params>(cyclotomicOrder, q, psi, device);
    }
    ~Params() { fhe_ctx_destroy(m_ctx); }
    Params(const Params&)            = delete;
    Params& operator=(const Params&) = delete;
    uint32_t GetRingDimension() const { return 1u << fhe_ctx_logn(m_ctx); }
    uint32_t GetCyclotomicOrder() const { return 2u << fhe_ctx_logn(m_ctx); }
    const std::vector<uint64_t>& GetModuli() const { return m_moduli; }
    const std::vector<uint64_t>& GetRoots() const { return m_roots; }
    fhe_ctx* ctx() const { return m_ctx; }

private:
    fhe_ctx* m_ctx = nullptr;
    std::vector<uint64_t> m_moduli, m_roots;
};

class DCRTPolyHip {
public:
    // towers over the context limbs limbIdx (empty = limbs [0, nLimbs))
    DCRTPolyHip(std::shared_ptr<Params> params, uint32_t nLimbs, Format format, uint32_t batch = 1,
                std::vector<uint32_t> limbIdx = {})
        : m_params(std::move(params)), m_limbs(nLimbs), m_batch(batch), m_format(format), m_idx(std::move(limbIdx)) {
        if (!m_idx.empty() && m_idx.size() != nLimbs)
            throw Error("limb index list does not match the number of towers");
        void* p = nullptr;
        check(fhe_malloc(m_params->ctx(), bytes(), &p));
        m_data = static_cast<uint64_t*>(p);
    }
    ~DCRTPolyHip() {
        if (m_data)
            fhe_free(m_params->ctx(), m_data);
    }
    DCRTPolyHip(DCRTPolyHip&& o) noexcept { *this = std::move(o); }
    DCRTPolyHip& operator=(DCRTPolyHip&& o) noexcept {
        if (this != &o) {
            if (m_data)
                fhe_free(m_params->ctx(), m_data);
            m_params = std::move(o.m_params);
            m_limbs = o.m_limbs, m_batch = o.m_batch, m_format = o.m_format, m_idx = std::move(o.m_idx);
            m_data   = o.m_data;
            o.m_data = nullptr;
        }
        return *this;
    }
    DCRTPolyHip(const DCRTPolyHip& o) : DCRTPolyHip(o.m_params, o.m_limbs, o.m_format, o.m_batch, o.m_idx) {
        check(fhe_memcpy_d2d(m_params->ctx(), m_data, o.m_data, bytes(), nullptr));
    }

    // host <-> device (SetValues / GetValues of the limbs, poly.h:166-195)
    void SetValues(const std::vector<uint64_t>& host, Format format) {
        if (host.size() != words())
            throw Error("SetValues: size mismatch");
        check(fhe_memcpy_h2d(m_params->ctx(), m_data, host.data(), bytes(), nullptr));
        check(fhe_stream_sync(m_params->ctx(), nullptr));
        m_format = format;
    }
    std::vector<uint64_t> GetValues() const {
        std::vector<uint64_t> h(words());
        check(fhe_memcpy_d2h(m_params->ctx(), h.data(), m_data, bytes(), nullptr));
        check(fhe_stream_sync(m_params->ctx(), nullptr));
        return h;
    }

    Format GetFormat() const { return m_format; }
    uint32_t GetNumOfElements() const { return m_limbs; }
    uint32_t GetRingDimension() const { return m_params->GetRingDimension(); }
    uint32_t GetBatch() const { return m_batch; }
    const std::shared_ptr<Params>& GetParams() const { return m_params; }
    uint64_t* data() { return m_data; }
    const uint64_t* data() const { return m_data; }

    // DCRTPolyImpl::SwitchFormat (dcrtpoly-impl.h:1932-1940), SetFormat (ilelement.h:447-450)
    void SwitchFormat(void* stream = nullptr) {
        if (m_format == COEFFICIENT)
            check(fhe_ntt_fwd(m_params->ctx(), m_data, idx(), m_limbs, m_batch, stream));
        else
            check(fhe_ntt_inv(m_params->ctx(), m_data, idx(), m_limbs, m_batch, stream));
        m_format = m_format == COEFFICIENT ? EVALUATION : COEFFICIENT;
    }
    void SetFormat(Format f, void* stream = nullptr) {
        if (f != m_format)
            SwitchFormat(stream);
    }

    // Plus / Minus / Times and the in-place operators (dcrtpoly.h:131-189, dcrtpoly-impl.h:362-408)
    DCRTPolyHip Plus(const DCRTPolyHip& r) const { return bin(fhe_add, r); }
    DCRTPolyHip Minus(const DCRTPolyHip& r) const { return bin(fhe_sub, r); }
    DCRTPolyHip Times(const DCRTPolyHip& r) const { return bin(fhe_mul, r); }
    DCRTPolyHip& operator+=(const DCRTPolyHip& r) { return binEq(fhe_add, r); }
    DCRTPolyHip& operator-=(const DCRTPolyHip& r) { return binEq(fhe_sub, r); }
    DCRTPolyHip& operator*=(const DCRTPolyHip& r) { return binEq(fhe_mul, r); }
    // Times(const std::vector<NativeInteger>&)  dcrtpoly-impl.h:582-601
    DCRTPolyHip Times(const std::vector<uint64_t>& perLimb) const {
        if (perLimb.size() != m_limbs)
            throw Error("tower size mismatch; cannot multiply");
        DCRTPolyHip out(m_params, m_limbs, m_format, m_batch, m_idx);
        check(fhe_mul_const(m_params->ctx(), out.m_data, m_data, perLimb.data(), idx(), m_limbs, m_batch, nullptr));
        return out;
    }
    DCRTPolyHip Negate() const {  // dcrtpoly-impl.h:347-354
        DCRTPolyHip out(m_params, m_limbs, m_format, m_batch, m_idx);
        check(fhe_neg(m_params->ctx(), out.m_data, m_data, idx(), m_limbs, m_batch, nullptr));
        return out;
    }
    // AutomorphismTransform(k)  dcrtpoly-impl.h:314-333 ("Automorphism index not odd" is thrown for even k)
    DCRTPolyHip AutomorphismTransform(uint32_t k) const {
        DCRTPolyHip out(m_params, m_limbs, m_format, m_batch, m_idx);
        check(fhe_automorph(m_params->ctx(), out.m_data, m_data, k, m_format == EVALUATION, idx(), m_limbs, m_batch, nullptr));
        return out;
    }
    // DropLastElementAndScale (CKKS rescale)  dcrtpoly-impl.h:693-712; tower must use context limbs [0, limbs)
    void DropLastElementAndScale() {
        if (!m_idx.empty())
            throw Error("DropLastElementAndScale: tower must use the leading context limbs");
        if (m_format != EVALUATION)
            throw Error("DropLastElementAndScale: EVALUATION format expected");
        DCRTPolyHip out(m_params, m_limbs - 1, m_format, m_batch);
        size_t wsb = fhe_rescale_workspace_bytes(m_params->ctx(), m_limbs, m_batch);
        void* ws   = nullptr;
        check(fhe_malloc(m_params->ctx(), wsb, &ws));
        fhe_status s = fhe_rescale(m_params->ctx(), m_data, m_limbs, m_batch, out.m_data, ws, wsb, nullptr);
        fhe_stream_sync(m_params->ctx(), nullptr);
        fhe_free(m_params->ctx(), ws);
        check(s);
        *this = std::move(out);
    }
    // ModReduce (BGV modulus switch by the last limb, plaintext modulus t)  dcrtpoly-impl.h:736-755
    void ModReduce(uint64_t t) {
        if (!m_idx.empty())
            throw Error("ModReduce: tower must use the leading context limbs");
        DCRTPolyHip out(m_params, m_limbs - 1, m_format, m_batch);
        size_t wsb = fhe_rescale_workspace_bytes(m_params->ctx(), m_limbs, m_batch);
        void* ws   = nullptr;
        check(fhe_malloc(m_params->ctx(), wsb, &ws));
        fhe_status s = fhe_mod_reduce(m_params->ctx(), m_data, m_limbs, t, m_format == EVALUATION, m_batch, out.m_data, ws, wsb, nullptr);
        fhe_stream_sync(m_params->ctx(), nullptr);
        fhe_free(m_params->ctx(), ws);
        check(s);
        *this = std::move(out);
    }
    // ExpandCRTBasis / ExpandCRTBasisReverseOrder to this basis + the context limbs `extra` (dcrtpoly-impl.h:1088-1148)
    DCRTPolyHip ExpandCRTBasis(const std::vector<uint32_t>& extra, Format resultFormat, bool reverseOrder = false) const {
        std::vector<uint32_t> src = m_idx;
        if (src.empty())
            for (uint32_t i = 0; i < m_limbs; ++i)
                src.push_back(i);
        fhe_conv* cv = nullptr;
        check(fhe_conv_create(m_params->ctx(), src.data(), m_limbs, extra.data(), (uint32_t)extra.size(), &cv));
        std::vector<uint32_t> all = reverseOrder ? extra : src;
        all.insert(all.end(), reverseOrder ? src.begin() : extra.begin(), reverseOrder ? src.end() : extra.end());
        DCRTPolyHip out(m_params, (uint32_t)all.size(), resultFormat, m_batch, all);
        size_t wsb = fhe_expand_crt_basis_workspace_bytes(cv, m_batch);
        void* ws   = nullptr;
        check(fhe_malloc(m_params->ctx(), wsb, &ws));
        fhe_status s = fhe_expand_crt_basis(cv, m_data, m_format == EVALUATION, out.m_data, resultFormat == EVALUATION,
                                            reverseOrder, m_batch, ws, wsb, nullptr);
        fhe_stream_sync(m_params->ctx(), nullptr);
        fhe_free(m_params->ctx(), ws);
        fhe_conv_destroy(cv);
        check(s);
        return out;
    }
    // ApproxSwitchCRTBasis / SwitchCRTBasis to the context limbs `target` (dcrtpoly-impl.h:888-932, 1008-1085)
    DCRTPolyHip SwitchCRTBasis(const std::vector<uint32_t>& target, bool exact) const {
        if (m_format != COEFFICIENT)
            throw Error("SwitchCRTBasis: COEFFICIENT format expected");
        std::vector<uint32_t> src = m_idx;
        if (src.empty())
            for (uint32_t i = 0; i < m_limbs; ++i)
                src.push_back(i);
        fhe_conv* cv = nullptr;
        check(fhe_conv_create(m_params->ctx(), src.data(), m_limbs, target.data(), (uint32_t)target.size(), &cv));
        DCRTPolyHip out(m_params, (uint32_t)target.size(), COEFFICIENT, m_batch, target);
        fhe_status s = exact ? fhe_switch_basis_exact(cv, m_data, m_limbs, 0, out.m_data, (uint32_t)target.size(), 0, m_batch, nullptr)
                             : fhe_approx_switch_basis(cv, m_data, m_limbs, 0, out.m_data, (uint32_t)target.size(), 0, m_batch, nullptr);
        fhe_stream_sync(m_params->ctx(), nullptr);
        fhe_conv_destroy(cv);
        check(s);
        return out;
    }

private:
    typedef fhe_status (*BinFn)(fhe_ctx*, uint64_t*, const uint64_t*, const uint64_t*, const uint32_t*, uint32_t, uint32_t, void*);
    void compatible(const DCRTPolyHip& r) const {
        if (r.m_limbs != m_limbs || r.m_batch != m_batch)
            throw Error("tower size mismatch");  // dcrtpoly.h:154-162
        if (r.m_format != m_format)
            throw Error("format mismatch");
    }
    DCRTPolyHip bin(BinFn f, const DCRTPolyHip& r) const {
        compatible(r);
        DCRTPolyHip out(m_params, m_limbs, m_format, m_batch, m_idx);
        check(f(m_params->ctx(), out.m_data, m_data, r.m_data, idx(), m_limbs, m_batch, nullptr));
        return out;
    }
    DCRTPolyHip& binEq(BinFn f, const DCRTPolyHip& r) {
        compatible(r);
        check(f(m_params->ctx(), m_data, m_data, r.m_data, idx(), m_limbs, m_batch, nullptr));
        return *this;
    }
    const uint32_t* idx() const { return m_idx.empty() ? nullptr : m_idx.data(); }
    size_t words() const { return (size_t)m_batch * m_limbs * m_params->GetRingDimension(); }
    size_t bytes() const { return words() * sizeof(uint64_t); }

    std::shared_ptr<Params> m_params;
    uint32_t m_limbs = 0, m_batch = 0;
    Format m_format  = EVALUATION;
    std::vector<uint32_t> m_idx;
    uint64_t* m_data = nullptr;
};

// KeySwitchHYBRID over a context holding Q then P limbs (keyswitch-hybrid.cpp:308-435)
class KeySwitchHybrid {
public:
    KeySwitchHybrid(std::shared_ptr<Params> params, uint32_t sizeQ, uint32_t sizeP, uint32_t numPartQ)
        : m_params(std::move(params)), m_sizeQ(sizeQ), m_sizeP(sizeP) {
        check(fhe_ks_plan_create(m_params->ctx(), sizeQ, sizeP, numPartQ, &m_plan));
    }
    ~KeySwitchHybrid() {
        for (auto& kv : m_rot)
            fhe_ks_key_destroy(kv.second.second);
        for (void* d : m_diag)
            fhe_free(m_params->ctx(), d);
        fhe_ks_key_destroy(m_key);
        if (m_ws)
            fhe_free(m_params->ctx(), m_ws);
        fhe_ks_plan_destroy(m_plan);
    }
    // EvalKeyRelin b/a vectors, host uint64[numPartQ][sizeQ+sizeP][N] (evalkeyrelin.h:141,171)
    void SetEvalKey(const std::vector<uint64_t>& keyB, const std::vector<uint64_t>& keyA) {
        fhe_ks_key_destroy(m_key);
        m_key = nullptr;
        check(fhe_ks_key_upload(m_plan, keyB.data(), keyA.data(), &m_key));
    }
    // KeySwitchCore(a, evalKey)
    std::pair<DCRTPolyHip, DCRTPolyHip> KeySwitchCore(const DCRTPolyHip& a) {
        DCRTPolyHip o0(m_params, a.GetNumOfElements(), EVALUATION, a.GetBatch()), o1(m_params, a.GetNumOfElements(), EVALUATION, a.GetBatch());
        reserve(a.GetNumOfElements(), a.GetBatch());
        check(fhe_keyswitch_hybrid(m_plan, m_key, a.data(), a.GetNumOfElements(), a.GetBatch(), o0.data(), o1.data(), m_ws, m_wsBytes, nullptr));
        return {std::move(o0), std::move(o1)};
    }
    // LeveledSHEBase::EvalMult(ct1, ct2, evalKey) for 2-element ciphertexts (base-leveledshe.cpp:201-214)
    std::pair<DCRTPolyHip, DCRTPolyHip> EvalMult(const DCRTPolyHip& a0, const DCRTPolyHip& a1, const DCRTPolyHip& b0, const DCRTPolyHip& b1) {
        DCRTPolyHip c0(m_params, a0.GetNumOfElements(), EVALUATION, a0.GetBatch()), c1(m_params, a0.GetNumOfElements(), EVALUATION, a0.GetBatch());
        reserve(a0.GetNumOfElements(), a0.GetBatch());
        check(fhe_ckks_eval_mult(m_plan, m_key, a0.data(), a1.data(), b0.data(), b1.data(), a0.GetNumOfElements(), a0.GetBatch(),
                                 c0.data(), c1.data(), m_ws, m_wsBytes, nullptr));
        return {std::move(c0), std::move(c1)};
    }

    // DCRTPolyImpl::ApproxModDown with the context's P (dcrtpoly-impl.h:966-1005; t > 0: the BGV form): x over Q_l u P
    DCRTPolyHip ApproxModDown(const DCRTPolyHip& x, uint32_t sizeQl, uint64_t t = 0) {
        DCRTPolyHip out(m_params, sizeQl, EVALUATION, x.GetBatch());
        reserve(sizeQl, x.GetBatch());
        check(t ? fhe_approx_mod_down_bgv(m_plan, x.data(), sizeQl, t, x.GetBatch(), out.data(), m_ws, m_wsBytes, nullptr)
                : fhe_approx_mod_down(m_plan, x.data(), sizeQl, x.GetBatch(), out.data(), m_ws, m_wsBytes, nullptr));
        return out;
    }
    // KeySwitchHYBRID::KeySwitchExt for one element (keyswitch-hybrid.cpp:217-243): c * [P] on the Q_l limbs, zeros on P
    DCRTPolyHip KeySwitchExt(const DCRTPolyHip& c) {
        DCRTPolyHip out(m_params, c.GetNumOfElements() + m_sizeP, EVALUATION, c.GetBatch(), extLimbs(c.GetNumOfElements()));
        check(fhe_ks_ext(m_plan, c.data(), c.GetNumOfElements(), c.GetBatch(), out.data(), nullptr));
        return out;
    }
    // KeySwitchHYBRID::KeySwitchDown (keyswitch-hybrid.cpp:245-278)
    std::pair<DCRTPolyHip, DCRTPolyHip> KeySwitchDown(const DCRTPolyHip& x0, const DCRTPolyHip& x1) {
        const uint32_t sizeQl = x0.GetNumOfElements() - m_sizeP;
        DCRTPolyHip o0(m_params, sizeQl, EVALUATION, x0.GetBatch()), o1(m_params, sizeQl, EVALUATION, x0.GetBatch());
        reserve(sizeQl, x0.GetBatch());
        check(fhe_ks_down(m_plan, x0.data(), x1.data(), sizeQl, x0.GetBatch(), o0.data(), o1.data(), m_ws, m_wsBytes, nullptr));
        return {std::move(o0), std::move(o1)};
    }
    // LeveledSHEBase::EvalAutomorphism / EvalRotate (base-leveledshe.cpp:381-422) with the key set by SetRotationKey
    std::pair<DCRTPolyHip, DCRTPolyHip> EvalRotate(const DCRTPolyHip& c0, const DCRTPolyHip& c1, int32_t index) {
        auto it = m_rot.find(index);
        if (it == m_rot.end())
            throw Error("EvalRotate: no rotation key for index " + std::to_string(index));
        const uint32_t sizeQl = c0.GetNumOfElements(), batch = c0.GetBatch();
        DCRTPolyHip o0(m_params, sizeQl, EVALUATION, batch), o1(m_params, sizeQl, EVALUATION, batch);
        reserve(sizeQl, batch);
        check(fhe_eval_automorphism(m_plan, it->second.second, c0.data(), c1.data(), it->second.first, sizeQl, batch, o0.data(),
                                    o1.data(), m_ws, m_wsBytes, nullptr));
        return {std::move(o0), std::move(o1)};
    }
    // hoisted rotations: EvalFastRotationPrecompute once (base-leveledshe.cpp:425-430), then EvalFastRotation per index
    // (:432-463); the digits live in this object's workspace until the next call that uses it
    void EvalFastRotationPrecompute(const DCRTPolyHip& c1) {
        reserve(c1.GetNumOfElements(), c1.GetBatch());
        check(fhe_ks_precompute(m_plan, c1.data(), c1.GetNumOfElements(), c1.GetBatch(), m_ws, m_wsBytes, nullptr));
    }
    std::pair<DCRTPolyHip, DCRTPolyHip> EvalFastRotation(const DCRTPolyHip& c0, const DCRTPolyHip& c1, int32_t index) {
        auto it = m_rot.find(index);
        if (it == m_rot.end())
            throw Error("EvalFastRotation: no rotation key for index " + std::to_string(index));
        const uint32_t sizeQl = c0.GetNumOfElements(), batch = c0.GetBatch();
        DCRTPolyHip o0(m_params, sizeQl, EVALUATION, batch), o1(m_params, sizeQl, EVALUATION, batch);
        check(fhe_eval_fast_rotation(m_plan, it->second.second, c0.data(), c1.data(), it->second.first, sizeQl, batch, o0.data(),
                                     o1.data(), m_ws, m_wsBytes, nullptr));
        return {std::move(o0), std::move(o1)};
    }
    uint32_t AutomorphismIndex(int32_t index) const { return m_rot.at(index).first; }
    fhe_ks_plan* plan() const { return m_plan; }
    const fhe_ks_key* key() const { return m_key; }

    // ---- BSGS linear transform with double hoisting (FHECKKSRNS::EvalLinearTransform, ckksrns-fhe.cpp:1832-1882) ----
    // rotation key of `index` slots (EvalRotateKeyGen's key for FindAutomorphismIndex2nComplex(index, 2N))
    void SetRotationKey(int32_t index, const std::vector<uint64_t>& keyB, const std::vector<uint64_t>& keyA) {
        const uint32_t k = fhe_param_find_automorphism_index_2n_complex(index, 2u * m_params->GetRingDimension());
        if (k == 0)
            throw Error("m should be a power of two.");
        fhe_ks_key* h = nullptr;
        check(fhe_ks_key_upload(m_plan, keyB.data(), keyA.data(), &h));
        auto it = m_rot.find(index);
        if (it != m_rot.end())
            fhe_ks_key_destroy(it->second.second);
        m_rot[index] = {k, h};
    }
    // an encoded diagonal (EvalLinearTransformPrecompute's aux plaintext): host rows [sizeQl+sizeP][N], EVALUATION
    // format over the limbs {0..sizeQl-1, sizeQ..sizeQ+sizeP-1}; stays on the device until the object dies
    const uint64_t* UploadDiagonal(const std::vector<uint64_t>& rows) {
        void* d = nullptr;
        check(fhe_malloc(m_params->ctx(), rows.size() * sizeof(uint64_t), &d));
        m_diag.push_back(d);
        check(fhe_memcpy_h2d(m_params->ctx(), d, rows.data(), rows.size() * sizeof(uint64_t), nullptr));
        check(fhe_stream_sync(m_params->ctx(), nullptr));  // the copy reads the caller's vector: it may die right after the call
        return static_cast<const uint64_t*>(d);
    }
    // A[i] = diagonal i (nullptr = absent); baby step bStep, giant steps ceil(|A| / bStep)
    std::pair<DCRTPolyHip, DCRTPolyHip> EvalLinearTransform(const std::vector<const uint64_t*>& A, uint32_t bStep,
                                                             const DCRTPolyHip& c0, const DCRTPolyHip& c1) {
        const uint32_t slots = (uint32_t)A.size(), gStep = (slots + bStep - 1) / bStep;
        std::vector<uint32_t> inK(bStep, 0), outK(gStep, 0);
        std::vector<const fhe_ks_key*> inKeys(bStep, nullptr), outKeys(gStep, nullptr);
        auto key = [&](int32_t index, uint32_t& k, const fhe_ks_key*& h) {
            auto it = m_rot.find(index);
            if (it == m_rot.end())
                throw Error("EvalLinearTransform: no rotation key for index " + std::to_string(index));
            k = it->second.first, h = it->second.second;
        };
        for (uint32_t j = 1; j < bStep; ++j)
            key((int32_t)j, inK[j], inKeys[j]);
        for (uint32_t i = 1; i < gStep; ++i)
            key((int32_t)(bStep * i), outK[i], outKeys[i]);
        std::vector<const uint64_t*> diag((size_t)gStep * bStep, nullptr);
        for (uint32_t i = 0; i < slots; ++i)
            diag[i] = A[i];
        const uint32_t sizeQl = c0.GetNumOfElements(), batch = c0.GetBatch();
        DCRTPolyHip o0(m_params, sizeQl, EVALUATION, batch), o1(m_params, sizeQl, EVALUATION, batch);
        const size_t need = fhe_ckks_bsgs_workspace_bytes(m_plan, sizeQl, batch, bStep, gStep);
        void* ws          = nullptr;
        check(fhe_malloc(m_params->ctx(), need, &ws));
        const fhe_status st = fhe_ckks_bsgs_transform(m_plan, c0.data(), c1.data(), sizeQl, batch, bStep, inK.data(), inKeys.data(),
                                                      gStep, outK.data(), outKeys.data(), diag.data(), o0.data(), o1.data(), ws,
                                                      need, nullptr);
        fhe_stream_sync(m_params->ctx(), nullptr);
        fhe_free(m_params->ctx(), ws);
        check(st);
        return {std::move(o0), std::move(o1)};
    }

private:
    void reserve(uint32_t sizeQl, uint32_t batch) {
        size_t need = fhe_ks_workspace_bytes(m_plan, sizeQl, batch);
        if (need > m_wsBytes) {
            if (m_ws)
                fhe_free(m_params->ctx(), m_ws);
            check(fhe_malloc(m_params->ctx(), need, &m_ws));
            m_wsBytes = need;
        }
    }
    std::vector<uint32_t> extLimbs(uint32_t sizeQl) const {  // context limbs of Q_l u P
        std::vector<uint32_t> v;
        for (uint32_t i = 0; i < sizeQl; ++i)
            v.push_back(i);
        for (uint32_t j = 0; j < m_sizeP; ++j)
            v.push_back(m_sizeQ + j);
        return v;
    }
    std::shared_ptr<Params> m_params;
    uint32_t m_sizeQ, m_sizeP;
    fhe_ks_plan* m_plan = nullptr;
    fhe_ks_key* m_key   = nullptr;
    std::map<int32_t, std::pair<uint32_t, fhe_ks_key*>> m_rot;  // rotation index -> (automorphism index, key)
    std::vector<void*> m_diag;
    void* m_ws          = nullptr;
    size_t m_wsBytes    = 0;
};

// BFV multiplication in the BEHZ RNS variant (CryptoParametersBFVRNS's BEHZ tables, bfvrns-cryptoparameters.cpp:673-850;
// DCRTPolyImpl::FastBaseConvqToBskMontgomery / FastRNSFloorq / FastBaseConvSK, dcrtpoly-impl.h:1694-1929;
// LeveledSHEBFVRNS::EvalMult, bfvrns-leveledshe.cpp:198-445).  The context holds Q (qLimbs) and Bsk (bskLimbs, m_sk last).
class BfvBehz {
public:
    // the Bsk moduli / roots the reference picks for (N, Q, t) (bfvrns-cryptoparameters.cpp:682-711)
    static void SelectBsk(uint32_t cyclotomicOrder, const std::vector<uint64_t>& q, uint64_t t, std::vector<uint64_t>& bsk,
                          std::vector<uint64_t>& psi) {
        uint32_t logN = 0;
        while ((2u << logN) < cyclotomicOrder)
            ++logN;
        bsk.assign(q.size() + 1, 0), psi.assign(q.size() + 1, 0);
        if (fhe_param_behz_bsk(logN, (uint32_t)q.size(), q.data(), t, bsk.data(), psi.data()) != q.size() + 1)
            throw Error("BEHZ: no auxiliary basis for these parameters");
    }
    BfvBehz(std::shared_ptr<Params> params, const std::vector<uint32_t>& qLimbs, const std::vector<uint32_t>& bskLimbs, uint64_t t)
        : m_params(std::move(params)), m_numQ((uint32_t)qLimbs.size()) {
        if (bskLimbs.size() != qLimbs.size() + 1)
            throw Error("BEHZ: Bsk must hold one limb more than Q");
        check(fhe_behz_create(m_params->ctx(), qLimbs.data(), m_numQ, bskLimbs.data(), t, &m_plan));
    }
    ~BfvBehz() { fhe_behz_destroy(m_plan); }
    BfvBehz(const BfvBehz&)            = delete;
    BfvBehz& operator=(const BfvBehz&) = delete;
    // EvalMultNoRelin: (a0, a1) x (b0, b1) -> (d0, d1, d2), inputs EVALUATION over Q, outputs COEFFICIENT like the reference's
    std::vector<DCRTPolyHip> EvalMultNoRelin(const DCRTPolyHip& a0, const DCRTPolyHip& a1, const DCRTPolyHip& b0, const DCRTPolyHip& b1) {
        const uint32_t batch = a0.GetBatch();
        std::vector<DCRTPolyHip> d;
        for (int i = 0; i < 3; ++i)
            d.emplace_back(m_params, m_numQ, COEFFICIENT, batch);
        const size_t need = fhe_bfv_eval_mult_behz_workspace_bytes(m_plan, batch);
        void* ws          = nullptr;
        check(fhe_malloc(m_params->ctx(), need, &ws));
        const fhe_status st = fhe_bfv_eval_mult_behz(m_plan, a0.data(), a1.data(), b0.data(), b1.data(), d[0].data(), d[1].data(),
                                                     d[2].data(), 0, batch, ws, need, nullptr);
        fhe_stream_sync(m_params->ctx(), nullptr);
        fhe_free(m_params->ctx(), ws);
        check(st);
        return d;
    }
    // cc->EvalMult: EvalMultNoRelin + HYBRID relinearisation with `ks`'s evaluation key (base-leveledshe.cpp:201-214)
    std::pair<DCRTPolyHip, DCRTPolyHip> EvalMult(KeySwitchHybrid& ks, const DCRTPolyHip& a0, const DCRTPolyHip& a1,
                                                 const DCRTPolyHip& b0, const DCRTPolyHip& b1) {
        const uint32_t batch = a0.GetBatch();
        DCRTPolyHip c0(m_params, m_numQ, EVALUATION, batch), c1(m_params, m_numQ, EVALUATION, batch);
        const size_t need = fhe_bfv_eval_mult_relin_workspace_bytes(m_plan, ks.plan(), batch);
        void* ws          = nullptr;
        check(fhe_malloc(m_params->ctx(), need, &ws));
        const fhe_status st = fhe_bfv_eval_mult_relin_behz(m_plan, ks.plan(), ks.key(), a0.data(), a1.data(), b0.data(), b1.data(),
                                                           c0.data(), c1.data(), batch, ws, need, nullptr);
        fhe_stream_sync(m_params->ctx(), nullptr);
        fhe_free(m_params->ctx(), ws);
        check(st);
        return {std::move(c0), std::move(c1)};
    }

private:
    std::shared_ptr<Params> m_params;
    uint32_t m_numQ;
    fhe_behz* m_plan = nullptr;
};

}  // namespace fhehip
#endif
