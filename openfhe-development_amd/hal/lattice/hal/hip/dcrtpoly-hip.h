// dcrtpoly-hip.h — lbcrypto::DCRTPolyHipImpl: the DCRTPolyInterface implementation of the MI355X (HIP) backend.
//
// Contract: src/core/include/lattice/hal/dcrtpoly-interface.h:83-1604 (DCRTPolyInterface<Derived, BigVec, NativeVec, PolyImpl>),
// selected by lattice/hal/hip/lat-backend-hip.h in place of lattice/hal/lat-backend.h:39-61; the default implementation it
// stands in for is lattice/hal/default/dcrtpoly.h:59-398.  pke / binfhe sources are unchanged.
//
// Data model.  A tower lives EITHER in device memory as uint64_t[nLimbs][N] (one allocation, the layout of include/fhe_hip.h)
// OR in a host mirror, an object of the reference's own DCRTPolyImpl (a std::vector<PolyImpl<NativeVector>>), or in both:
//   * the hot members run as kernels of libfhe_hip.so on the device copy: SwitchFormat, + - * (tower x tower), Negate,
//     Times(vector<NativeInteger>) / TimesNoCheck / *= NativeInteger, AutomorphismTransform, ApproxSwitchCRTBasis,
//     ApproxModUp, ApproxModDown, SwitchCRTBasis, DropLastElementAndScale, DropLastElement(s), CloneTowers, and the
//     "ModRaise" constructor from one NativePoly; their results stay on the device;
//   * GetAllElements() / GetElementAtIndex() / operator[] hand out the host mirror, which is filled from the device on
//     demand (lazily, once); mutable access and every member not listed above run on the mirror with the reference's own
//     code (DCRTPolyImpl is a member, not a copy of its source) and drop the device copy;
//   * the (params, format) pair always lives in the mirror object, also while its limbs are empty.
// The table arguments of the CRT members are the reference's own (CryptoParametersRNS getters): conversion plans are
// built from them with fhe_conv_create_custom and cached by content.  Everything that the device library cannot take
// (ring outside 2^4..2^17, a modulus >= 2^60 or != 1 mod 2N, a missing root of unity, BGV's t > 0 in ApproxModDown ...)
// falls back to the mirror, so behaviour — including the exceptions thrown — is the default backend's.
// Without a usable device (hiprt::Available() == false) the class IS the default backend with one indirection.
//
// Threads.  Every host thread enqueues on a HIP stream of its own (hiprt::Op): towers handed from one thread to another — pke's
// OpenMP loops, a batch of ciphertexts spread over threads — are ordered with device-side waits, never by blocking the host.
// Counters.  Every member that touches words opens a hiprt::MemberScope: tests assert, member by member, that the evaluation
// paths executed on the device and what ran on the host mirror (fhe_hal_member_stats); FHE_HAL_REQUIRE_DEVICE=1 turns a host-mirror
// execution of a member that has a device path into an exception.
#ifndef LBCRYPTO_INC_LATTICE_HAL_HIP_DCRTPOLY_HIP_H
#define LBCRYPTO_INC_LATTICE_HAL_HIP_DCRTPOLY_HIP_H

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "lattice/hal/dcrtpoly-interface.h"
#include "lattice/hal/default/dcrtpoly.h"
#include "lattice/hal/hip/hip-runtime.h"

#define FHE_HAL_MEMBER() hiprt::MemberScope fhe_hal_scope_(__func__)

namespace lbcrypto {

template <typename VecType>
class DCRTPolyHipImpl final : public DCRTPolyInterface<DCRTPolyHipImpl<VecType>, VecType, NativeVector, PolyImpl> {
public:
    using Vector                = VecType;
    using Integer               = typename VecType::Integer;
    using Params                = ILDCRTParams<Integer>;
    using PolyType              = PolyImpl<NativeVector>;
    using PolyLargeType         = PolyImpl<VecType>;
    using DCRTPolyType          = DCRTPolyHipImpl<VecType>;
    using HostType              = DCRTPolyImpl<VecType>;  // the reference's default implementation = the host mirror
    using DCRTPolyInterfaceType = DCRTPolyInterface<DCRTPolyHipImpl<VecType>, VecType, NativeVector, PolyImpl>;
    using Precomputations       = typename DCRTPolyInterfaceType::CRTBasisExtensionPrecomputations;
    using DggType               = typename DCRTPolyInterfaceType::DggType;
    using DugType               = typename DCRTPolyInterfaceType::DugType;
    using TugType               = typename DCRTPolyInterfaceType::TugType;
    using BugType               = typename DCRTPolyInterfaceType::BugType;

    static_assert(sizeof(NativeInteger) == sizeof(uint64_t), "a NativeVector is a flat array of 64-bit residues");

    // ---------------------------------------------------------------------------------------------------------------
    // construction / assignment (dcrtpoly.h:75-132): values are produced on the host; they move to the device when the
    // first device member touches the object
    // ---------------------------------------------------------------------------------------------------------------
    DCRTPolyHipImpl() = default;
    DCRTPolyHipImpl(const DCRTPolyType& e) {
        CopyFrom(e);
    }
    DCRTPolyType& operator=(const DCRTPolyType& rhs) override {
        if (this != &rhs)
            CopyFrom(rhs);
        return *this;
    }
    DCRTPolyHipImpl(DCRTPolyType&& e) noexcept
        : m_h{std::move(e.m_h)}, m_d{std::move(e.m_d)}, m_lazy{std::move(e.m_lazy)}, m_hostValid{e.m_hostValid}, m_zero{e.m_zero}, m_k{e.m_k} {}
    DCRTPolyType& operator=(DCRTPolyType&& rhs) noexcept override {
        m_h         = std::move(rhs.m_h);
        m_d         = std::move(rhs.m_d);
        m_lazy      = std::move(rhs.m_lazy);
        m_hostValid = rhs.m_hostValid;
        m_zero      = rhs.m_zero;
        m_k         = rhs.m_k;
        return *this;
    }
    explicit DCRTPolyHipImpl(HostType&& h) noexcept : m_h{std::move(h)} {}
    explicit DCRTPolyHipImpl(const HostType& h) : m_h{h} {}

    DCRTPolyHipImpl(const PolyLargeType& e, const std::shared_ptr<Params>& params) : m_h{e, params} {}
    DCRTPolyType& operator=(const PolyLargeType& rhs) {
        FHE_HAL_MEMBER();
        Hm(__func__, true) = rhs;
        return *this;
    }
    // the "ModRaise" constructor (dcrtpoly-impl.h:87-93): one polynomial modulo q_0 lifted, centred, into every limb
    DCRTPolyHipImpl(const PolyType& e, const std::shared_ptr<Params>& params) {
        hiprt::MemberScope scope("ModRaise");
        if (!ModRaiseOnDevice(e, params)) {
            m_h = HostType(e, params);
            hiprt::CountHost("ModRaise", params->GetRingDimension());
        }
    }
    // dcrtpoly-impl.h:96-107: every limb is rhs, brought to the limb's modulus (SwitchModulus) — the ModRaise constructor's loop.  One
    // limb crosses PCIe instead of the whole tower (MultByMonomialInPlace builds its monomial this way in every bootstrap).
    DCRTPolyType& operator=(const PolyType& rhs) {
        FHE_HAL_MEMBER();
        const auto P = m_h.GetParams();
        static const bool onHost = std::getenv("FHE_HAL_ASSIGN_ON_HOST") != nullptr;  // (A/B switch of session l)
        if (!onHost && P && m_h.GetFormat() == Format::COEFFICIENT && rhs.GetFormat() == Format::COEFFICIENT && ModRaiseOnDevice(rhs, P))
            return *this;
        Hm(__func__, true) = rhs;
        return *this;
    }
    explicit DCRTPolyHipImpl(const std::vector<PolyType>& elements) : m_h{elements} {}
    // a zero tower stays unmaterialised (m_zero) until its first use decides where it lives: accumulators of the evaluation
    // path (`DCRTPoly first(params, EVALUATION, true); first += ...`) never cross PCIe
    DCRTPolyHipImpl(const std::shared_ptr<Params>& params, Format format = Format::EVALUATION,
                    bool initializeElementToZero = false)
        : m_h{params, format, initializeElementToZero && !hiprt::Available()} {
        if (initializeElementToZero && hiprt::Available()) {
            m_hostValid = false;
            m_zero      = true;
            m_k         = hiprt::ThreadWidth();  // (an accumulator of a wide evaluation is wide)
        }
    }
    // the sampling constructors (dcrtpoly-impl.h:126-205).  Default: the reference's host samplers on the reference's PRNG stream (same seed,
    // same words as the default backend), the tower uploaded at its first device use.  FHE_HAL_DEVICE_SAMPLER=1: device kernels on a
    // counter-based generator (SampleOnDevice below) — a bootstrapping key set is 5-6 GB of uniform words that then never cross PCIe.
    DCRTPolyHipImpl(const DggType& dgg, const std::shared_ptr<Params>& p, Format f = Format::EVALUATION) {
        if (!SampleOnDevice(1, p, f, dgg.GetStd()))
            m_h = HostType(dgg, p, f);
    }
    DCRTPolyHipImpl(const BugType& bug, const std::shared_ptr<Params>& p, Format f = Format::EVALUATION) : m_h{bug, p, f} {}
    DCRTPolyHipImpl(const TugType& tug, const std::shared_ptr<Params>& p, Format f = Format::EVALUATION, uint32_t h = 0) {
        if (h != 0 || !SampleOnDevice(2, p, f, 0.0))  // (a fixed Hamming weight is the host generator's)
            m_h = HostType(tug, p, f, h);
    }
    DCRTPolyHipImpl(DugType& dug, const std::shared_ptr<Params>& p, Format f = Format::EVALUATION) {
        if (!SampleOnDevice(0, p, f, 0.0))
            m_h = HostType(dug, p, f);
    }

    DCRTPolyType& operator=(std::initializer_list<uint64_t> rhs) override {
        FHE_HAL_MEMBER();
        Hm(__func__, true) = rhs;
        return *this;
    }
    DCRTPolyType& operator=(uint64_t val) {
        FHE_HAL_MEMBER();
        Hm(__func__, true) = val;
        return *this;
    }
    DCRTPolyType& operator=(const std::vector<int64_t>& rhs) {
        FHE_HAL_MEMBER();
        Hm(__func__, true) = rhs;
        return *this;
    }
    DCRTPolyType& operator=(const std::vector<int32_t>& rhs) {
        FHE_HAL_MEMBER();
        Hm(__func__, true) = rhs;
        return *this;
    }
    DCRTPolyType& operator=(std::initializer_list<std::string> rhs) {
        FHE_HAL_MEMBER();
        Hm(__func__, true) = rhs;
        return *this;
    }

    DCRTPolyType CloneWithNoise(const DiscreteGaussianGeneratorImpl<VecType>& dgg, Format format) const override {
        FHE_HAL_MEMBER();
        return Wrap(Hc().CloneWithNoise(dgg, format));
    }
    // dcrtpoly-impl.h:207-214
    DCRTPolyType CloneTowers(uint32_t startTower, uint32_t endTower) const {
        FHE_HAL_MEMBER();
        Settle();
        // (words that exist on the host only — a secret key's towers as the sampler produced them — are cloned where they are: an upload
        // here was read back limb by limb by key generation, 0.85 GB of PCIe over the reference's unit tests)
        // (a tower with a device copy is cloned on the device whether or not the mirror is valid too: the secret key during key generation
        // has both, and its clone's next member is a transform)
        if (m_d && endTower < NumLimbs() && startTower <= endTower) {
            const auto& P = m_h.GetParams();
            auto params   = std::make_shared<Params>(P->GetCyclotomicOrder(), P->GetParamPartition(startTower, endTower));
            const size_t N = P->GetRingDimension(), n = endTower - startTower + 1, L = NumLimbs();
            hiprt::Op op;
            auto d        = hiprt::Alloc((size_t)m_k * n * N);
            uint64_t* dst = op.W(d);
            for (uint32_t k = 0; k < m_k; ++k)  // (each tower of a wide one)
                hiprt::D2D(op, dst + (size_t)k * n * N, op.R(m_d) + ((size_t)k * L + startTower) * N, n * N * 8, "CloneTowers");
            hiprt::CountDevice();
            return FromDevice(params, m_h.GetFormat(), std::move(d), m_k);
        }
        if (!m_d && m_hostValid) {
            hiprt::CountHost(__func__, RingOf(m_h), /*hostData=*/true);
            return DCRTPolyType(m_h.CloneTowers(startTower, endTower));
        }
        return Wrap(Hc().CloneTowers(startTower, endTower));
    }

    bool operator==(const DCRTPolyType& rhs) const override {
        FHE_HAL_MEMBER();
        return Hc() == rhs.Hc();
    }

    // ---------------------------------------------------------------------------------------------------------------
    // tower arithmetic (dcrtpoly.h:131-189, dcrtpoly-impl.h:347-408, 582-620)
    // ---------------------------------------------------------------------------------------------------------------
    DCRTPolyType& operator+=(const DCRTPolyType& rhs) override {
        FHE_HAL_MEMBER();
        if (AddLazily(rhs, false))
            return *this;
        if (!BinaryInPlace(rhs, hiprt::api().add, false))
            Hm() += rhs.Hc();
        return *this;
    }
    // dcrtpoly-impl.h:383-399: every limb += NativeInteger(rhs) as a constant polynomial (PolyImpl::Plus(Integer), poly-impl.h:211-218)
    DCRTPolyType& operator+=(const Integer& rhs) override {
        FHE_HAL_MEMBER();
        if (!AddScalarInPlace(NativeInteger{rhs}, false))
            Hm() += rhs;
        return *this;
    }
    DCRTPolyType& operator+=(const NativeInteger& rhs) override {
        FHE_HAL_MEMBER();
        if (!AddScalarInPlace(rhs, false))
            Hm() += rhs;
        return *this;
    }
    DCRTPolyType& operator-=(const DCRTPolyType& rhs) override {
        FHE_HAL_MEMBER();
        if (AddLazily(rhs, true))
            return *this;
        if (!BinaryInPlace(rhs, hiprt::api().sub, false))
            Hm() -= rhs.Hc();
        return *this;
    }
    // dcrtpoly-impl.h:411-427: every word of every limb -= NativeInteger(rhs) (poly.h:249-252: ModSubEq, both formats)
    DCRTPolyType& operator-=(const Integer& rhs) override {
        FHE_HAL_MEMBER();
        if (!AddScalarInPlace(NativeInteger{rhs}, true))
            Hm() -= rhs;
        return *this;
    }
    DCRTPolyType& operator-=(const NativeInteger& rhs) override {
        FHE_HAL_MEMBER();
        if (!AddScalarInPlace(rhs, true))
            Hm() -= rhs;
        return *this;
    }
    DCRTPolyType& operator*=(const DCRTPolyType& rhs) override {
        FHE_HAL_MEMBER();
        // (EVALUATION only on the device: for COEFFICIENT operands the reference's PolyImpl::operator*= throws, poly.h:254-263)
        if (!BinaryInPlace(rhs, hiprt::api().mul, true))
            Hm() *= rhs.Hc();
        return *this;
    }
    DCRTPolyType& operator*=(const Integer& rhs) override {  // dcrtpoly-impl.h:604-611: NativeInteger val{rhs}, every limb *= val
        FHE_HAL_MEMBER();
        std::vector<NativeInteger> c(NumLimbs(), NativeInteger{rhs});
        if (!TimesConstInPlace(c))
            Hm() *= rhs;
        return *this;
    }
    DCRTPolyType& operator*=(const NativeInteger& rhs) override {
        FHE_HAL_MEMBER();
        std::vector<NativeInteger> c(NumLimbs(), rhs);
        if (!TimesConstInPlace(c))
            Hm() *= rhs;
        return *this;
    }

    DCRTPolyType Negate() const override {
        FHE_HAL_MEMBER();
        hiprt::Resolved r;
        if (OnDeviceWide(&r)) {
            hiprt::Op op;
            auto d = hiprt::Alloc(Words());
            hiprt::Check(hiprt::api().neg(r.ctx, op.W(d), op.R(m_d), r.idx[0].data(), NumLimbs(), m_k, op.s), "Negate");
            hiprt::CountDevice();
            return FromDevice(m_h.GetParams(), m_h.GetFormat(), std::move(d), m_k);
        }
        return Wrap(Hc().Negate());
    }
    DCRTPolyType operator-() const override {
        FHE_HAL_MEMBER();
        return DCRTPolyType(m_h.GetParams(), m_h.GetFormat(), true) -= *this;
    }

    std::vector<DCRTPolyType> BaseDecompose(usint baseBits, bool evalModeAnswer) const override {
        FHE_HAL_MEMBER();
        return WrapAll(Hc().BaseDecompose(baseBits, evalModeAnswer));
    }
    std::vector<DCRTPolyType> PowersOfBase(usint baseBits) const override {
        FHE_HAL_MEMBER();
        return WrapAll(Hc().PowersOfBase(baseBits));
    }
    // dcrtpoly-impl.h:230-285 (the digit decomposition of KeySwitchBV): one launch per source limb cuts its digits and lifts each into every
    // limb of its tower, one transform takes all towers to EVALUATION (fhe_crt_decompose); the result towers are windows of one buffer
    std::vector<DCRTPolyType> CRTDecompose(uint32_t baseBits) const {
        FHE_HAL_MEMBER();
        std::vector<DCRTPolyType> out;
        if (CRTDecomposeOnDevice(baseBits, &out))
            return out;
        return WrapAll(Hc().CRTDecompose(baseBits));
    }
    bool CRTDecomposeOnDevice(uint32_t baseBits, std::vector<DCRTPolyType>* out) const {
        const auto& P = m_h.GetParams();
        if (!hiprt::Available() || m_k != 1 || !P || NumLimbs() == 0 || NumLimbs() != P->GetParams().size())
            return false;
        hiprt::Resolved r;
        if (!ResolveSets(P->GetRingDimension(), {P}, &r) || !Upload())
            return false;
        const uint32_t L      = NumLimbs();
        const size_t N        = P->GetRingDimension();
        const uint32_t towers = hiprt::api().crt_decompose_towers(r.ctx, r.idx[0].data(), L, baseBits);
        if (towers == 0)
            return false;  // (a digit size the library leaves to the host: windows beyond the 64-bit word)
        hiprt::Op op;
        hiprt::Buf coef = m_d;
        if (m_h.GetFormat() != Format::COEFFICIENT) {  // (:231-233: the coefficient copy)
            coef = hiprt::Alloc((size_t)L * N);
            hiprt::Check(hiprt::api().ntt_inv_oop(r.ctx, op.R(m_d), op.W(coef), r.idx[0].data(), L, 1, op.s), "CRTDecompose");
        }
        auto all = hiprt::Alloc((size_t)towers * L * N);
        hiprt::Check(hiprt::api().crt_decompose(r.ctx, op.R(coef), r.idx[0].data(), L, baseBits, op.W(all), op.s), "CRTDecompose");
        out->clear();
        out->reserve(towers);
        for (uint32_t t = 0; t < towers; ++t)  // (windows of the one allocation, as the towers of a wide evaluation: copy-on-write covers them)
            out->push_back(FromDevice(P, Format::EVALUATION, hiprt::View(all, (size_t)t * L * N, (size_t)L * N)));
        hiprt::CountDevice();
        return true;
    }

    // dcrtpoly-impl.h:314-333 -> poly-impl.h:310-376 (EVALUATION: gather through PrecomputeAutoMap, COEFFICIENT: signed permutation)
    DCRTPolyType AutomorphismTransform(uint32_t i) const override {
        FHE_HAL_MEMBER();
        hiprt::Resolved r;
        if ((i & 1u) && OnDeviceWide(&r)) {
            hiprt::Op op;
            auto d = hiprt::Alloc(Words());
            hiprt::Check(hiprt::api().automorph(r.ctx, op.W(d), op.R(m_d), i, m_h.GetFormat() == Format::EVALUATION ? 1 : 0,
                                                r.idx[0].data(), NumLimbs(), m_k, op.s),
                         "AutomorphismTransform");
            hiprt::CountDevice();
            return FromDevice(m_h.GetParams(), m_h.GetFormat(), std::move(d), m_k);
        }
        return Wrap(Hc().AutomorphismTransform(i));
    }
    DCRTPolyType AutomorphismTransform(uint32_t i, const std::vector<uint32_t>& vec) const override {
        FHE_HAL_MEMBER();
        // The device kernel computes the permutation of PrecomputeAutoMap(N, i) (nbtheory2.cpp:264-275) on the fly: `vec` is compared
        // with that table word for word (memoised per (N, i)); any other map is applied by the reference's own code.
        const uint32_t N = m_h.GetParams()->GetRingDimension();
        if (m_h.GetFormat() == Format::EVALUATION && hiprt::Available() && hiprt::IsAutoMap(N, i, vec))
            return AutomorphismTransform(i);
        return Wrap(Hc().AutomorphismTransform(i, vec));
    }

    DCRTPolyType Plus(const Integer& rhs) const override {  // dcrtpoly-impl.h:509-517
        FHE_HAL_MEMBER();
        DCRTPolyType out(*this);
        if (out.AddScalarInPlace(NativeInteger{rhs}, false))
            return out;
        return Wrap(Hc().Plus(rhs));
    }
    DCRTPolyType Plus(const std::vector<Integer>& rhs) const {  // dcrtpoly-impl.h:520-527
        FHE_HAL_MEMBER();
        DCRTPolyType out;
        if (AddConstOnDevice(rhs, false, &out))
            return out;
        return Wrap(Hc().Plus(rhs));
    }
    DCRTPolyType Plus(const DCRTPolyType& rhs) const override {
        FHE_HAL_MEMBER();
        DCRTPolyType out;
        if (Binary(rhs, hiprt::api().add, false, &out))
            return out;
        return Wrap(Hc().Plus(rhs.Hc()));
    }
    DCRTPolyType Minus(const DCRTPolyType& rhs) const override {
        FHE_HAL_MEMBER();
        DCRTPolyType out;
        if (Binary(rhs, hiprt::api().sub, false, &out))
            return out;
        return Wrap(Hc().Minus(rhs.Hc()));
    }
    DCRTPolyType Minus(const Integer& rhs) const override {  // dcrtpoly-impl.h:530-538
        FHE_HAL_MEMBER();
        DCRTPolyType out(*this);
        if (out.AddScalarInPlace(NativeInteger{rhs}, true))
            return out;
        return Wrap(Hc().Minus(rhs));
    }
    DCRTPolyType Minus(const std::vector<Integer>& rhs) const {  // dcrtpoly-impl.h:541-548
        FHE_HAL_MEMBER();
        DCRTPolyType out;
        if (AddConstOnDevice(rhs, true, &out))
            return out;
        return Wrap(Hc().Minus(rhs));
    }
    DCRTPolyType Times(const DCRTPolyType& rhs) const override {
        FHE_HAL_MEMBER();
        DCRTPolyType out;
        if (Binary(rhs, hiprt::api().mul, true, &out))
            return out;
        return Wrap(Hc().Times(rhs.Hc()));
    }
    DCRTPolyType Times(const Integer& rhs) const override {  // dcrtpoly-impl.h:551-559
        FHE_HAL_MEMBER();
        DCRTPolyType out(*this);
        std::vector<NativeInteger> c(NumLimbs(), NativeInteger{rhs});
        if (out.TimesConstInPlace(c))
            return out;
        return Wrap(Hc().Times(rhs));
    }
    DCRTPolyType Times(const std::vector<Integer>& rhs) const {  // dcrtpoly-impl.h:572-580: limb i times NativeInteger(rhs[i])
        FHE_HAL_MEMBER();
        if (rhs.size() >= NumLimbs()) {
            std::vector<NativeInteger> c(NumLimbs());
            for (uint32_t i = 0; i < NumLimbs(); ++i)
                c[i] = NativeInteger(rhs[i]);
            if (m_k > 1) {  // a wide tower: the product is recorded, the sum it usually joins is computed in one launch (m_lazy)
                DCRTPolyType lazy(*this);
                if (lazy.ScaleLazily(c))
                    return lazy;
            }
            // (a clone multiplied by the constants its original was multiplied by before — the level adjustments of pke's weighted sums
            // — takes that result: DevBuf::memo)
            const hiprt::Buf src = m_d;
            std::vector<uint64_t> memoKey;
            hiprt::Resolved r;
            if (src && m_k == 1 && !m_hostValid && !src->parent && src->Owners(src, 2) >= 1 && OnDevice(&r)) {
                memoKey.reserve(2 + 2 * (size_t)NumLimbs());
                memoKey.push_back(2), memoKey.push_back(NumLimbs());
                memoKey.insert(memoKey.end(), r.idx[0].begin(), r.idx[0].end());
                for (const auto& v : c)
                    memoKey.push_back(v.ConvertToInt<uint64_t>());
                if (auto hit = hiprt::MemoFind(src, memoKey))
                    return FromDevice(m_h.GetParams(), m_h.GetFormat(), std::move(hit));
            }
            DCRTPolyType out(*this);
            if (out.TimesConstInPlace(c)) {
                if (!memoKey.empty() && out.m_d != src)
                    hiprt::MemoStore(src, std::move(memoKey), out.m_d);
                return out;
            }
        }
        return Wrap(Hc().Times(rhs));
    }
    // dcrtpoly-impl.h:562-570 -> PolyImpl::Times(SignedNativeInt) (poly-impl.h:235-251): limb i times |rhs| mod q_i, or q_i minus that
    DCRTPolyType Times(NativeInteger::SignedNativeInt rhs) const override {
        FHE_HAL_MEMBER();
        const auto& P = m_h.GetParams();
        if (P && NumLimbs() == P->GetParams().size()) {
            const uint64_t mag = rhs < 0 ? (uint64_t)0 - (uint64_t)rhs : (uint64_t)rhs;
            std::vector<NativeInteger> c(NumLimbs());
            for (uint32_t i = 0; i < NumLimbs(); ++i) {
                const uint64_t q = P->GetParams()[i]->GetModulus().template ConvertToInt<uint64_t>();
                const uint64_t m = mag % q;
                c[i]             = NativeInteger(rhs < 0 ? (m ? q - m : 0) : m);
            }
            DCRTPolyType out(*this);
            if (out.TimesConstInPlace(c))
                return out;
        }
        return Wrap(Hc().Times(rhs));
    }
#if NATIVEINT != 64
    DCRTPolyType Times(int64_t rhs) const {
        return Times(static_cast<NativeInteger::SignedNativeInt>(rhs));
    }
#endif
    // dcrtpoly-impl.h:582-601
    DCRTPolyType Times(const std::vector<NativeInteger>& rhs) const {
        FHE_HAL_MEMBER();
        if (rhs.size() == NumLimbs()) {
            DCRTPolyType out(*this);
            if (out.TimesConstInPlace(rhs))
                return out;
        }
        return Wrap(Hc().Times(rhs));
    }
    DCRTPolyType TimesNoCheck(const std::vector<NativeInteger>& rhs) const {
        FHE_HAL_MEMBER();
        // (fewer factors than limbs: the reference leaves the trailing limbs of the result UNFILLED, dcrtpoly-impl.h:594-601 — such a
        // tower has no device form; KeySwitchGenInternal meets it with old keys that carry more limbs than [P]_q has entries)
        if (rhs.size() >= NumLimbs()) {
            DCRTPolyType out(*this);
            if (out.TimesConstInPlace(rhs))
                return out;
        }
        return Wrap(Hc().TimesNoCheck(rhs));
    }

    DCRTPolyType MultiplicativeInverse() const override {
        FHE_HAL_MEMBER();
        return Wrap(Hc().MultiplicativeInverse());
    }
    bool InverseExists() const override {
        FHE_HAL_MEMBER();
        return Hc().InverseExists();
    }
    bool IsEmpty() const override {
        return (m_zero || m_lazy || (m_d && !m_hostValid)) ? false : m_h.IsEmpty();
    }

    void SetValuesToZero() override {
        FHE_HAL_MEMBER();
        if (hiprt::Available() && m_h.GetParams() && NumLimbs() == m_h.GetParams()->GetParams().size()) {
            std::lock_guard<std::mutex> lk(m_lock.m);
            auto P      = m_h.GetParams();
            m_h         = HostType(P, m_h.GetFormat(), false);
            m_d.reset();
            m_lazy.reset();
            m_hostValid = false;
            m_zero      = true;
            return;
        }
        Hm().SetValuesToZero();
    }
    void AddILElementOne() override {
        FHE_HAL_MEMBER();
        Hm().AddILElementOne();
    }
    // dcrtpoly-impl.h:669-689: the device copy keeps its leading limbs, the mirror object keeps the metadata in step
    void DropLastElement() override {
        CompactForDrop(1);
        DropLastMeta();
    }
    void DropLastElements(size_t i) override {
        CompactForDrop(i);
        std::lock_guard<std::mutex> lk(m_lock.m);
        m_h.DropLastElements(i);
    }
    // (metadata only: a narrow tower's device copy keeps its leading limbs; members whose result buffer already has the new height)
    void DropLastMeta() {
        std::lock_guard<std::mutex> lk(m_lock.m);
        m_h.DropLastElement();
    }
    // a wide tower stays dense ([m_k][limbs][N]): its towers move together when limbs are dropped
    void CompactForDrop(size_t drop) {
        Settle();  // (a pending sum of wide towers is computed at its full height)
        if (m_k == 1 || !m_d || drop == 0 || drop >= NumLimbs())
            return;
        const size_t N = m_h.GetParams()->GetRingDimension(), L = NumLimbs(), l = L - drop;
        hiprt::Op op;
        auto d        = hiprt::Alloc((size_t)m_k * l * N);
        uint64_t* dst = op.W(d);
        for (uint32_t k = 0; k < m_k; ++k)
            hiprt::D2D(op, dst + (size_t)k * l * N, op.R(m_d) + (size_t)k * L * N, l * N * 8, "wide tower: limbs dropped");
        m_d = std::move(d);
    }
    // dcrtpoly-impl.h:693-712 (CKKS rescale)
    void DropLastElementAndScale(const std::vector<NativeInteger>& QlQlInvModqlDivqlModq,
                                 const std::vector<NativeInteger>& qlInvModq) override {
        FHE_HAL_MEMBER();
        if (!RescaleOnDevice(QlQlInvModqlDivqlModq, qlInvModq))
            Hm().DropLastElementAndScale(QlQlInvModqlDivqlModq, qlInvModq);
    }
    void ModReduce(const NativeInteger& t, const std::vector<NativeInteger>& tModqPrecon, const NativeInteger& negtInvModq,
                   const NativeInteger& negtInvModqPrecon, const std::vector<NativeInteger>& qlInvModq,
                   const std::vector<NativeInteger>& qlInvModqPrecon) override {
        FHE_HAL_MEMBER();
        if (ModReduceOnDevice(t, negtInvModq, qlInvModq))
            return;
        Hm().ModReduce(t, tModqPrecon, negtInvModq, negtInvModqPrecon, qlInvModq, qlInvModqPrecon);
    }

    PolyLargeType CRTInterpolate() const override {
        FHE_HAL_MEMBER();
        return Hc().CRTInterpolate();
    }
    PolyType DecryptionCRTInterpolate(PlaintextModulus ptm) const override {
        FHE_HAL_MEMBER();
        return Hc().DecryptionCRTInterpolate(ptm);
    }
    PolyType ToNativePoly() const override {
        FHE_HAL_MEMBER();
        return Hc().ToNativePoly();
    }
    PolyLargeType CRTInterpolateIndex(usint i) const override {
        FHE_HAL_MEMBER();
        return Hc().CRTInterpolateIndex(i);
    }
    Integer GetWorkingModulus() const override {
        return m_h.GetWorkingModulus();
    }
    // dcrtpoly-impl.h:630-647: one limb of `element`, in COEFFICIENT form, scaled to `modulus` through double precision
    void SetValuesModSwitch(const DCRTPolyType& element, const NativeInteger& modulus) override {
        FHE_HAL_MEMBER();
        if (!ModSwitchOnDevice(element, modulus))
            Hm().SetValuesModSwitch(element.Hc(), modulus);
    }
    std::shared_ptr<Params> GetExtendedCRTBasis(const std::shared_ptr<Params>& paramsP) const override {
        return m_h.GetExtendedCRTBasis(paramsP);
    }
    // dcrtpoly-impl.h:868-885
    void TimesQovert(const std::shared_ptr<Params>& paramsQ, const std::vector<NativeInteger>& tInvModq, const NativeInteger& t,
                     const NativeInteger& NegQModt, const NativeInteger& NegQModtPrecon) override {
        FHE_HAL_MEMBER();
        hiprt::Resolved r;
        if (tInvModq.size() >= NumLimbs() && t > NativeInteger(1) && NegQModt < t && OnDevice(&r)) {
            std::vector<uint64_t> ti(NumLimbs());
            for (uint32_t i = 0; i < NumLimbs(); ++i)
                ti[i] = tInvModq[i].ConvertToInt<uint64_t>();
            hiprt::Op op;
            auto dst = WriteTarget();
            hiprt::Check(hiprt::api().times_q_over_t(r.ctx, op.W(dst), op.R(m_d), t.ConvertToInt<uint64_t>(), NegQModt.ConvertToInt<uint64_t>(),
                                                     ti.data(), r.idx[0].data(), NumLimbs(), 1, op.s),
                         "TimesQovert");
            m_d = std::move(dst);
            hiprt::CountDevice();
            DeviceIsNewer(m_h.GetFormat());
            return;
        }
        Hm().TimesQovert(paramsQ, tInvModq, t, NegQModt, NegQModtPrecon);
    }

    // ---------------------------------------------------------------------------------------------------------------
    // CRT basis conversions (dcrtpoly-impl.h:888-1085)
    // ---------------------------------------------------------------------------------------------------------------
    DCRTPolyType ApproxSwitchCRTBasis(const std::shared_ptr<Params>& paramsQ, const std::shared_ptr<Params>& paramsP,
                                      const std::vector<NativeInteger>& QHatInvModq,
                                      const std::vector<NativeInteger>& QHatInvModqPrecon,
                                      const std::vector<std::vector<NativeInteger>>& QHatModp,
                                      const std::vector<DoubleNativeInt>& modpBarrettMu) const override {
        FHE_HAL_MEMBER();
        DCRTPolyType out;
        if (SwitchBasisOnDevice(paramsQ, paramsP, QHatInvModq, QHatModp, nullptr, nullptr, &out))
            return out;
        return Wrap(Hc().ApproxSwitchCRTBasis(paramsQ, paramsP, QHatInvModq, QHatInvModqPrecon, QHatModp, modpBarrettMu));
    }
    void ApproxModUp(const std::shared_ptr<Params>& paramsQ, const std::shared_ptr<Params>& paramsP,
                     const std::shared_ptr<Params>& paramsQP, const std::vector<NativeInteger>& QHatInvModq,
                     const std::vector<NativeInteger>& QHatInvModqPrecon,
                     const std::vector<std::vector<NativeInteger>>& QHatModp,
                     const std::vector<DoubleNativeInt>& modpBarrettMu) override {
        FHE_HAL_MEMBER();
        if (!ModUpOnDevice(paramsQ, paramsP, paramsQP, QHatInvModq, QHatModp))
            Hm().ApproxModUp(paramsQ, paramsP, paramsQP, QHatInvModq, QHatInvModqPrecon, QHatModp, modpBarrettMu);
    }
    DCRTPolyType ApproxModDown(const std::shared_ptr<Params>& paramsQ, const std::shared_ptr<Params>& paramsP,
                               const std::vector<NativeInteger>& PInvModq, const std::vector<NativeInteger>& PInvModqPrecon,
                               const std::vector<NativeInteger>& PHatInvModp,
                               const std::vector<NativeInteger>& PHatInvModpPrecon,
                               const std::vector<std::vector<NativeInteger>>& PHatModq,
                               const std::vector<DoubleNativeInt>& modqBarrettMu, const std::vector<NativeInteger>& tInvModp,
                               const std::vector<NativeInteger>& tInvModpPrecon, const NativeInteger& t,
                               const std::vector<NativeInteger>& tModqPrecon) const override {
        FHE_HAL_MEMBER();
        DCRTPolyType out;
        if (ModDownOnDevice(paramsQ, paramsP, PInvModq, PHatInvModp, PHatModq, tInvModp, t, &out))
            return out;
        return Wrap(Hc().ApproxModDown(paramsQ, paramsP, PInvModq, PInvModqPrecon, PHatInvModp, PHatInvModpPrecon, PHatModq,
                                       modqBarrettMu, tInvModp, tInvModpPrecon, t, tModqPrecon));
    }
    DCRTPolyType SwitchCRTBasis(const std::shared_ptr<Params>& paramsP, const std::vector<NativeInteger>& QHatInvModq,
                                const std::vector<NativeInteger>& QHatInvModqPrecon,
                                const std::vector<std::vector<NativeInteger>>& QHatModp,
                                const std::vector<std::vector<NativeInteger>>& alphaQModp,
                                const std::vector<DoubleNativeInt>& modpBarrettMu,
                                const std::vector<double>& qInv) const override {
        FHE_HAL_MEMBER();
        DCRTPolyType out;
        // (:1008-1085 index QHatModp as [j][i]: transposed with respect to ApproxSwitchCRTBasis)
        if (SwitchBasisOnDevice(m_h.GetParams(), paramsP, QHatInvModq, QHatModp, &alphaQModp, &qInv, &out, /*transposed=*/true))
            return out;
        return Wrap(Hc().SwitchCRTBasis(paramsP, QHatInvModq, QHatInvModqPrecon, QHatModp, alphaQModp, modpBarrettMu, qInv));
    }
    void ExpandCRTBasis(const std::shared_ptr<Params>& paramsQP, const std::shared_ptr<Params>& paramsP,
                        const std::vector<NativeInteger>& QHatInvModq, const std::vector<NativeInteger>& QHatInvModqPrecon,
                        const std::vector<std::vector<NativeInteger>>& QHatModp,
                        const std::vector<std::vector<NativeInteger>>& alphaQModp,
                        const std::vector<DoubleNativeInt>& modpBarrettMu, const std::vector<double>& qInv,
                        Format resultFormat) override {
        FHE_HAL_MEMBER();
        if (ExpandOnDevice(paramsQP, paramsP, QHatInvModq, QHatInvModqPrecon, QHatModp, alphaQModp, modpBarrettMu, qInv, resultFormat, false))
            return;
        Hm().ExpandCRTBasis(paramsQP, paramsP, QHatInvModq, QHatInvModqPrecon, QHatModp, alphaQModp, modpBarrettMu, qInv,
                            resultFormat);
    }
    void ExpandCRTBasisReverseOrder(const std::shared_ptr<Params>& paramsQP, const std::shared_ptr<Params>& paramsP,
                                    const std::vector<NativeInteger>& QHatInvModq,
                                    const std::vector<NativeInteger>& QHatInvModqPrecon,
                                    const std::vector<std::vector<NativeInteger>>& QHatModp,
                                    const std::vector<std::vector<NativeInteger>>& alphaQModp,
                                    const std::vector<DoubleNativeInt>& modpBarrettMu, const std::vector<double>& qInv,
                                    Format resultFormat) override {
        FHE_HAL_MEMBER();
        if (ExpandOnDevice(paramsQP, paramsP, QHatInvModq, QHatInvModqPrecon, QHatModp, alphaQModp, modpBarrettMu, qInv, resultFormat, true))
            return;
        Hm().ExpandCRTBasisReverseOrder(paramsQP, paramsP, QHatInvModq, QHatInvModqPrecon, QHatModp, alphaQModp, modpBarrettMu,
                                        qInv, resultFormat);
    }
    void FastExpandCRTBasisPloverQ(const Precomputations& pre) override {
        FHE_HAL_MEMBER();
        // dcrtpoly-impl.h:1151-1164: Q -> Pl (approximate, tables mPlQHatInvModq / qInvModp), Pl -> Ql (exact), result [Ql | Pl]
        if (hiprt::Available()) {
            DCRTPolyType partPl = ApproxSwitchCRTBasis(m_h.GetParams(), pre.paramsPl, pre.mPlQHatInvModq, pre.mPlQHatInvModqPrecon,
                                                       pre.qInvModp, pre.modpBarrettMu);
            DCRTPolyType partQl = partPl.SwitchCRTBasis(pre.paramsQl, pre.PlHatInvModp, pre.PlHatInvModpPrecon, pre.PlHatModq,
                                                        pre.alphaPlModq, pre.modqBarrettMu, pre.pInv);
            if (partPl.IsDeviceResident() && partQl.IsDeviceResident()) {
                const Format f = m_h.GetFormat();
                *this = AssembleRows(pre.paramsQlPl, f, {RowPiece{&partQl, 0, partQl.NumLimbs()}, RowPiece{&partPl, 0, partPl.NumLimbs()}});
                return;
            }
        }
        // (the struct is the interface's nested type; the mirror's is the same layout under its own name)
        typename HostType::Precomputations hp{pre.paramsQlPl,        pre.paramsPl,           pre.paramsQl,
                                              pre.mPlQHatInvModq,    pre.mPlQHatInvModqPrecon, pre.qInvModp,
                                              pre.modpBarrettMu,     pre.PlHatInvModp,       pre.PlHatInvModpPrecon,
                                              pre.PlHatModq,         pre.alphaPlModq,        pre.modqBarrettMu,
                                              pre.pInv};
        Hm().FastExpandCRTBasisPloverQ(hp);
    }
    void ExpandCRTBasisQlHat(const std::shared_ptr<Params>& paramsQ, const std::vector<NativeInteger>& QlHatModq,
                             const std::vector<NativeInteger>& QlHatModqPrecon, const usint sizeQ) override {
        FHE_HAL_MEMBER();
        // dcrtpoly-impl.h:1167-1187: limb i times QlHatModq[i], the limbs [sizeQl, sizeQ) zero
        const uint32_t sizeQl = NumLimbs();
        if (hiprt::Available() && QlHatModq.size() >= sizeQl && sizeQ >= sizeQl && paramsQ->GetParams().size() == sizeQ) {
            DCRTPolyType scaled(*this);
            if (scaled.TimesConstInPlace(QlHatModq)) {
                *this = AssembleRows(paramsQ, m_h.GetFormat(), {RowPiece{&scaled, 0, sizeQl}, RowPiece{nullptr, 0, sizeQ - sizeQl}});
                return;
            }
        }
        Hm().ExpandCRTBasisQlHat(paramsQ, QlHatModq, QlHatModqPrecon, sizeQ);
    }

    // dcrtpoly-impl.h:1190-1467 (BFV decryption, HPS): the tower scaled by t/Q and rounded -> one polynomial modulo t
    PolyType ScaleAndRound(const NativeInteger& t, const std::vector<NativeInteger>& tQHatInvModqDivqModt,
                           const std::vector<NativeInteger>& tQHatInvModqDivqModtPrecon,
                           const std::vector<NativeInteger>& tQHatInvModqBDivqModt,
                           const std::vector<NativeInteger>& tQHatInvModqBDivqModtPrecon,
                           const std::vector<double>& tQHatInvModqDivqFrac,
                           const std::vector<double>& tQHatInvModqBDivqFrac) const override {
        FHE_HAL_MEMBER();
        PolyType out;
        if (ScaleAndRoundNativeOnDevice(t, &tQHatInvModqDivqModt, &tQHatInvModqBDivqModt, &tQHatInvModqDivqFrac, &tQHatInvModqBDivqFrac, nullptr,
                                        NativeInteger(0), nullptr, nullptr, &out))
            return out;
        hiprt::CountHost(__func__, RingOf(m_h));
        return Hc().ScaleAndRound(t, tQHatInvModqDivqModt, tQHatInvModqDivqModtPrecon, tQHatInvModqBDivqModt,
                                  tQHatInvModqBDivqModtPrecon, tQHatInvModqDivqFrac, tQHatInvModqBDivqFrac);
    }
    DCRTPolyType ApproxScaleAndRound(const std::shared_ptr<Params>& paramsP,
                                     const std::vector<std::vector<NativeInteger>>& tPSHatInvModsDivsModp,
                                     const std::vector<DoubleNativeInt>& modpBarretMu) const override {
        FHE_HAL_MEMBER();
        DCRTPolyType out;
        if (ScaleAndRoundOnDevice(paramsP, tPSHatInvModsDivsModp, nullptr, &out))
            return out;
        return Wrap(Hc().ApproxScaleAndRound(paramsP, tPSHatInvModsDivsModp, modpBarretMu));
    }
    DCRTPolyType ScaleAndRound(const std::shared_ptr<Params>& paramsOutput,
                               const std::vector<std::vector<NativeInteger>>& tOSHatInvModsDivsModo,
                               const std::vector<double>& tOSHatInvModsDivsFrac,
                               const std::vector<DoubleNativeInt>& modoBarretMu) const override {
        FHE_HAL_MEMBER();
        DCRTPolyType out;
        if (ScaleAndRoundOnDevice(paramsOutput, tOSHatInvModsDivsModo, &tOSHatInvModsDivsFrac, &out))
            return out;
        return Wrap(Hc().ScaleAndRound(paramsOutput, tOSHatInvModsDivsModo, tOSHatInvModsDivsFrac, modoBarretMu));
    }
    // dcrtpoly-impl.h:1631-1671 (BFV decryption, BEHZ)
    PolyType ScaleAndRound(const std::vector<NativeInteger>& moduliQ, const NativeInteger& t, const NativeInteger& tgamma,
                           const std::vector<NativeInteger>& tgammaQHatModq,
                           const std::vector<NativeInteger>& tgammaQHatModqPrecon,
                           const std::vector<NativeInteger>& negInvqModtgamma,
                           const std::vector<NativeInteger>& negInvqModtgammaPrecon) const override {
        FHE_HAL_MEMBER();
        PolyType out;
        if (ScaleAndRoundNativeOnDevice(t, nullptr, nullptr, nullptr, nullptr, &moduliQ, tgamma, &tgammaQHatModq, &negInvqModtgamma, &out))
            return out;
        hiprt::CountHost(__func__, RingOf(m_h));
        return Hc().ScaleAndRound(moduliQ, t, tgamma, tgammaQHatModq, tgammaQHatModqPrecon, negInvqModtgamma,
                                  negInvqModtgammaPrecon);
    }
    // dcrtpoly-impl.h:1674-1689
    void ScaleAndRoundPOverQ(const std::shared_ptr<Params>& paramsQ, const std::vector<NativeInteger>& pInvModq) override {
        FHE_HAL_MEMBER();
        if (!POverQOnDevice(paramsQ, pInvModq))
            Hm().ScaleAndRoundPOverQ(paramsQ, pInvModq);
    }
    void FastBaseConvqToBskMontgomery(
        const std::shared_ptr<Params>& paramsQBsk, const std::vector<NativeInteger>& moduliQ,
        const std::vector<NativeInteger>& moduliBsk, const std::vector<DoubleNativeInt>& modbskBarrettMu,
        const std::vector<NativeInteger>& mtildeQHatInvModq, const std::vector<NativeInteger>& mtildeQHatInvModqPrecon,
        const std::vector<std::vector<NativeInteger>>& QHatModbsk, const std::vector<uint64_t>& QHatModmtilde,
        const std::vector<NativeInteger>& QModbsk, const std::vector<NativeInteger>& QModbskPrecon,
        const uint64_t& negQInvModmtilde, const std::vector<NativeInteger>& mtildeInvModbsk,
        const std::vector<NativeInteger>& mtildeInvModbskPrecon) override {
        FHE_HAL_MEMBER();
        // the plan computes with THESE tables (hiprt::BehzPlan, one per member and table content)
        const size_t numQ = moduliQ.size(), numBsk = moduliBsk.size();
        std::vector<uint64_t> tabs;
        if (mtildeQHatInvModq.size() >= numQ && QHatModbsk.size() >= numQ && QHatModmtilde.size() >= numQ && QModbsk.size() >= numBsk &&
            mtildeInvModbsk.size() >= numBsk && PushMatrix(tabs, QHatModbsk, numQ, numBsk, true)) {
            tabs.clear();
            PushVector(tabs, mtildeQHatInvModq, numQ);
            PushMatrix(tabs, QHatModbsk, numQ, numBsk);
            tabs.insert(tabs.end(), QHatModmtilde.begin(), QHatModmtilde.begin() + numQ);
            PushVector(tabs, QModbsk, numBsk);
            tabs.push_back(negQInvModmtilde);
            PushVector(tabs, mtildeInvModbsk, numBsk);
            if (BehzOnDevice(0, paramsQBsk, moduliQ, moduliBsk, 0, tabs))
                return;
        }
        Hm().FastBaseConvqToBskMontgomery(paramsQBsk, moduliQ, moduliBsk, modbskBarrettMu, mtildeQHatInvModq, mtildeQHatInvModqPrecon,
                                          QHatModbsk, QHatModmtilde, QModbsk, QModbskPrecon, negQInvModmtilde, mtildeInvModbsk,
                                          mtildeInvModbskPrecon);
    }
    void FastRNSFloorq(const NativeInteger& t, const std::vector<NativeInteger>& moduliQ,
                       const std::vector<NativeInteger>& moduliBsk, const std::vector<DoubleNativeInt>& modbskBarrettMu,
                       const std::vector<NativeInteger>& tQHatInvModq, const std::vector<NativeInteger>& tQHatInvModqPrecon,
                       const std::vector<std::vector<NativeInteger>>& QHatModbsk,
                       const std::vector<std::vector<NativeInteger>>& qInvModbsk, const std::vector<NativeInteger>& tQInvModbsk,
                       const std::vector<NativeInteger>& tQInvModbskPrecon) override {
        FHE_HAL_MEMBER();
        const size_t numQ = moduliQ.size(), numBsk = moduliBsk.size();
        std::vector<uint64_t> tabs;
        if (tQHatInvModq.size() >= numQ && tQInvModbsk.size() >= numBsk && PushMatrix(tabs, QHatModbsk, numQ, numBsk, true) &&
            PushMatrix(tabs, qInvModbsk, numQ, numBsk, true)) {
            tabs.clear();
            PushVector(tabs, tQHatInvModq, numQ);
            PushMatrix(tabs, QHatModbsk, numQ, numBsk);
            PushMatrix(tabs, qInvModbsk, numQ, numBsk);
            PushVector(tabs, tQInvModbsk, numBsk);
            if (BehzOnDevice(1, m_h.GetParams(), moduliQ, moduliBsk, t.ConvertToInt<uint64_t>(), tabs))
                return;
        }
        Hm().FastRNSFloorq(t, moduliQ, moduliBsk, modbskBarrettMu, tQHatInvModq, tQHatInvModqPrecon, QHatModbsk, qInvModbsk,
                           tQInvModbsk, tQInvModbskPrecon);
    }
    void FastBaseConvSK(const std::shared_ptr<Params>& paramsQ, const std::vector<DoubleNativeInt>& modqBarrettMu,
                        const std::vector<NativeInteger>& moduliBsk, const std::vector<DoubleNativeInt>& modbskBarrettMu,
                        const std::vector<NativeInteger>& BHatInvModb, const std::vector<NativeInteger>& BHatInvModbPrecon,
                        const std::vector<NativeInteger>& BHatModmsk, const NativeInteger& BInvModmsk,
                        const NativeInteger& BInvModmskPrecon, const std::vector<std::vector<NativeInteger>>& BHatModq,
                        const std::vector<NativeInteger>& BModq, const std::vector<NativeInteger>& BModqPrecon) override {
        FHE_HAL_MEMBER();
        const size_t numQ = paramsQ->GetParams().size(), numBsk = moduliBsk.size(), numB = numBsk ? numBsk - 1 : 0;
        std::vector<uint64_t> tabs;
        if (numB == numQ && BHatInvModb.size() >= numB && BHatModmsk.size() >= numB && BModq.size() >= numQ &&
            PushMatrix(tabs, BHatModq, numB, numQ, true)) {
            tabs.clear();
            PushVector(tabs, BHatInvModb, numB);
            PushVector(tabs, BHatModmsk, numB);
            tabs.push_back(BInvModmsk.ConvertToInt<uint64_t>());
            PushMatrix(tabs, BHatModq, numB, numQ);
            PushVector(tabs, BModq, numQ);
            std::vector<NativeInteger> moduliQ(numQ);
            for (size_t i = 0; i < numQ; ++i)
                moduliQ[i] = paramsQ->GetParams()[i]->GetModulus();
            if (BehzOnDevice(2, paramsQ, moduliQ, moduliBsk, 0, tabs))
                return;
        }
        Hm().FastBaseConvSK(paramsQ, modqBarrettMu, moduliBsk, modbskBarrettMu, BHatInvModb, BHatInvModbPrecon, BHatModmsk,
                            BInvModmsk, BInvModmskPrecon, BHatModq, BModq, BModqPrecon);
    }

    // dcrtpoly-impl.h:1932-1940 -> ChineseRemainderTransformFTT (transformnat-impl.h:303-374, 512-625)
    void SwitchFormat(uint32_t thread_limit = 0) override {
        FHE_HAL_MEMBER();
        hiprt::Resolved r;
        if (OnDeviceWide(&r)) {
            const bool toCoeff = m_h.GetFormat() == Format::EVALUATION;
            hiprt::Op op;
            if (SharedWords()) {  // words shared with a copy (or a window of a wide tower): transform into a buffer of its own
                auto d = hiprt::Alloc(Words());
                auto f = toCoeff ? hiprt::api().ntt_inv_oop : hiprt::api().ntt_fwd_oop;
                hiprt::Check(f(r.ctx, op.R(m_d), op.W(d), r.idx[0].data(), NumLimbs(), m_k, op.s), "SwitchFormat");
                m_d = std::move(d);
            }
            else {
                auto f = toCoeff ? hiprt::api().ntt_inv : hiprt::api().ntt_fwd;
                hiprt::Check(f(r.ctx, op.W(m_d), r.idx[0].data(), NumLimbs(), m_k, op.s), "SwitchFormat");
            }
            hiprt::CountDevice();
            DeviceIsNewer(toCoeff ? Format::COEFFICIENT : Format::EVALUATION);
            return;
        }
        Hm().SwitchFormat(thread_limit);
    }
    void SwitchModulusAtIndex(size_t index, const Integer& modulus, const Integer& rootOfUnity) override {
        FHE_HAL_MEMBER();
        Hm().SwitchModulusAtIndex(index, modulus, rootOfUnity);
    }

    template <class Archive>
    void save(Archive& ar, std::uint32_t const version) const {
        Hc().save(ar, version);
    }
    template <class Archive>
    void load(Archive& ar, std::uint32_t const version) {
        m_d.reset();
        m_lazy.reset();
        m_hostValid = true;
        m_zero      = false;
        m_h.load(ar, version);
    }
    static const std::string GetElementName() {
        return "DCRTPolyImpl";
    }
    std::string SerializedObjectName() const override {
        return "DCRTPoly";
    }
    static uint32_t SerializedVersion() {
        return 1;
    }

    inline Format GetFormat() const final {
        return Meta().GetFormat();
    }
    void OverrideFormat(const Format f) final {
        std::lock_guard<std::mutex> lk(m_lock.m);
        m_h.OverrideFormat(f);
    }
    inline const std::shared_ptr<Params>& GetParams() const {
        return Meta().GetParams();
    }
    // the limbs as host objects: filled from the device on demand (const) / the device copy is dropped (mutable access)
    usint GetNumOfElements() const {  // (the interface's version reaches GetAllElements(): a device -> host copy for a count)
        return NumLimbs();
    }
    const std::vector<PolyType>& GetAllElements() const {
        FHE_HAL_MEMBER();
        const auto& limbs = Hc().GetAllElements();
        if (m_k > 1 && m_d && !limbs.empty() && limbs[0].GetLength() > 0) {
            // the limbs of a wide tower read on the host are tower 0's.  Bootstrapping's ModRaise hands limb 0 straight back to the
            // ModRaise constructor (ckksrns-fhe.cpp: `DCRTPoly tmp(dcrt.GetElementAtIndex(0), paramsRaised)`): remember whose limb it is
            LastWideRead() = WideRead{&limbs[0].GetValues()[0], m_d, m_k, NumLimbs()};
        }
        return limbs;
    }
    std::vector<PolyType>& GetAllElements() {
        FHE_HAL_MEMBER();
        return Hm(__func__, true).GetAllElements();
    }
    void SetElementAtIndex(usint index, const PolyType& element) {
        FHE_HAL_MEMBER();
        Hm(__func__, true).SetElementAtIndex(index, element);
    }
    void SetElementAtIndex(usint index, PolyType&& element) {
        FHE_HAL_MEMBER();
        Hm(__func__, true).SetElementAtIndex(index, std::move(element));
    }

    // ---- row-level helpers for the backend's own overrides of pke's limb loops (keyswitch/keyswitch-hybrid.h in this directory) --
    // A piece = rows [first, first + n) of `src` (nullptr: n rows of zeros).  The pieces, laid out one after the other, become
    // the limbs of a new tower over `params` in format `f` — what pke writes as a loop of
    // `result.SetElementAtIndex(i, src.GetElementAtIndex(j))` (keyswitch-hybrid.cpp:356-376, :228-237).
    // transform: the source rows are in the OTHER format than the assembled tower and are transformed while they are placed
    // (pke's `partsCt.SetFormat(COEFFICIENT)` / `partsCtCompl.SetFormat(EVALUATION)` around the copies, :362, :369)
    struct RowPiece {
        const DCRTPolyType* src;
        uint32_t first, n;
        bool transform = false;
    };
    static DCRTPolyType AssembleRows(const std::shared_ptr<Params>& params, Format f, const std::vector<RowPiece>& pieces) {
        FHE_HAL_MEMBER();
        const size_t N   = params->GetRingDimension();
        uint32_t total   = 0;
        bool deviceOk    = hiprt::Available();
        for (const auto& pc : pieces) {
            total += pc.n;
            if (pc.src && (pc.first + pc.n > pc.src->NumLimbs()))
                OPENFHE_THROW("AssembleRows: row range outside the source tower");
        }
        if (total != params->GetParams().size())
            OPENFHE_THROW("AssembleRows: pieces do not cover the parameter set");
        hiprt::Resolved r;
        deviceOk = deviceOk && ResolveSets(params->GetRingDimension(), {params}, &r);
        for (const auto& pc : pieces)
            if (deviceOk && pc.src)
                deviceOk = pc.src->Upload();
        if (deviceOk) {
            hiprt::Op op;
            auto d       = hiprt::Alloc((size_t)total * N);
            uint64_t* dp = op.W(d);
            uint32_t at  = 0;
            for (const auto& pc : pieces) {
                if (pc.src && pc.n && pc.transform) {
                    auto fn = f == Format::EVALUATION ? hiprt::api().ntt_fwd_oop : hiprt::api().ntt_inv_oop;
                    hiprt::Check(fn(r.ctx, op.R(pc.src->m_d) + (size_t)pc.first * N, dp + (size_t)at * N, r.idx[0].data() + at, pc.n, 1, op.s),
                                 "AssembleRows");
                }
                else if (pc.src && pc.n)
                    hiprt::D2D(op, dp + (size_t)at * N, op.R(pc.src->m_d) + (size_t)pc.first * N, (size_t)pc.n * N * 8, "AssembleRows");
                else if (pc.n)
                    hiprt::Check(hiprt::api().memset_zero(r.ctx, dp + (size_t)at * N, (size_t)pc.n * N * 8, op.s), "AssembleRows");
                at += pc.n;
            }
            hiprt::CountDevice();
            return FromDevice(params, f, std::move(d));
        }
        HostType h(params, f, true);  // host: the reference's own loop, on an object of the reference's class
        uint32_t at = 0;
        for (const auto& pc : pieces) {
            for (uint32_t i = 0; i < pc.n; ++i, ++at)
                if (pc.src) {
                    PolyType e = pc.src->Hc().GetElementAtIndex(pc.first + i);
                    if (pc.transform)
                        e.SetFormat(f);
                    h.SetElementAtIndex(at, std::move(e));
                }
        }
        return Wrap(std::move(h));
    }
    // { sum_j x[j][i] * k0[j][idx(i)],  sum_j x[j][i] * k1[j][idx(i)] },  idx(i) = i < sizeQl ? i : i + keySkip — the two sums
    // of EvalFastKeySwitchCoreExt (keyswitch-hybrid.cpp:419-430) in one pass over the digits (fhe_inner_product)
    static std::vector<DCRTPolyType> InnerProduct(const std::vector<DCRTPolyType>& x, const std::vector<DCRTPolyType>& k0,
                                                  const std::vector<DCRTPolyType>& k1, uint32_t sizeQl, uint32_t keySkip) {
        FHE_HAL_MEMBER();
        const auto& params = x[0].GetParams();
        const uint32_t rows = x[0].NumLimbs(), n = (uint32_t)x.size();
        hiprt::Resolved r;
        bool deviceOk = n >= 1 && k0.size() >= n && k1.size() >= n && x[0].OnDevice(&r);
        for (uint32_t j = 0; deviceOk && j < n; ++j)
            deviceOk = x[j].NumLimbs() == rows && x[j].GetFormat() == Format::EVALUATION && k0[j].NumLimbs() >= rows + keySkip &&
                       k1[j].NumLimbs() >= rows + keySkip && x[j].Upload() && k0[j].Upload() && k1[j].Upload();
        std::vector<DCRTPolyType> out;
        if (deviceOk) {
            const size_t N = params->GetRingDimension();
            hiprt::Op op;
            std::vector<const uint64_t*> px(n), p0(n), p1(n);
            for (uint32_t j = 0; j < n; ++j)
                px[j] = op.R(x[j].m_d), p0[j] = op.R(k0[j].m_d), p1[j] = op.R(k1[j].m_d);
            std::vector<uint32_t> keyRow(rows);
            for (uint32_t i = 0; i < rows; ++i)
                keyRow[i] = i < sizeQl ? i : i + keySkip;
            auto d0 = hiprt::Alloc((size_t)rows * N), d1 = hiprt::Alloc((size_t)rows * N);
            hiprt::Check(hiprt::api().inner_product(r.ctx, n, px.data(), p0.data(), p1.data(), keyRow.data(), r.idx[0].data(), rows, 1,
                                                    op.W(d0), op.W(d1), op.s),
                         "InnerProduct");
            hiprt::CountDevice();
            out.push_back(FromDevice(params, Format::EVALUATION, std::move(d0)));
            out.push_back(FromDevice(params, Format::EVALUATION, std::move(d1)));
            return out;
        }
        out.emplace_back(params, Format::EVALUATION, true);
        out.emplace_back(params, Format::EVALUATION, true);
        for (uint32_t j = 0; j < n; ++j) {
            out[0].MultAccRows(0, x[j], 0, k0[j], 0, sizeQl);
            out[0].MultAccRows(sizeQl, x[j], sizeQl, k0[j], sizeQl + keySkip, rows - sizeQl);
            out[1].MultAccRows(0, x[j], 0, k1[j], 0, sizeQl);
            out[1].MultAccRows(sizeQl, x[j], sizeQl, k1[j], sizeQl + keySkip, rows - sizeQl);
        }
        return out;
    }
    // this[outFirst + i] += a[aFirst + i] * b[bFirst + i], i < n, EVALUATION — the accumulation of EvalFastKeySwitchCoreExt
    // (keyswitch-hybrid.cpp:419-430) on whole row ranges
    void MultAccRows(uint32_t outFirst, const DCRTPolyType& a, uint32_t aFirst, const DCRTPolyType& b, uint32_t bFirst, uint32_t n) {
        FHE_HAL_MEMBER();
        if (outFirst + n > NumLimbs() || aFirst + n > a.NumLimbs() || bFirst + n > b.NumLimbs())
            OPENFHE_THROW("MultAccRows: row range outside a tower");
        hiprt::Resolved r;
        if (OnDevice(&r) && a.Upload() && b.Upload()) {
            const size_t N = m_h.GetParams()->GetRingDimension();
            Unshare();
            hiprt::Op op;
            hiprt::Check(hiprt::api().mul_add(r.ctx, op.W(m_d) + (size_t)outFirst * N, op.R(a.m_d) + (size_t)aFirst * N,
                                              op.R(b.m_d) + (size_t)bFirst * N, r.idx[0].data() + outFirst, n, 1, op.s),
                         "MultAccRows");
            hiprt::CountDevice();
            DeviceIsNewer(m_h.GetFormat());
            return;
        }
        HostType& h = Hm();
        for (uint32_t i = 0; i < n; ++i)
            h.SetElementAtIndex(outFirst + i, h.GetElementAtIndex(outFirst + i) +
                                                  a.Hc().GetElementAtIndex(aFirst + i) * b.Hc().GetElementAtIndex(bFirst + i));
    }

    // { a0*b0, a0*b1 + a1*b0, a1*b1 }: LeveledSHEBase::EvalMultCore for two 2-element ciphertexts (base-leveledshe.cpp:619-623) in one
    // pass over the four towers (fhe_tensor); b0 == nullptr: EvalSquareCore's { a0*a0, 2*a0*a1, a1*a1 } (:646-664).  Empty when the
    // towers cannot take the device path (the caller then runs the reference's five tower operations).
    static std::vector<DCRTPolyType> Tensor(const DCRTPolyType& a0, const DCRTPolyType& a1, const DCRTPolyType* b0, const DCRTPolyType* b1) {
        FHE_HAL_MEMBER();
        std::vector<DCRTPolyType> out;
        hiprt::Resolved r;
        if (!a0.Compatible(a1, true) || (b0 && (!a0.Compatible(*b0, true) || !a0.Compatible(*b1, true))) || !a0.OnDeviceWide(&r) || !a1.Upload() ||
            (b0 && (!b0->Upload() || !b1->Upload())))
            return out;
        const uint32_t L = a0.NumLimbs();
        const uint32_t k = std::max(std::max(a0.m_k, a1.m_k), b0 ? std::max(b0->m_k, b1->m_k) : 1u);
        const hiprt::Buf x0 = a0.Widened(k), x1 = a1.Widened(k), y0 = b0 ? b0->Widened(k) : nullptr, y1 = b0 ? b1->Widened(k) : nullptr;
        const size_t words = (size_t)k * a0.TowerWords();
        hiprt::Op op;
        auto d0 = hiprt::Alloc(words), d1 = hiprt::Alloc(words), d2 = hiprt::Alloc(words);
        if (b0)
            hiprt::Check(hiprt::api().tensor(r.ctx, op.R(x0), op.R(x1), op.R(y0), op.R(y1), op.W(d0), op.W(d1), op.W(d2),
                                             r.idx[0].data(), L, k, op.s),
                         "EvalMultCore");
        else
            hiprt::Check(hiprt::api().tensor_square(r.ctx, op.R(x0), op.R(x1), op.W(d0), op.W(d1), op.W(d2), r.idx[0].data(), L, k, op.s),
                         "EvalSquareCore");
        hiprt::CountDevice();
        const auto& P = a0.m_h.GetParams();
        out.push_back(FromDevice(P, Format::EVALUATION, std::move(d0), k));
        out.push_back(FromDevice(P, Format::EVALUATION, std::move(d1), k));
        out.push_back(FromDevice(P, Format::EVALUATION, std::move(d2), k));
        return out;
    }

    // ---- the two elements of a ciphertext in ONE launch each (hal/keyswitch-hybrid-hip.cpp's definitions of pke's element loops):
    // pke applies += / -= / * constants / DropLastElementAndScale to cv[0] and cv[1] one after the other (base-leveledshe.cpp:562-606,
    // ckksrns-leveledshe.cpp:172-191, :748-759); a single ciphertext's tower fills less than half of the chip, so the second launch
    // costs as much as the first.  false: the towers cannot take the paired device path — the caller runs the element loop.
    static bool PairAddInPlace(DCRTPolyType& a0, DCRTPolyType& a1, const DCRTPolyType& b0, const DCRTPolyType& b1, bool subtract) {
        hiprt::MemberScope scope(subtract ? "operator-=" : "operator+=");
        hiprt::Resolved r;
        if (a0.m_k != 1 || a1.m_k != 1 || b0.m_k != 1 || b1.m_k != 1)
            return false;  // (wide towers: one launch per element is already K towers)
        if (&a0 == &a1 || !a0.Compatible(a1, false) || !a0.Compatible(b0, false) || !a0.Compatible(b1, false) || !a0.OnDevice(&r) || !a1.Upload() ||
            !b0.Upload() || !b1.Upload() || a0.m_d == a1.m_d || b0.m_d == b1.m_d)
            return false;
        const auto& A = hiprt::api();
        hiprt::Op op;
        auto d0 = a0.WriteTarget(), d1 = a1.WriteTarget();
        const uint64_t *pa0 = op.R(a0.m_d), *pa1 = op.R(a1.m_d), *pb0 = op.R(b0.m_d), *pb1 = op.R(b1.m_d);
        hiprt::Check((subtract ? A.sub_pair : A.add_pair)(r.ctx, op.W(d0), op.W(d1), pa0, pa1, pb0, pb1, r.idx[0].data(), a0.NumLimbs(), op.s),
                     "DCRTPoly arithmetic on both elements");
        a0.m_d = std::move(d0), a1.m_d = std::move(d1);
        hiprt::CountDevice();
        a0.DeviceIsNewer(a0.m_h.GetFormat()), a1.DeviceIsNewer(a1.m_h.GetFormat());
        return true;
    }
    // a_e = a_e * factors (limb i times NativeInteger(factors[i]), dcrtpoly-impl.h:572-580) on both elements
    static bool PairTimesInPlace(DCRTPolyType& a0, DCRTPolyType& a1, const std::vector<Integer>& factors) {
        hiprt::MemberScope scope("Times");
        hiprt::Resolved r;
        const uint32_t L = a0.NumLimbs();
        if (a0.m_k != 1 || a1.m_k != 1)
            return false;
        if (&a0 == &a1 || factors.size() < L || !a0.Compatible(a1, false) || !a0.OnDevice(&r) || !a1.Upload() || a0.m_d == a1.m_d)
            return false;
        std::vector<uint64_t> k(L);
        for (uint32_t i = 0; i < L; ++i)
            k[i] = NativeInteger(factors[i]).template ConvertToInt<uint64_t>();
        hiprt::Op op;
        auto d0 = a0.WriteTarget(), d1 = a1.WriteTarget();
        const uint64_t *pa0 = op.R(a0.m_d), *pa1 = op.R(a1.m_d);
        hiprt::Check(hiprt::api().mul_const_pair(r.ctx, op.W(d0), op.W(d1), pa0, pa1, k.data(), r.idx[0].data(), L, op.s),
                     "DCRTPoly Times(constants) on both elements");
        a0.m_d = std::move(d0), a1.m_d = std::move(d1);
        hiprt::CountDevice();
        a0.DeviceIsNewer(a0.m_h.GetFormat()), a1.DeviceIsNewer(a1.m_h.GetFormat());
        return true;
    }
    // DropLastElementAndScale (dcrtpoly-impl.h:693-712) of both elements with the same tables: the four launches of fhe_rescale_limbs,
    // each over both towers.  (Elements that are clones of towers rescaled before take the remembered results, see RescaleOnDevice.)
    static bool PairRescaleInPlace(DCRTPolyType& a0, DCRTPolyType& a1, const std::vector<NativeInteger>& QlQlInvModqlDivqlModq,
                                   const std::vector<NativeInteger>& qlInvModq) {
        hiprt::MemberScope scope("DropLastElementAndScale");
        const uint32_t L = a0.NumLimbs();
        if (a0.m_k != 1 || a1.m_k != 1)
            return false;
        if (&a0 == &a1 || L < 2 || !a0.Compatible(a1, true) || QlQlInvModqlDivqlModq.size() < L - 1 || qlInvModq.size() < L - 1)
            return false;
        hiprt::Resolved r;
        if (!a0.OnDevice(&r) || !a1.Upload() || a0.m_d == a1.m_d)
            return false;
        const size_t N   = a0.m_h.GetParams()->GetRingDimension();
        const uint32_t l = L - 1;
        std::vector<uint64_t> a(l), b(l);
        for (uint32_t i = 0; i < l; ++i) {
            a[i] = QlQlInvModqlDivqlModq[i].ConvertToInt<uint64_t>();
            b[i] = qlInvModq[i].ConvertToInt<uint64_t>();
        }
        // clones of towers rescaled before: both results remembered -> taken; one of them -> element by element (the member looks it up)
        const hiprt::Buf s0 = a0.m_d, s1 = a1.m_d;
        const bool c0 = s0->Owners(s0, 2) >= 1, c1 = s1->Owners(s1, 2) >= 1;  // another tower holds these words (windows do not count)
        const bool remember = c0 && c1 && !s0->parent && !s1->parent;
        std::vector<uint64_t> memoKey;
        if (c0 || c1) {
            if (!remember)
                return false;
            memoKey = RescaleMemoKey(r, L, a, b);
            auto h0 = hiprt::MemoFind(s0, memoKey), h1 = hiprt::MemoFind(s1, memoKey);
            if (h0 && h1) {
                a0.m_d = std::move(h0), a1.m_d = std::move(h1);
                a0.DeviceIsNewer(Format::EVALUATION), a1.DeviceIsNewer(Format::EVALUATION);
                a0.DropLastMeta(), a1.DropLastMeta();
                return true;
            }
            if (h0 || h1)
                return false;
        }
        const auto& A = hiprt::api();
        hiprt::Op op;
        const uint64_t *x0 = op.R(a0.m_d), *x1 = op.R(a1.m_d);
        const size_t wsBytes = A.rescale_workspace_bytes(r.ctx, L, 2);
        auto ws = hiprt::Alloc(wsBytes / 8), o0 = hiprt::Alloc((size_t)l * N), o1 = hiprt::Alloc((size_t)l * N);
        hiprt::Check(A.rescale_limbs_pair(r.ctx, x0, x1, r.idx[0].data(), L, a.data(), b.data(), op.W(o0), op.W(o1), op.W(ws, false), wsBytes, op.s),
                     "DropLastElementAndScale on both elements");
        if (remember) {
            hiprt::MemoStore(s0, memoKey, o0);
            hiprt::MemoStore(s1, std::move(memoKey), o1);
        }
        a0.m_d = std::move(o0), a1.m_d = std::move(o1);
        hiprt::CountDevice();
        a0.DeviceIsNewer(Format::EVALUATION), a1.DeviceIsNewer(Format::EVALUATION);
        a0.DropLastMeta(), a1.DropLastMeta();
        return true;
    }
    static std::vector<uint64_t> RescaleMemoKey(const hiprt::Resolved& r, uint32_t L, const std::vector<uint64_t>& a, const std::vector<uint64_t>& b) {
        std::vector<uint64_t> key;
        key.reserve(2 + 3 * (size_t)L);
        key.push_back(1), key.push_back(L);
        key.insert(key.end(), r.idx[0].begin(), r.idx[0].end());
        key.insert(key.end(), a.begin(), a.end());
        key.insert(key.end(), b.begin(), b.end());
        return key;
    }

    // the host mirror (synchronised), for code that wants the default implementation's object
    const HostType& Host() const {
        return Hc();
    }
    // true while the authoritative copy of the words is the device buffer
    bool IsDeviceResident() const {
        return (m_d || m_lazy) && !m_hostValid;
    }
    // ---- for the backend's composite hooks (hal/keyswitch-hybrid-hip.cpp: whole pke operations as ONE library call) ----
    // the tower's device words (uploaded if they are on the host), nullptr when the tower cannot live on the device
    hiprt::Buf DeviceWords() const {
        const auto& P = m_h.GetParams();
        if (!hiprt::Available() || !P || NumLimbs() == 0 || NumLimbs() != P->GetParams().size() || !Upload())
            return nullptr;
        return m_d;
    }
    // the same for an operation that UPDATES the words in place: a private copy if they were shared with another tower, the device
    // copy becomes the authoritative one
    hiprt::Buf DeviceWordsForUpdate() {
        if (!DeviceWords())
            return nullptr;
        Unshare();
        DeviceIsNewer(m_h.GetFormat());
        return m_d;
    }
    // a tower over `params` in format f whose words are the device buffer d ([limbs][N])
    static DCRTPolyType FromDeviceWords(const std::shared_ptr<Params>& params, Format f, hiprt::Buf d, uint32_t k = 1) {
        return FromDevice(params, f, std::move(d), k);
    }
    uint32_t Width() const {
        return m_k;
    }
    // replaces this tower's device words by a window of a packed buffer holding the same values (evaluation keys packed for the
    // library's plans: no second copy stays behind)
    void AdoptDeviceWords(hiprt::Buf d) const {
        std::lock_guard<std::mutex> lk(m_lock.m);
        auto P      = m_h.GetParams();
        m_h         = HostType(P, m_h.GetFormat(), false);  // (the device words are the tower now: no stale host copy stays behind)
        m_d         = std::move(d);
        m_lazy.reset();
        m_hostValid = false;
        m_zero      = false;
    }

private:
    struct Lock {  // a mutex that does not travel with copies
        std::mutex m;
        Lock() = default;
        Lock(const Lock&) {}
        Lock& operator=(const Lock&) {
            return *this;
        }
    };

    mutable HostType m_h;           // metadata always; words valid iff m_hostValid
    mutable hiprt::Buf m_d;         // device words [nLimbs][N] (may hold more rows than nLimbs after DropLastElement)
    // A WIDE tower (lockstep evaluation) whose value is a weighted sum not yet computed: value = sum_t k_t (.) words_t per limb.  pke's
    // weighted sums (internalEvalLinearWSumMutable, ckksrns-advancedshe.cpp:97-136: the inner loops of bootstrapping's Chebyshev
    // evaluation) are `EvalMultInPlace(ct_i, c_i); EvalAddInPlaceNoCheck(ct_0, ct_i)` per term — Times(vector<Integer>) and operator+=
    // here: the product and the sums are recorded, and the first member that needs the words computes the whole sum in ONE launch that
    // reads every term once (fhe_lincomb) instead of 2n - 1 launches moving 5n - 3 towers.  While m_lazy is set m_d is null and the
    // mirror holds (params, format) only; the terms' buffers are shared references, so nobody writes them in place meanwhile
    // (WriteTarget / Unshare copy shared words first).  Exact modular arithmetic: the residues are the reference's.
    struct LazyTerm {
        hiprt::Buf words;
        std::vector<uint64_t> k;  // factor of limb i (reduced modulo q_i), at least NumLimbs() entries
    };
    struct LazySum {
        std::vector<LazyTerm> terms;
    };
    mutable std::shared_ptr<const LazySum> m_lazy;
    mutable bool m_hostValid{true};
    mutable bool m_zero{false};     // an all-zero tower not yet materialised on either side (then !m_hostValid && !m_d)
    // WIDE tower (hiprt::WidthScope): the device buffer holds m_k towers [m_k][nLimbs][N], dense, of m_k ciphertexts with equal metadata
    // that pke evaluates in lockstep.  A wide tower has no host form: its mirror carries (params, format) and, when something reads limbs
    // for their metadata, the words of tower 0; members without a wide device path throw.
    uint32_t m_k{1};
    mutable Lock m_lock;

    // (params, format, limb count) live in the mirror object, which SyncHost() never replaces (see there)
    const HostType& Meta() const {
        return m_h;
    }
    uint32_t NumLimbs() const {
        return (uint32_t)m_h.GetAllElements().size();
    }
    size_t Words() const {
        return (size_t)m_k * NumLimbs() * m_h.GetParams()->GetRingDimension();
    }
    size_t TowerWords() const {  // one tower of a wide one
        return (size_t)NumLimbs() * m_h.GetParams()->GetRingDimension();
    }
    static uint32_t RingOf(const HostType& h) {
        return h.GetParams() ? h.GetParams()->GetRingDimension() : 0;
    }
    static DCRTPolyType Wrap(HostType&& h, const char* who = __builtin_FUNCTION()) {
        hiprt::CountHost(who, RingOf(h));
        return DCRTPolyType(std::move(h));
    }
    static std::vector<DCRTPolyType> WrapAll(std::vector<HostType>&& v, const char* who = __builtin_FUNCTION()) {
        hiprt::CountHost(who, v.empty() ? 0 : RingOf(v[0]));
        std::vector<DCRTPolyType> r;
        r.reserve(v.size());
        for (auto& h : v)
            r.emplace_back(std::move(h));
        return r;
    }
    static DCRTPolyType FromDevice(const std::shared_ptr<Params>& p, Format f, hiprt::Buf d, uint32_t k = 1) {
        DCRTPolyType r;
        r.m_h         = HostType(p, f, false);
        r.m_d         = std::move(d);
        r.m_hostValid = false;
        r.m_k         = k;
        return r;
    }
    static void LimbsOf(const std::shared_ptr<Params>& p, std::vector<uint64_t>& q, std::vector<uint64_t>& psi) {
        const auto& v = p->GetParams();
        q.resize(v.size());
        psi.resize(v.size());
        for (size_t i = 0; i < v.size(); ++i) {
            q[i]   = v[i]->GetModulus().template ConvertToInt<uint64_t>();
            psi[i] = v[i]->GetRootOfUnity().template ConvertToInt<uint64_t>();
        }
    }
    // resolves the context limbs of several parameter sets at once (one lock, one context generation)
    static bool ResolveSets(uint32_t N, const std::vector<std::shared_ptr<Params>>& sets, hiprt::Resolved* r) {
        if (!hiprt::Available())
            return false;
        std::vector<std::vector<uint64_t>> q(sets.size()), psi(sets.size());
        std::vector<hiprt::LimbSet> ls(sets.size());
        for (size_t i = 0; i < sets.size(); ++i) {
            LimbsOf(sets[i], q[i], psi[i]);
            ls[i] = hiprt::LimbSet{q[i].data(), psi[i].data(), (uint32_t)q[i].size()};
        }
        return hiprt::Resolve(N, ls, r);
    }
    static void PushVector(std::vector<uint64_t>& out, const std::vector<NativeInteger>& v, size_t n) {
        for (size_t i = 0; i < n; ++i)
            out.push_back(v[i].ConvertToInt<uint64_t>());
    }
    // appends m[0..rows)[0..cols) row-major; checkOnly: only tells whether the matrix has that many entries
    static bool PushMatrix(std::vector<uint64_t>& out, const std::vector<std::vector<NativeInteger>>& m, size_t rows, size_t cols,
                           bool checkOnly = false) {
        if (m.size() < rows)
            return false;
        for (size_t i = 0; i < rows; ++i)
            if (m[i].size() < cols)
                return false;
        if (!checkOnly)
            for (size_t i = 0; i < rows; ++i)
                PushVector(out, m[i], cols);
        return true;
    }

    struct WideRead {
        const void* limb0 = nullptr;  // host address of the words of limb 0 handed out
        hiprt::Buf words;             // the wide tower's device words [k][limbs][N]
        uint32_t k = 1, limbs = 0;
    };
    static WideRead& LastWideRead() {
        static thread_local WideRead w;
        return w;
    }
    // ---- pending weighted sums (m_lazy) -------------------------------------------------------------------------------
    static bool LazySums() {
        static const bool on = !(std::getenv("FHE_HAL_LAZY_SUMS") && std::string(std::getenv("FHE_HAL_LAZY_SUMS")) == "0");
        return on && hiprt::Available();
    }
    static constexpr size_t kMaxLazyTerms = 64;
    uint64_t LimbModulus(uint32_t i) const {
        return m_h.GetParams()->GetParams()[i]->GetModulus().template ConvertToInt<uint64_t>();
    }
    // this tower as the terms of a sum: its pending terms, or its device words once
    // (the description is read through a local reference taken under the lock: another host thread holding the same const tower may
    // settle the sum — m_lazy.reset() — at any time)
    std::shared_ptr<const LazySum> PendingSum() const {
        std::lock_guard<std::mutex> lk(m_lock.m);
        return m_lazy;
    }
    bool TermsOf(std::vector<LazyTerm>* out, bool negate) const {
        const uint32_t L = NumLimbs();
        if (const auto sum = PendingSum()) {
            for (const auto& t : sum->terms) {
                LazyTerm c{t.words, std::vector<uint64_t>(L)};
                for (uint32_t i = 0; i < L; ++i)
                    c.k[i] = (negate && t.k[i]) ? LimbModulus(i) - t.k[i] : t.k[i];
                out->push_back(std::move(c));
            }
            return true;
        }
        if (m_zero)
            return true;  // (nothing to add)
        if (!Upload())
            return false;
        LazyTerm c{m_d, std::vector<uint64_t>(L)};
        for (uint32_t i = 0; i < L; ++i)
            c.k[i] = negate ? LimbModulus(i) - 1 : 1;
        out->push_back(std::move(c));
        return true;
    }
    void BecomeLazy(std::vector<LazyTerm>&& terms, uint32_t k) {
        std::lock_guard<std::mutex> lk(m_lock.m);
        auto sum   = std::make_shared<LazySum>();
        sum->terms = std::move(terms);
        auto P      = m_h.GetParams();
        m_h         = HostType(P, m_h.GetFormat(), false);
        m_d.reset();
        m_lazy      = std::move(sum);
        m_hostValid = false;
        m_zero      = false;
        m_k         = k;
    }
    // this *= k per limb, recorded (wide towers with device words or a pending sum only)
    bool ScaleLazily(const std::vector<NativeInteger>& k) {
        const uint32_t L = NumLimbs();
        const auto& P    = m_h.GetParams();
        if (!LazySums() || m_k == 1 || k.size() < L || !P || L != P->GetParams().size() || (!PendingSum() && !(m_d && !m_hostValid)))
            return false;
        hiprt::Resolved r;
        if (!ResolveSets(P->GetRingDimension(), {P}, &r))
            return false;
        std::vector<LazyTerm> terms;
        if (!TermsOf(&terms, false))
            return false;
        for (auto& t : terms)
            for (uint32_t i = 0; i < L; ++i) {
                const uint64_t q = LimbModulus(i);
                t.k[i] = (uint64_t)((unsigned __int128)t.k[i] * (k[i].ConvertToInt<uint64_t>() % q) % q);
            }
        BecomeLazy(std::move(terms), m_k);
        return true;
    }
    // this += / -= rhs recorded as terms, when either side is a pending sum (wide towers)
    bool AddLazily(const DCRTPolyType& rhs, bool minus) {
        // (only sums that join a pending sum are recorded: recording EVERY sum of wide towers was measured — 47.9 against 51.4
        // bootstraps/s, 43.1 against 38.3 GB of operands, profiles/r05_sweeps.md: a lone a += b settles at once and pays the
        // weighted-sum kernel for a plain addition)
        if (!LazySums() || (!PendingSum() && !rhs.PendingSum()) || !Compatible(rhs, false))
            return false;
        const uint32_t k = std::max(m_k, rhs.m_k);
        if (k == 1 || (m_k != k && !m_zero) || rhs.m_k != k)
            return false;  // (a narrow operand would have to be replicated: the plain path does that after the sums are computed)
        std::vector<LazyTerm> terms;
        if (!TermsOf(&terms, false) || !rhs.TermsOf(&terms, minus) || terms.empty() || terms.size() > kMaxLazyTerms)
            return false;
        BecomeLazy(std::move(terms), k);
        return true;
    }
    // the pending sum computed: one fhe_lincomb launch per 16 terms, every term read once (m_lock held by the caller)
    void SettleLocked() const {
        if (!m_lazy)
            return;
        const auto& P    = m_h.GetParams();
        const uint32_t L = NumLimbs();
        const size_t N   = P->GetRingDimension();
        hiprt::Resolved r;
        if (!ResolveSets(N, {P}, &r))
            OPENFHE_THROW("HIP backend: a pending weighted sum lost its device context");
        const auto sum = m_lazy;
        const size_t n = sum->terms.size();
        std::vector<uint64_t> consts(n * L);
        std::vector<const uint64_t*> ptrs(n);
        hiprt::Op op;
        for (size_t t = 0; t < n; ++t) {
            ptrs[t] = op.R(sum->terms[t].words);
            for (uint32_t i = 0; i < L; ++i)
                consts[t * L + i] = sum->terms[t].k[i];
        }
        auto d = hiprt::Alloc((size_t)m_k * L * N);
        hiprt::Check(hiprt::api().lincomb(r.ctx, op.W(d), ptrs.data(), consts.data(), (uint32_t)n, r.idx[0].data(), L, m_k, 0, op.s),
                     "DCRTPoly weighted sum");
        hiprt::CountDevice();
        m_d = std::move(d);
        m_lazy.reset();
        m_hostValid = false;
    }
    void Settle() const {
        if (!PendingSum())
            return;
        hiprt::MemberScope scope("WeightedSum");
        std::lock_guard<std::mutex> lk(m_lock.m);
        SettleLocked();
    }
    // ---- the two copies ---------------------------------------------------------------------------------------------
    void CopyFrom(const DCRTPolyType& e) {
        std::lock_guard<std::mutex> lk(e.m_lock.m);
        m_zero = e.m_zero;
        m_k    = e.m_k;
        m_d.reset();
        m_lazy = e.m_lazy;
        if (m_lazy) {  // (a copy of a pending sum is the same pending sum: the description is immutable and shared)
            m_h         = HostType(e.m_h.GetParams(), e.m_h.GetFormat(), false);
            m_hostValid = false;
            return;
        }
        if (e.m_d && e.m_h.GetParams() && e.m_h.GetAllElements().size() == e.m_h.GetParams()->GetParams().size()) {
            // a source with a device copy (even next to a valid mirror): copy-on-write, the mirror of the copy holds (params,
            // format) only.  Device words are never modified while shared: every writer goes through WriteTarget / Unshare
            m_d         = e.m_d;  // shared until one of the two is written (WriteTarget / Unshare)
            m_h         = HostType(e.m_h.GetParams(), e.m_h.GetFormat(), false);
            m_hostValid = false;
            return;
        }
        m_h         = e.m_h;
        m_hostValid = e.m_hostValid;
    }
    // where an in-place operation writes: the tower's own buffer, or a fresh one while the words are shared with a copy
    // Words that another tower may read: a buffer with several owners (copies), or a WINDOW of a larger buffer (a tower taken out of a
    // wide one without a copy, UnpackTower / PackWide below; evaluation keys inside their packed buffer) — the larger buffer's other
    // windows and the wide tower itself are other towers' words, so a window is never written in place.
    bool SharedWords() const {
        return m_d && (m_d.use_count() > 1 || m_d->parent);
    }
    hiprt::Buf WriteTarget() const {
        return SharedWords() ? hiprt::Alloc(Words()) : m_d;
    }
    // a private copy of shared words (operations that read-modify-write in place)
    void Unshare() {
        if (SharedWords()) {
            hiprt::Op op;
            auto d = hiprt::Alloc(Words());
            hiprt::D2D(op, op.W(d), op.R(m_d), Words() * 8, "DCRTPoly copy");
            m_d = std::move(d);
        }
    }
    // host words valid (fills the mirror from the device if needed).  The mirror object itself stays in place — only its limbs are
    // filled in — so that (params, format, limb count) can be read by other threads of a const tower while one of them synchronises
    void SyncHost(const char* who) const {
        std::lock_guard<std::mutex> lk(m_lock.m);
        if (m_lazy) {
            hiprt::MemberScope scope("WeightedSum");
            SettleLocked();
        }
        if (m_hostValid)
            return;
        const auto& P    = m_h.GetParams();
        const Format f   = m_h.GetFormat();
        const uint32_t L = (uint32_t)m_h.GetAllElements().size();
        const size_t N   = P->GetRingDimension();
        if (m_zero) {
            for (uint32_t i = 0; i < L; ++i)
                m_h.SetElementAtIndex(i, PolyType(P->GetParams()[i], f, true));
            m_hostValid = true;
            m_zero      = false;
            return;
        }
        std::vector<uint64_t> stage((size_t)L * N);
        {
            hiprt::Op op;
            hiprt::Check(hiprt::api().d2h(hiprt::AnyCtx(), stage.data(), op.R(m_d), stage.size() * 8, op.s), "DCRTPoly device -> host");
            op.HostSync();
        }
        hiprt::CountD2H(stage.size() * 8);
        hiprt::CountHostRead(who);
        for (uint32_t i = 0; i < L; ++i) {
            NativeVector v(N, P->GetParams()[i]->GetModulus());
            std::memcpy(&v[0], stage.data() + (size_t)i * N, N * 8);
            PolyType poly(P->GetParams()[i], f, false);
            poly.SetValues(std::move(v), f);
            m_h.SetElementAtIndex(i, std::move(poly));
        }
        m_hostValid = true;
    }
    const HostType& Hc(const char* who = __builtin_FUNCTION()) const {
        hiprt::TraceMember(who);
        SyncHost(who);
        return m_h;
    }
    // mutable host access: the device copy is stale afterwards.  hostData: the caller PRODUCES words on the host (an encoder or a sampler
    // filling limbs); while the tower has no device copy that is not a fall-back of anything and is counted apart
    HostType& Hm(const char* who = __builtin_FUNCTION(), bool hostData = false) {
        if (m_k > 1)
            OPENFHE_THROW(std::string("HIP backend: DCRTPoly::") + who + " on a wide tower (" + std::to_string(m_k) +
                          " ciphertexts in lockstep): the member has no wide device path and a wide tower has no host form");
        const bool hadDevice = m_d != nullptr;
        SyncHost(who);
        m_d.reset();
        hiprt::CountHost(who, RingOf(m_h), hostData && !hadDevice);
        return m_h;
    }
    // device words valid (uploads the mirror if needed); r.idx[0] = context limbs of this tower
    // members with a wide device path (they pass m_k as the library's batch) ask with OnDeviceWide
    bool OnDevice(hiprt::Resolved* r, const char* who = __builtin_FUNCTION()) const {
        if (m_k > 1)
            OPENFHE_THROW(std::string("HIP backend: DCRTPoly::") + who + " on a wide tower: the member has no wide device path");
        return OnDeviceWide(r, who);
    }
    bool OnDeviceWide(hiprt::Resolved* r, const char* who = __builtin_FUNCTION()) const {
        hiprt::TraceMember(who);
        const auto& P = m_h.GetParams();
        if (!P || NumLimbs() == 0 || NumLimbs() != P->GetParams().size())
            return false;
        if (!ResolveSets(P->GetRingDimension(), {P}, r))
            return false;
        return Upload();
    }
    bool Upload() const {
        std::lock_guard<std::mutex> lk(m_lock.m);
        if (m_lazy) {
            hiprt::MemberScope scope("WeightedSum");
            SettleLocked();
        }
        if (m_d)
            return true;
        const uint32_t L = NumLimbs();
        const size_t N   = m_h.GetParams()->GetRingDimension();
        if (m_zero) {
            hiprt::Op op;
            auto d = hiprt::Alloc((size_t)m_k * L * N);
            hiprt::Check(hiprt::api().memset_zero(hiprt::AnyCtx(), op.W(d), (size_t)m_k * L * N * 8, op.s), "DCRTPoly zero tower");
            m_d    = std::move(d);
            m_zero = false;
            return true;
        }
        for (const auto& e : m_h.GetAllElements())
            if (e.IsEmpty() || e.GetLength() != N)
                return false;  // an unfilled tower: leave it (and its exceptions) to the host code
        hiprt::Op op;
        auto d       = hiprt::Alloc((size_t)m_k * L * N);
        uint64_t* dp = op.W(d);
        for (uint32_t k = 0; k < m_k; ++k)  // (a wide tower whose words were read on the host as tower 0 — a zero accumulator: every tower gets them)
            for (uint32_t i = 0; i < L; ++i)
                hiprt::Check(hiprt::api().h2d(hiprt::AnyCtx(), dp + ((size_t)k * L + i) * N, &m_h.GetAllElements()[i].GetValues()[0], N * 8, op.s),
                             "DCRTPoly host -> device");
        op.HostSync();  // (the host vectors may change or go away as soon as this returns)
        hiprt::CountH2D((size_t)m_k * L * N * 8);
        m_d = std::move(d);
        return true;
    }
    // after a kernel wrote the device copy in place: the mirror keeps (params, format) only
    void DeviceIsNewer(Format f) {
        std::lock_guard<std::mutex> lk(m_lock.m);
        auto P      = m_h.GetParams();
        m_h         = HostType(P, f, false);
        m_hostValid = false;
    }

    // ---- device members ---------------------------------------------------------------------------------------------
    using BinFn = decltype(hiprt::Api::add);
    // both towers over the same limbs in the same format (Times: EVALUATION): everything else goes to the host code
    bool Compatible(const DCRTPolyType& rhs, bool evalOnly) const {
        const auto &A = m_h.GetParams(), &B = rhs.m_h.GetParams();
        if (!A || !B || A->GetRingDimension() != B->GetRingDimension() || m_h.GetFormat() != rhs.m_h.GetFormat())
            return false;
        if (evalOnly && m_h.GetFormat() != Format::EVALUATION)
            return false;
        const auto &a = A->GetParams(), &b = B->GetParams();
        if (a.size() != b.size() || a.size() != NumLimbs() || b.size() != rhs.NumLimbs())
            return false;
        if (A.get() == B.get())
            return true;
        for (size_t i = 0; i < a.size(); ++i)
            if (a[i]->GetModulus() != b[i]->GetModulus())
                return false;
        return true;
    }
    // The words of this tower replicated to width k ([k][limbs][N]): a plaintext or a constant tower that meets a wide ciphertext tower.
    // Remembered on the buffer (the encodings of bootstrapping's constants meet every wide ciphertext again).
    hiprt::Buf Widened(uint32_t k) const {
        if (m_k == k)
            return m_d;
        if (m_k != 1)
            OPENFHE_THROW("HIP backend: towers of different widths (" + std::to_string(m_k) + ", " + std::to_string(k) + ") in one operation");
        const std::vector<uint64_t> key{3, k, NumLimbs()};
        if (auto hit = hiprt::MemoFind(m_d, key))
            return hit;
        const size_t w = TowerWords();
        hiprt::Op op;
        auto d        = hiprt::Alloc((size_t)k * w);
        uint64_t* dst = op.W(d);
        for (uint32_t i = 0; i < k; ++i)
            hiprt::D2D(op, dst + (size_t)i * w, op.R(m_d), w * 8, "tower replicated for a wide operation");
        hiprt::MemoStore(m_d, key, d);
        return d;
    }
    bool Binary(const DCRTPolyType& rhs, BinFn fn, bool evalOnly, DCRTPolyType* out) const {
        hiprt::Resolved r;
        if (!Compatible(rhs, evalOnly) || !OnDeviceWide(&r) || !rhs.Upload())
            return false;
        const uint32_t k = std::max(m_k, rhs.m_k);
        const hiprt::Buf a = Widened(k), b = rhs.Widened(k);
        hiprt::Op op;
        auto d = hiprt::Alloc((size_t)k * TowerWords());
        hiprt::Check(fn(r.ctx, op.W(d), op.R(a), op.R(b), r.idx[0].data(), NumLimbs(), k, op.s), "DCRTPoly arithmetic");
        hiprt::CountDevice();
        *out = FromDevice(m_h.GetParams(), m_h.GetFormat(), std::move(d), k);
        return true;
    }
    bool BinaryInPlace(const DCRTPolyType& rhs, BinFn fn, bool evalOnly) {
        hiprt::Resolved r;
        if (!Compatible(rhs, evalOnly) || !OnDeviceWide(&r) || !rhs.Upload())
            return false;
        if (rhs.m_k > m_k) {  // (a narrow tower updated by a wide one becomes wide)
            m_d = Widened(rhs.m_k);
            m_k = rhs.m_k;
        }
        const hiprt::Buf b = rhs.Widened(m_k);
        hiprt::Op op;
        auto dst             = WriteTarget();
        const uint64_t* lhsP = op.R(m_d);  // (before W: an in-place target is both)
        const uint64_t* rhsP = op.R(b);
        hiprt::Check(fn(r.ctx, op.W(dst), lhsP, rhsP, r.idx[0].data(), NumLimbs(), m_k, op.s), "DCRTPoly arithmetic");
        m_d = std::move(dst);
        hiprt::CountDevice();
        DeviceIsNewer(m_h.GetFormat());
        return true;
    }
    // limb i plus / minus NativeInteger(k[i]) as a constant polynomial (PolyImpl::Plus / Minus(Integer), poly-impl.h:211-225)
    bool AddConstOnDevice(const std::vector<Integer>& k, bool minus, DCRTPolyType* out) const {
        hiprt::Resolved r;
        if (k.size() < NumLimbs() || !OnDeviceWide(&r))
            return false;
        std::vector<uint64_t> c(NumLimbs());
        for (uint32_t i = 0; i < NumLimbs(); ++i)
            c[i] = NativeInteger(k[i]).template ConvertToInt<uint64_t>();
        hiprt::Op op;
        auto d = hiprt::Alloc(Words());
        if (minus)
            hiprt::Check(hiprt::api().sub_const(r.ctx, op.W(d), op.R(m_d), c.data(), r.idx[0].data(), NumLimbs(), m_k, op.s), "DCRTPoly Minus(constants)");
        else
            hiprt::Check(hiprt::api().add_const(r.ctx, op.W(d), op.R(m_d), c.data(), r.idx[0].data(), NumLimbs(), m_k,
                                                m_h.GetFormat() == Format::COEFFICIENT ? 1 : 0, op.s),
                         "DCRTPoly Plus(constants)");
        hiprt::CountDevice();
        *out = FromDevice(m_h.GetParams(), m_h.GetFormat(), std::move(d), m_k);
        return true;
    }
    // += / -= one scalar on every limb (dcrtpoly-impl.h:383-427): Plus(Integer) touches coefficient 0 only in COEFFICIENT format
    // (poly-impl.h:211-218), -= subtracts from every word in both formats (poly.h:249-252)
    bool AddScalarInPlace(const NativeInteger& v, bool minus) {
        hiprt::Resolved r;
        if (!OnDeviceWide(&r))
            return false;
        std::vector<uint64_t> c(NumLimbs(), v.ConvertToInt<uint64_t>());
        hiprt::Op op;
        auto dst            = WriteTarget();
        const uint64_t* src = op.R(m_d);
        if (minus)
            hiprt::Check(hiprt::api().sub_const(r.ctx, op.W(dst), src, c.data(), r.idx[0].data(), NumLimbs(), m_k, op.s), "DCRTPoly -= scalar");
        else
            hiprt::Check(hiprt::api().add_const(r.ctx, op.W(dst), src, c.data(), r.idx[0].data(), NumLimbs(), m_k,
                                                m_h.GetFormat() == Format::COEFFICIENT ? 1 : 0, op.s),
                         "DCRTPoly += scalar");
        m_d = std::move(dst);
        hiprt::CountDevice();
        DeviceIsNewer(m_h.GetFormat());
        return true;
    }
    bool TimesConstInPlace(const std::vector<NativeInteger>& c) {
        hiprt::Resolved r;
        if (c.size() < NumLimbs() || !OnDeviceWide(&r))
            return false;
        std::vector<uint64_t> k(NumLimbs());
        for (uint32_t i = 0; i < NumLimbs(); ++i)
            k[i] = c[i].ConvertToInt<uint64_t>();
        hiprt::Op op;
        auto dst            = WriteTarget();
        const uint64_t* src = op.R(m_d);
        hiprt::Check(hiprt::api().mul_const(r.ctx, op.W(dst), src, k.data(), r.idx[0].data(), NumLimbs(), m_k, op.s),
                     "DCRTPoly Times(constants)");
        m_d = std::move(dst);
        hiprt::CountDevice();
        DeviceIsNewer(m_h.GetFormat());
        return true;
    }
    static void Flatten(const std::vector<std::vector<NativeInteger>>& m, size_t rows, size_t cols, bool transposed,
                        std::vector<uint64_t>& out) {
        out.resize(rows * cols);
        for (size_t i = 0; i < rows; ++i)
            for (size_t j = 0; j < cols; ++j)
                out[i * cols + j] = (transposed ? m[j][i] : m[i][j]).ConvertToInt<uint64_t>();
    }
    // ExpandCRTBasis / ExpandCRTBasisReverseOrder (dcrtpoly-impl.h:1086-1148): this (Q) -> [Q | P] (or [P | Q]) in resultFormat; the
    // EVALUATION form of the Q limbs is reused when the input has one, the P limbs come from the exact conversion of the
    // COEFFICIENT form and are transformed on their way into place
    bool ExpandOnDevice(const std::shared_ptr<Params>& paramsQP, const std::shared_ptr<Params>& paramsP,
                        const std::vector<NativeInteger>& QHatInvModq, const std::vector<NativeInteger>& QHatInvModqPrecon,
                        const std::vector<std::vector<NativeInteger>>& QHatModp, const std::vector<std::vector<NativeInteger>>& alphaQModp,
                        const std::vector<DoubleNativeInt>& modpBarrettMu, const std::vector<double>& qInv, Format resultFormat,
                        bool reverse) {
        const uint32_t sizeQ = NumLimbs(), sizeP = (uint32_t)paramsP->GetParams().size();
        if (!hiprt::Available() || paramsQP->GetParams().size() != sizeQ + sizeP)
            return false;
        const bool wasEval = m_h.GetFormat() == Format::EVALUATION;
        DCRTPolyType coeff(*this);
        if (wasEval)
            coeff.SwitchFormat();
        DCRTPolyType partP;
        if (!coeff.SwitchBasisOnDevice(coeff.m_h.GetParams(), paramsP, QHatInvModq, QHatModp, &alphaQModp, &qInv, &partP, /*transposed=*/true))
            return false;
        const bool toEval       = resultFormat == Format::EVALUATION;
        const DCRTPolyType& qsrc = (toEval && wasEval) ? *this : coeff;
        RowPiece pq{&qsrc, 0, sizeQ, toEval && !wasEval}, pp{&partP, 0, sizeP, toEval};
        DCRTPolyType out = reverse ? AssembleRows(paramsQP, resultFormat, {pp, pq}) : AssembleRows(paramsQP, resultFormat, {pq, pp});
        *this            = std::move(out);
        return true;
    }
    // ScaleAndRound / ApproxScaleAndRound, DCRTPoly -> DCRTPoly (dcrtpoly-impl.h:1470-1628): the caller's table [sizeO][sizeI+1]
    // and fractions [sizeI]; the output basis is the leading limbs of this tower iff the first moduli agree (:1527-1534)
    bool ScaleAndRoundOnDevice(const std::shared_ptr<Params>& paramsO, const std::vector<std::vector<NativeInteger>>& tab,
                               const std::vector<double>* frac, DCRTPolyType* out) const {
        const uint32_t sizeQP = NumLimbs(), sizeO = (uint32_t)paramsO->GetParams().size();
        if (hiprt::Available() && sizeO >= 1 && sizeQP == sizeO && tab.size() >= sizeO) {
            // no input limbs (the leveled technique at its top level, bfvrns-leveledshe.cpp): nu = 0.5, alpha = 0, every sum is
            // the single product x_j * tab[j][0]  (dcrtpoly-impl.h:1549-1567)
            std::vector<NativeInteger> c(sizeO);
            for (uint32_t j = 0; j < sizeO; ++j) {
                if (tab[j].empty())
                    return false;
                c[j] = tab[j][0];
            }
            DCRTPolyType scaled(*this);
            if (!scaled.TimesConstInPlace(c))
                return false;
            scaled.m_h = HostType(paramsO, m_h.GetFormat(), false);
            *out       = std::move(scaled);
            return true;
        }
        if (!hiprt::Available() || sizeO == 0 || sizeQP <= sizeO || sizeQP - sizeO >= hiprt::kMaxDeviceLimbs || sizeO > hiprt::kMaxDeviceLimbs)
            return false;
        const uint32_t sizeI = sizeQP - sizeO;
        if (tab.size() < sizeO || (frac && frac->size() < sizeI))
            return false;
        for (uint32_t j = 0; j < sizeO; ++j)
            if (tab[j].size() < sizeI + 1)
                return false;
        hiprt::Resolved r;
        const auto& mine = m_h.GetParams();
        if (!ResolveSets(mine->GetRingDimension(), {mine, paramsO}, &r) || !Upload())
            return false;
        std::vector<uint64_t> flat;
        Flatten(tab, sizeO, sizeI + 1, false, flat);
        fhe_sr_plan* plan      = hiprt::SrPlan(r.ctx, sizeI, r.idx[1], flat.data(), frac ? frac->data() : nullptr);
        const bool outputFirst = paramsO->GetParams()[0]->GetModulus() == mine->GetParams()[0]->GetModulus();
        const size_t N         = mine->GetRingDimension();
        hiprt::Op op;
        auto d = hiprt::Alloc((size_t)sizeO * N);
        hiprt::Check(hiprt::api().scale_and_round(plan, op.R(m_d), outputFirst ? 1 : 0, op.W(d), 1, op.s), "ScaleAndRound");
        hiprt::CountDevice();
        *out = FromDevice(paramsO, m_h.GetFormat(), std::move(d));
        return true;
    }
    // ScaleAndRound -> NativePoly modulo t, the two decryption overloads (dcrtpoly-impl.h:1190-1467 HPS with the caller's four tables;
    // :1631-1671 BEHZ with moduliQ / tgamma / two tables): N words come back over PCIe instead of the whole tower going out
    bool ScaleAndRoundNativeOnDevice(const NativeInteger& t, const std::vector<NativeInteger>* tabModt, const std::vector<NativeInteger>* tabBModt,
                                     const std::vector<double>* frac, const std::vector<double>* bfrac,
                                     const std::vector<NativeInteger>* moduliQ, const NativeInteger& tgamma,
                                     const std::vector<NativeInteger>* tgammaQHatModq, const std::vector<NativeInteger>* negInvqModtgamma,
                                     PolyType* out) const {
        hiprt::Resolved r;
        const uint32_t sizeQ = NumLimbs();
        if (m_h.GetFormat() != Format::COEFFICIENT || sizeQ == 0 || !OnDevice(&r))
            return false;
        const auto& P  = m_h.GetParams();
        const size_t N = P->GetRingDimension();
        std::vector<uint64_t> a(sizeQ), b(sizeQ);
        const uint64_t tt = t.ConvertToInt<uint64_t>();
        hiprt::Op op;
        auto d = hiprt::Alloc(N);
        if (moduliQ) {  // BEHZ
            if (moduliQ->size() < sizeQ || tgammaQHatModq->size() < sizeQ || negInvqModtgamma->size() < sizeQ)
                return false;
            for (uint32_t i = 0; i < sizeQ; ++i) {
                if ((*moduliQ)[i] != P->GetParams()[i]->GetModulus())
                    return false;  // (the reference multiplies modulo the caller's moduli: only the tower's own are the device's)
                a[i] = (*tgammaQHatModq)[i].ConvertToInt<uint64_t>();
                b[i] = (*negInvqModtgamma)[i].ConvertToInt<uint64_t>();
            }
            hiprt::Check(hiprt::api().scale_and_round_behz_decrypt(r.ctx, op.R(m_d), r.idx[0].data(), sizeQ, tgamma.ConvertToInt<uint64_t>(),
                                                                   a.data(), b.data(), 1, op.W(d), op.s),
                         "ScaleAndRound (BEHZ decryption)");
        }
        else {
            if (tabModt->size() < sizeQ || frac->size() < sizeQ)
                return false;
            const bool haveB = tabBModt->size() >= sizeQ && bfrac->size() >= sizeQ;
            for (uint32_t i = 0; i < sizeQ; ++i) {
                a[i] = (*tabModt)[i].ConvertToInt<uint64_t>();
                b[i] = haveB ? (*tabBModt)[i].ConvertToInt<uint64_t>() : 0;
            }
            if (hiprt::api().scale_and_round_native(r.ctx, op.R(m_d), r.idx[0].data(), sizeQ, tt, a.data(), haveB ? b.data() : nullptr, frac->data(),
                                                    haveB ? bfrac->data() : nullptr, 1, op.W(d), op.s) != FHE_OK)
                return false;  // (a branch of the reference that needs the split tables the caller did not provide)
        }
        typename PolyType::Vector coefficients(N, tt);
        hiprt::Check(hiprt::api().d2h(hiprt::AnyCtx(), &coefficients[0], op.R(d), N * 8, op.s), "ScaleAndRound result");
        op.HostSync();
        hiprt::CountD2H(N * 8);
        hiprt::CountDevice();
        // (:1458-1465: the root of unity is set to ONE, "as the calculation is expensive")
        PolyType result(std::make_shared<typename PolyType::Params>(P->GetCyclotomicOrder(), tt, 1));
        result.SetValues(std::move(coefficients), Format::COEFFICIENT);
        *out = std::move(result);
        return true;
    }
    // ScaleAndRoundPOverQ (dcrtpoly-impl.h:1674-1689): this over Q u {p} -> Q: x_i = (x_i - SwitchModulus(x_last -> q_i)) * [p^-1]_{q_i}; the
    // format is whatever the tower's is (the reference switches the last limb's modulus in that format, the callers use COEFFICIENT)
    bool POverQOnDevice(const std::shared_ptr<Params>& paramsQ, const std::vector<NativeInteger>& pInvModq) {
        const uint32_t L = NumLimbs();
        if (L < 2 || paramsQ->GetParams().size() != L - 1 || pInvModq.size() < L - 1 || m_h.GetFormat() != Format::COEFFICIENT)
            return false;
        hiprt::Resolved r;
        if (!OnDevice(&r))
            return false;
        const auto& mine = m_h.GetParams();
        for (uint32_t i = 0; i + 1 < L; ++i)
            if (mine->GetParams()[i]->GetModulus() != paramsQ->GetParams()[i]->GetModulus())
                return false;
        // fhe_scale_and_round_p_over_q derives [p^-1]_{q_i} itself: the caller's table must be that
        const uint64_t p = mine->GetParams()[L - 1]->GetModulus().template ConvertToInt<uint64_t>();
        for (uint32_t i = 0; i + 1 < L; ++i) {
            const uint64_t q = mine->GetParams()[i]->GetModulus().template ConvertToInt<uint64_t>();
            if ((unsigned __int128)(p % q) * pInvModq[i].ConvertToInt<uint64_t>() % q != 1)
                return false;
        }
        const size_t N = mine->GetRingDimension();
        hiprt::Op op;
        auto d = hiprt::Alloc((size_t)(L - 1) * N);
        hiprt::Check(hiprt::api().scale_and_round_p_over_q(r.ctx, op.R(m_d), r.idx[0].data(), L - 1, op.W(d), 1, op.s), "ScaleAndRoundPOverQ");
        hiprt::CountDevice();
        *this = FromDevice(paramsQ, Format::COEFFICIENT, std::move(d));
        return true;
    }
    // SetValuesModSwitch (dcrtpoly-impl.h:630-647): this (one limb, modulus `modulus`) = round(element (one limb) * modulus / q) in double
    bool ModSwitchOnDevice(const DCRTPolyType& element, const NativeInteger& modulus) {
        if (NumLimbs() != 1 || element.NumLimbs() != 1 || !m_h.GetParams() || !element.m_h.GetParams() ||
            m_h.GetParams()->GetRingDimension() != element.m_h.GetParams()->GetRingDimension() || m_h.GetFormat() != Format::COEFFICIENT ||
            m_h.GetParams()->GetParams().size() != 1 || m_h.GetParams()->GetParams()[0]->GetModulus() != modulus)  // (PolyImpl::SetValues throws)
            return false;
        DCRTPolyType input(element);  // (:637-638: a copy of the limb, in COEFFICIENT form)
        if (input.GetFormat() != Format::COEFFICIENT)
            input.SwitchFormat();
        hiprt::Resolved r;
        if (!input.IsDeviceResident() || !input.OnDevice(&r))
            return false;
        const size_t N      = m_h.GetParams()->GetRingDimension();
        const uint64_t from = element.m_h.GetParams()->GetParams()[0]->GetModulus().template ConvertToInt<uint64_t>();
        hiprt::Op op;
        auto d = hiprt::Alloc(N);
        hiprt::Check(hiprt::api().mod_switch_round(r.ctx, op.R(input.m_d), from, modulus.ConvertToInt<uint64_t>(), op.W(d), N, op.s),
                     "SetValuesModSwitch");
        hiprt::CountDevice();
        // (:646 m_vectors[0].SetValues(tmp, COEFFICIENT): the limb keeps its parameter object, its words now carry `modulus`; the tower's
        // format flag is not touched by the reference)
        m_d = std::move(d);
        DeviceIsNewer(m_h.GetFormat());
        return true;
    }
    // the BEHZ trio (dcrtpoly-impl.h:1694-1929); which: 0 = FastBaseConvqToBskMontgomery (this over Q -> Q u Bsk, EVALUATION),
    // 1 = FastRNSFloorq (in place on Q u Bsk, COEFFICIENT), 2 = FastBaseConvSK (Q u Bsk -> Q, COEFFICIENT).  `tabs` = the member's
    // table arguments flattened (hiprt::BehzPlan): the device plan computes with the caller's values.
    bool BehzOnDevice(int which, const std::shared_ptr<Params>& params, const std::vector<NativeInteger>& moduliQ,
                      const std::vector<NativeInteger>& moduliBsk, uint64_t t, const std::vector<uint64_t>& tabs) {
        const uint32_t numQ = (uint32_t)moduliQ.size(), numBsk = (uint32_t)moduliBsk.size();
        if (!hiprt::Available() || numQ == 0 || numBsk != numQ + 1)
            return false;
        const auto mine = m_h.GetParams();
        const size_t N  = mine->GetRingDimension();
        hiprt::Resolved r;
        std::shared_ptr<Params> qbsk;  // the parameter set Q u Bsk whose moduli must be the caller's moduliQ / moduliBsk
        if (which == 0) {              // params = paramsQBsk
            if (NumLimbs() != numQ || params->GetParams().size() != numQ + numBsk)
                return false;
            if (!ResolveSets(N, {params, mine}, &r) || !Upload())
                return false;
            for (uint32_t i = 0; i < numQ; ++i)
                if (r.idx[0][i] != r.idx[1][i])
                    return false;
            qbsk = params;
        }
        else {  // this tower is over Q u Bsk
            if (NumLimbs() != numQ + numBsk || m_h.GetFormat() != Format::COEFFICIENT)
                return false;
            if (!ResolveSets(N, {mine}, &r) || !Upload())
                return false;
            qbsk = mine;
        }
        for (uint32_t i = 0; i < numQ; ++i)
            if (qbsk->GetParams()[i]->GetModulus() != moduliQ[i])
                return false;
        for (uint32_t j = 0; j < numBsk; ++j)
            if (qbsk->GetParams()[numQ + j]->GetModulus() != moduliBsk[j])
                return false;
        std::vector<uint32_t> qIdx(r.idx[0].begin(), r.idx[0].begin() + numQ), bskIdx(r.idx[0].begin() + numQ, r.idx[0].end());
        fhe_behz* plan = hiprt::BehzPlan(r.ctx, qIdx, bskIdx, which, t, tabs);
        if (!plan)
            return false;
        const auto& A = hiprt::api();
        hiprt::Op op;
        if (which == 0) {
            const bool wasEval = m_h.GetFormat() == Format::EVALUATION;
            auto d             = hiprt::Alloc((size_t)(numQ + numBsk) * N);
            hiprt::D2D(op, op.W(d), op.R(m_d), (size_t)numQ * N * 8, "FastBaseConvqToBskMontgomery");
            const size_t wsB = A.behz_workspace_bytes(plan, 1);
            auto ws          = hiprt::Alloc(wsB / 8 + 1);
            hiprt::Check(A.behz_q_to_bsk(plan, d->p, wasEval ? 1 : 0, 1, op.W(ws, false), wsB, op.s), "FastBaseConvqToBskMontgomery");
            hiprt::CountDevice();
            *this = FromDevice(params, Format::EVALUATION, std::move(d));
            return true;
        }
        if (which == 1) {
            Unshare();
            hiprt::Check(A.behz_floorq(plan, op.W(m_d), 1, op.s), "FastRNSFloorq");
            hiprt::CountDevice();
            DeviceIsNewer(Format::COEFFICIENT);
            return true;
        }
        auto d = hiprt::Alloc((size_t)numQ * N);
        hiprt::Check(A.behz_conv_sk(plan, op.R(m_d), op.W(d), 1, op.s), "FastBaseConvSK");
        hiprt::CountDevice();
        *this = FromDevice(params, Format::COEFFICIENT, std::move(d));
        return true;
    }
    // ApproxSwitchCRTBasis (alpha == nullptr) / SwitchCRTBasis on the device; this tower over paramsQ -> *out over paramsP
    bool SwitchBasisOnDevice(const std::shared_ptr<Params>& paramsQ, const std::shared_ptr<Params>& paramsP,
                             const std::vector<NativeInteger>& QHatInvModq, const std::vector<std::vector<NativeInteger>>& QHatModp,
                             const std::vector<std::vector<NativeInteger>>* alpha, const std::vector<double>* qInv,
                             DCRTPolyType* out, bool transposed = false) const {
        // (:892: sizeQ = min(limbs of this tower, limbs of paramsQ) — the last digit of a lower level is shorter than its params)
        const uint32_t sizeQ = std::min<uint32_t>(NumLimbs(), (uint32_t)paramsQ->GetParams().size());
        const uint32_t sizeP = (uint32_t)paramsP->GetParams().size();
        if (sizeQ == 0 || sizeQ > hiprt::kMaxDeviceLimbs || sizeP == 0 || sizeP > hiprt::kMaxDeviceLimbs || sizeQ != NumLimbs() || QHatInvModq.size() < sizeQ)
            return false;
        if ((transposed ? QHatModp.size() < sizeP : QHatModp.size() < sizeQ))
            return false;
        for (uint32_t i = 0; i < (transposed ? sizeP : sizeQ); ++i)
            if (QHatModp[i].size() < (transposed ? sizeQ : sizeP))
                return false;
        if (alpha && (alpha->size() < sizeQ + 1 || qInv->size() < sizeQ))
            return false;
        hiprt::Resolved r;
        const auto& mine = m_h.GetParams();
        if (!ResolveSets(mine->GetRingDimension(), {mine, paramsP}, &r) || !Upload())
            return false;
        std::vector<uint64_t> hi(sizeQ), hm, al;
        for (uint32_t i = 0; i < sizeQ; ++i)
            hi[i] = QHatInvModq[i].ConvertToInt<uint64_t>();
        Flatten(QHatModp, sizeQ, sizeP, transposed, hm);
        if (alpha)
            Flatten(*alpha, sizeQ + 1, sizeP, false, al);
        fhe_conv* cv = hiprt::ConvPlan(r.ctx, r.idx[0], r.idx[1], hi.data(), hm.data(), alpha ? al.data() : nullptr,
                                       alpha ? qInv->data() : nullptr);
        const size_t N = mine->GetRingDimension();
        hiprt::Op op;
        auto d  = hiprt::Alloc((size_t)sizeP * N);
        auto fn = alpha ? hiprt::api().switch_basis_exact : hiprt::api().approx_switch_basis;
        hiprt::Check(fn(cv, op.R(m_d), sizeQ, 0, op.W(d), sizeP, 0, 1, op.s), "SwitchCRTBasis");
        hiprt::CountDevice();
        *out = FromDevice(paramsP, m_h.GetFormat(), std::move(d));
        return true;
    }
    // ApproxModUp (dcrtpoly-impl.h:935-963): this (Q) -> Q u P in EVALUATION
    bool ModUpOnDevice(const std::shared_ptr<Params>& paramsQ, const std::shared_ptr<Params>& paramsP,
                       const std::shared_ptr<Params>& paramsQP, const std::vector<NativeInteger>& QHatInvModq,
                       const std::vector<std::vector<NativeInteger>>& QHatModp) {
        const uint32_t sizeQ = NumLimbs(), sizeP = (uint32_t)paramsP->GetParams().size();
        if (paramsQP->GetParams().size() != sizeQ + sizeP)
            return false;
        const bool wasEval = m_h.GetFormat() == Format::EVALUATION;
        DCRTPolyType coeff(*this);
        if (wasEval)
            coeff.SwitchFormat();
        DCRTPolyType partP;
        if (!coeff.SwitchBasisOnDevice(paramsQ, paramsP, QHatInvModq, QHatModp, nullptr, nullptr, &partP) || !partP.IsDeviceResident())
            return false;
        hiprt::Resolved r;
        if (!ResolveSets(paramsQP->GetRingDimension(), {paramsQP, paramsP, m_h.GetParams()}, &r))
            return false;
        DCRTPolyType& qpart = wasEval ? *this : coeff;  // the EVALUATION copy of the Q limbs is kept when there is one
        if (!qpart.Upload())
            return false;
        const size_t N = paramsQP->GetRingDimension();
        hiprt::Op op;
        auto d       = hiprt::Alloc((size_t)(sizeQ + sizeP) * N);
        uint64_t* dp = op.W(d);
        hiprt::D2D(op, dp, op.R(qpart.m_d), (size_t)sizeQ * N * 8, "ApproxModUp");
        hiprt::D2D(op, dp + (size_t)sizeQ * N, op.R(partP.m_d), (size_t)sizeP * N * 8, "ApproxModUp");
        if (!wasEval)
            hiprt::Check(hiprt::api().ntt_fwd(r.ctx, dp, r.idx[2].data(), sizeQ, 1, op.s), "ApproxModUp");
        hiprt::Check(hiprt::api().ntt_fwd(r.ctx, dp + (size_t)sizeQ * N, r.idx[1].data(), sizeP, 1, op.s), "ApproxModUp");
        hiprt::CountDevice();
        *this = FromDevice(paramsQP, Format::EVALUATION, std::move(d));
        return true;
    }
    // ApproxModDown (dcrtpoly-impl.h:966-1005): this over Q_l u P -> *out over Q_l, EVALUATION (t > 0: BGV's factors)
    bool ModDownOnDevice(const std::shared_ptr<Params>& paramsQ, const std::shared_ptr<Params>& paramsP,
                         const std::vector<NativeInteger>& PInvModq, const std::vector<NativeInteger>& PHatInvModp,
                         const std::vector<std::vector<NativeInteger>>& PHatModq, const std::vector<NativeInteger>& tInvModp,
                         const NativeInteger& t, DCRTPolyType* out) const {
        const uint32_t sizeP = (uint32_t)paramsP->GetParams().size(), L = NumLimbs();
        if (L <= sizeP || sizeP > hiprt::kMaxDeviceLimbs || m_h.GetFormat() != Format::EVALUATION)
            return false;
        const bool bgv = t > NativeInteger(0);  // BGV: the P part times -t^-1 before, the switched part times t after the conversion
        if (bgv && tInvModp.size() < sizeP)
            return false;
        const uint32_t sizeQ = L - sizeP;
        if (sizeQ > paramsQ->GetParams().size() || PInvModq.size() < sizeQ || PHatInvModp.size() < sizeP || PHatModq.size() < sizeP)
            return false;
        for (uint32_t j = 0; j < sizeP; ++j)
            if (PHatModq[j].size() < sizeQ)
                return false;
        // the Q limbs of this tower are the first sizeQ limbs of paramsQ (:991-994 drops the others)
        const auto& mine = m_h.GetParams();
        for (uint32_t i = 0; i < sizeQ; ++i)
            if (mine->GetParams()[i]->GetModulus() != paramsQ->GetParams()[i]->GetModulus())
                return false;
        hiprt::Resolved r;
        if (!ResolveSets(mine->GetRingDimension(), {mine, paramsP}, &r) || !Upload())
            return false;
        std::vector<uint32_t> idxQ(r.idx[0].begin(), r.idx[0].begin() + sizeQ);
        const size_t N = mine->GetRingDimension();
        const auto& A  = hiprt::api();
        hiprt::Op op;
        const uint64_t* self = op.R(m_d);
        // P part to COEFFICIENT (:978-985)
        auto pcoef   = hiprt::Alloc((size_t)sizeP * N);
        uint64_t* pc = op.W(pcoef);
        hiprt::Check(A.ntt_inv_oop(r.ctx, self + (size_t)sizeQ * N, pc, r.idx[1].data(), sizeP, 1, op.s), "ApproxModDown");
        if (bgv) {  // :982-984
            std::vector<uint64_t> ti(sizeP);
            for (uint32_t j = 0; j < sizeP; ++j)
                ti[j] = tInvModp[j].ConvertToInt<uint64_t>();
            hiprt::Check(A.mul_const(r.ctx, pc, pc, ti.data(), r.idx[1].data(), sizeP, 1, op.s), "ApproxModDown");
        }
        // P -> Q_l (:987-988), with the reference's PHatInvModp / PHatModq tables
        std::vector<uint64_t> hi(sizeP), hm((size_t)sizeP * sizeQ);
        for (uint32_t j = 0; j < sizeP; ++j) {
            hi[j] = PHatInvModp[j].ConvertToInt<uint64_t>();
            for (uint32_t i = 0; i < sizeQ; ++i)
                hm[(size_t)j * sizeQ + i] = PHatModq[j][i].ConvertToInt<uint64_t>();
        }
        fhe_conv* cv = hiprt::ConvPlan(r.ctx, r.idx[1], idxQ, hi.data(), hm.data(), nullptr, nullptr);
        auto sw      = hiprt::Alloc((size_t)sizeQ * N);
        uint64_t* sp = op.W(sw);
        hiprt::Check(A.approx_switch_basis(cv, pc, sizeP, 0, sp, sizeQ, 0, 1, op.s), "ApproxModDown");
        if (bgv) {  // :998-1000
            std::vector<uint64_t> tq(sizeQ, t.ConvertToInt<uint64_t>());
            hiprt::Check(A.mul_const(r.ctx, sp, sp, tq.data(), idxQ.data(), sizeQ, 1, op.s), "ApproxModDown");
        }
        hiprt::Check(A.ntt_fwd(r.ctx, sp, idxQ.data(), sizeQ, 1, op.s), "ApproxModDown");  // :1001
        // (x_i - switched_i) * [P^-1]_{q_i}   (:1002)
        std::vector<uint64_t> pinv(sizeQ);
        for (uint32_t i = 0; i < sizeQ; ++i)
            pinv[i] = PInvModq[i].ConvertToInt<uint64_t>();
        hiprt::Check(A.sub(r.ctx, sp, self, sp, idxQ.data(), sizeQ, 1, op.s), "ApproxModDown");
        hiprt::Check(A.mul_const(r.ctx, sp, sp, pinv.data(), idxQ.data(), sizeQ, 1, op.s), "ApproxModDown");
        hiprt::CountDevice();
        // the result's params: paramsQ, shortened to sizeQ limbs like `ans.DropLastElements(diffQ)` does (:991-994)
        DCRTPolyType ans = FromDevice(paramsQ, Format::EVALUATION, std::move(sw));
        const uint32_t diffQ = (uint32_t)paramsQ->GetParams().size() - sizeQ;
        if (diffQ > 0)
            ans.DropLastElements(diffQ);
        *out = std::move(ans);
        return true;
    }
    // ModReduce (dcrtpoly-impl.h:736-755), BGV's modulus switch: delta = INTT(last limb) * (-t^-1 mod q_l); every remaining limb
    // x_i = (x_i + t * SwitchModulus(delta -> q_i) [to EVALUATION if the tower is]) * q_l^-1
    bool ModReduceOnDevice(const NativeInteger& t, const NativeInteger& negtInvModq, const std::vector<NativeInteger>& qlInvModq) {
        const uint32_t L = NumLimbs();
        if (L < 2 || qlInvModq.size() < L - 1)
            return false;
        hiprt::Resolved r;
        if (!OnDevice(&r))
            return false;
        const size_t N          = m_h.GetParams()->GetRingDimension();
        const uint32_t l        = L - 1;
        const uint32_t lastLimb = r.idx[0][l];
        const bool eval         = m_h.GetFormat() == Format::EVALUATION;
        std::vector<uint32_t> idx(r.idx[0].begin(), r.idx[0].begin() + l);
        std::vector<uint64_t> tq(l, t.ConvertToInt<uint64_t>()), qi(l);
        for (uint32_t i = 0; i < l; ++i)
            qi[i] = qlInvModq[i].ConvertToInt<uint64_t>();
        const uint64_t nt = negtInvModq.ConvertToInt<uint64_t>();
        const auto& A     = hiprt::api();
        hiprt::Op op;
        const uint64_t* self = op.R(m_d);
        auto delta           = hiprt::Alloc(N);
        auto tmp             = hiprt::Alloc((size_t)l * N);
        uint64_t *dl = op.W(delta), *tp = op.W(tmp);
        if (eval)
            hiprt::Check(A.ntt_inv_oop(r.ctx, self + (size_t)l * N, dl, &lastLimb, 1, 1, op.s), "ModReduce");
        else
            hiprt::D2D(op, dl, self + (size_t)l * N, N * 8, "ModReduce");
        hiprt::Check(A.mul_const(r.ctx, dl, dl, &nt, &lastLimb, 1, 1, op.s), "ModReduce");
        hiprt::Check(A.switch_modulus(r.ctx, tp, idx.data(), l, dl, 1, 0, lastLimb, 1, op.s), "ModReduce");
        if (eval)
            hiprt::Check(A.ntt_fwd(r.ctx, tp, idx.data(), l, 1, op.s), "ModReduce");
        hiprt::Check(A.mul_const(r.ctx, tp, tp, tq.data(), idx.data(), l, 1, op.s), "ModReduce");
        hiprt::Check(A.add(r.ctx, tp, self, tp, idx.data(), l, 1, op.s), "ModReduce");
        hiprt::Check(A.mul_const(r.ctx, tp, tp, qi.data(), idx.data(), l, 1, op.s), "ModReduce");
        m_d = std::move(tmp);
        hiprt::CountDevice();
        DeviceIsNewer(m_h.GetFormat());
        DropLastElement();
        return true;
    }
    // DropLastElementAndScale, EVALUATION format (dcrtpoly-impl.h:693-712)
    bool RescaleOnDevice(const std::vector<NativeInteger>& QlQlInvModqlDivqlModq, const std::vector<NativeInteger>& qlInvModq) {
        const uint32_t L = NumLimbs();
        if (L < 2 || m_h.GetFormat() != Format::EVALUATION || QlQlInvModqlDivqlModq.size() < L - 1 || qlInvModq.size() < L - 1)
            return false;
        hiprt::Resolved r;
        if (!OnDeviceWide(&r))
            return false;
        const size_t N     = m_h.GetParams()->GetRingDimension();
        const uint32_t l   = L - 1;
        std::vector<uint64_t> a(l), b(l);
        for (uint32_t i = 0; i < l; ++i) {
            a[i] = QlQlInvModqlDivqlModq[i].ConvertToInt<uint64_t>();
            b[i] = qlInvModq[i].ConvertToInt<uint64_t>();
        }
        // a clone of a tower that was rescaled with these tables before takes that result (pke's weighted sums rescale a fresh clone of
        // every power in every sum): the same words through the same member with the same tables
        const hiprt::Buf src = m_d;
        std::vector<uint64_t> memoKey;
        // (a clone = another TOWER holding these words: references held by windows of a packed wide buffer do not count — round 5 counted
        // them, so every wide buffer looked cloned, its results were remembered and a benchmark's next pass found them: round-5 advisor)
        if (!src->parent && src->Owners(src, /*src and m_d*/ 2) >= 1) {
            memoKey = RescaleMemoKey(r, L, a, b);
            if (auto hit = hiprt::MemoFind(src, memoKey)) {
                m_d = std::move(hit);
                DeviceIsNewer(Format::EVALUATION);
                DropLastMeta();
                return true;
            }
        }
        // the whole member is ONE library call with the caller's tables (fhe_rescale_limbs: :696-709; 4 launches on the rings of two
        // static passes: the INTT of the last limb, then SwitchModulus on the way into the forward transform and the `* qlInvModq +`
        // on its way out)
        const auto& A = hiprt::api();
        hiprt::Op op;
        const uint64_t* self = op.R(m_d);
        const size_t wsBytes = A.rescale_workspace_bytes(r.ctx, L, m_k);
        auto ws              = hiprt::Alloc(wsBytes / 8);
        auto tmp             = hiprt::Alloc((size_t)m_k * l * N);
        hiprt::Check(A.rescale_limbs(r.ctx, self, r.idx[0].data(), L, a.data(), b.data(), m_k, op.W(tmp), op.W(ws, false), wsBytes, op.s),
                     "DropLastElementAndScale");
        if (!memoKey.empty())
            hiprt::MemoStore(src, std::move(memoKey), tmp);
        m_d = std::move(tmp);
        hiprt::CountDevice();
        DeviceIsNewer(Format::EVALUATION);
        DropLastMeta();  // :698 (a narrow tower's device copy keeps its leading limbs; the result buffer already has the new height)
        return true;
    }
    // kind 0: uniform residues, 1: discrete Gaussian (Peikert's inversion, the reference's table), 2: uniform ternary; COEFFICIENT words on the
    // device, transformed there when EVALUATION is asked for
    bool SampleOnDevice(int kind, const std::shared_ptr<Params>& params, Format format, double sigma) {
        if (!hiprt::DeviceSamplerEnabled() || !params || params->GetParams().empty() || (kind == 1 && !(sigma > 1.000000001 && sigma < 300.0)))
            return false;
        hiprt::Resolved r;
        if (!ResolveSets(params->GetRingDimension(), {params}, &r))
            return false;
        hiprt::MemberScope scope("DeviceSampler");
        const size_t N   = params->GetRingDimension();
        const uint32_t L = (uint32_t)params->GetParams().size();
        uint64_t seed;
        uint32_t sid;
        hiprt::DeviceSamplerStream(&seed, &sid);
        {
            hiprt::Op op;
            auto d = hiprt::Alloc((size_t)L * N);
            const auto& A = hiprt::api();
            const fhe_status st = kind == 0   ? A.sample_uniform(r.ctx, op.W(d), r.idx[0].data(), L, 1, seed, sid, op.s)
                                  : kind == 1 ? A.sample_gaussian(r.ctx, op.W(d), r.idx[0].data(), L, 1, sigma, seed, sid, op.s)
                                              : A.sample_ternary(r.ctx, op.W(d), r.idx[0].data(), L, 1, seed, sid, op.s);
            hiprt::Check(st, "DeviceSampler");
            hiprt::CountDevice();
            m_h         = HostType(params, Format::COEFFICIENT, false);
            m_d         = std::move(d);
            m_hostValid = false;
            m_zero      = false;
            m_k         = 1;
        }
        if (format == Format::EVALUATION)
            SwitchFormat();
        return true;
    }
    bool ModRaiseOnDevice(const PolyType& e, const std::shared_ptr<Params>& params) {
        if (!hiprt::Available() || e.IsEmpty() || e.GetFormat() != Format::COEFFICIENT || params->GetParams().empty() ||
            e.GetModulus() != params->GetParams()[0]->GetModulus() || e.GetLength() != params->GetRingDimension())
            return false;
        hiprt::Resolved r;
        if (!ResolveSets(params->GetRingDimension(), {params}, &r))
            return false;
        const size_t N   = params->GetRingDimension();
        const uint32_t L = (uint32_t)params->GetParams().size();
        WideRead& wr     = LastWideRead();
        if (wr.words && wr.limb0 == &e.GetValues()[0]) {
            // limb 0 of a WIDE tower: every one of its k towers is raised from its own limb 0, which never left the device
            const uint32_t k = wr.k;
            hiprt::Op op;
            auto d = hiprt::Alloc((size_t)k * L * N);
            hiprt::Check(hiprt::api().switch_modulus(r.ctx, op.W(d), r.idx[0].data(), L, op.R(wr.words), wr.limbs, 0, r.idx[0][0], k, op.s), "ModRaise");
            hiprt::CountDevice();
            m_h         = HostType(params, Format::COEFFICIENT, false);
            m_d         = std::move(d);
            m_hostValid = false;
            m_zero      = false;
            m_k         = k;
            wr          = WideRead{};
            return true;
        }
        // (a host polynomial is the same for every tower of a wide evaluation — MultByMonomialInPlace's monomial: a narrow tower, replicated
        // when it meets a wide one)
        hiprt::Op op;
        auto src = hiprt::Alloc(N);
        auto d   = hiprt::Alloc((size_t)L * N);
        hiprt::Check(hiprt::api().h2d(r.ctx, op.W(src), &e.GetValues()[0], N * 8, op.s), "ModRaise");
        hiprt::Check(hiprt::api().switch_modulus(r.ctx, op.W(d), r.idx[0].data(), L, src->p, 1, 0, r.idx[0][0], 1, op.s), "ModRaise");
        op.HostSync();
        hiprt::CountH2D(N * 8);
        hiprt::CountDevice();
        m_h         = HostType(params, Format::COEFFICIENT, false);
        m_d         = std::move(d);
        m_hostValid = false;
        m_zero      = false;
        m_k         = 1;
        return true;
    }
    // ---- wide towers: K towers with equal (params, format) as one ([K][limbs][N]) and back --------------------------------------------
public:
    // Wide from birth (round 5): the towers of a lockstep group live as windows of ONE allocation.  UnpackTower hands out tower i as a
    // window of the wide buffer (no copy; the window keeps the buffer alive and is copied only if somebody writes it in place,
    // SharedWords); PackWide of K towers that are consecutive windows of one buffer is that buffer again (no copy), and when it has to
    // copy it leaves the K source towers behind as windows of the packed buffer, so that packing the same operands again — the next
    // operation of a pipeline, the next pass of a benchmark — costs nothing.  Values never change: only which allocation holds them.
    // Re-pointing the sources is OPT-IN (`adopt`, round-6 fix of a round-5 advisor finding): PackWide takes its sources as const, and a
    // reader of a shared const ciphertext on another host thread (OpenFHE treats const reads as thread-safe) must never see its buffer
    // swapped and the old one recycled under it.  Only a caller that OWNS the sources for the duration of the call — the lockstep drivers,
    // which pack disjoint groups of their own ciphertexts — passes adopt = true; and even then a source whose buffer is shared with a copy
    // (use_count > 1) is left alone.  FHE_HAL_WIDE_VIEWS=0 restores the copying forms everywhere.
    static DCRTPolyType PackWide(const std::vector<const DCRTPolyType*>& towers, bool adopt = false) {
        hiprt::MemberScope scope("PackWide");
        if (towers.empty())
            OPENFHE_THROW("PackWide: no towers");
        const DCRTPolyType& t0 = *towers[0];
        hiprt::Resolved r;
        if (!t0.OnDevice(&r))
            OPENFHE_THROW("PackWide: the towers cannot live on the device");
        const uint32_t k = (uint32_t)towers.size();
        const size_t w   = t0.TowerWords();
        for (uint32_t i = 0; i < k; ++i) {
            const DCRTPolyType& t = *towers[i];
            if (t.m_k != 1 || !t0.Compatible(t, false) || !t.Upload())
                OPENFHE_THROW("PackWide: towers of different shapes");
        }
        static const bool views = !(std::getenv("FHE_HAL_WIDE_VIEWS") && std::string(std::getenv("FHE_HAL_WIDE_VIEWS")) == "0");
        if (views) {  // already consecutive windows of one buffer?
            std::vector<hiprt::Buf> held(k);
            bool contiguous = true;
            for (uint32_t i = 0; i < k && contiguous; ++i) {
                std::lock_guard<std::mutex> lk(towers[i]->m_lock.m);
                held[i]    = towers[i]->m_d;
                contiguous = held[i] && held[i]->parent && held[i]->parent == held[0]->parent && held[i]->words == w &&
                             held[i]->p == held[0]->p + (size_t)i * w;
            }
            if (contiguous) {
                const auto& parent = held[0]->parent;
                const size_t off   = (size_t)(held[0]->p - parent->p);
                hiprt::CountDevice();
                auto d = (off == 0 && parent->words == (size_t)k * w && !parent->parent) ? parent : hiprt::View(parent, off, (size_t)k * w);
                return FromDevice(t0.m_h.GetParams(), t0.m_h.GetFormat(), std::move(d), k);
            }
        }
        hiprt::Op op;
        auto d        = hiprt::Alloc((size_t)k * w);
        uint64_t* dst = op.W(d);
        for (uint32_t i = 0; i < k; ++i)
            hiprt::D2D(op, dst + (size_t)i * w, op.R(towers[i]->m_d), w * 8, "towers packed into a wide one");
        if (views && adopt)
            for (uint32_t i = 0; i < k; ++i) {  // the sources become windows of the packed buffer (same words)
                std::lock_guard<std::mutex> lk(towers[i]->m_lock.m);
                if (towers[i]->m_d && !towers[i]->m_lazy && towers[i]->m_d.use_count() == 1)
                    towers[i]->m_d = hiprt::View(d, (size_t)i * w, w);
            }
        hiprt::CountDevice();
        return FromDevice(t0.m_h.GetParams(), t0.m_h.GetFormat(), std::move(d), k);
    }
    // A tower handed out by UnpackTower is a WINDOW of the wide buffer: it keeps the whole [K][limbs][N] allocation alive for as long as it
    // lives (a caller that keeps one output of a group of 16 pins 16x its size).  Detach() gives a long-lived output a buffer of its own
    // (one device-to-device copy); towers that are not windows are left as they are.  (Round-5 advisor.)
    void Detach() {
        hiprt::MemberScope scope("Detach");
        hiprt::Buf src;
        {
            std::lock_guard<std::mutex> lk(m_lock.m);
            src = m_d;
        }
        if (!src || !src->parent)
            return;
        hiprt::Op op;
        auto d = hiprt::Alloc(src->words);
        hiprt::D2D(op, op.W(d), op.R(src), src->words * 8, "a window detached from its wide buffer");
        hiprt::CountDevice();
        std::lock_guard<std::mutex> lk(m_lock.m);
        m_d = std::move(d);
    }
    DCRTPolyType UnpackTower(uint32_t i) const {
        hiprt::MemberScope scope("UnpackTower");
        hiprt::Resolved r;
        if (i >= m_k || !OnDeviceWide(&r))
            OPENFHE_THROW("UnpackTower: no such tower");
        const size_t w = TowerWords();
        static const bool views = !(std::getenv("FHE_HAL_WIDE_VIEWS") && std::string(std::getenv("FHE_HAL_WIDE_VIEWS")) == "0");
        if (views) {
            hiprt::Buf src;
            {
                std::lock_guard<std::mutex> lk(m_lock.m);
                src = m_d;
            }
            hiprt::CountDevice();
            return FromDevice(m_h.GetParams(), m_h.GetFormat(), hiprt::View(src, (size_t)i * w, w), 1);
        }
        hiprt::Op op;
        auto d = hiprt::Alloc(w);
        hiprt::D2D(op, op.W(d), op.R(m_d) + (size_t)i * w, w * 8, "tower taken out of a wide one");
        hiprt::CountDevice();
        return FromDevice(m_h.GetParams(), m_h.GetFormat(), std::move(d), 1);
    }
};

}  // namespace lbcrypto

#endif
