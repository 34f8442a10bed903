// keyswitch-hybrid-hip.cpp — the HIP backend's hook into HYBRID key switching, without touching pke.
//
// Three members of KeySwitchHYBRID are limb-by-limb loops over GetElementAtIndex / SetElementAtIndex, which DCRTPolyInterface
// can only serve from the host mirror: the digit decomposition of EvalKeySwitchPrecomputeCore (keyswitch-hybrid.cpp:314-379),
// the inner product of EvalFastKeySwitchCoreExt (:402-435) and KeySwitchExt (:217-243).  This file defines the same three
// members with the same arithmetic on whole device towers (DCRTPolyHipImpl::AssembleRows / MultAccRows; the same CRT tables
// through DCRTPoly::ApproxSwitchCRTBasis; exact modular sums, whose order is free).  The HIP build compiles the reference's
// own keyswitch-hybrid.cpp unmodified, marks its definitions of exactly these three symbols WEAK in the object file
// (objcopy --weaken-symbol, openfhe-development_amd/hal/Makefile) and links this file's strong ones: every caller — the
// reference's KeySwitchCore / EvalFastKeySwitchCore, base-leveledshe.cpp, ckksrns-fhe.cpp — reaches them through the vtable
// or the PLT.  In a checkout of openfhe-development the same effect is three `#ifdef WITH_HIP` lines in keyswitch-hybrid.cpp.
// tests/test_hal_shim.py compares every ciphertext limb of the result with the stock backend's.
#include "keyswitch/keyswitch-hybrid.h"

#include <cmath>
#include <memory>
#include <vector>

#include "ciphertext.h"
#include "key/evalkeyrelin.h"
#include "scheme/ckksrns/ckksrns-cryptoparameters.h"
#include "schemerns/rns-cryptoparameters.h"

namespace lbcrypto {

using RowPiece = DCRTPoly::RowPiece;

// KeySwitchHYBRID::KeySwitchExt (keyswitch-hybrid.cpp:217-243): element k over Q_l u P = [ c_k * [P]_{q_i} | 0 ]
Ciphertext<DCRTPoly> KeySwitchHYBRID::KeySwitchExt(ConstCiphertext<DCRTPoly> ciphertext, bool addFirst) const {
    const auto cryptoParams = std::dynamic_pointer_cast<CryptoParametersCKKSRNS>(ciphertext->GetCryptoParameters());
    const auto& cv          = ciphertext->GetElements();
    const auto& PModq       = cryptoParams->GetPModq();
    const auto paramsP      = cryptoParams->GetParamsP();
    const auto paramsQlP    = cv[0].GetExtendedCRTBasis(paramsP);
    const uint32_t sizeQl   = cv[0].GetParams()->GetParams().size();
    const uint32_t sizeP    = paramsP->GetParams().size();

    std::vector<DCRTPoly> elements(cv.size());
    for (uint32_t k = 0; k < cv.size(); ++k) {
        if (addFirst || k > 0) {
            const DCRTPoly cMult = cv[k].TimesNoCheck(PModq);
            elements[k]          = DCRTPoly::AssembleRows(paramsQlP, Format::EVALUATION, {RowPiece{&cMult, 0, sizeQl}, RowPiece{nullptr, 0, sizeP}});
        }
        else {
            elements[k] = DCRTPoly(paramsQlP, Format::EVALUATION, true);
        }
    }
    auto result = ciphertext->CloneEmpty();
    result->SetElements(std::move(elements));
    return result;
}

// KeySwitchHYBRID::EvalKeySwitchPrecomputeCore (keyswitch-hybrid.cpp:314-379): digit decomposition and ModUp of every digit
std::shared_ptr<std::vector<DCRTPoly>> KeySwitchHYBRID::EvalKeySwitchPrecomputeCore(
    const DCRTPoly& c, std::shared_ptr<CryptoParametersBase<DCRTPoly>> cryptoParamsBase) const {
    const auto cryptoParams = std::dynamic_pointer_cast<CryptoParametersRNS>(cryptoParamsBase);
    const auto paramsQl     = c.GetParams();
    const auto paramsP      = cryptoParams->GetParamsP();
    const auto paramsQlP    = c.GetExtendedCRTBasis(paramsP);
    const uint32_t sizeQl   = paramsQl->GetParams().size();
    const uint32_t sizeP    = paramsP->GetParams().size();
    const uint32_t alpha    = cryptoParams->GetNumPerPartQ();
    uint32_t numPartQl      = std::ceil(static_cast<double>(sizeQl) / alpha);  // :329-333
    if (numPartQl > cryptoParams->GetNumberOfQPartitions())
        numPartQl = cryptoParams->GetNumberOfQPartitions();

    auto result = std::make_shared<std::vector<DCRTPoly>>(numPartQl);
    for (uint32_t part = 0; part < numPartQl; ++part) {
        // the digit's parameter set: the precomputed one, shortened for the last digit of a lower level (:338-355)
        std::shared_ptr<ParmType> paramsPart = cryptoParams->GetParamsPartQ(part);
        if (part == numPartQl - 1) {
            const uint32_t sizePartQl = sizeQl - alpha * part;
            std::vector<NativeInteger> moduli(sizePartQl), roots(sizePartQl);
            for (uint32_t i = 0; i < sizePartQl; ++i) {
                moduli[i] = paramsPart->GetParams()[i]->GetModulus();
                roots[i]  = paramsPart->GetParams()[i]->GetRootOfUnity();
            }
            paramsPart = std::make_shared<ParmType>(paramsPart->GetCyclotomicOrder(), moduli, roots);
        }
        const uint32_t sizePartQl   = paramsPart->GetParams().size();
        const uint32_t startPartIdx = alpha * part, endPartIdx = startPartIdx + sizePartQl;

        // the digit's rows of c in COEFFICIENT format (:357-362: copy, then SetFormat)
        DCRTPoly partsCt = DCRTPoly::AssembleRows(paramsPart, Format::COEFFICIENT, {RowPiece{&c, startPartIdx, sizePartQl, true}});
        auto partsCtCompl = partsCt.ApproxSwitchCRTBasis(cryptoParams->GetParamsPartQ(part), cryptoParams->GetParamsComplPartQ(sizeQl - 1, part),
                                                         cryptoParams->GetPartQlHatInvModq(part, sizePartQl - 1),
                                                         cryptoParams->GetPartQlHatInvModqPrecon(part, sizePartQl - 1),
                                                         cryptoParams->GetPartQlHatModp(sizeQl - 1, part),
                                                         cryptoParams->GetmodComplPartqBarrettMu(sizeQl - 1, part));  // :363-368
        // [ complement rows below the digit | the digit's own rows of c | the remaining complement rows ]  (:369-378: the
        // complement goes to EVALUATION on its way into place)
        (*result)[part] = DCRTPoly::AssembleRows(paramsQlP, Format::EVALUATION,
                                                 {RowPiece{&partsCtCompl, 0, startPartIdx, true}, RowPiece{&c, startPartIdx, sizePartQl},
                                                  RowPiece{&partsCtCompl, startPartIdx, sizeQl + sizeP - endPartIdx, true}});
    }
    return result;
}

// KeySwitchHYBRID::EvalFastKeySwitchCoreExt (keyswitch-hybrid.cpp:402-435): both halves of sum_j digit_j * key_j over Q_l u P;
// key limb idx = i for i < sizeQl, i + (sizeQ - sizeQl) for the P limbs (:425)
std::shared_ptr<std::vector<DCRTPoly>> KeySwitchHYBRID::EvalFastKeySwitchCoreExt(
    const std::shared_ptr<std::vector<DCRTPoly>> digits, const EvalKey<DCRTPoly> evalKey,
    const std::shared_ptr<ParmType> paramsQl) const {
    const uint32_t sizeQl  = paramsQl->GetParams().size();
    auto&& cryptoParams    = std::dynamic_pointer_cast<CryptoParametersRNS>(evalKey->GetCryptoParameters());
    const uint32_t sizeQ   = cryptoParams->GetElementParams()->GetParams().size();
    const auto& av = evalKey->GetAVector();
    const auto& bv = evalKey->GetBVector();

    return std::make_shared<std::vector<DCRTPoly>>(DCRTPoly::InnerProduct(*digits, bv, av, sizeQl, sizeQ - sizeQl));
}

}  // namespace lbcrypto

// ---- LeveledSHECKKSRNS::EvalFastRotationExt (ckksrns-leveledshe.cpp:534-582), hooked the same way (weak symbol in the
// reference's object): its `psiC0.SetElementAtIndex(i, cMult.GetElementAtIndex(i))` loop (:563-568) becomes one AssembleRows ----
#include "cryptocontext.h"
#include "math/nbtheory.h"
#include "scheme/ckksrns/ckksrns-leveledshe.h"

namespace lbcrypto {

Ciphertext<DCRTPoly> LeveledSHECKKSRNS::EvalFastRotationExt(ConstCiphertext<DCRTPoly>& ciphertext, uint32_t index,
                                                            const std::shared_ptr<std::vector<DCRTPoly>> digits, bool addFirst,
                                                            const std::map<uint32_t, EvalKey<DCRTPoly>>& evalKeys) const {
    const auto cryptoParams  = std::dynamic_pointer_cast<CryptoParametersCKKSRNS>(ciphertext->GetCryptoParameters());
    const uint32_t M         = cryptoParams->GetElementParams()->GetCyclotomicOrder();
    const uint32_t autoIndex = FindAutomorphismIndex2nComplex(index, M);  // :545-546
    auto evalKeyIterator     = evalKeys.find(autoIndex);
    if (evalKeyIterator == evalKeys.end())
        OPENFHE_THROW("EvalKey for index [" + std::to_string(autoIndex) + "] is not found.");
    auto& evalKey = evalKeyIterator->second;

    const auto& cv      = ciphertext->GetElements();
    const auto paramsQl = cv[0].GetParams();
    const auto cc       = ciphertext->GetCryptoContext();
    auto cTilda         = *cc->GetScheme()->EvalFastKeySwitchCoreExt(digits, evalKey, paramsQl);  // :558

    if (addFirst) {  // :560-570: cTilda[0] += [ c0 * [P]_{q_i} | 0 ]
        const DCRTPoly cMult  = cv[0].TimesNoCheck(cryptoParams->GetPModq());
        const uint32_t sizeQl = paramsQl->GetParams().size();
        const uint32_t sizeP  = cTilda[0].GetParams()->GetParams().size() - sizeQl;
        cTilda[0] += DCRTPoly::AssembleRows(cTilda[0].GetParams(), Format::EVALUATION, {RowPiece{&cMult, 0, sizeQl}, RowPiece{nullptr, 0, sizeP}});
    }

    const uint32_t N = cryptoParams->GetElementParams()->GetRingDimension();
    std::vector<uint32_t> vec(N);
    PrecomputeAutoMap(N, autoIndex, &vec);  // :572-574
    cTilda[0] = cTilda[0].AutomorphismTransform(autoIndex, vec);
    cTilda[1] = cTilda[1].AutomorphismTransform(autoIndex, vec);

    auto result = ciphertext->CloneEmpty();
    result->SetElements(std::move(cTilda));
    return result;
}

// ---- scalar operations of CKKS: LeveledSHECKKSRNS::EvalAddInPlace / EvalSubInPlace(ciphertext, double)
// (ckksrns-leveledshe.cpp:60-68, :112-120) and EvalMultCoreInPlace(ciphertext, double) (:748-759), hooked the same way.  The
// reference reads the moduli from the LIMB OBJECTS of element 0 (GetElementForEvalAddOrSub :219-224, GetElementForEvalMult
// :445-449), which would pull the whole element over PCIe for its metadata, and adds the constants in a loop over host limbs.
// Here the reference's own GetElementFor* runs on a ciphertext that carries the same metadata and an element of the same
// shape without words, and the constants are applied to the device tower (DCRTPoly::Plus / Minus / Times(vector<Integer>) =
// the same per-limb NativeInteger arithmetic, dcrtpoly-impl.h:520-548, :572-580). ----
static ConstCiphertext<DCRTPoly> ShapeOf(const Ciphertext<DCRTPoly>& ciphertext) {
    const auto& e0 = ciphertext->GetElements()[0];
    auto shape     = ciphertext->CloneEmpty();
    std::vector<DCRTPoly> elements;
    elements.emplace_back(e0.GetParams(), e0.GetFormat(), false);
    shape->SetElements(std::move(elements));
    return shape;
}

void LeveledSHECKKSRNS::EvalAddInPlace(Ciphertext<DCRTPoly>& ciphertext, double operand) const {
    const auto elmnts = GetElementForEvalAddOrSub(ShapeOf(ciphertext), operand);
    auto& cv          = ciphertext->GetElements();
    cv[0]             = cv[0].Plus(elmnts);
}

void LeveledSHECKKSRNS::EvalSubInPlace(Ciphertext<DCRTPoly>& ciphertext, double operand) const {
    const auto elmnts = GetElementForEvalAddOrSub(ShapeOf(ciphertext), operand);
    auto& cv          = ciphertext->GetElements();
    cv[0]             = cv[0].Minus(elmnts);
}

// ckksrns-leveledshe.cpp:172-191: the rescale of both elements with the level's tables, one library call per level (the same four
// launches as one element's)
void LeveledSHECKKSRNS::ModReduceInternalInPlace(Ciphertext<DCRTPoly>& ciphertext, size_t levels) const {
    const auto cryptoParams = std::dynamic_pointer_cast<CryptoParametersCKKSRNS>(ciphertext->GetCryptoParameters());
    auto& cv                = ciphertext->GetElements();
    const size_t sizeQ = cryptoParams->GetElementParams()->GetParams().size(), sizeQl = cv[0].GetNumOfElements(), diffQl = sizeQ - sizeQl;
    ciphertext->SetNoiseScaleDeg(ciphertext->GetNoiseScaleDeg() - levels / cryptoParams->GetCompositeDegree());
    ciphertext->SetLevel(ciphertext->GetLevel() + levels);
    for (size_t i = 0; i < levels; ++i) {
        const auto& scale = cryptoParams->GetQlQlInvModqlDivqlModq(diffQl + i);
        const auto& inv   = cryptoParams->GetqlInvModq(diffQl + i);
        if (cv.size() != 2 || !DCRTPoly::PairRescaleInPlace(cv[0], cv[1], scale, inv))
            for (auto& dcrtpoly : cv)
                dcrtpoly.DropLastElementAndScale(scale, inv);
        ciphertext->SetScalingFactor(ciphertext->GetScalingFactor() / cryptoParams->GetModReduceFactor(sizeQl - 1 - i));
    }
}

void LeveledSHECKKSRNS::EvalMultCoreInPlace(Ciphertext<DCRTPoly>& ciphertext, double operand) const {
    const auto factors = GetElementForEvalMult(ShapeOf(ciphertext), operand);
    auto& cv           = ciphertext->GetElements();
    if (cv.size() != 2 || !DCRTPoly::PairTimesInPlace(cv[0], cv[1], factors))  // (both elements in one launch)
        for (uint32_t i = 0; i < cv.size(); ++i)
            cv[i] = cv[i] * factors;
    ciphertext->SetNoiseScaleDeg(ciphertext->GetNoiseScaleDeg() + 1);
    const auto cryptoParams = std::dynamic_pointer_cast<CryptoParametersCKKSRNS>(ciphertext->GetCryptoParameters());
    ciphertext->SetScalingFactor(ciphertext->GetScalingFactor() * cryptoParams->GetScalingFactorReal(ciphertext->GetLevel()));
}

}  // namespace lbcrypto

// ---- whole pke operations as ONE call of the device library --------------------------------------------------------------------
// Through the class surface an EvalMult is ~40 tower operations (each a launch or two, each with its own buffers); the device library
// has the same sequence as one composite over a key-switching plan (fhe_keyswitch_hybrid_acc: digit decomposition, ModUp of every
// digit, inner product with the key, both ModDowns, the two accumulations fused into the last kernel).  The hooks below route
//   LeveledSHEBase<DCRTPoly>::EvalMultCore / EvalSquareCore  (base-leveledshe.cpp:607-664)  -> one tensor kernel
//   LeveledSHEBase<DCRTPoly>::EvalMult(ct, ct, key) / EvalSquare(ct, key)  (:201-214, :281-293) -> + ONE composite key switch
// to it.  The composite derives its CRT tables from the moduli; the member-by-member path computes with the tables pke passes (the
// reference's CryptoParameters).  So the FIRST use of a composite at a level runs both and compares every word on the device
// (checksums): only a composite that reproduced the member-by-member result is used from then on (hiprt::DomainChecked).
// LeveledSHEBase's members are template instantiations (weak symbols in the reference's object): the explicit specialisations here
// are the strong definitions the library links.
#include "hip-hooks.h"
#include "schemebase/base-leveledshe.h"

// the reference's own KeySwitchCore.  Build of hal/Makefile on the UNMODIFIED sources: a second name objcopy gives the reference's
// definition in its object file; build on sources that carry integration/with_hip.patch (FHE_HIP_PATCHED_PKE): the member
// KeySwitchCoreReference the patch compiles that body under.
#ifdef FHE_HIP_PATCHED_PKE
#include "keyswitch/keyswitch-hybrid.h"
static inline std::shared_ptr<std::vector<lbcrypto::DCRTPoly>> fhe_ref_KeySwitchCore(const lbcrypto::KeySwitchHYBRID* self, const lbcrypto::DCRTPoly& a,
                                                                                     const lbcrypto::EvalKey<lbcrypto::DCRTPoly> evalKey) {
    return self->KeySwitchCoreReference(a, evalKey);
}
#else
extern "C" std::shared_ptr<std::vector<lbcrypto::DCRTPoly>> fhe_ref_KeySwitchCore(const lbcrypto::KeySwitchHYBRID* self, const lbcrypto::DCRTPoly& a,
                                                                                 const lbcrypto::EvalKey<lbcrypto::DCRTPoly> evalKey);
#endif

namespace lbcrypto {
namespace {
void LimbsOfParams(const std::shared_ptr<DCRTPoly::Params>& p, std::vector<uint64_t>& q, std::vector<uint64_t>& psi) {
    const auto& v = p->GetParams();
    q.resize(v.size()), psi.resize(v.size());
    for (size_t i = 0; i < v.size(); ++i) {
        q[i]   = v[i]->GetModulus().ConvertToInt<uint64_t>();
        psi[i] = v[i]->GetRootOfUnity().ConvertToInt<uint64_t>();
    }
}
// the key-switching domain of a parameter set, and the level of a tower in it (0: the tower is not a prefix of Q)
std::shared_ptr<hiprt::KsDomain> DomainOf(const std::shared_ptr<CryptoParametersRNS>& cp, const DCRTPoly& c, uint32_t* sizeQl) {
    // (a noise scale other than 1 — BGV — makes KeySwitchHYBRID pass the plaintext modulus to ApproxModDown, keyswitch-hybrid.cpp:385-398:
    // the composite is the t = 0 form)
    if (!hiprt::Available() || !cp || cp->GetKeySwitchTechnique() != HYBRID || cp->GetNoiseScale() != 1)
        return nullptr;
    std::vector<uint64_t> q, psiQ, p, psiP, ql, psiQl;
    LimbsOfParams(cp->GetElementParams(), q, psiQ);
    LimbsOfParams(cp->GetParamsP(), p, psiP);
    LimbsOfParams(c.GetParams(), ql, psiQl);
    if (ql.empty() || ql.size() > q.size() || !std::equal(ql.begin(), ql.end(), q.begin()) || !std::equal(psiQl.begin(), psiQl.end(), psiQ.begin()))
        return nullptr;
    const uint32_t numPartQ = cp->GetNumPartQ();
    if (numPartQ == 0 || cp->GetNumPerPartQ() != (q.size() + numPartQ - 1) / numPartQ)
        return nullptr;  // (a digit partition the plan does not derive the same way)
    *sizeQl = (uint32_t)ql.size();
    return hiprt::GetKsDomain(c.GetParams()->GetRingDimension(), hiprt::LimbSet{q.data(), psiQ.data(), (uint32_t)q.size()},
                              hiprt::LimbSet{p.data(), psiP.data(), (uint32_t)p.size()}, numPartQ);
}
bool KeyBuffers(const EvalKey<DCRTPoly>& evalKey, size_t numPartQ, size_t limbs, std::vector<hiprt::Buf>& b, std::vector<hiprt::Buf>& a) {
    const auto &bv = evalKey->GetBVector(), &av = evalKey->GetAVector();
    if (bv.size() != numPartQ || av.size() != numPartQ)
        return false;
    for (size_t j = 0; j < numPartQ; ++j) {
        if (bv[j].GetNumOfElements() != limbs || av[j].GetNumOfElements() != limbs || bv[j].GetFormat() != Format::EVALUATION ||
            av[j].GetFormat() != Format::EVALUATION)
            return false;
        b.push_back(bv[j].DeviceWords());
        a.push_back(av[j].DeviceWords());
        if (!b.back() || !a.back())
            return false;
    }
    return true;
}
// acc0 += ks0(c), acc1 += ks1(c) in one call; false: the composite cannot take these towers
bool CompositeKeySwitchAcc(hiprt::KsDomain& dom, uint32_t sizeQl, const std::vector<hiprt::Buf>& kb, const std::vector<hiprt::Buf>& ka, DCRTPoly& acc0,
                           DCRTPoly& acc1, const DCRTPoly& c) {
    const uint32_t width = c.Width();  // (a wide ciphertext: its K towers are the composite's batch)
    if (acc0.Width() != width || acc1.Width() != width)
        return false;
    auto b0 = acc0.DeviceWordsForUpdate(), b1 = acc1.DeviceWordsForUpdate();
    auto bc = c.DeviceWords();
    if (!b0 || !b1 || !bc)
        return false;
    const auto& A = hiprt::api();
    hiprt::Op op;
    hiprt::PackedKey pk = hiprt::DomainKey(dom, kb, ka, op);
    if (!pk.key)
        return false;
    const size_t wsB = A.ks_workspace_bytes(hiprt::DomainPlan(dom), sizeQl, width);
    auto ws          = hiprt::Alloc(wsB / 8 + 1);
    op.R(pk.b), op.R(pk.a);
    hiprt::Check(A.keyswitch_hybrid_acc(hiprt::DomainPlan(dom), pk.key.get(), op.R(bc), sizeQl, width, op.W(b0), op.W(b1), op.W(ws, false), wsB, op.s),
                 "EvalMult: composite key switch");
    hiprt::CountDevice("EvalMult.KeySwitchAccumulate");
    hiprt::CountComposite();
    return true;
}
[[noreturn]] void WideNeedsCheckedComposite(const char* what) {
    OPENFHE_THROW(std::string("HIP backend: a wide evaluation (several ciphertexts in lockstep) reached ") + what +
                  " at a level whose composite has not been checked against the member-by-member path yet: evaluate one ciphertext of the "
                  "same shape on its own first");
}
// cv[0] += ks0(cv[2]); cv[1] += ks1(cv[2])  (base-leveledshe.cpp:207-211) — composite when it applies and has been checked at this level
void KeySwitchAccumulate(const Ciphertext<DCRTPoly>& ciphertext, const EvalKey<DCRTPoly>& evalKey) {
    auto& cv = ciphertext->GetElements();
    uint32_t sizeQl = 0;
    std::shared_ptr<hiprt::KsDomain> dom;
    std::vector<hiprt::Buf> kb, ka;
    const auto cp = std::dynamic_pointer_cast<CryptoParametersRNS>(evalKey->GetCryptoParameters());
    bool usable   = cv.size() == 3 && cv[0].GetFormat() == Format::EVALUATION && cv[1].GetFormat() == Format::EVALUATION &&
                  cv[2].GetFormat() == Format::EVALUATION && cv[0].GetNumOfElements() == cv[2].GetNumOfElements() &&
                  cv[1].GetNumOfElements() == cv[2].GetNumOfElements();
    if (usable)
        dom = DomainOf(cp, cv[2], &sizeQl);
    usable = usable && dom && hiprt::DomainChecked(*dom, hiprt::kKeySwitchAcc, sizeQl) != 2 &&
             KeyBuffers(evalKey, cp->GetNumPartQ(), cp->GetParamsQP()->GetParams().size(), kb, ka);
    if (usable && hiprt::DomainChecked(*dom, hiprt::kKeySwitchAcc, sizeQl) == 1 && CompositeKeySwitchAcc(*dom, sizeQl, kb, ka, cv[0], cv[1], cv[2]))
        return;
    if (cv[2].Width() > 1)
        WideNeedsCheckedComposite("EvalMult's key switch");
    // the reference's lines (member by member, with the reference's tables) ...
    DCRTPoly try0, try1;
    bool tried = false;
    if (usable) {  // ... and, at the first use of this level, the composite on copies, compared word for word on the device
        try0 = cv[0], try1 = cv[1];
        tried = CompositeKeySwitchAcc(*dom, sizeQl, kb, ka, try0, try1, cv[2]);
    }
    // (KeySwitchCore itself is a composite too, below: the check needs the reference's own sequence)
    const auto scheme = ciphertext->GetCryptoContext()->GetScheme();
    auto ab = usable ? scheme->EvalFastKeySwitchCore(scheme->EvalKeySwitchPrecomputeCore(cv[2], evalKey->GetCryptoParameters()), evalKey,
                                                     cv[2].GetParams())  // (= KeySwitchHYBRID::KeySwitchCore, keyswitch-hybrid.cpp:308-312)
                     : scheme->KeySwitchCore(cv[2], evalKey);
    cv[0] += (*ab)[0];
    cv[1] += (*ab)[1];
    if (tried) {
        auto r0 = cv[0].DeviceWords(), r1 = cv[1].DeviceWords(), t0 = try0.DeviceWords(), t1 = try1.DeviceWords();
        const bool same = r0 && r1 && t0 && t1 && hiprt::Checksums(hiprt::DomainCtx(*dom), r0, sizeQl) == hiprt::Checksums(hiprt::DomainCtx(*dom), t0, sizeQl) &&
                          hiprt::Checksums(hiprt::DomainCtx(*dom), r1, sizeQl) == hiprt::Checksums(hiprt::DomainCtx(*dom), t1, sizeQl);
        hiprt::DomainSetChecked(*dom, hiprt::kKeySwitchAcc, sizeQl, same);
    }
}
}  // namespace

// KeySwitchHYBRID::KeySwitchCore (keyswitch-hybrid.cpp:308-312) — every relinearisation, rotation and conjugation of pke ends here — as one
// composite (fhe_keyswitch_hybrid); the reference's definition (precompute + fast key switch, member by member) is fhe_ref_KeySwitchCore
std::shared_ptr<std::vector<DCRTPoly>> KeySwitchHYBRID::KeySwitchCore(const DCRTPoly& a, const EvalKey<DCRTPoly> evalKey) const {
    hiprt::MemberScope scope("KeySwitchCore");
    uint32_t sizeQl = 0;
    std::vector<hiprt::Buf> kb, ka;
    const auto cp = std::dynamic_pointer_cast<CryptoParametersRNS>(evalKey->GetCryptoParameters());
    auto dom      = a.GetFormat() == Format::EVALUATION ? DomainOf(cp, a, &sizeQl) : nullptr;
    const int st  = dom ? hiprt::DomainChecked(*dom, hiprt::kKeySwitch, sizeQl) : 2;
    static const bool dbg = std::getenv("FHE_HAL_DEBUG") != nullptr;
    if (st == 2 || !KeyBuffers(evalKey, cp->GetNumPartQ(), cp->GetParamsQP()->GetParams().size(), kb, ka)) {
        if (dbg)
            fprintf(stderr, "hal debug: KeySwitchCore runs the reference's sequence: domain %p level %u state %d format %d\n", (void*)dom.get(), sizeQl, st,
                    (int)a.GetFormat());
        return fhe_ref_KeySwitchCore(this, a, evalKey);
    }
    auto bc = a.DeviceWords();
    if (!bc)
        return fhe_ref_KeySwitchCore(this, a, evalKey);
    if (dbg && st == 0)
        fprintf(stderr, "hal debug: KeySwitchCore first use at level %u\n", sizeQl);
    std::shared_ptr<std::vector<DCRTPoly>> mine;
    {
        const auto& A = hiprt::api();
        hiprt::Op op;
        hiprt::PackedKey pk = hiprt::DomainKey(*dom, kb, ka, op);
        if (!pk.key)
            return fhe_ref_KeySwitchCore(this, a, evalKey);
        const size_t N       = a.GetParams()->GetRingDimension();
        const uint32_t width = a.Width();  // (a wide ciphertext: its K towers are the composite's batch)
        const size_t wsB     = A.ks_workspace_bytes(hiprt::DomainPlan(*dom), sizeQl, width);
        auto ws = hiprt::Alloc(wsB / 8 + 1), o0 = hiprt::Alloc((size_t)width * sizeQl * N), o1 = hiprt::Alloc((size_t)width * sizeQl * N);
        op.R(pk.b), op.R(pk.a);
        hiprt::Check(A.keyswitch_hybrid(hiprt::DomainPlan(*dom), pk.key.get(), op.R(bc), sizeQl, width, op.W(o0), op.W(o1), op.W(ws, false), wsB, op.s),
                     "KeySwitchCore: composite key switch");
        hiprt::CountDevice("KeySwitchCore");
        hiprt::CountComposite();
        mine = std::make_shared<std::vector<DCRTPoly>>();
        mine->push_back(DCRTPoly::FromDeviceWords(a.GetParams(), Format::EVALUATION, std::move(o0), width));
        mine->push_back(DCRTPoly::FromDeviceWords(a.GetParams(), Format::EVALUATION, std::move(o1), width));
    }
    if (st == 1)
        return mine;
    if (a.Width() > 1)
        WideNeedsCheckedComposite("KeySwitchCore");
    auto ref = fhe_ref_KeySwitchCore(this, a, evalKey);  // first use at this level: both, compared word for word on the device
    auto r0 = (*ref)[0].DeviceWords(), r1 = (*ref)[1].DeviceWords();
    const bool same = r0 && r1 && (*ref)[0].GetNumOfElements() == sizeQl &&
                      hiprt::Checksums(hiprt::DomainCtx(*dom), r0, sizeQl) == hiprt::Checksums(hiprt::DomainCtx(*dom), (*mine)[0].DeviceWords(), sizeQl) &&
                      hiprt::Checksums(hiprt::DomainCtx(*dom), r1, sizeQl) == hiprt::Checksums(hiprt::DomainCtx(*dom), (*mine)[1].DeviceWords(), sizeQl);
    hiprt::DomainSetChecked(*dom, hiprt::kKeySwitch, sizeQl, same);
    return ref;
}

// base-leveledshe.cpp:562-579, :589-606: += / -= of the elements, both elements of a two-element pair in one launch
template <>
void LeveledSHEBase<DCRTPoly>::EvalAddCoreInPlace(Ciphertext<DCRTPoly>& ciphertext1, ConstCiphertext<DCRTPoly>& ciphertext2) const {
    VerifyNumOfTowers(ciphertext1, ciphertext2);
    auto& cv1       = ciphertext1->GetElements();
    const auto& cv2 = ciphertext2->GetElements();
    if (cv1.size() == 2 && cv2.size() == 2 && DCRTPoly::PairAddInPlace(cv1[0], cv1[1], cv2[0], cv2[1], false))
        return;
    const uint32_t c1Size = cv1.size(), c2Size = cv2.size(), cSmallSize = std::min(c1Size, c2Size);
    cv1.reserve(c2Size);
    uint32_t i = 0;
    for (; i < cSmallSize; ++i)
        cv1[i] += cv2[i];
    for (; i < c2Size; ++i)
        cv1.emplace_back(cv2[i]);
}
template <>
void LeveledSHEBase<DCRTPoly>::EvalSubCoreInPlace(Ciphertext<DCRTPoly>& ciphertext1, ConstCiphertext<DCRTPoly>& ciphertext2) const {
    VerifyNumOfTowers(ciphertext1, ciphertext2);
    auto& cv1       = ciphertext1->GetElements();
    const auto& cv2 = ciphertext2->GetElements();
    if (cv1.size() == 2 && cv2.size() == 2 && DCRTPoly::PairAddInPlace(cv1[0], cv1[1], cv2[0], cv2[1], true))
        return;
    const uint32_t c1Size = cv1.size(), c2Size = cv2.size(), cSmallSize = std::min(c1Size, c2Size);
    cv1.reserve(c2Size);
    uint32_t i = 0;
    for (; i < cSmallSize; ++i)
        cv1[i] -= cv2[i];
    for (; i < c2Size; ++i)
        cv1.emplace_back(cv2[i].Negate());
}
// base-leveledshe.cpp:201-214
template <>
Ciphertext<DCRTPoly> LeveledSHEBase<DCRTPoly>::EvalMult(ConstCiphertext<DCRTPoly>& ciphertext1, ConstCiphertext<DCRTPoly>& ciphertext2,
                                                        const EvalKey<DCRTPoly> evalKey) const {
    auto ciphertext = EvalMult(ciphertext1, ciphertext2);
    hiprt::MemberScope scope("EvalMult.KeySwitchAccumulate");
    KeySwitchAccumulate(ciphertext, evalKey);
    ciphertext->GetElements().resize(2);
    return ciphertext;
}
// base-leveledshe.cpp:281-293
template <>
Ciphertext<DCRTPoly> LeveledSHEBase<DCRTPoly>::EvalSquare(ConstCiphertext<DCRTPoly>& ciphertext, const EvalKey<DCRTPoly> evalKey) const {
    auto csquare = EvalSquare(ciphertext);
    hiprt::MemberScope scope("EvalMult.KeySwitchAccumulate");
    KeySwitchAccumulate(csquare, evalKey);
    csquare->GetElements().resize(2);
    return csquare;
}

// base-leveledshe.cpp:607-644: the product of two 2-element ciphertexts as one tensor kernel; every other shape as in the reference
template <>
Ciphertext<DCRTPoly> LeveledSHEBase<DCRTPoly>::EvalMultCore(ConstCiphertext<DCRTPoly>& ctxt1, ConstCiphertext<DCRTPoly>& ctxt2) const {
    VerifyNumOfTowers(ctxt1, ctxt2);
    auto& cv1 = ctxt1->GetElements();
    auto& cv2 = ctxt2->GetElements();
    const uint32_t n1 = cv1.size(), n2 = cv2.size(), nr = n1 + n2 - 1;
    std::vector<DCRTPoly> cvr;
    if (n1 == 2 && n2 == 2)
        cvr = DCRTPoly::Tensor(cv1[0], cv1[1], &cv2[0], &cv2[1]);
    if (cvr.empty()) {
        cvr.reserve(nr);
        for (uint32_t k = 0; k < nr; ++k) {  // element k = sum over i + j = k of cv1[i] * cv2[j]  (:619-638)
            for (uint32_t i = 0; i < n1; ++i) {
                if (k < i || k - i >= n2)
                    continue;
                if (cvr.size() == k)
                    cvr.emplace_back(cv1[i] * cv2[k - i]);
                else
                    cvr[k] += (cv1[i] * cv2[k - i]);
            }
        }
    }
    auto result = ctxt1->CloneEmpty();
    result->SetElements(std::move(cvr));
    result->SetNoiseScaleDeg(ctxt1->GetNoiseScaleDeg() + ctxt2->GetNoiseScaleDeg());
    result->SetScalingFactor(ctxt1->GetScalingFactor() * ctxt2->GetScalingFactor());
    result->SetScalingFactorInt(
        ctxt1->GetScalingFactorInt().ModMul(ctxt2->GetScalingFactorInt(), ctxt1->GetCryptoParameters()->GetPlaintextModulus()));
    return result;
}

}  // namespace lbcrypto

// ---- KeySwitchHYBRID::KeySwitchGenInternal (keyswitch-hybrid.cpp:51-195), hooked like the members above (weak symbols in the reference's
// object).  The reference assembles every key element limb by limb on the host (2 * numPartQ * (sizeQ + sizeP) SetElementAtIndex per
// key; a bootstrapping key set is 60-70 keys).  Here the samplers still run on the host, in the reference's order (a, then e, per digit:
// the deterministic test PRNG gives the same words), and the arithmetic runs on whole device towers:
//   sNewExt = [ sNew (EVALUATION) | SwitchModulus(sNew limb 0 -> p_j) to EVALUATION ]                         (:60-81)
//   b_part  = -a * sNewExt + ns * e + [ PModq_i * sOld_i on the limbs of digit `part`, 0 elsewhere ]            (:100-118)
// (exact modular sums and products: any order gives the reference's residues).
namespace lbcrypto {
namespace {
// [ 0 ... 0 | rows [first, first+n) of src | 0 ... 0 ] over paramsQP
DCRTPoly RowsInZeros(const std::shared_ptr<DCRTPoly::Params>& paramsQP, const DCRTPoly& src, uint32_t first, uint32_t n) {
    const uint32_t total = paramsQP->GetParams().size();
    return DCRTPoly::AssembleRows(paramsQP, Format::EVALUATION,
                                  {RowPiece{nullptr, 0, first}, RowPiece{&src, first, n}, RowPiece{nullptr, 0, total - first - n}});
}
}  // namespace

EvalKey<DCRTPoly> KeySwitchHYBRID::KeySwitchGenInternal(const PrivateKey<DCRTPoly> oldKey, const PrivateKey<DCRTPoly> newKey) const {
    return KeySwitchHYBRID::KeySwitchGenInternal(oldKey, newKey, nullptr);
}

EvalKey<DCRTPoly> KeySwitchHYBRID::KeySwitchGenInternal(const PrivateKey<DCRTPoly> oldKey, const PrivateKey<DCRTPoly> newKey,
                                                        const EvalKey<DCRTPoly> ekPrev) const {
    hiprt::MemberScope scope("KeySwitchGenInternal");
    const auto cryptoParams = std::dynamic_pointer_cast<CryptoParametersRNS>(newKey->GetCryptoParameters());
    const auto& paramsQ     = cryptoParams->GetElementParams();
    const auto& paramsQP    = cryptoParams->GetParamsQP();
    const auto& paramsP     = cryptoParams->GetParamsP();
    const uint32_t sizeQ = paramsQ->GetParams().size(), sizeP = paramsP->GetParams().size();

    // sNew over Q extended to Q u P (:60-81): the P limbs are limb 0 in COEFFICIENT form, lifted centred to every p_j (the ModRaise
    // constructor of the backend class is exactly PolyImpl::SwitchModulus into every limb, dcrtpoly-impl.h:87-93)
    DCRTPoly sNewEval = newKey->GetPrivateElement();
    sNewEval.SetFormat(Format::EVALUATION);
    DCRTPoly sNew0 = newKey->GetPrivateElement().CloneTowers(0, 0);
    sNew0.SetFormat(Format::COEFFICIENT);
    // (one polynomial modulo q_0 -> every limb of Q u P, centred; the Q limbs of that lift are not used)
    DCRTPoly lifted(sNew0.GetElementAtIndex(0), paramsQP);
    lifted.SetFormat(Format::EVALUATION);
    const DCRTPoly sNewExt = DCRTPoly::AssembleRows(paramsQP, Format::EVALUATION, {RowPiece{&sNewEval, 0, sizeQ}, RowPiece{&lifted, sizeQ, sizeP}});

    const auto ns              = cryptoParams->GetNoiseScale();
    const uint32_t numPerPartQ = cryptoParams->GetNumPerPartQ();
    const uint32_t numPartQ    = cryptoParams->GetNumPartQ();
    std::vector<DCRTPoly> av(numPartQ), bv(numPartQ);
    DugType dug;
    auto dgg = cryptoParams->GetDiscreteGaussianGenerator();
    const auto& sOld  = oldKey->GetPrivateElement();
    const DCRTPoly sOldP = sOld.TimesNoCheck(cryptoParams->GetPModq());  // [P]_{q_i} * sOld_i on every Q limb (:113)

    // The reference's loop header, verbatim (:96): the samplers read a thread-local PRNG, so the keys are reproduced word for word only
    // if the same thread draws the same digit; and OpenMP's private clause hands every thread a DEFAULT-constructed dgg (not a copy of
    // the parameters' one), which is therefore what the reference's evaluation-key noise is drawn with.
#pragma omp parallel for num_threads(OpenFHEParallelControls.GetThreadLimit(numPartQ)) private(dug, dgg)
    for (uint32_t part = 0; part < numPartQ; ++part) {
        DCRTPoly a = (ekPrev == nullptr) ? DCRTPoly(dug, paramsQP, Format::EVALUATION) : ekPrev->GetAVector()[part];
        DCRTPoly e(dgg, paramsQP, Format::EVALUATION);
        const uint32_t startPartIdx = numPerPartQ * part;
        const uint32_t endPartIdx   = (sizeQ > (startPartIdx + numPerPartQ)) ? (startPartIdx + numPerPartQ) : sizeQ;
        DCRTPoly b = (a * sNewExt).Negate();
        if (ns != 1)
            e *= NativeInteger(ns);
        b += e;
        b += RowsInZeros(paramsQP, sOldP, startPartIdx, endPartIdx - startPartIdx);
        av[part] = std::move(a);
        bv[part] = std::move(b);
    }
    EvalKeyRelin<DCRTPoly> ek(std::make_shared<EvalKeyRelinImpl<DCRTPoly>>(newKey->GetCryptoContext()));
    ek->SetAVector(std::move(av));
    ek->SetBVector(std::move(bv));
    ek->SetKeyTag(newKey->GetKeyTag());
    return ek;
}

// the public-key variant (:132-195): a = newp1 * u + ns * e1,  b = newp0 * u + ns * e0 + [ PModq_i * sOld_i on the digit's limbs ]
EvalKey<DCRTPoly> KeySwitchHYBRID::KeySwitchGenInternal(const PrivateKey<DCRTPoly> oldKey, const PublicKey<DCRTPoly> newKey) const {
    hiprt::MemberScope scope("KeySwitchGenInternal");
    const auto cryptoParams = std::dynamic_pointer_cast<CryptoParametersRNS>(newKey->GetCryptoParameters());
    const auto& paramsQ     = cryptoParams->GetElementParams();
    const auto& paramsQP    = cryptoParams->GetParamsQP();
    const uint32_t sizeQ    = paramsQ->GetParams().size();
    const auto ns              = cryptoParams->GetNoiseScale();
    const uint32_t numPerPartQ = cryptoParams->GetNumPerPartQ();
    const uint32_t numPartQ    = cryptoParams->GetNumPartQ();
    std::vector<DCRTPoly> av(numPartQ), bv(numPartQ);
    TugType tug;
    auto dgg = cryptoParams->GetDiscreteGaussianGenerator();
    const auto& sOld  = oldKey->GetPrivateElement();
    const auto& newp0 = newKey->GetPublicElements().at(0);
    const auto& newp1 = newKey->GetPublicElements().at(1);
    const DCRTPoly sOldP = sOld.TimesNoCheck(cryptoParams->GetPModq());

#pragma omp parallel for num_threads(OpenFHEParallelControls.GetThreadLimit(numPartQ)) private(dgg, tug)  // (:157, see above)
    for (uint32_t part = 0; part < numPartQ; ++part) {
        DCRTPoly u = (cryptoParams->GetSecretKeyDist() == GAUSSIAN) ? DCRTPoly(dgg, paramsQP, Format::EVALUATION) :
                                                                      DCRTPoly(tug, paramsQP, Format::EVALUATION);
        DCRTPoly e0(dgg, paramsQP, Format::EVALUATION);
        DCRTPoly e1(dgg, paramsQP, Format::EVALUATION);
        const uint32_t startPartIdx = numPerPartQ * part;
        const uint32_t endPartIdx   = (sizeQ > startPartIdx + numPerPartQ) ? (startPartIdx + numPerPartQ) : sizeQ;
        if (ns != 1) {
            e0 *= NativeInteger(ns);
            e1 *= NativeInteger(ns);
        }
        DCRTPoly a = newp1 * u;
        a += e1;
        DCRTPoly b = newp0 * u;
        b += e0;
        b += RowsInZeros(paramsQP, sOldP, startPartIdx, endPartIdx - startPartIdx);
        av[part] = std::move(a);
        bv[part] = std::move(b);
    }
    EvalKeyRelin<DCRTPoly> ek = std::make_shared<EvalKeyRelinImpl<DCRTPoly>>(newKey->GetCryptoContext());
    ek->SetAVector(std::move(av));
    ek->SetBVector(std::move(bv));
    ek->SetKeyTag(newKey->GetKeyTag());
    return ek;
}

}  // namespace lbcrypto

// ---- the linear transforms of CKKS bootstrapping as ONE library call per level ------------------------------------------------------
// FHECKKSRNS::EvalLinearTransform (ckksrns-fhe.cpp:1832-1882), EvalCoeffsToSlots (:1884-2039) and EvalSlotsToCoeffs (:2041-2198) are, level
// by level, the same baby-step/giant-step shape: hoisted inner rotations in the extended basis, a plaintext multiply-accumulate per
// giant step, KeySwitchDown, an outer rotation, one final KeySwitchDown.  Through the class surface a level is ~400 tower operations;
// the device library runs it as one composite with every stage batched over all giant steps (fhe_ckks_bsgs_transform, double hoisting).
// The hooks below keep the reference's level structure (rotation amounts per level by the reference's formulas, ModReduce between
// levels through the scheme) and hand every level to the composite.  The reference's own definitions stay in the library under the
// names fhe_ref_* (objcopy --redefine-sym, hal/Makefile): they are the fall-back for everything the composite does not take, and the
// FIRST use of a hooked function at a level computes both and compares every word and the ciphertext metadata (hiprt::DomainChecked).
#include "scheme/ckksrns/ckksrns-fhe.h"
#include "scheme/ckksrns/ckksrns-utils.h"

#ifdef FHE_HIP_PATCHED_PKE  // (sources with integration/with_hip.patch: the reference's bodies are the *Reference members)
static inline lbcrypto::Ciphertext<lbcrypto::DCRTPoly> fhe_ref_EvalLinearTransform(const lbcrypto::FHECKKSRNS* self,
                                                                                   const std::vector<lbcrypto::ReadOnlyPlaintext>& A,
                                                                                   lbcrypto::ConstCiphertext<lbcrypto::DCRTPoly>& ct) {
    return self->EvalLinearTransformReference(A, ct);
}
static inline lbcrypto::Ciphertext<lbcrypto::DCRTPoly> fhe_ref_EvalCoeffsToSlots(const lbcrypto::FHECKKSRNS* self,
                                                                                 const std::vector<std::vector<lbcrypto::ReadOnlyPlaintext>>& A,
                                                                                 lbcrypto::ConstCiphertext<lbcrypto::DCRTPoly>& ctxt) {
    return self->EvalCoeffsToSlotsReference(A, ctxt);
}
static inline lbcrypto::Ciphertext<lbcrypto::DCRTPoly> fhe_ref_EvalSlotsToCoeffs(const lbcrypto::FHECKKSRNS* self,
                                                                                 const std::vector<std::vector<lbcrypto::ReadOnlyPlaintext>>& A,
                                                                                 lbcrypto::ConstCiphertext<lbcrypto::DCRTPoly>& ctxt) {
    return self->EvalSlotsToCoeffsReference(A, ctxt);
}
#else
extern "C" {
lbcrypto::Ciphertext<lbcrypto::DCRTPoly> fhe_ref_EvalLinearTransform(const lbcrypto::FHECKKSRNS* self, const std::vector<lbcrypto::ReadOnlyPlaintext>& A,
                                                                     lbcrypto::ConstCiphertext<lbcrypto::DCRTPoly>& ct);
lbcrypto::Ciphertext<lbcrypto::DCRTPoly> fhe_ref_EvalCoeffsToSlots(const lbcrypto::FHECKKSRNS* self,
                                                                   const std::vector<std::vector<lbcrypto::ReadOnlyPlaintext>>& A,
                                                                   lbcrypto::ConstCiphertext<lbcrypto::DCRTPoly>& ctxt);
lbcrypto::Ciphertext<lbcrypto::DCRTPoly> fhe_ref_EvalSlotsToCoeffs(const lbcrypto::FHECKKSRNS* self,
                                                                   const std::vector<std::vector<lbcrypto::ReadOnlyPlaintext>>& A,
                                                                   lbcrypto::ConstCiphertext<lbcrypto::DCRTPoly>& ctxt);
}
#endif

namespace lbcrypto {
namespace {
constexpr uint32_t kNoSkip = 0xffffffffu;
// result = sum_i Rot_{rotOut[i]}( sum_j Rot_{rotIn[j]}(ct) * A[i * |rotIn| + j] ) — terms with index == skip or beyond A absent —
// with the reference's metadata (EvalMultExt, :2723-2732: noise-scale degree and scaling factor of the plaintexts join the ciphertext's).
// false: the composite does not take this level (the caller runs the reference's function instead).
bool BsgsLevelOnDevice(Ciphertext<DCRTPoly>& ct, const std::vector<int32_t>& rotIn, const std::vector<int32_t>& rotOut,
                       const std::vector<ReadOnlyPlaintext>& A, uint32_t skip) {
    // (FHE_HAL_DEBUG=1 names the exit a level left the composite through)
    auto Why = [](int site) {
        static const bool dbg = std::getenv("FHE_HAL_DEBUG") != nullptr;
        if (dbg)
            fprintf(stderr, "hal debug: BsgsLevelOnDevice declined at exit %d (%s)\n", site, hiprt::api().last_error());
        return false;
    };
    const auto cc = ct->GetCryptoContext();
    const auto cp = std::dynamic_pointer_cast<CryptoParametersCKKSRNS>(ct->GetCryptoParameters());
    auto& cv      = ct->GetElements();
    if (!cp || cv.size() != 2 || cv[0].GetFormat() != Format::EVALUATION || cv[1].GetFormat() != Format::EVALUATION || A.empty() || !A[0])
        return Why(1);
    uint32_t sizeQl = 0;
    auto dom        = DomainOf(cp, cv[0], &sizeQl);
    if (!dom || cv[1].GetNumOfElements() != sizeQl)
        return Why(2);
    const uint32_t M = cc->GetCyclotomicOrder(), nIn = rotIn.size(), nOut = rotOut.size();
    const size_t sizeP = cp->GetParamsP()->GetParams().size(), sizeQP = cp->GetParamsQP()->GetParams().size();
    const auto& keyMap = cc->GetEvalAutomorphismKeyMap(ct->GetKeyTag());
    hiprt::Op op;
    std::vector<hiprt::PackedKey> keep;
    auto keysOf = [&](const std::vector<int32_t>& rot, std::vector<uint32_t>& k, std::vector<const fhe_ks_key*>& keys) {
        k.assign(rot.size(), 0), keys.assign(rot.size(), nullptr);
        for (size_t j = 0; j < rot.size(); ++j) {
            if (rot[j] == 0)
                continue;
            k[j]    = FindAutomorphismIndex2nComplex(rot[j], M);
            auto it = keyMap.find(k[j]);
            std::vector<hiprt::Buf> kb, ka;
            if (it == keyMap.end() || !KeyBuffers(it->second, cp->GetNumPartQ(), sizeQP, kb, ka))
                return Why(3);
            keep.push_back(hiprt::DomainKey(*dom, kb, ka, op));
            if (!keep.back().key)
                return Why(4);
            op.R(keep.back().b), op.R(keep.back().a);
            keys[j] = keep.back().key.get();
        }
        return true;
    };
    std::vector<uint32_t> inK, outK;
    std::vector<const fhe_ks_key*> inKeys, outKeys;
    if (!keysOf(rotIn, inK, inKeys) || !keysOf(rotOut, outK, outKeys))
        return Why(5);
    // the encoded diagonals: towers over Q_l u P in EVALUATION (EvalLinearTransformPrecompute's aux plaintexts), device resident after
    // their first use
    std::vector<const uint64_t*> diag((size_t)nOut * nIn, nullptr);
    std::vector<hiprt::Buf> diagKeep;
    const auto& ql = cv[0].GetParams()->GetParams();
    for (uint32_t i = 0; i < nOut; ++i)
        for (uint32_t j = 0; j < nIn; ++j) {
            const size_t idx = (size_t)i * nIn + j;
            if (idx == skip || idx >= A.size())
                continue;
            if (!A[idx])
                return Why(6);
            const DCRTPoly& pt = A[idx]->GetElement<DCRTPoly>();
            const auto& pl     = pt.GetParams()->GetParams();
            if (pt.GetFormat() != Format::EVALUATION || pl.size() != sizeQl + sizeP || pl[0]->GetModulus() != ql[0]->GetModulus() ||
                pl[sizeQl - 1]->GetModulus() != ql[sizeQl - 1]->GetModulus() ||
                pl[sizeQl]->GetModulus() != cp->GetParamsP()->GetParams()[0]->GetModulus())
                return Why(7);
            diagKeep.push_back(pt.DeviceWords());
            if (!diagKeep.back())
                return Why(8);
            diag[idx] = op.R(diagKeep.back());
        }
    auto b0 = cv[0].DeviceWords(), b1 = cv[1].DeviceWords();
    if (!b0 || !b1)
        return Why(9);
    const uint32_t width = cv[0].Width();  // (a wide ciphertext: its K towers are the composite's batch, the diagonals are shared)
    if (cv[1].Width() != width)
        return Why(9);
    const auto& Api  = hiprt::api();
    const size_t N   = cv[0].GetParams()->GetRingDimension();
    const size_t wsB = Api.bsgs_workspace_bytes(hiprt::DomainPlan(*dom), sizeQl, width, nIn, nOut);
    auto ws = hiprt::Alloc(wsB / 8 + 1), o0 = hiprt::Alloc((size_t)width * sizeQl * N), o1 = hiprt::Alloc((size_t)width * sizeQl * N);
    if (Api.bsgs_transform(hiprt::DomainPlan(*dom), op.R(b0), op.R(b1), sizeQl, width, nIn, inK.data(), inKeys.data(), nOut, outK.data(), outKeys.data(),
                           diag.data(), op.W(o0), op.W(o1), op.W(ws, false), wsB, op.s) != FHE_OK)
        return Why(10);
    hiprt::CountDevice("Bootstrap.BsgsLevel");
    hiprt::CountComposite();
    auto result = ct->CloneEmpty();
    std::vector<DCRTPoly> elements;
    elements.push_back(DCRTPoly::FromDeviceWords(cv[0].GetParams(), Format::EVALUATION, std::move(o0), width));
    elements.push_back(DCRTPoly::FromDeviceWords(cv[0].GetParams(), Format::EVALUATION, std::move(o1), width));
    result->SetElements(std::move(elements));
    result->SetNoiseScaleDeg(ct->GetNoiseScaleDeg() + A[0]->GetNoiseScaleDeg());
    result->SetScalingFactor(ct->GetScalingFactor() * A[0]->GetScalingFactor());
    ct = std::move(result);
    return true;
}
bool SameCiphertext(const hiprt::KsDomain& dom, const Ciphertext<DCRTPoly>& a, const Ciphertext<DCRTPoly>& b) {
    if (!a || !b || a->GetElements().size() != b->GetElements().size() || a->GetLevel() != b->GetLevel() ||
        a->GetNoiseScaleDeg() != b->GetNoiseScaleDeg() || a->GetScalingFactor() != b->GetScalingFactor() || a->GetSlots() != b->GetSlots())
        return false;
    for (size_t e = 0; e < a->GetElements().size(); ++e) {
        const auto &x = a->GetElements()[e], &y = b->GetElements()[e];
        auto bx = x.DeviceWords(), by = y.DeviceWords();
        if (!bx || !by || x.GetNumOfElements() != y.GetNumOfElements() || x.GetFormat() != y.GetFormat() ||
            hiprt::Checksums(hiprt::DomainCtx(dom), bx, x.GetNumOfElements()) != hiprt::Checksums(hiprt::DomainCtx(dom), by, y.GetNumOfElements()))
            return false;
    }
    return true;
}
// runs `composite` (false: not applicable) with the first-use check against `reference`
template <typename Composite, typename Reference>
Ciphertext<DCRTPoly> CheckedComposite(ConstCiphertext<DCRTPoly>& ctxt, Composite composite, Reference reference) {
    uint32_t sizeQl = 0;
    const auto cp   = std::dynamic_pointer_cast<CryptoParametersCKKSRNS>(ctxt->GetCryptoParameters());
    auto dom        = (cp && ctxt->GetElements().size() == 2) ? DomainOf(cp, ctxt->GetElements()[0], &sizeQl) : nullptr;
    const int state = dom ? hiprt::DomainChecked(*dom, hiprt::kBsgs, sizeQl) : 2;
    static const bool dbg = std::getenv("FHE_HAL_DEBUG") != nullptr;
    if (dbg)
        fprintf(stderr, "hal debug: CheckedComposite domain %p level %u state %d elements %zu\n", (void*)dom.get(), sizeQl, state,
                ctxt->GetElements().size());
    const bool wide = ctxt->GetElements()[0].Width() > 1;
    if (wide && state != 1)
        WideNeedsCheckedComposite("a linear transform of bootstrapping");
    if (state == 2)
        return reference();
    Ciphertext<DCRTPoly> mine;
    if (!composite(mine)) {
        if (wide)
            WideNeedsCheckedComposite("a linear transform of bootstrapping (the composite declined)");
        return reference();
    }
    if (state == 1)
        return mine;
    auto ref = reference();
    hiprt::DomainSetChecked(*dom, hiprt::kBsgs, sizeQl, SameCiphertext(*dom, mine, ref));
    return ref;
}
}  // namespace

Ciphertext<DCRTPoly> FHECKKSRNS::EvalLinearTransform(const std::vector<ReadOnlyPlaintext>& A, ConstCiphertext<DCRTPoly>& ct) const {
    hiprt::MemberScope scope("Bootstrap.EvalLinearTransform");
    return CheckedComposite(
        ct,
        [&](Ciphertext<DCRTPoly>& out) {
            const uint32_t slots = A.size();  // (:1835-1838)
            const auto& p        = GetBootPrecom(slots);
            const uint32_t bStep = (p.m_paramsEnc.g == 0) ? std::ceil(std::sqrt(slots)) : p.m_paramsEnc.g;
            const uint32_t gStep = std::ceil(static_cast<double>(slots) / bStep);
            std::vector<int32_t> rotIn(bStep), rotOut(gStep);
            for (uint32_t j = 0; j < bStep; ++j)
                rotIn[j] = j;  // (:1846-1847: fast rotations by 1 .. bStep-1; the unrotated term is KeySwitchExt, :1855)
            for (uint32_t i = 0; i < gStep; ++i)
                rotOut[i] = bStep * i;  // (:1871)
            out = ct->Clone();
            return BsgsLevelOnDevice(out, rotIn, rotOut, A, kNoSkip);
        },
        [&] { return fhe_ref_EvalLinearTransform(this, A, ct); });
}

Ciphertext<DCRTPoly> FHECKKSRNS::EvalCoeffsToSlots(const std::vector<std::vector<ReadOnlyPlaintext>>& A, ConstCiphertext<DCRTPoly>& ctxt) const {
    hiprt::MemberScope scope("Bootstrap.EvalCoeffsToSlots");
    return CheckedComposite(
        ctxt,
        [&](Ciphertext<DCRTPoly>& out) {
            const uint32_t slots = ctxt->GetSlots();
            const auto& p        = GetBootPrecom(slots).m_paramsEnc;
            const auto cc        = ctxt->GetCryptoContext();
            const uint32_t M4    = cc->GetCyclotomicOrder() / 4;
            const int32_t flagRem = p.remCollapse != 0 ? 1 : 0, stop = flagRem ? 0 : -1;  // (:1895-1903)
            const auto cp         = std::dynamic_pointer_cast<CryptoParametersCKKSRNS>(cc->GetCryptoParameters());
            out = ctxt->Clone();
            bool firstLevel = true;
            auto level = [&](const std::vector<int32_t>& rotIn, const std::vector<int32_t>& rotOut, const std::vector<ReadOnlyPlaintext>& Al, uint32_t skip) {
                if (!firstLevel)
                    cc->GetScheme()->ModReduceInternalInPlace(out, cp->GetCompositeDegree());  // (:1936-1937, :1989)
                firstLevel = false;
                return BsgsLevelOnDevice(out, rotIn, rotOut, Al, skip);
            };
            // levels lvlb-1 down to stop+1 (:1911-1918, :1933-1985): rotations scaled by 2^((s - flagRem) * layersCollapse + remCollapse)
            const int32_t offset = static_cast<int32_t>((p.numRotations + 1) / 2) - 1;
            for (int32_t s = p.lvlb - 1; s > stop; --s) {
                const int32_t scale = 1 << ((s - flagRem) * p.layersCollapse + p.remCollapse);
                std::vector<int32_t> rotIn(p.g), rotOut(p.b);
                for (uint32_t i = 0; i < p.b; ++i)
                    rotOut[i] = ReduceRotation(scale * p.g * i, M4);
                for (uint32_t j = 0; j < p.g; ++j)
                    rotIn[j] = ReduceRotation(scale * (static_cast<int32_t>(j) - offset), slots);
                if (!level(rotIn, rotOut, A[s], p.numRotations))
                    return false;
            }
            if (flagRem) {  // the remainder level (:1920-1926, :1988-2037)
                const int32_t offsetRem = static_cast<int32_t>((p.numRotationsRem + 1) / 2) - 1;
                std::vector<int32_t> rotIn(p.gRem), rotOut(p.bRem);
                for (uint32_t i = 0; i < p.bRem; ++i)
                    rotOut[i] = ReduceRotation(p.gRem * i, M4);
                for (uint32_t j = 0; j < p.gRem; ++j)
                    rotIn[j] = ReduceRotation(static_cast<int32_t>(j) - offsetRem, slots);
                if (!level(rotIn, rotOut, A[stop], p.numRotationsRem))
                    return false;
            }
            return true;
        },
        [&] { return fhe_ref_EvalCoeffsToSlots(this, A, ctxt); });
}

Ciphertext<DCRTPoly> FHECKKSRNS::EvalSlotsToCoeffs(const std::vector<std::vector<ReadOnlyPlaintext>>& A, ConstCiphertext<DCRTPoly>& ctxt) const {
    hiprt::MemberScope scope("Bootstrap.EvalSlotsToCoeffs");
    return CheckedComposite(
        ctxt,
        [&](Ciphertext<DCRTPoly>& out) {
            const uint32_t slots = ctxt->GetSlots();
            const auto& p        = GetBootPrecom(slots).m_paramsDec;
            const auto cc        = ctxt->GetCryptoContext();
            const uint32_t M4    = cc->GetCyclotomicOrder() / 4;
            const int32_t flagRem = (p.remCollapse == 0) ? 0 : 1, smax = p.lvlb - flagRem;  // (:2049-2058)
            const auto cp         = std::dynamic_pointer_cast<CryptoParametersCKKSRNS>(cc->GetCryptoParameters());
            out = ctxt->Clone();
            bool firstLevel = true;
            auto level = [&](const std::vector<int32_t>& rotIn, const std::vector<int32_t>& rotOut, const std::vector<ReadOnlyPlaintext>& Al, uint32_t skip) {
                if (!firstLevel)
                    cc->GetScheme()->ModReduceInternalInPlace(out, cp->GetCompositeDegree());  // (:2090-2091, :2143)
                firstLevel = false;
                return BsgsLevelOnDevice(out, rotIn, rotOut, Al, skip);
            };
            const int32_t offset = static_cast<int32_t>((p.numRotations + 1) / 2) - 1;
            for (int32_t s = 0; s < smax; ++s) {  // (:2061-2067, :2089-2140): rotations scaled by 2^(s * layersCollapse)
                const int32_t scale = 1 << (s * p.layersCollapse);
                std::vector<int32_t> rotIn(p.g), rotOut(p.b);
                for (uint32_t j = 0; j < p.g; ++j)
                    rotIn[j] = ReduceRotation((static_cast<int32_t>(j) - offset) * scale, M4);
                for (uint32_t i = 0; i < p.b; ++i)
                    rotOut[i] = ReduceRotation((p.g * i) * scale, M4);
                if (!level(rotIn, rotOut, A[s], p.numRotations))
                    return false;
            }
            if (flagRem) {  // (:2069-2076, :2142-2193)
                const int32_t scaleRem  = 1 << (smax * p.layersCollapse);
                const int32_t offsetRem = static_cast<int32_t>((p.numRotationsRem + 1) / 2) - 1;
                std::vector<int32_t> rotIn(p.gRem), rotOut(p.bRem);
                for (uint32_t j = 0; j < p.gRem; ++j)
                    rotIn[j] = ReduceRotation((static_cast<int32_t>(j) - offsetRem) * scaleRem, M4);
                for (uint32_t i = 0; i < p.bRem; ++i)
                    rotOut[i] = ReduceRotation((p.gRem * i) * scaleRem, M4);
                if (!level(rotIn, rotOut, A[smax], p.numRotationsRem))
                    return false;
            }
            return true;
        },
        [&] { return fhe_ref_EvalSlotsToCoeffs(this, A, ctxt); });
}

}  // namespace lbcrypto
