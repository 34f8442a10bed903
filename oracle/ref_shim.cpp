// ref_shim.cpp — TEST INFRASTRUCTURE ONLY.
//
// A C-ABI window onto the REFERENCE ITSELF (lbcrypto::DCRTPoly / ChineseRemainderTransformFTT /
// CryptoContext compiled unmodified from /root/reference into oracle/_ref/).  Used to
//   (1) pin oracle/fhe_oracle.c against the real reference on random inputs,
//   (2) generate tests/golden/* fixtures (tests/golden/make_golden.py),
//   (3) serve as bench.py's cpu_baseline with kind "reference" (OpenMP, all host cores).
// Our own code: it only CALLS the reference's public class surface; no reference source is copied.
#include <cstdint>
#include <cstring>
#include <memory>
#include <random>
#include <vector>

#include "openfhe.h"
#include "utils/utilities-int.h"
#include <chrono>

using namespace lbcrypto;

namespace {
using ParmType = DCRTPoly::Params;

std::shared_ptr<ParmType> make_params(uint32_t N, uint32_t L, const uint64_t* q, const uint64_t* psi) {
    std::vector<NativeInteger> m(L), r(L);
    for (uint32_t i = 0; i < L; ++i) {
        m[i] = NativeInteger(q[i]);
        r[i] = NativeInteger(psi[i]);
    }
    return std::make_shared<ParmType>(2 * N, m, r);
}

DCRTPoly make_poly(const std::shared_ptr<ParmType>& params, const uint64_t* x, Format f) {
    uint32_t N = params->GetRingDimension();
    uint32_t L = params->GetParams().size();
    DCRTPoly p(params, f, true);
    for (uint32_t i = 0; i < L; ++i) {
        NativePoly e(params->GetParams()[i], f, true);
        for (uint32_t j = 0; j < N; ++j)
            e[j] = NativeInteger(x[(size_t)i * N + j]);
        p.SetElementAtIndex(i, std::move(e));
    }
    return p;
}

void export_poly(const DCRTPoly& p, uint64_t* out) {
    uint32_t N = p.GetRingDimension();
    uint32_t L = p.GetNumOfElements();
    for (uint32_t i = 0; i < L; ++i) {
        const auto& e = p.GetElementAtIndex(i);
        for (uint32_t j = 0; j < N; ++j)
            out[(size_t)i * N + j] = e[j].ConvertToInt<uint64_t>();
    }
}

std::vector<NativeInteger> vecNI(const uint64_t* v, size_t n) {
    std::vector<NativeInteger> r(n);
    for (size_t i = 0; i < n; ++i)
        r[i] = NativeInteger(v[i]);
    return r;
}
std::vector<NativeInteger> precon(const std::vector<NativeInteger>& v, const uint64_t* mod) {
    std::vector<NativeInteger> r(v.size());
    for (size_t i = 0; i < v.size(); ++i)
        r[i] = v[i].PrepModMulConst(NativeInteger(mod[i]));
    return r;
}
std::vector<DoubleNativeInt> mu128(const uint64_t* mod, size_t n) {
    std::vector<DoubleNativeInt> r(n);
    for (size_t i = 0; i < n; ++i)
        r[i] = (BigInteger(1).LShiftEq(128) / BigInteger(mod[i])).ConvertToInt<DoubleNativeInt>();
    return r;
}
}  // namespace

extern "C" {

// ---- number theory ----
uint64_t ref_last_prime(uint32_t bits, uint64_t m) { return LastPrime<NativeInteger>(bits, m).ConvertToInt<uint64_t>(); }
uint64_t ref_first_prime(uint32_t bits, uint64_t m) { return FirstPrime<NativeInteger>(bits, m).ConvertToInt<uint64_t>(); }
uint64_t ref_previous_prime(uint64_t q, uint64_t m) { return PreviousPrime<NativeInteger>(NativeInteger(q), m).ConvertToInt<uint64_t>(); }
uint64_t ref_next_prime(uint64_t q, uint64_t m) { return NextPrime<NativeInteger>(NativeInteger(q), m).ConvertToInt<uint64_t>(); }
uint64_t ref_root_of_unity(uint32_t m, uint64_t q) { return RootOfUnity<NativeInteger>(m, NativeInteger(q)).ConvertToInt<uint64_t>(); }
void ref_dcrt_params(uint32_t order, uint32_t L, uint32_t bits, uint64_t* q, uint64_t* psi) {
    ILDCRTParams<BigInteger> p(order, L, bits);
    for (uint32_t i = 0; i < L; ++i) {
        q[i]   = p.GetParams()[i]->GetModulus().ConvertToInt<uint64_t>();
        psi[i] = p.GetParams()[i]->GetRootOfUnity().ConvertToInt<uint64_t>();
    }
}
void ref_precompute_auto_map(uint32_t n, uint32_t k, uint32_t* out) {
    std::vector<uint32_t> v(n);
    PrecomputeAutoMap(n, k, &v);
    std::memcpy(out, v.data(), sizeof(uint32_t) * n);
}
uint32_t ref_find_automorphism_index_2n_complex(int32_t i, uint32_t m) { return FindAutomorphismIndex2nComplex(i, m); }

// ---- scalar ops ----
uint64_t ref_mod_mul_fast_const(uint64_t a, uint64_t b, uint64_t q) {
    NativeInteger B(b), Q(q);
    return NativeInteger(a).ModMulFastConst(B, Q, B.PrepModMulConst(Q)).ConvertToInt<uint64_t>();
}
uint64_t ref_prep_mod_mul_const(uint64_t b, uint64_t q) { return NativeInteger(b).PrepModMulConst(NativeInteger(q)).ConvertToInt<uint64_t>(); }
uint64_t ref_compute_mu(uint64_t q) { return NativeInteger(q).ComputeMu().ConvertToInt<uint64_t>(); }
uint64_t ref_mod_mul_fast(uint64_t a, uint64_t b, uint64_t q) {
    NativeInteger Q(q);
    return NativeInteger(a).ModMulFast(NativeInteger(b), Q, Q.ComputeMu()).ConvertToInt<uint64_t>();
}
uint64_t ref_barrett128(uint64_t lo, uint64_t hi, uint64_t q) {
    DoubleNativeInt a  = ((DoubleNativeInt)hi << 64) | lo;
    DoubleNativeInt mu = (BigInteger(1).LShiftEq(128) / BigInteger(q)).ConvertToInt<DoubleNativeInt>();
    return BarrettUint128ModUint64(a, q, mu);
}

// ---- single-limb NTT through ChineseRemainderTransformFTT (the FTT hook, math-hal.h:105-106) ----
void ref_ntt(uint64_t q, uint64_t psi, uint32_t N, uint64_t* x, int inverse) {
    NativeVector v(N, NativeInteger(q));
    for (uint32_t i = 0; i < N; ++i)
        v[i] = NativeInteger(x[i]);
    if (inverse)
        ChineseRemainderTransformFTT<NativeVector>().InverseTransformFromBitReverseInPlace(NativeInteger(psi), 2 * N, &v);
    else
        ChineseRemainderTransformFTT<NativeVector>().ForwardTransformToBitReverseInPlace(NativeInteger(psi), 2 * N, &v);
    for (uint32_t i = 0; i < N; ++i)
        x[i] = v[i].ConvertToInt<uint64_t>();
}

// ---- DCRTPoly session: keeps `batch` reference DCRTPoly objects alive so SwitchFormat / operator*=
//      can be timed without conversion overhead (cpu_baseline kind "reference") ----
struct RefTowers {
    std::shared_ptr<ParmType> params;
    std::vector<DCRTPoly> polys;
};
void* ref_towers_create(uint32_t N, uint32_t L, const uint64_t* q, const uint64_t* psi, const uint64_t* x,
                        uint32_t batch, int evalFormat) {
    auto* t   = new RefTowers;
    t->params = make_params(N, L, q, psi);
    for (uint32_t b = 0; b < batch; ++b)
        t->polys.push_back(make_poly(t->params, x + (size_t)b * L * N, evalFormat ? Format::EVALUATION : Format::COEFFICIENT));
    return t;
}
void ref_towers_destroy(void* h) { delete static_cast<RefTowers*>(h); }
void ref_towers_switch_format(void* h) {  // DCRTPolyImpl::SwitchFormat, dcrtpoly-impl.h:1932-1940
    for (auto& p : static_cast<RefTowers*>(h)->polys)
        p.SwitchFormat();
}
void ref_towers_mul_eq(void* h, void* other) {  // operator*=  dcrtpoly-impl.h:395-408
    auto& a = static_cast<RefTowers*>(h)->polys;
    auto& b = static_cast<RefTowers*>(other)->polys;
    for (size_t i = 0; i < a.size(); ++i)
        a[i] *= b[i];
}
void ref_towers_add_eq(void* h, void* other) {
    auto& a = static_cast<RefTowers*>(h)->polys;
    auto& b = static_cast<RefTowers*>(other)->polys;
    for (size_t i = 0; i < a.size(); ++i)
        a[i] += b[i];
}
void ref_towers_sub_eq(void* h, void* other) {
    auto& a = static_cast<RefTowers*>(h)->polys;
    auto& b = static_cast<RefTowers*>(other)->polys;
    for (size_t i = 0; i < a.size(); ++i)
        a[i] -= b[i];
}
void ref_towers_export(void* h, uint64_t* out) {
    auto* t    = static_cast<RefTowers*>(h);
    uint32_t N = t->params->GetRingDimension(), L = t->params->GetParams().size();
    for (size_t b = 0; b < t->polys.size(); ++b)
        export_poly(t->polys[b], out + b * (size_t)L * N);
}

// ---- automorphism, modulus switch ----
void ref_automorph(uint32_t N, uint32_t L, const uint64_t* q, const uint64_t* psi, const uint64_t* in, uint64_t* out,
                   uint32_t k, int evalFormat, int usePrecomp) {
    auto params = make_params(N, L, q, psi);
    auto p      = make_poly(params, in, evalFormat ? Format::EVALUATION : Format::COEFFICIENT);
    if (usePrecomp) {
        std::vector<uint32_t> pre(N);
        PrecomputeAutoMap(N, k, &pre);
        export_poly(p.AutomorphismTransform(k, pre), out);
    }
    else {
        export_poly(p.AutomorphismTransform(k), out);
    }
}
void ref_switch_modulus(uint64_t* v, uint32_t n, uint64_t oldq, uint64_t newq) {
    NativeVector x(n, NativeInteger(oldq));
    for (uint32_t i = 0; i < n; ++i)
        x[i] = NativeInteger(v[i]);
    x.SwitchModulus(NativeInteger(newq));
    for (uint32_t i = 0; i < n; ++i)
        v[i] = x[i].ConvertToInt<uint64_t>();
}

// ---- basis conversions with caller-provided tables ----
// QHatModp is [sizeQ][sizeP] (as ApproxSwitchCRTBasis indexes it)
void ref_approx_switch_crt_basis(uint32_t N, uint32_t sizeQ, const uint64_t* q, const uint64_t* psiQ, const uint64_t* x,
                                 const uint64_t* QHatInvModq, const uint64_t* QHatModp, uint32_t sizeP,
                                 const uint64_t* p, const uint64_t* psiP, uint64_t* out) {
    auto pq = make_params(N, sizeQ, q, psiQ);
    auto pp = make_params(N, sizeP, p, psiP);
    auto X  = make_poly(pq, x, Format::COEFFICIENT);
    auto hi = vecNI(QHatInvModq, sizeQ);
    std::vector<std::vector<NativeInteger>> hm(sizeQ);
    for (uint32_t i = 0; i < sizeQ; ++i)
        hm[i] = vecNI(QHatModp + (size_t)i * sizeP, sizeP);
    auto r = X.ApproxSwitchCRTBasis(pq, pp, hi, precon(hi, q), hm, mu128(p, sizeP));
    export_poly(r, out);
}
// QHatModp is [sizeP][sizeQ]; alphaQModp [sizeQ+1][sizeP]
void ref_switch_crt_basis(uint32_t N, uint32_t sizeQ, const uint64_t* q, const uint64_t* psiQ, const uint64_t* x,
                          const uint64_t* QHatInvModq, const uint64_t* QHatModp_pq, const uint64_t* alphaQModp,
                          uint32_t sizeP, const uint64_t* p, const uint64_t* psiP, const double* qInv, uint64_t* out) {
    auto pq = make_params(N, sizeQ, q, psiQ);
    auto pp = make_params(N, sizeP, p, psiP);
    auto X  = make_poly(pq, x, Format::COEFFICIENT);
    auto hi = vecNI(QHatInvModq, sizeQ);
    std::vector<std::vector<NativeInteger>> hm(sizeP), al(sizeQ + 1);
    for (uint32_t j = 0; j < sizeP; ++j)
        hm[j] = vecNI(QHatModp_pq + (size_t)j * sizeQ, sizeQ);
    for (uint32_t a = 0; a <= sizeQ; ++a)
        al[a] = vecNI(alphaQModp + (size_t)a * sizeP, sizeP);
    std::vector<double> qi(qInv, qInv + sizeQ);
    auto r = X.SwitchCRTBasis(pp, hi, precon(hi, q), hm, al, mu128(p, sizeP), qi);
    export_poly(r, out);
}
// ExpandCRTBasis / ExpandCRTBasisReverseOrder with caller tables (same layouts as ref_switch_crt_basis); out [(nQ+nP)][N]
void ref_expand_crt_basis(uint32_t N, uint32_t sizeQ, const uint64_t* q, const uint64_t* psiQ, const uint64_t* x, int inEval,
                          const uint64_t* QHatInvModq, const uint64_t* QHatModp_pq, const uint64_t* alphaQModp, uint32_t sizeP,
                          const uint64_t* p, const uint64_t* psiP, const double* qInv, int resultEval, int reverse,
                          uint64_t* out) {
    auto pq = make_params(N, sizeQ, q, psiQ);
    auto pp = make_params(N, sizeP, p, psiP);
    std::vector<uint64_t> all(sizeQ + sizeP), allPsi(sizeQ + sizeP);
    for (uint32_t i = 0; i < sizeQ + sizeP; ++i) {
        const bool fromQ = reverse ? i >= sizeP : i < sizeQ;
        const uint32_t k = reverse ? (fromQ ? i - sizeP : i) : (fromQ ? i : i - sizeQ);
        all[i]    = fromQ ? q[k] : p[k];
        allPsi[i] = fromQ ? psiQ[k] : psiP[k];
    }
    auto pqp = make_params(N, sizeQ + sizeP, all.data(), allPsi.data());
    auto X   = make_poly(pq, x, inEval ? Format::EVALUATION : Format::COEFFICIENT);
    auto hi  = vecNI(QHatInvModq, sizeQ);
    std::vector<std::vector<NativeInteger>> hm(sizeP), al(sizeQ + 1);
    for (uint32_t j = 0; j < sizeP; ++j)
        hm[j] = vecNI(QHatModp_pq + (size_t)j * sizeQ, sizeQ);
    for (uint32_t a = 0; a <= sizeQ; ++a)
        al[a] = vecNI(alphaQModp + (size_t)a * sizeP, sizeP);
    std::vector<double> qi(qInv, qInv + sizeQ);
    const Format rf = resultEval ? Format::EVALUATION : Format::COEFFICIENT;
    if (reverse)
        X.ExpandCRTBasisReverseOrder(pqp, pp, hi, precon(hi, q), hm, al, mu128(p, sizeP), qi, rf);
    else
        X.ExpandCRTBasis(pqp, pp, hi, precon(hi, q), hm, al, mu128(p, sizeP), qi, rf);
    export_poly(X, out);
}
// DCRTPolyImpl::ApproxModUp (dcrtpoly-impl.h:935-963) with caller tables; QHatModp [sizeQ][sizeP]; out [(sizeQ+sizeP)][N] EVALUATION
void ref_approx_mod_up(uint32_t N, uint32_t sizeQ, const uint64_t* q, const uint64_t* psiQ, const uint64_t* x, int inEval,
                       const uint64_t* QHatInvModq, const uint64_t* QHatModp, uint32_t sizeP, const uint64_t* p,
                       const uint64_t* psiP, uint64_t* out) {
    auto pq = make_params(N, sizeQ, q, psiQ);
    auto pp = make_params(N, sizeP, p, psiP);
    std::vector<uint64_t> all(q, q + sizeQ), allPsi(psiQ, psiQ + sizeQ);
    all.insert(all.end(), p, p + sizeP);
    allPsi.insert(allPsi.end(), psiP, psiP + sizeP);
    auto pqp = make_params(N, sizeQ + sizeP, all.data(), allPsi.data());
    auto X   = make_poly(pq, x, inEval ? Format::EVALUATION : Format::COEFFICIENT);
    auto hi  = vecNI(QHatInvModq, sizeQ);
    std::vector<std::vector<NativeInteger>> hm(sizeQ);
    for (uint32_t i = 0; i < sizeQ; ++i)
        hm[i] = vecNI(QHatModp + (size_t)i * sizeP, sizeP);
    X.ApproxModUp(pq, pp, pqp, hi, precon(hi, q), hm, mu128(p, sizeP));
    export_poly(X, out);
}
// DCRTPolyImpl::ExpandCRTBasisQlHat (dcrtpoly-impl.h:1167-1187): x [sizeQl][N] -> out [sizeQ][N]
void ref_expand_crt_basis_ql_hat(uint32_t N, uint32_t sizeQ, const uint64_t* q, const uint64_t* psi, const uint64_t* x,
                                 uint32_t sizeQl, int evalFormat, const uint64_t* QlHatModq, uint64_t* out) {
    auto pq  = make_params(N, sizeQ, q, psi);
    auto pql = make_params(N, sizeQl, q, psi);
    auto X   = make_poly(pql, x, evalFormat ? Format::EVALUATION : Format::COEFFICIENT);
    auto h   = vecNI(QlHatModq, sizeQl);
    X.ExpandCRTBasisQlHat(pq, h, precon(h, q), sizeQ);
    export_poly(X, out);
}
// PolyImpl::MultAccEqNoCheck per limb (poly.h:323 -> mubintvecnat.cpp:132-142): acc[i] += v[i] * consts[i]
void ref_mult_acc(uint32_t N, uint32_t L, const uint64_t* q, const uint64_t* psi, uint64_t* acc, const uint64_t* v,
                  const uint64_t* consts) {
    auto pq = make_params(N, L, q, psi);
    auto A  = make_poly(pq, acc, Format::EVALUATION);
    auto V  = make_poly(pq, v, Format::EVALUATION);
    for (uint32_t i = 0; i < L; ++i) {
        NativePoly e = A.GetElementAtIndex(i);
        e.MultAccEqNoCheck(V.GetElementAtIndex(i), NativeInteger(consts[i]));
        A.SetElementAtIndex(i, std::move(e));
    }
    export_poly(A, acc);
}
// DCRTPolyImpl::Plus / Minus(const std::vector<Integer>&) (dcrtpoly-impl.h:520-548); fmt 0 = EVALUATION, 1 = COEFFICIENT
void ref_plus_minus_const(uint32_t N, uint32_t L, const uint64_t* q, const uint64_t* psi, const uint64_t* a, const uint64_t* consts,
                          int fmt, int minus, uint64_t* out) {
    auto pq = make_params(N, L, q, psi);
    auto A  = make_poly(pq, a, fmt ? Format::COEFFICIENT : Format::EVALUATION);
    std::vector<DCRTPoly::Integer> k(L);
    for (uint32_t i = 0; i < L; ++i)
        k[i] = DCRTPoly::Integer(std::to_string(consts[i]));
    DCRTPoly R = minus ? A.Minus(k) : A.Plus(k);
    export_poly(R, out);
}
// the "ModRaise" constructor DCRTPolyImpl(const PolyType&, params) (dcrtpoly-impl.h:87-93): x [N] modulo q[0], COEFFICIENT
void ref_mod_raise(uint32_t N, uint32_t L, const uint64_t* q, const uint64_t* psi, const uint64_t* x, uint64_t* out) {
    auto pq = make_params(N, L, q, psi);
    NativePoly e(pq->GetParams()[0], Format::COEFFICIENT, true);
    for (uint32_t j = 0; j < N; ++j)
        e[j] = NativeInteger(x[j]);
    DCRTPoly X(e, pq);
    export_poly(X, out);
}
// DCRTPolyImpl::CRTDecompose(baseBits) (dcrtpoly-impl.h:230-285): x [L][N] in the given format; out [towers][L][N] (EVALUATION), returns the
// number of towers
uint32_t ref_crt_decompose(uint32_t N, uint32_t L, const uint64_t* q, const uint64_t* psi, const uint64_t* x, int inEval, uint32_t baseBits,
                           uint64_t* out) {
    auto pq = make_params(N, L, q, psi);
    DCRTPoly X = make_poly(pq, x, inEval ? Format::EVALUATION : Format::COEFFICIENT);
    auto R     = X.CRTDecompose(baseBits);
    for (size_t t = 0; t < R.size() && out; ++t)
        export_poly(R[t], out + t * (size_t)L * N);
    return (uint32_t)R.size();
}
// FastExpandCRTBasisPloverQ (COEFFICIENT): qInvModp [sizeQ][sizePl]; PlHatModq_qp [sizeQl][sizePl]; alphaPlModq
// [sizePl+1][sizeQl]; out [(sizeQl+sizePl)][N]
void ref_fast_expand_crt_basis_p_over_q(uint32_t N, uint32_t sizeQ, const uint64_t* q, const uint64_t* psiQ, const uint64_t* x,
                                        const uint64_t* mPlQHatInvModq, const uint64_t* qInvModp, uint32_t sizePl,
                                        const uint64_t* pl, const uint64_t* psiPl, const uint64_t* PlHatInvModp,
                                        const uint64_t* PlHatModq_qp, const uint64_t* alphaPlModq, uint32_t sizeQl,
                                        const uint64_t* ql, const uint64_t* psiQl, const double* pInv, uint64_t* out) {
    auto pq  = make_params(N, sizeQ, q, psiQ);
    auto ppl = make_params(N, sizePl, pl, psiPl);
    auto pql = make_params(N, sizeQl, ql, psiQl);
    std::vector<uint64_t> all(sizeQl + sizePl), allPsi(sizeQl + sizePl);
    for (uint32_t i = 0; i < sizeQl + sizePl; ++i) {
        all[i]    = i < sizeQl ? ql[i] : pl[i - sizeQl];
        allPsi[i] = i < sizeQl ? psiQl[i] : psiPl[i - sizeQl];
    }
    auto pqlpl = make_params(N, sizeQl + sizePl, all.data(), allPsi.data());
    auto X     = make_poly(pq, x, Format::COEFFICIENT);
    auto m1    = vecNI(mPlQHatInvModq, sizeQ);
    std::vector<std::vector<NativeInteger>> qinvp(sizeQ), hm(sizeQl), al(sizePl + 1);
    for (uint32_t i = 0; i < sizeQ; ++i)
        qinvp[i] = vecNI(qInvModp + (size_t)i * sizePl, sizePl);
    for (uint32_t i = 0; i < sizeQl; ++i)
        hm[i] = vecNI(PlHatModq_qp + (size_t)i * sizePl, sizePl);
    for (uint32_t a = 0; a <= sizePl; ++a)
        al[a] = vecNI(alphaPlModq + (size_t)a * sizeQl, sizeQl);
    auto hi2 = vecNI(PlHatInvModp, sizePl);
    std::vector<double> pi(pInv, pInv + sizePl);
    DCRTPoly::CRTBasisExtensionPrecomputations pre(pqlpl, ppl, pql, m1, precon(m1, q), qinvp, mu128(pl, sizePl), hi2,
                                                   precon(hi2, pl), hm, al, mu128(ql, sizeQl), pi);
    X.FastExpandCRTBasisPloverQ(pre);
    export_poly(X, out);
}
// ApproxModDown with caller tables; x [(sizeQl+sizeP)][N] EVALUATION; PHatModq [sizeP][sizeQl]; t = 0: CKKS/BFV form,
// t > 0: BGV form with tInvModp / tModqPrecon as CryptoParametersBGVRNS builds them
void ref_approx_mod_down(uint32_t N, uint32_t sizeQl, const uint64_t* q, const uint64_t* psiQ, uint32_t sizeP, const uint64_t* p,
                         const uint64_t* psiP, const uint64_t* x, const uint64_t* PInvModq, const uint64_t* PHatInvModp,
                         const uint64_t* PHatModq, uint64_t t, uint64_t* out) {
    std::vector<uint64_t> all(sizeQl + sizeP), allPsi(sizeQl + sizeP);
    for (uint32_t i = 0; i < sizeQl + sizeP; ++i) {
        all[i]    = i < sizeQl ? q[i] : p[i - sizeQl];
        allPsi[i] = i < sizeQl ? psiQ[i] : psiP[i - sizeQl];
    }
    auto pqp = make_params(N, sizeQl + sizeP, all.data(), allPsi.data());
    auto pq  = make_params(N, sizeQl, q, psiQ);
    auto pp  = make_params(N, sizeP, p, psiP);
    auto X   = make_poly(pqp, x, Format::EVALUATION);
    auto pinv = vecNI(PInvModq, sizeQl);
    auto phi  = vecNI(PHatInvModp, sizeP);
    std::vector<std::vector<NativeInteger>> phm(sizeP);
    for (uint32_t j = 0; j < sizeP; ++j)
        phm[j] = vecNI(PHatModq + (size_t)j * sizeQl, sizeQl);
    std::vector<NativeInteger> tInvModp, tInvModpPrecon, tModqPrecon;
    const NativeInteger T(t);
    if (t > 0) {
        for (uint32_t j = 0; j < sizeP; ++j) {
            tInvModp.push_back(T.ModInverse(NativeInteger(p[j])));
            tInvModpPrecon.push_back(tInvModp[j].PrepModMulConst(NativeInteger(p[j])));
        }
        for (uint32_t i = 0; i < sizeQl; ++i)
            tModqPrecon.push_back(T.PrepModMulConst(NativeInteger(q[i])));
    }
    auto r = X.ApproxModDown(pq, pp, pinv, precon(pinv, q), phi, precon(phi, p), phm, mu128(q, sizeQl), tInvModp,
                             tInvModpPrecon, T, tModqPrecon);
    export_poly(r, out);
}
// DropLastElementAndScale with caller tables (EVAL in/out)
void ref_drop_last_element_and_scale(uint32_t N, uint32_t sizeQl, const uint64_t* q, const uint64_t* psi,
                                     const uint64_t* x, const uint64_t* tabA, const uint64_t* tabB, uint64_t* out) {
    auto pq = make_params(N, sizeQl, q, psi);
    auto X  = make_poly(pq, x, Format::EVALUATION);
    X.DropLastElementAndScale(vecNI(tabA, sizeQl - 1), vecNI(tabB, sizeQl - 1));
    export_poly(X, out);
}

// DCRTPoly::ModReduce with the tables CryptoParametersBGVRNS::PrecomputeCRTTables derives (bgvrns-cryptoparameters.cpp)
void ref_mod_reduce(uint32_t N, uint32_t sizeQl, const uint64_t* q, const uint64_t* psi, const uint64_t* x, uint64_t t,
                    int evalFormat, uint64_t* out) {
    auto pq = make_params(N, sizeQl, q, psi);
    auto X  = make_poly(pq, x, evalFormat ? Format::EVALUATION : Format::COEFFICIENT);
    const NativeInteger T(t), ql(q[sizeQl - 1]);
    std::vector<NativeInteger> tModqPrecon(sizeQl - 1), qlInvModq(sizeQl - 1), qlInvModqPrecon(sizeQl - 1);
    for (uint32_t i = 0; i + 1 < sizeQl; ++i) {
        const NativeInteger qi(q[i]);
        tModqPrecon[i]     = T.PrepModMulConst(qi);
        qlInvModq[i]       = ql.ModInverse(qi);
        qlInvModqPrecon[i] = qlInvModq[i].PrepModMulConst(qi);
    }
    const NativeInteger negtInvModq = ql - T.ModInverse(ql);
    X.ModReduce(T, tModqPrecon, negtInvModq, negtInvModq.PrepModMulConst(ql), qlInvModq, qlInvModqPrecon);
    export_poly(X, out);
}

// ---- CKKS session: the reference's own context, keys, ciphertexts (config 3 shape) ----
struct RefCkks {
    CryptoContext<DCRTPoly> cc;
    KeyPair<DCRTPoly> kp;
    std::vector<Ciphertext<DCRTPoly>> cts;
};

void* ref_ckks_create(uint32_t ringDim, uint32_t multDepth, uint32_t scalingModSize, uint32_t firstModSize,
                      uint32_t numLargeDigits, int scalTech) {
    CCParams<CryptoContextCKKSRNS> parameters;
    parameters.SetSecurityLevel(HEStd_NotSet);
    parameters.SetRingDim(ringDim);
    parameters.SetMultiplicativeDepth(multDepth);
    parameters.SetScalingModSize(scalingModSize);
    parameters.SetFirstModSize(firstModSize);
    parameters.SetKeySwitchTechnique(HYBRID);
    parameters.SetScalingTechnique(static_cast<ScalingTechnique>(scalTech));
    if (numLargeDigits > 0)
        parameters.SetNumLargeDigits(numLargeDigits);
    auto* s = new RefCkks;
    s->cc   = GenCryptoContext(parameters);
    s->cc->Enable(PKE);
    s->cc->Enable(KEYSWITCH);
    s->cc->Enable(LEVELEDSHE);
    s->kp = s->cc->KeyGen();
    s->cc->EvalMultKeyGen(s->kp.secretKey);
    return s;
}
void ref_ckks_destroy(void* h) {
    auto* s = static_cast<RefCkks*>(h);
    s->cc->ClearEvalMultKeys();
    delete s;
}
// info[0]=N, [1]=sizeQ, [2]=sizeP, [3]=numPartQ, [4]=numPerPartQ(alpha)
void ref_ckks_info(void* h, uint32_t* info) {
    auto* s       = static_cast<RefCkks*>(h);
    const auto cp = std::dynamic_pointer_cast<CryptoParametersRNS>(s->cc->GetCryptoParameters());
    info[0]       = cp->GetElementParams()->GetRingDimension();
    info[1]       = cp->GetElementParams()->GetParams().size();
    info[2]       = cp->GetParamsP()->GetParams().size();
    info[3]       = cp->GetNumPartQ();
    info[4]       = cp->GetNumPerPartQ();
}
void ref_ckks_get_moduli(void* h, uint64_t* q, uint64_t* psiQ, uint64_t* p, uint64_t* psiP) {
    auto* s       = static_cast<RefCkks*>(h);
    const auto cp = std::dynamic_pointer_cast<CryptoParametersRNS>(s->cc->GetCryptoParameters());
    const auto& Q = cp->GetElementParams()->GetParams();
    const auto& P = cp->GetParamsP()->GetParams();
    for (size_t i = 0; i < Q.size(); ++i) {
        q[i]    = Q[i]->GetModulus().ConvertToInt<uint64_t>();
        psiQ[i] = Q[i]->GetRootOfUnity().ConvertToInt<uint64_t>();
    }
    for (size_t i = 0; i < P.size(); ++i) {
        p[i]    = P[i]->GetModulus().ConvertToInt<uint64_t>();
        psiP[i] = P[i]->GetRootOfUnity().ConvertToInt<uint64_t>();
    }
}
// relinearisation key: b/a vectors, each [numPartQ][sizeQ+sizeP][N]
void ref_ckks_get_relin_key(void* h, uint64_t* keyB, uint64_t* keyA) {
    auto* s        = static_cast<RefCkks*>(h);
    const auto& ek = CryptoContextImpl<DCRTPoly>::GetEvalMultKeyVector(s->kp.secretKey->GetKeyTag())[0];
    const auto& av = ek->GetAVector();
    const auto& bv = ek->GetBVector();
    size_t stride  = (size_t)av[0].GetNumOfElements() * av[0].GetRingDimension();
    for (size_t j = 0; j < av.size(); ++j) {
        export_poly(bv[j], keyB + j * stride);
        export_poly(av[j], keyA + j * stride);
    }
}
// PrecomputeCRTTables outputs for cross-checking the oracle's / product's table builders
void ref_ckks_get_tables(void* h, uint64_t* PInvModq, uint64_t* PHatInvModp, uint64_t* PHatModq /*[sizeP][sizeQ]*/) {
    auto* s       = static_cast<RefCkks*>(h);
    const auto cp = std::dynamic_pointer_cast<CryptoParametersRNS>(s->cc->GetCryptoParameters());
    size_t sizeQ = cp->GetElementParams()->GetParams().size(), sizeP = cp->GetParamsP()->GetParams().size();
    for (size_t i = 0; i < sizeQ; ++i)
        PInvModq[i] = cp->GetPInvModq()[i].ConvertToInt<uint64_t>();
    for (size_t j = 0; j < sizeP; ++j) {
        PHatInvModp[j] = cp->GetPHatInvModp()[j].ConvertToInt<uint64_t>();
        for (size_t i = 0; i < sizeQ; ++i)
            PHatModq[j * sizeQ + i] = cp->GetPHatModq()[j][i].ConvertToInt<uint64_t>();
    }
}
// PartQlHatInvModq(part, sizePartQl-1) and PartQlHatModp(sizeQl-1, part) ([sizePartQl][sizeCompl]); returns sizeCompl
uint32_t ref_ckks_get_part_tables(void* h, uint32_t part, uint32_t sizeQl, uint64_t* hatInv, uint64_t* hatModp,
                                  uint64_t* complModuli) {
    auto* s       = static_cast<RefCkks*>(h);
    const auto cp = std::dynamic_pointer_cast<CryptoParametersRNS>(s->cc->GetCryptoParameters());
    uint32_t alpha = cp->GetNumPerPartQ();
    uint32_t numPartQl = (sizeQl + alpha - 1) / alpha;
    if (numPartQl > cp->GetNumberOfQPartitions())
        numPartQl = cp->GetNumberOfQPartitions();
    uint32_t sizePartQl = (part == numPartQl - 1) ? sizeQl - alpha * part : alpha;
    const auto& hi = cp->GetPartQlHatInvModq(part, sizePartQl - 1);
    for (uint32_t i = 0; i < sizePartQl; ++i)
        hatInv[i] = hi[i].ConvertToInt<uint64_t>();
    const auto& hm    = cp->GetPartQlHatModp(sizeQl - 1, part);
    const auto& compl_ = cp->GetParamsComplPartQ(sizeQl - 1, part)->GetParams();
    uint32_t nc       = compl_.size();
    for (uint32_t i = 0; i < sizePartQl; ++i)
        for (uint32_t j = 0; j < nc; ++j)
            hatModp[(size_t)i * nc + j] = hm[i][j].ConvertToInt<uint64_t>();
    for (uint32_t j = 0; j < nc; ++j)
        complModuli[j] = compl_[j]->GetModulus().ConvertToInt<uint64_t>();
    return nc;
}
// rescale tables at the level with sizeQl limbs (ckksrns-cryptoparameters.cpp:60-81): index k = sizeQ - sizeQl
void ref_ckks_get_rescale_tables(void* h, uint32_t sizeQl, uint64_t* tabA, uint64_t* tabB) {
    auto* s       = static_cast<RefCkks*>(h);
    const auto cp = std::dynamic_pointer_cast<CryptoParametersCKKSRNS>(s->cc->GetCryptoParameters());
    size_t sizeQ  = cp->GetElementParams()->GetParams().size();
    const auto& A = cp->GetQlQlInvModqlDivqlModq(sizeQ - sizeQl);
    const auto& B = cp->GetqlInvModq(sizeQ - sizeQl);
    for (size_t i = 0; i + 1 < sizeQl; ++i) {
        tabA[i] = A[i].ConvertToInt<uint64_t>();
        tabB[i] = B[i].ConvertToInt<uint64_t>();
    }
}
// encrypt a deterministic message; returns ciphertext index in the session
int ref_ckks_encrypt(void* h, uint32_t seed, uint32_t level) {
    auto* s        = static_cast<RefCkks*>(h);
    uint32_t slots = s->cc->GetRingDimension() / 2;
    std::mt19937_64 gen(seed);
    std::uniform_real_distribution<double> dist(-1.0, 1.0);
    std::vector<double> v(slots);
    for (auto& e : v)
        e = dist(gen);
    Plaintext pt = s->cc->MakeCKKSPackedPlaintext(v, 1, level);
    s->cts.push_back(s->cc->Encrypt(s->kp.publicKey, pt));
    return static_cast<int>(s->cts.size()) - 1;
}
// info[0]=#elements, [1]=sizeQl, [2]=noiseScaleDeg, [3]=level
void ref_ct_info(void* h, int ct, uint32_t* info) {
    auto& c = static_cast<RefCkks*>(h)->cts[ct];
    info[0] = c->GetElements().size();
    info[1] = c->GetElements()[0].GetNumOfElements();
    info[2] = c->GetNoiseScaleDeg();
    info[3] = c->GetLevel();
}
void ref_ct_export(void* h, int ct, uint32_t elem, uint64_t* out) {
    export_poly(static_cast<RefCkks*>(h)->cts[ct]->GetElements()[elem], out);
}
int ref_ckks_eval_mult(void* h, int a, int b) {  // cryptocontext.h:1871-1879
    auto* s = static_cast<RefCkks*>(h);
    s->cts.push_back(s->cc->EvalMult(s->cts[a], s->cts[b]));
    return static_cast<int>(s->cts.size()) - 1;
}
int ref_ckks_eval_mult_no_relin(void* h, int a, int b) {
    auto* s = static_cast<RefCkks*>(h);
    s->cts.push_back(s->cc->EvalMultNoRelin(s->cts[a], s->cts[b]));
    return static_cast<int>(s->cts.size()) - 1;
}
int ref_ckks_eval_square_no_relin(void* h, int a) {  // EvalSquareCore through LeveledSHEBase::EvalSquare (base-leveledshe.cpp:646-700)
    auto* c = static_cast<RefCkks*>(h);
    c->cts.push_back(c->cc->GetScheme()->EvalSquare(c->cts[a]));
    return (int)c->cts.size() - 1;
}
int ref_ckks_rescale(void* h, int a) {
    auto* s = static_cast<RefCkks*>(h);
    s->cts.push_back(s->cc->Rescale(s->cts[a]));
    return static_cast<int>(s->cts.size()) - 1;
}
// time `reps` EvalMult calls on (a,b); returns seconds per call (cpu_baseline kind "reference")
double ref_ckks_time_eval_mult(void* h, int a, int b, int reps) {
    auto* s = static_cast<RefCkks*>(h);
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) {
        auto c = s->cc->EvalMult(s->cts[a], s->cts[b]);
        (void)c;
    }
    auto t1 = std::chrono::steady_clock::now();
    return std::chrono::duration<double>(t1 - t0).count() / reps;
}
void ref_ckks_decrypt(void* h, int ct, double* out, uint32_t n) {
    auto* s = static_cast<RefCkks*>(h);
    Plaintext pt;
    s->cc->Decrypt(s->kp.secretKey, s->cts[ct], &pt);
    pt->SetLength(n);
    auto v = pt->GetRealPackedValue();
    for (uint32_t i = 0; i < n && i < v.size(); ++i)
        out[i] = v[i];
}
// rotations: EvalAtIndexKeyGen + key export + EvalRotate / hoisted EvalFastRotation
void ref_ckks_rotate_keygen(void* h, const int32_t* indices, uint32_t n) {
    auto* s = static_cast<RefCkks*>(h);
    s->cc->EvalRotateKeyGen(s->kp.secretKey, std::vector<int32_t>(indices, indices + n));
}
// returns the automorphism index k of rotation `index`; fills the key vectors [numPartQ][sizeQ+sizeP][N]
uint32_t ref_ckks_get_rot_key(void* h, int32_t index, uint64_t* keyB, uint64_t* keyA) {
    auto* s     = static_cast<RefCkks*>(h);
    uint32_t M  = s->cc->GetCyclotomicOrder();
    uint32_t k  = FindAutomorphismIndex2nComplex(index, M);
    auto& km    = CryptoContextImpl<DCRTPoly>::GetEvalAutomorphismKeyMap(s->kp.secretKey->GetKeyTag());
    const auto& ek = km.at(k);
    const auto& av = ek->GetAVector();
    const auto& bv = ek->GetBVector();
    size_t stride  = (size_t)av[0].GetNumOfElements() * av[0].GetRingDimension();
    for (size_t j = 0; j < av.size(); ++j) {
        export_poly(bv[j], keyB + j * stride);
        export_poly(av[j], keyA + j * stride);
    }
    return k;
}
int ref_ckks_eval_rotate(void* h, int ct, int32_t index) {
    auto* s = static_cast<RefCkks*>(h);
    s->cts.push_back(s->cc->EvalRotate(s->cts[ct], index));
    return static_cast<int>(s->cts.size()) - 1;
}
int ref_ckks_eval_fast_rotate(void* h, int ct, int32_t index) {  // hoisted: precompute digits once, then rotate
    auto* s     = static_cast<RefCkks*>(h);
    auto digits = s->cc->EvalFastRotationPrecompute(s->cts[ct]);
    s->cts.push_back(s->cc->EvalFastRotation(s->cts[ct], index, s->cc->GetCyclotomicOrder(), digits));
    return static_cast<int>(s->cts.size()) - 1;
}
// double hoisting building blocks: EvalFastRotationExt (result in Q_l u P) and KeySwitchDown
int ref_ckks_eval_fast_rotate_ext(void* h, int ct, int32_t index, int addFirst) {
    auto* s     = static_cast<RefCkks*>(h);
    auto digits = s->cc->EvalFastRotationPrecompute(s->cts[ct]);
    s->cts.push_back(s->cc->EvalFastRotationExt(s->cts[ct], index, digits, addFirst != 0));
    return static_cast<int>(s->cts.size()) - 1;
}
int ref_ckks_key_switch_down(void* h, int ctExt) {
    auto* s = static_cast<RefCkks*>(h);
    s->cts.push_back(s->cc->KeySwitchDown(s->cts[ctExt]));
    return static_cast<int>(s->cts.size()) - 1;
}

// ---- BSGS linear transform with double hoisting: the reference's own FHECKKSRNS::EvalLinearTransform ----
// (the FHE object is a protected member of SchemeBase: reached through a derived accessor, not modified)
namespace {
struct SchemeFheAccess : SchemeBase<DCRTPoly> {
    static std::shared_ptr<FHEBase<DCRTPoly>> get(const SchemeBase<DCRTPoly>& s) { return s.*(&SchemeFheAccess::m_FHE); }
};
std::shared_ptr<FHECKKSRNS> fhe_of(const CryptoContext<DCRTPoly>& cc) {
    return std::dynamic_pointer_cast<FHECKKSRNS>(SchemeFheAccess::get(*cc->GetScheme()));
}
struct RefLt {
    std::vector<ReadOnlyPlaintext> A;
    uint32_t slots, bStep;
};
}  // namespace
// bootstrapping parameters only (no plaintext precomputation): level budget {1,1}, baby step = dim1 = bStep
void* ref_ckks_lt_create(void* h, uint32_t slots, uint32_t bStep, const double* m /*[slots][slots] (re, im)*/, uint32_t L) {
    auto* s = static_cast<RefCkks*>(h);
    s->cc->Enable(ADVANCEDSHE);
    s->cc->Enable(FHE);
    s->cc->EvalBootstrapSetup({1, 1}, {bStep, bStep}, slots, 0, false);
    std::vector<std::vector<std::complex<double>>> M(slots, std::vector<std::complex<double>>(slots));
    for (uint32_t i = 0; i < slots; ++i)
        for (uint32_t j = 0; j < slots; ++j)
            M[i][j] = {m[2 * ((size_t)i * slots + j)], m[2 * ((size_t)i * slots + j) + 1]};
    auto* lt  = new RefLt;
    lt->slots = slots, lt->bStep = bStep;
    lt->A     = fhe_of(s->cc)->EvalLinearTransformPrecompute(*s->cc, M, 1.0, L);
    return lt;
}
void ref_ckks_lt_destroy(void* l) { delete static_cast<RefLt*>(l); }
// diagonal i as [sizeQl+sizeP][N] EVALUATION residues; returns its number of limbs
uint32_t ref_ckks_lt_get_diag(void* l, uint32_t i, uint64_t* out) {
    auto* lt = static_cast<RefLt*>(l);
    auto pt  = lt->A[i]->GetElement<DCRTPoly>();
    pt.SetFormat(Format::EVALUATION);
    if (out)
        export_poly(pt, out);
    return pt.GetNumOfElements();
}
int ref_ckks_eval_linear_transform(void* h, void* l, int ct) {
    auto* s  = static_cast<RefCkks*>(h);
    auto* lt = static_cast<RefLt*>(l);
    ConstCiphertext<DCRTPoly> c = s->cts[ct];
    s->cts.push_back(fhe_of(s->cc)->EvalLinearTransform(lt->A, c));
    return static_cast<int>(s->cts.size()) - 1;
}
double ref_ckks_time_linear_transform(void* h, void* l, int ct, int reps) {
    auto* s  = static_cast<RefCkks*>(h);
    auto* lt = static_cast<RefLt*>(l);
    ConstCiphertext<DCRTPoly> c = s->cts[ct];
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) {
        auto out = fhe_of(s->cc)->EvalLinearTransform(lt->A, c);
        (void)out;
    }
    auto t1 = std::chrono::steady_clock::now();
    return std::chrono::duration<double>(t1 - t0).count() / reps;
}
// an encryption with `slots` packed values (sparse packing when slots < N/2)
int ref_ckks_encrypt_slots(void* h, const double* vals /*[slots] (re, im)*/, uint32_t level, uint32_t slots) {
    auto* s = static_cast<RefCkks*>(h);
    std::vector<std::complex<double>> v(slots);
    for (uint32_t i = 0; i < slots; ++i)
        v[i] = {vals[2 * i], vals[2 * i + 1]};
    auto pt = s->cc->MakeCKKSPackedPlaintext(v, 1, level, nullptr, slots);
    s->cts.push_back(s->cc->Encrypt(s->kp.publicKey, pt));
    return static_cast<int>(s->cts.size()) - 1;
}
void ref_ckks_decrypt_complex(void* h, int ct, double* out, uint32_t n) {
    auto* s = static_cast<RefCkks*>(h);
    Plaintext pt;
    s->cc->Decrypt(s->kp.secretKey, s->cts[ct], &pt);
    pt->SetLength(n);
    auto v = pt->GetCKKSPackedValue();
    for (uint32_t i = 0; i < n && i < v.size(); ++i)
        out[2 * i] = v[i].real(), out[2 * i + 1] = v[i].imag();
}

// ---- EvalCoeffsToSlots (the multi-level, FFT-like linear transform of bootstrapping) on the reference itself ----
namespace {
struct RefC2S {
    std::vector<std::vector<ReadOnlyPlaintext>> A;
    ckks_boot_params p;
};
}  // namespace
// level budget `budget` for encoding; plaintexts via EvalCoeffsToSlotsPrecompute with the roots-of-unity tables that
// EvalBootstrapSetup builds for it (ckksrns-fhe.cpp:144-166); L as in that function (Q limbs left after the transform)
static void* ref_ckks_fft_create(void* h, uint32_t slots, uint32_t budget, uint32_t L, bool decode);
void* ref_ckks_c2s_create(void* h, uint32_t slots, uint32_t budget, uint32_t L) {
    return ref_ckks_fft_create(h, slots, budget, L, false);
}
// the decoding direction: EvalSlotsToCoeffsPrecompute / EvalSlotsToCoeffs (ckksrns-fhe.cpp:1670-1830, 2041-2198)
void* ref_ckks_s2c_create(void* h, uint32_t slots, uint32_t budget, uint32_t L) {
    return ref_ckks_fft_create(h, slots, budget, L, true);
}
static void* ref_ckks_fft_create(void* h, uint32_t slots, uint32_t budget, uint32_t L, bool decode) {
    auto* s = static_cast<RefCkks*>(h);
    s->cc->Enable(ADVANCEDSHE);
    s->cc->Enable(FHE);
    s->cc->EvalBootstrapSetup({budget, budget}, {0, 0}, slots, 0, false);
    const uint32_t m = 4 * slots;
    std::vector<uint32_t> rotGroup(slots);
    uint32_t fivePows = 1;
    for (uint32_t i = 0; i < slots; ++i) {
        rotGroup[i] = fivePows;
        fivePows    = (fivePows * 5) & (m - 1);
    }
    std::vector<std::complex<double>> ksiPows(m + 1);
    for (uint32_t j = 0; j < m; ++j)
        ksiPows[j] = {std::cos(2 * M_PI * j / m), std::sin(2 * M_PI * j / m)};
    ksiPows[m] = ksiPows[0];
    auto* c = new RefC2S;
    c->A    = decode ? fhe_of(s->cc)->EvalSlotsToCoeffsPrecompute(*s->cc, ksiPows, rotGroup, false, 1.0, L, false) :
                       fhe_of(s->cc)->EvalCoeffsToSlotsPrecompute(*s->cc, ksiPows, rotGroup, false, 1.0, L, false);
    c->p    = GetCollapsedFFTParams(slots, budget, 0);
    return c;
}
void ref_ckks_c2s_destroy(void* c) { delete static_cast<RefC2S*>(c); }
void ref_ckks_c2s_params(void* c, uint32_t* out /*[9]*/) {
    const auto& p = static_cast<RefC2S*>(c)->p;
    const uint32_t v[9] = {p.lvlb, p.layersCollapse, p.remCollapse, p.numRotations, p.b, p.g, p.numRotationsRem, p.bRem, p.gRem};
    std::copy(v, v + 9, out);
}
// plaintext A[s][idx] as [limbs][N] EVALUATION residues; returns its number of limbs, 0 if the slot is empty
uint32_t ref_ckks_c2s_get_diag(void* c, uint32_t s, uint32_t idx, uint64_t* out) {
    auto& A = static_cast<RefC2S*>(c)->A;
    if (s >= A.size() || idx >= A[s].size() || !A[s][idx])
        return 0;
    auto pt = A[s][idx]->GetElement<DCRTPoly>();
    pt.SetFormat(Format::EVALUATION);
    if (out)
        export_poly(pt, out);
    return pt.GetNumOfElements();
}
int ref_ckks_eval_slots_to_coeffs(void* h, void* c, int ct) {
    auto* s = static_cast<RefCkks*>(h);
    ConstCiphertext<DCRTPoly> x = s->cts[ct];
    s->cts.push_back(fhe_of(s->cc)->EvalSlotsToCoeffs(static_cast<RefC2S*>(c)->A, x));
    return static_cast<int>(s->cts.size()) - 1;
}
int ref_ckks_eval_coeffs_to_slots(void* h, void* c, int ct) {
    auto* s = static_cast<RefCkks*>(h);
    ConstCiphertext<DCRTPoly> x = s->cts[ct];
    s->cts.push_back(fhe_of(s->cc)->EvalCoeffsToSlots(static_cast<RefC2S*>(c)->A, x));
    return static_cast<int>(s->cts.size()) - 1;
}
int ref_omp_threads() { return OpenFHEParallelControls.GetNumThreads(); }

// ---- ScaleAndRound family with caller tables ----
// x is [sizeI+sizeO][N] COEFF over `moduli` (in x's limb order); output basis = first sizeO limbs if outputFirst else last sizeO
void ref_scale_and_round(uint32_t N, uint32_t sizeI, uint32_t sizeO, int outputFirst, const uint64_t* moduli,
                         const uint64_t* roots, const uint64_t* x, const uint64_t* tab, const double* frac, uint64_t* out) {
    auto pAll = make_params(N, sizeI + sizeO, moduli, roots);
    uint32_t off = outputFirst ? 0 : sizeI;
    auto pOut = make_params(N, sizeO, moduli + off, roots + off);
    auto X    = make_poly(pAll, x, Format::COEFFICIENT);
    std::vector<std::vector<NativeInteger>> t(sizeO);
    for (uint32_t j = 0; j < sizeO; ++j)
        t[j] = vecNI(tab + (size_t)j * (sizeI + 1), sizeI + 1);
    std::vector<double> f(frac, frac + sizeI);
    export_poly(X.ScaleAndRound(pOut, t, f, mu128(moduli + off, sizeO)), out);
}
void ref_approx_scale_and_round(uint32_t N, uint32_t sizeQ, uint32_t sizeP, const uint64_t* moduli, const uint64_t* roots,
                                const uint64_t* x, const uint64_t* tab, uint64_t* out) {
    auto pAll = make_params(N, sizeQ + sizeP, moduli, roots);
    auto pP   = make_params(N, sizeP, moduli + sizeQ, roots + sizeQ);
    auto X    = make_poly(pAll, x, Format::COEFFICIENT);
    std::vector<std::vector<NativeInteger>> t(sizeP);
    for (uint32_t j = 0; j < sizeP; ++j)
        t[j] = vecNI(tab + (size_t)j * (sizeQ + 1), sizeQ + 1);
    export_poly(X.ApproxScaleAndRound(pP, t, mu128(moduli + sizeQ, sizeP)), out);
}
void ref_scale_and_round_p_over_q(uint32_t N, uint32_t sizeQ, const uint64_t* moduli, const uint64_t* roots,
                                  const uint64_t* x, const uint64_t* pInvModq, uint64_t* out) {
    auto pAll = make_params(N, sizeQ + 1, moduli, roots);
    auto pQ   = make_params(N, sizeQ, moduli, roots);
    auto X    = make_poly(pAll, x, Format::COEFFICIENT);
    X.ScaleAndRoundPOverQ(pQ, vecNI(pInvModq, sizeQ));
    export_poly(X, out);
}

void ref_times_q_over_t(uint32_t N, uint32_t L, const uint64_t* q, const uint64_t* psi, uint64_t* x, uint64_t t, uint64_t negQModt,
                        const uint64_t* tInvModq) {
    auto pq = make_params(N, L, q, psi);
    auto X  = make_poly(pq, x, Format::COEFFICIENT);
    const NativeInteger T(t), NQ(negQModt);
    X.TimesQovert(pq, vecNI(tInvModq, L), T, NQ, NQ.PrepModMulConst(T));
    export_poly(X, x);
}
void ref_set_values_mod_switch(uint32_t N, uint64_t qFrom, uint64_t psiFrom, const uint64_t* x, uint64_t qTo, uint64_t psiTo, uint64_t* out) {
    auto pf = make_params(N, 1, &qFrom, &psiFrom), pt = make_params(N, 1, &qTo, &psiTo);
    auto X  = make_poly(pf, x, Format::COEFFICIENT);
    DCRTPoly Y(pt, Format::COEFFICIENT, true);
    Y.SetValuesModSwitch(X, NativeInteger(qTo));
    export_poly(Y, out);
}

// ScaleAndRound -> NativePoly mod t (decryption) with caller tables; out [N]
void ref_scale_and_round_native(uint32_t N, uint32_t sizeQ, const uint64_t* q, const uint64_t* psi, const uint64_t* x, uint64_t t,
                                const uint64_t* tabModt, const uint64_t* tabBModt, const double* frac, const double* bfrac,
                                uint64_t* out) {
    auto pq = make_params(N, sizeQ, q, psi);
    auto X  = make_poly(pq, x, Format::COEFFICIENT);
    const NativeInteger T(t);
    auto a = vecNI(tabModt, sizeQ), b = vecNI(tabBModt, sizeQ);
    std::vector<NativeInteger> ap(sizeQ), bp(sizeQ);
    for (uint32_t i = 0; i < sizeQ; ++i) {
        ap[i] = a[i].PrepModMulConst(T);
        bp[i] = b[i].PrepModMulConst(T);
    }
    std::vector<double> f(frac, frac + sizeQ), bf(bfrac, bfrac + sizeQ);
    auto r = X.ScaleAndRound(T, a, ap, b, bp, f, bf);
    for (uint32_t k = 0; k < N; ++k)
        out[k] = r[k].ConvertToInt<uint64_t>();
}
void ref_scale_and_round_behz_decrypt(uint32_t N, uint32_t sizeQ, const uint64_t* q, const uint64_t* psi, const uint64_t* x,
                                      uint64_t t, uint64_t tgamma, const uint64_t* tgammaQHatModq,
                                      const uint64_t* negInvqModtgamma, uint64_t* out) {
    auto pq = make_params(N, sizeQ, q, psi);
    auto X  = make_poly(pq, x, Format::COEFFICIENT);
    const NativeInteger TG(tgamma);
    auto a = vecNI(tgammaQHatModq, sizeQ), b = vecNI(negInvqModtgamma, sizeQ);
    std::vector<NativeInteger> bp(sizeQ);
    for (uint32_t i = 0; i < sizeQ; ++i)
        bp[i] = b[i].PrepModMulConst(TG);
    auto r = X.ScaleAndRound(vecNI(q, sizeQ), NativeInteger(t), TG, a, precon(a, q), b, bp);
    for (uint32_t k = 0; k < N; ++k)
        out[k] = r[k].ConvertToInt<uint64_t>();
}

// ---- BFV / BEHZ session: the reference's own CryptoParametersBFVRNS tables ----
struct RefBfv {
    CryptoContext<DCRTPoly> cc;
    KeyPair<DCRTPoly> kp;
    std::vector<Ciphertext<DCRTPoly>> cts;
};
void* ref_bfv_create(uint32_t ringDim, uint64_t t, uint32_t multDepth, uint32_t scalingModSize, int multTech) {
    CCParams<CryptoContextBFVRNS> parameters;
    parameters.SetSecurityLevel(HEStd_NotSet);
    parameters.SetRingDim(ringDim);
    parameters.SetPlaintextModulus(t);
    parameters.SetMultiplicativeDepth(multDepth);
    parameters.SetScalingModSize(scalingModSize);
    parameters.SetMultiplicationTechnique(static_cast<MultiplicationTechnique>(multTech));
    auto* s = new RefBfv;
    s->cc   = GenCryptoContext(parameters);
    s->cc->Enable(PKE);
    s->cc->Enable(KEYSWITCH);
    s->cc->Enable(LEVELEDSHE);
    return s;
}
// BFV with HYBRID key switching (relinearisation path of config 5)
void* ref_bfv_create_hybrid(uint32_t ringDim, uint64_t t, uint32_t multDepth, uint32_t scalingModSize, uint32_t numLargeDigits) {
    CCParams<CryptoContextBFVRNS> parameters;
    parameters.SetSecurityLevel(HEStd_NotSet);
    parameters.SetRingDim(ringDim);
    parameters.SetPlaintextModulus(t);
    parameters.SetMultiplicativeDepth(multDepth);
    parameters.SetScalingModSize(scalingModSize);
    parameters.SetMultiplicationTechnique(BEHZ);
    parameters.SetKeySwitchTechnique(HYBRID);
    if (numLargeDigits > 0)
        parameters.SetNumLargeDigits(numLargeDigits);
    auto* s = new RefBfv;
    s->cc   = GenCryptoContext(parameters);
    s->cc->Enable(PKE);
    s->cc->Enable(KEYSWITCH);
    s->cc->Enable(LEVELEDSHE);
    s->kp = s->cc->KeyGen();
    s->cc->EvalMultKeyGen(s->kp.secretKey);
    return s;
}
// info[0]=sizeP, [1]=numPartQ, [2]=numPerPartQ
void ref_bfv_hybrid_info(void* h, uint32_t* info) {
    const auto cp = std::dynamic_pointer_cast<CryptoParametersRNS>(static_cast<RefBfv*>(h)->cc->GetCryptoParameters());
    info[0]       = cp->GetParamsP()->GetParams().size();
    info[1]       = cp->GetNumPartQ();
    info[2]       = cp->GetNumPerPartQ();
}
void ref_bfv_get_p(void* h, uint64_t* p, uint64_t* psiP) {
    const auto cp = std::dynamic_pointer_cast<CryptoParametersRNS>(static_cast<RefBfv*>(h)->cc->GetCryptoParameters());
    const auto& P = cp->GetParamsP()->GetParams();
    for (size_t i = 0; i < P.size(); ++i) {
        p[i]    = P[i]->GetModulus().ConvertToInt<uint64_t>();
        psiP[i] = P[i]->GetRootOfUnity().ConvertToInt<uint64_t>();
    }
}
void ref_bfv_get_relin_key(void* h, uint64_t* keyB, uint64_t* keyA) {
    auto* s        = static_cast<RefBfv*>(h);
    const auto& ek = CryptoContextImpl<DCRTPoly>::GetEvalMultKeyVector(s->kp.secretKey->GetKeyTag())[0];
    const auto& av = ek->GetAVector();
    const auto& bv = ek->GetBVector();
    size_t stride  = (size_t)av[0].GetNumOfElements() * av[0].GetRingDimension();
    for (size_t j = 0; j < av.size(); ++j) {
        export_poly(bv[j], keyB + j * stride);
        export_poly(av[j], keyA + j * stride);
    }
}
int ref_bfv_eval_mult(void* h, int a, int b) {  // cc->EvalMult: EvalMultNoRelin + SetFormat + KeySwitchCore + adds
    auto* s = static_cast<RefBfv*>(h);
    s->cts.push_back(s->cc->EvalMult(s->cts[a], s->cts[b]));
    return static_cast<int>(s->cts.size()) - 1;
}
double ref_bfv_time_eval_mult(void* h, int a, int b, int reps) {  // with relinearisation
    auto* s = static_cast<RefBfv*>(h);
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) {
        auto c = s->cc->EvalMult(s->cts[a], s->cts[b]);
        (void)c;
    }
    auto t1 = std::chrono::steady_clock::now();
    return std::chrono::duration<double>(t1 - t0).count() / reps;
}
void ref_bfv_destroy(void* h) {
    auto* s = static_cast<RefBfv*>(h);
    s->cc->ClearEvalMultKeys();
    delete s;
}
// ciphertext session (config 5): keys, deterministic packed messages, EvalMultNoRelin through the scheme layer
void ref_bfv_keygen(void* h) {
    auto* s = static_cast<RefBfv*>(h);
    s->kp   = s->cc->KeyGen();
}
int ref_bfv_encrypt(void* h, uint32_t seed) {
    auto* s    = static_cast<RefBfv*>(h);
    uint64_t t = s->cc->GetCryptoParameters()->GetPlaintextModulus();
    std::mt19937_64 gen(seed);
    std::vector<int64_t> v(s->cc->GetRingDimension());
    for (auto& e : v)
        e = static_cast<int64_t>(gen() % t) - static_cast<int64_t>(t / 2);
    s->cts.push_back(s->cc->Encrypt(s->kp.publicKey, s->cc->MakePackedPlaintext(v)));
    return static_cast<int>(s->cts.size()) - 1;
}
// info[0]=#elements, [1]=#limbs, [2]=format of element 0 (0 = EVALUATION, 1 = COEFFICIENT)
void ref_bfv_ct_info(void* h, int ct, uint32_t* info) {
    auto& c = static_cast<RefBfv*>(h)->cts[ct];
    info[0] = c->GetElements().size();
    info[1] = c->GetElements()[0].GetNumOfElements();
    info[2] = c->GetElements()[0].GetFormat() == Format::EVALUATION ? 0 : 1;
}
void ref_bfv_ct_export(void* h, int ct, uint32_t elem, uint64_t* out) {
    export_poly(static_cast<RefBfv*>(h)->cts[ct]->GetElements()[elem], out);
}
int ref_bfv_eval_mult_no_relin(void* h, int a, int b) {  // bfvrns-leveledshe.cpp:198-445
    auto* s = static_cast<RefBfv*>(h);
    s->cts.push_back(s->cc->EvalMultNoRelin(s->cts[a], s->cts[b]));
    return static_cast<int>(s->cts.size()) - 1;
}
double ref_bfv_time_eval_mult_no_relin(void* h, int a, int b, int reps) {
    auto* s = static_cast<RefBfv*>(h);
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) {
        auto c = s->cc->EvalMultNoRelin(s->cts[a], s->cts[b]);
        (void)c;
    }
    auto t1 = std::chrono::steady_clock::now();
    return std::chrono::duration<double>(t1 - t0).count() / reps;
}
// info[0]=N, [1]=numQ, [2]=numBsk
void ref_bfv_info(void* h, uint32_t* info) {
    const auto cp = std::dynamic_pointer_cast<CryptoParametersBFVRNS>(static_cast<RefBfv*>(h)->cc->GetCryptoParameters());
    info[0]       = cp->GetElementParams()->GetRingDimension();
    info[1]       = cp->GetModuliQ().size();
    info[2]       = cp->GetModuliBsk().size();
}
void ref_bfv_get_moduli(void* h, uint64_t* q, uint64_t* psiQ, uint64_t* bsk, uint64_t* psiBsk) {
    const auto cp = std::dynamic_pointer_cast<CryptoParametersBFVRNS>(static_cast<RefBfv*>(h)->cc->GetCryptoParameters());
    const auto& P = cp->GetParamsQBsk()->GetParams();
    size_t numQ   = cp->GetModuliQ().size();
    for (size_t i = 0; i < P.size(); ++i) {
        uint64_t m = P[i]->GetModulus().ConvertToInt<uint64_t>(), r = P[i]->GetRootOfUnity().ConvertToInt<uint64_t>();
        if (i < numQ) {
            q[i]    = m;
            psiQ[i] = r;
        }
        else {
            bsk[i - numQ]    = m;
            psiBsk[i - numQ] = r;
        }
    }
}
// x [numQ][N] in `evalFormat` -> out [numQ+numBsk][N] EVALUATION   (dcrtpoly-impl.h:1694-1786)
void ref_bfv_behz_q_to_bsk(void* h, const uint64_t* x, int evalFormat, uint64_t* out) {
    const auto cp = std::dynamic_pointer_cast<CryptoParametersBFVRNS>(static_cast<RefBfv*>(h)->cc->GetCryptoParameters());
    auto a = make_poly(cp->GetElementParams(), x, evalFormat ? Format::EVALUATION : Format::COEFFICIENT);
    a.FastBaseConvqToBskMontgomery(cp->GetParamsQBsk(), cp->GetModuliQ(), cp->GetModuliBsk(), cp->GetModbskBarrettMu(),
                                   cp->GetmtildeQHatInvModq(), cp->GetmtildeQHatInvModqPrecon(), cp->GetQHatModbsk(),
                                   cp->GetQHatModmtilde(), cp->GetQModbsk(), cp->GetQModbskPrecon(),
                                   cp->GetNegQInvModmtilde(), cp->GetmtildeInvModbsk(), cp->GetmtildeInvModbskPrecon());
    export_poly(a, out);
}
// x [numQ+numBsk][N] COEFFICIENT, in place   (dcrtpoly-impl.h:1791-1840)
void ref_bfv_fast_rns_floorq(void* h, uint64_t* x) {
    const auto cp = std::dynamic_pointer_cast<CryptoParametersBFVRNS>(static_cast<RefBfv*>(h)->cc->GetCryptoParameters());
    auto a = make_poly(cp->GetParamsQBsk(), x, Format::COEFFICIENT);
    a.FastRNSFloorq(cp->GetPlaintextModulus(), cp->GetModuliQ(), cp->GetModuliBsk(), cp->GetModbskBarrettMu(),
                    cp->GettQHatInvModq(), cp->GettQHatInvModqPrecon(), cp->GetQHatModbsk(), cp->GetqInvModbsk(),
                    cp->GettQInvModbsk(), cp->GettQInvModbskPrecon());
    export_poly(a, x);
}
// x [numQ+numBsk][N] COEFFICIENT -> out [numQ][N]   (dcrtpoly-impl.h:1845-1929)
void ref_bfv_fast_base_conv_sk(void* h, const uint64_t* x, uint64_t* out) {
    const auto cp = std::dynamic_pointer_cast<CryptoParametersBFVRNS>(static_cast<RefBfv*>(h)->cc->GetCryptoParameters());
    auto a = make_poly(cp->GetParamsQBsk(), x, Format::COEFFICIENT);
    a.FastBaseConvSK(cp->GetElementParams(), cp->GetModqBarrettMu(), cp->GetModuliBsk(), cp->GetModbskBarrettMu(),
                     cp->GetBHatInvModb(), cp->GetBHatInvModbPrecon(), cp->GetBHatModmsk(), cp->GetBInvModmsk(),
                     cp->GetBInvModmskPrecon(), cp->GetBHatModq(), cp->GetBModq(), cp->GetBModqPrecon());
    export_poly(a, out);
}

}  // extern "C"
