/*
 * fhe_oracle.h — TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the reference algorithms on the lbcrypto::DCRTPoly hot path
 * (SURVEY.md §8a rows a1..a18).  It is the parity checker for the HIP library: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The product library
 * (libfhe_hip.so) never links, imports or falls back to anything in oracle/.
 *
 * Parity status: PINNED — checked against the reference's own known-answer vectors
 * (tests/golden/ fixtures, from /root/reference/src/core/unittest and src/pke/unittest) and against
 * the reference itself compiled from its sources into oracle/_ref/ (tests/test_oracle_vs_ref.py).
 *
 * All reference citations are relative to /root/reference/.
 * Data layout everywhere: a tower is uint64_t[nLimbs][N], limb-major, N contiguous.
 */
#ifndef FHE_ORACLE_H
#define FHE_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------- a1/a2: scalar modular arithmetic ---------------- */
uint64_t orc_mulmod(uint64_t a, uint64_t b, uint64_t q); /* exact a*b mod q (setup use) */
uint64_t orc_powmod(uint64_t a, uint64_t e, uint64_t q);
uint64_t orc_invmod(uint64_t a, uint64_t q);             /* q prime */
uint32_t orc_get_msb(uint64_t x);                        /* nbtheory.h:169-186 (1-indexed bit length) */
uint64_t orc_compute_mu(uint64_t q);                     /* ubintnat.h:642-647 */
uint64_t orc_mod_mul_fast(uint64_t a, uint64_t b, uint64_t q, uint64_t mu);            /* :1348-1361 */
uint64_t orc_prep_mod_mul_const(uint64_t b, uint64_t q);                               /* :1437-1444 */
uint64_t orc_mod_mul_fast_const(uint64_t a, uint64_t b, uint64_t q, uint64_t bPrecon); /* :1464-1469 */
uint64_t orc_mod_add_fast(uint64_t a, uint64_t b, uint64_t q);                         /* :737-743 */
uint64_t orc_mod_sub_fast(uint64_t a, uint64_t b, uint64_t q);                         /* :911-921 */
void     orc_barrett_mu128(uint64_t q, uint64_t mu[2]); /* floor(2^128/q): mu[0]=lo, mu[1]=hi */
uint64_t orc_barrett128(uint64_t a_lo, uint64_t a_hi, uint64_t q, uint64_t mu_lo, uint64_t mu_hi);
                                                        /* utilities-int.h:60-99 */

/* ---------------- number theory (host-side parameter reproduction) ---------------- */
int      orc_is_prime(uint64_t n);
uint64_t orc_first_prime(uint32_t nBits, uint64_t m);    /* nbtheory-impl.h:329-347 */
uint64_t orc_last_prime(uint32_t nBits, uint64_t m);     /* :349-373 */
uint64_t orc_next_prime(uint64_t q, uint64_t m);         /* :375-383 */
uint64_t orc_previous_prime(uint64_t q, uint64_t m);     /* :385-393 */
uint64_t orc_root_of_unity(uint64_t m, uint64_t q);      /* :183-231 (minimum primitive m-th root) */
uint32_t orc_reverse_bits(uint32_t x, uint32_t nbits);   /* nbtheory.h:135-157 */
void     orc_precompute_auto_map(uint32_t n, uint32_t k, uint32_t* precomp); /* nbtheory2.cpp:264-275 */
uint32_t orc_find_automorphism_index_2n_complex(int32_t i, uint32_t m);      /* nbtheory2.cpp:243-262 */
/* ILDCRTParams(order, depth, bits) modulus chain: ildcrtparams.h:100-117 */
void     orc_dcrt_params(uint32_t order, uint32_t nLimbs, uint32_t bits, uint64_t* q, uint64_t* psi);

/* ---------------- a3/a4/a5: negacyclic NTT ---------------- */
/* tables are uint64_t[N] each; nInv = N^-1 mod q.  transformnat-impl.h:714-756 */
void orc_ntt_precompute(uint64_t q, uint64_t psi, uint32_t N, uint64_t* tbl, uint64_t* tblPrecon,
                        uint64_t* tblInv, uint64_t* tblInvPrecon, uint64_t* nInv, uint64_t* nInvPrecon);
/* in place, natural -> bit-reversed.  transformnat-impl.h:303-374 */
void orc_ntt_fwd(uint64_t* x, uint32_t N, uint64_t q, const uint64_t* tbl, const uint64_t* tblPrecon);
/* in place, bit-reversed -> natural, 1/N folded in.  transformnat-impl.h:512-625 */
void orc_ntt_inv(uint64_t* x, uint32_t N, uint64_t q, const uint64_t* tblInv, const uint64_t* tblInvPrecon,
                 uint64_t nInv, uint64_t nInvPrecon);

/* A tower context = what DCRTPoly::Params + the static twiddle cache hold. */
typedef struct orc_ctx orc_ctx;
orc_ctx* orc_ctx_create(uint32_t N, uint32_t nLimbs, const uint64_t* q, const uint64_t* psi);
void     orc_ctx_destroy(orc_ctx*);
uint32_t orc_ctx_n(const orc_ctx*);
uint32_t orc_ctx_limbs(const orc_ctx*);
uint64_t orc_ctx_modulus(const orc_ctx*, uint32_t limb);
/* a6: DCRTPoly::SwitchFormat over a batch of towers x[batch][nSel][N]; limbIdx selects context limbs.
 * nThreads<=0 -> omp default (the reference's `num_threads(0)` behaviour, dcrtpoly-impl.h:1932-1940) */
void orc_ntt_fwd_tower(const orc_ctx*, uint64_t* x, const uint32_t* limbIdx, uint32_t nSel, uint32_t batch, int nThreads);
void orc_ntt_inv_tower(const orc_ctx*, uint64_t* x, const uint32_t* limbIdx, uint32_t nSel, uint32_t batch, int nThreads);

/* ---------------- a7: element-wise vector ops (mubintvecnat.cpp:132-142,229-339) ---------------- */
void orc_vec_add(uint64_t* out, const uint64_t* a, const uint64_t* b, size_t n, uint64_t q);
void orc_vec_sub(uint64_t* out, const uint64_t* a, const uint64_t* b, size_t n, uint64_t q);
void orc_vec_mul(uint64_t* out, const uint64_t* a, const uint64_t* b, size_t n, uint64_t q);
void orc_vec_mul_const(uint64_t* out, const uint64_t* a, uint64_t c, size_t n, uint64_t q);
void orc_vec_add_const(uint64_t* out, const uint64_t* a, uint64_t c, size_t n, uint64_t q, int coeff0Only);
void orc_vec_sub_const(uint64_t* out, const uint64_t* a, uint64_t c, size_t n, uint64_t q);
void orc_vec_inner_product(uint64_t* out, const uint64_t* const* x, const uint64_t* const* k, uint32_t nTerms, size_t n, uint64_t q);
void orc_vec_mult_acc(uint64_t* acc, const uint64_t* v, uint64_t c, size_t n, uint64_t q);
void orc_vec_neg(uint64_t* out, const uint64_t* a, size_t n, uint64_t q);

/* ---------------- a8: automorphism (poly-impl.h:310-376) ---------------- */
void orc_automorph_eval(uint64_t* out, const uint64_t* in, uint32_t N, const uint32_t* precomp);
void orc_automorph_eval_k(uint64_t* out, const uint64_t* in, uint32_t N, uint32_t k);
void orc_automorph_coeff(uint64_t* out, const uint64_t* in, uint32_t N, uint32_t k, uint64_t q);

/* ---------------- a9: centred modulus switch (mubintvecnat.cpp:109-122) ---------------- */
void orc_switch_modulus(uint64_t* v, size_t n, uint64_t oldq, uint64_t newq);

/* ---------------- a10: ApproxSwitchCRTBasis fast path (dcrtpoly-impl.h:888-915) ----------------
 * x[sizeQ][N] (COEFF) -> out[sizeP][N].  QHatModp is [sizeQ][sizeP]; mu128 is [sizeP][2] (lo,hi). */
void orc_approx_switch_crt_basis(const uint64_t* x, uint32_t sizeQ, uint32_t N, const uint64_t* q,
                                 const uint64_t* QHatInvModq, const uint64_t* QHatInvModqPrecon,
                                 const uint64_t* QHatModp, uint32_t sizeP, const uint64_t* p,
                                 const uint64_t* mu128, uint64_t* out);

/* ---------------- a16: exact SwitchCRTBasis (dcrtpoly-impl.h:1008-1085) ----------------
 * QHatModp here is [sizeP][sizeQ] (the reference indexes QHatModp[j][i] in this function);
 * alphaQModp is [sizeQ+1][sizeP]; qInv[i] = 1.0/q_i (double). */
void orc_switch_crt_basis(const uint64_t* x, uint32_t sizeQ, uint32_t N, const uint64_t* q,
                          const uint64_t* QHatInvModq, const uint64_t* QHatInvModqPrecon,
                          const uint64_t* QHatModp_pq, const uint64_t* alphaQModp, uint32_t sizeP,
                          const uint64_t* p, const uint64_t* mu128, const double* qInv, uint64_t* out);

/* ---------------- HYBRID key-switch tables (rns-cryptoparameters.cpp:80-350) ---------------- */
/* DCRTPolyImpl::ExpandCRTBasis[ReverseOrder] (dcrtpoly-impl.h:1088-1148) and FastExpandCRTBasisPloverQ (:1151-1164) */
void orc_expand_crt_basis(const orc_ctx* ctxQP, uint32_t nQ, uint32_t nP, const uint64_t* x, int inEval,
                          const uint64_t* QHatInvModq, const uint64_t* QHatInvModqPrecon, const uint64_t* QHatModp_pq,
                          const uint64_t* alphaQModp, const uint64_t* muP128, const double* qInv, int resultEval,
                          int reverse, uint64_t* out);
void orc_fast_expand_crt_basis_p_over_q(const uint64_t* x, uint32_t nQ, uint32_t N, const uint64_t* q,
                                        const uint64_t* mPlQHatInvModq, const uint64_t* mPlQHatInvModqPrecon,
                                        const uint64_t* qInvModp, uint32_t nPl, const uint64_t* pl, const uint64_t* muPl128,
                                        const uint64_t* PlHatInvModp, const uint64_t* PlHatInvModpPrecon,
                                        const uint64_t* PlHatModq_qp, const uint64_t* alphaPlModq, uint32_t nQl,
                                        const uint64_t* ql, const uint64_t* muQl128, const double* pInv, uint64_t* out);
/* DCRTPolyImpl::ApproxModUp (dcrtpoly-impl.h:935-963); QHatModp [nQ][nP]; out [(nQ+nP)][N] EVALUATION */
void orc_approx_mod_up(const orc_ctx* ctxQP, uint32_t nQ, uint32_t nP, const uint64_t* x, int inEval,
                       const uint64_t* QHatInvModq, const uint64_t* QHatInvModqPrecon, const uint64_t* QHatModp,
                       const uint64_t* muP128, uint64_t* out);
/* DCRTPolyImpl::ExpandCRTBasisQlHat (dcrtpoly-impl.h:1167-1187) */
void orc_expand_crt_basis_ql_hat(const uint64_t* x, uint32_t sizeQl, uint32_t N, const uint64_t* q, const uint64_t* QlHatModq,
                                 uint32_t sizeQ, uint64_t* out);
/* LeveledSHEBase::EvalSquareCore, 2-element ciphertext (base-leveledshe.cpp:646-664) */
void orc_eval_square_core(const uint64_t* a0, const uint64_t* a1, uint32_t nLimbs, uint32_t N, const uint64_t* q, uint64_t* d0,
                          uint64_t* d1, uint64_t* d2);
/* ModRaise constructor DCRTPolyImpl(const PolyType&, params) (dcrtpoly-impl.h:87-93) */
void orc_mod_raise(const uint64_t* x, uint32_t N, const uint64_t* q, uint32_t nLimbs, uint64_t* out);
uint32_t orc_crt_decompose(const orc_ctx* c, const uint64_t* x, uint32_t nLimbs, uint32_t baseBits, uint64_t* out);
typedef struct orc_hybrid orc_hybrid;
/* Q tower (sizeQ limbs) + P tower (sizeP limbs) given explicitly; numPartQ = dnum. */
orc_hybrid* orc_hybrid_create(uint32_t N, uint32_t sizeQ, const uint64_t* q, const uint64_t* psiQ,
                              uint32_t sizeP, const uint64_t* p, const uint64_t* psiP, uint32_t numPartQ);
void        orc_hybrid_destroy(orc_hybrid*);
uint32_t    orc_hybrid_alpha(const orc_hybrid*);
/* choose the P basis the way PrecomputeCRTTables does (rns-cryptoparameters.cpp:128-176);
 * returns sizeP, fills p/psiP (capacity >= 64). */
uint32_t    orc_hybrid_select_p(uint32_t N, uint32_t sizeQ, const uint64_t* q, uint32_t numPartQ, uint32_t auxBits,
                                uint64_t* p, uint64_t* psiP);
/* table getters (all copied out as flat u64 arrays) */
void orc_hybrid_get_PInvModq(const orc_hybrid*, uint64_t* out /*[sizeQ]*/);
void orc_hybrid_get_PHatInvModp(const orc_hybrid*, uint64_t* out /*[sizeP]*/);
void orc_hybrid_get_PHatModq(const orc_hybrid*, uint64_t* out /*[sizeP][sizeQ]*/);
/* digit `part` at level sizeQl: PartQlHatInvModq[part][sizePartQl-1] -> out[sizePartQl] */
uint32_t orc_hybrid_get_PartQlHatInvModq(const orc_hybrid*, uint32_t part, uint32_t sizeQl, uint64_t* out);
/* PartQlHatModp[sizeQl-1][part] -> out[sizePartQl][sizeCompl]; returns sizeCompl; complModuli[sizeCompl] */
uint32_t orc_hybrid_get_PartQlHatModp(const orc_hybrid*, uint32_t part, uint32_t sizeQl, uint64_t* out,
                                      uint64_t* complModuli);

/* a11..a13 composites on explicit arrays. Everything EVALUATION format in/out unless said. */
/* EvalKeySwitchPrecomputeCore (keyswitch-hybrid.cpp:314-379): c[sizeQl][N] EVAL ->
 * digits[numPartQl][sizeQl+sizeP][N] EVAL.  returns numPartQl. */
uint32_t orc_hybrid_precompute_digits(const orc_hybrid*, const uint64_t* c, uint32_t sizeQl, uint64_t* digits);
/* EvalFastKeySwitchCoreExt (:402-435): key a/b are [numPartQ][sizeQ+sizeP][N];
 * out0/out1 [sizeQl+sizeP][N] */
void orc_hybrid_inner_product(const orc_hybrid*, const uint64_t* digits, uint32_t numPartQl, uint32_t sizeQl,
                              const uint64_t* keyB, const uint64_t* keyA, uint64_t* out0, uint64_t* out1);
/* ApproxModDown (dcrtpoly-impl.h:966-1005), t=0 (CKKS): x[sizeQl+sizeP][N] EVAL -> out[sizeQl][N] EVAL */
void orc_hybrid_approx_mod_down(const orc_hybrid*, const uint64_t* x, uint32_t sizeQl, uint64_t* out);
/* same with the BGV factors t^-1 mod p_j / t mod q_i (dcrtpoly-impl.h:981-983, 996-998) */
void orc_hybrid_approx_mod_down_t(const orc_hybrid*, const uint64_t* x, uint32_t sizeQl, uint64_t t, uint64_t* out);
/* KeySwitchCore (:308-312): full a13. out0/out1 [sizeQl][N] */
void orc_hybrid_key_switch(const orc_hybrid*, const uint64_t* c, uint32_t sizeQl, const uint64_t* keyB,
                           const uint64_t* keyA, uint64_t* out0, uint64_t* out1);
/* a14 EvalMultCore (base-leveledshe.cpp:607-644) + relinearise (:201-214):
 * ct1 = (a0,a1), ct2 = (b0,b1), each [sizeQl][N] EVAL -> out (c0,c1) */
void orc_ckks_eval_mult_relin(const orc_hybrid*, const uint64_t* a0, const uint64_t* a1, const uint64_t* b0,
                              const uint64_t* b1, uint32_t sizeQl, const uint64_t* keyB, const uint64_t* keyA,
                              uint64_t* c0, uint64_t* c1);

/* LeveledSHEBase::EvalAutomorphism (base-leveledshe.cpp:381-422) == EvalFastRotation with freshly computed digits
 * (:432-463): out0 = Auto_k(c0 + ks0(c1)), out1 = Auto_k(ks1(c1)), all EVALUATION, key = the automorphism key of k */
/* EvalFastRotationExt (ckksrns-leveledshe.cpp:534-582): result stays in the extended basis, out [(sizeQl+sizeP)][N] */
void orc_eval_fast_rotation_ext(const orc_hybrid*, const uint64_t* c0, const uint64_t* c1, uint32_t sizeQl, uint32_t k,
                                int addFirst, const uint64_t* keyB, const uint64_t* keyA, uint64_t* out0, uint64_t* out1);
/* BSGS plaintext-matrix product with double hoisting (FHECKKSRNS::EvalLinearTransform, ckksrns-fhe.cpp:1832-1882; one level
 * of EvalCoeffsToSlots / EvalSlotsToCoeffs, :1884-2198).  inK[j] / outK[i] = automorphism index, 0 = no rotation;
 * diag[i*nIn+j] = plaintext rows [(sizeQl+sizeP)][N] EVALUATION or NULL (term absent). */
void orc_ckks_bsgs_transform(const orc_hybrid*, const uint64_t* c0, const uint64_t* c1, uint32_t sizeQl, uint32_t nIn,
                             const uint32_t* inK, const uint64_t* const* inKeyB, const uint64_t* const* inKeyA, uint32_t nOut,
                             const uint32_t* outK, const uint64_t* const* outKeyB, const uint64_t* const* outKeyA,
                             const uint64_t* const* diag, uint64_t* out0, uint64_t* out1);
void orc_eval_automorphism(const orc_hybrid*, const uint64_t* c0, const uint64_t* c1, uint32_t sizeQl, uint32_t k,
                           const uint64_t* keyB, const uint64_t* keyA, uint64_t* out0, uint64_t* out1);

/* ---------------- a15: DropLastElementAndScale (dcrtpoly-impl.h:693-712), EVAL in/out ----------------
 * x[sizeQl][N] -> out[sizeQl-1][N]; ctx limbs [0,sizeQl) are the tower. Tables are computed inside
 * the way ckksrns-cryptoparameters.cpp:60-81 does. */
void orc_drop_last_element_and_scale(const orc_ctx*, const uint64_t* x, uint32_t sizeQl, uint64_t* out);
/* DCRTPolyImpl::ModReduce (dcrtpoly-impl.h:736-755): BGV modulus switch by the last limb; x [sizeQl][N], out [sizeQl-1][N] */
void orc_mod_reduce(const orc_ctx*, const uint64_t* x, uint32_t sizeQl, uint64_t t, int evalFormat, uint64_t* out);
void orc_rescale_tables(const orc_ctx*, uint32_t sizeQl, uint64_t* QlQlInvModqlDivqlModq, uint64_t* qlInvModq);

/* ---------------- a17: ScaleAndRound family (dcrtpoly-impl.h:1470-1689) ----------------
 * x is a tower [sizeI+sizeO][N] (COEFF). tab is [sizeO][sizeI+1] as the reference indexes it (tab[j][i]);
 * outputFirst != 0: the output basis is the FIRST sizeO limbs of x (inputIndex = sizeO), else the LAST sizeO limbs. */
void orc_scale_and_round(const uint64_t* x, uint32_t sizeI, uint32_t sizeO, uint32_t N, int outputFirst,
                         const uint64_t* tab, const double* frac, const uint64_t* o, const uint64_t* mu128,
                         uint64_t* out);
/* ApproxScaleAndRound (:1470-1510): x [sizeQ+sizeP][N] -> out [sizeP][N], tab [sizeP][sizeQ+1] */
void orc_approx_scale_and_round(const uint64_t* x, uint32_t sizeQ, uint32_t sizeP, uint32_t N, const uint64_t* tab,
                                const uint64_t* p, const uint64_t* mu128, uint64_t* out);
/* TimesQovert (:868-885): x [L][N] in place;  SetValuesModSwitch (:630-647): one limb through double precision */
void orc_times_q_over_t(uint64_t* x, uint32_t L, uint32_t N, const uint64_t* q, uint64_t t, uint64_t negQModt, const uint64_t* tInvModq);
void orc_set_values_mod_switch(const uint64_t* x, uint32_t N, uint64_t qFrom, uint64_t qTo, uint64_t* out);
/* ScaleAndRoundPOverQ (:1674-1689): x [sizeQ+1][N] (last limb modulus pLast) -> out [sizeQ][N] */
void orc_scale_and_round_p_over_q(const uint64_t* x, uint32_t sizeQ, uint32_t N, const uint64_t* q, uint64_t pLast,
                                  const uint64_t* pInvModq, uint64_t* out);

/* ---------------- a18: BEHZ (dcrtpoly-impl.h:1694-1929; tables bfvrns-cryptoparameters.cpp:673-850) ---------------- */
/* ScaleAndRound -> NativePoly mod t (dcrtpoly-impl.h:1190-1467) and its BEHZ-decryption overload (:1631-1671);
 * x [sizeQ][N] COEFFICIENT, out [N]; tabBModt / bfrac may be NULL when the split branch cannot be taken */
void orc_scale_and_round_native(const uint64_t* x, uint32_t sizeQ, uint32_t N, const uint64_t* q, uint64_t t,
                                const uint64_t* tabModt, const uint64_t* tabBModt, const double* frac, const double* bfrac,
                                uint64_t* out);
void orc_scale_and_round_behz_decrypt(const uint64_t* x, uint32_t sizeQ, uint32_t N, const uint64_t* q, uint64_t tgamma,
                                      const uint64_t* tgammaQHatModq, const uint64_t* negInvqModtgamma, uint64_t* out);
typedef struct orc_behz orc_behz;
orc_behz* orc_behz_create(uint32_t N, uint32_t numQ, const uint64_t* q, uint64_t t);
void      orc_behz_destroy(orc_behz*);
uint32_t  orc_behz_num_bsk(const orc_behz*);                 /* numQ + 1 */
void      orc_behz_get_bsk(const orc_behz*, uint64_t* bsk, uint64_t* psiBsk);
/* coefficient-domain cores (the NTTs around them are orc_ntt_*): */
void orc_behz_q_to_bsk_montgomery(const orc_behz*, const uint64_t* xq /*[numQ][N]*/, uint64_t* outBsk /*[numBsk][N]*/);
void orc_behz_fast_rns_floorq(const orc_behz*, uint64_t* x /*[numQ+numBsk][N], in place*/);
void orc_behz_fast_base_conv_sk(const orc_behz*, const uint64_t* x /*[numQ+numBsk][N]*/, uint64_t* outQ /*[numQ][N]*/);
/* LeveledSHEBFVRNS::EvalMult, BEHZ branch (bfvrns-leveledshe.cpp:198-445) without relinearisation: ctxAll = Q then Bsk
 * limbs; inputs [numQ][N] EVALUATION, outputs [numQ][N] COEFFICIENT (as the reference leaves them). */
void orc_bfv_eval_mult_behz(const orc_behz*, const orc_ctx* ctxAll, const uint64_t* a0, const uint64_t* a1,
                            const uint64_t* b0, const uint64_t* b1, uint64_t* d0, uint64_t* d1, uint64_t* d2);

/* ---- f3: sampled towers (csrc/sampler_kernels.h) on Philox4x32-10: the reference's distributions and inversion table, NOT its Blake2
 * words (parity unpinned against the reference's stream by construction; pinned: Philox against the Random123 vectors, the table against
 * discretegaussiangenerator-impl.h:75-89) ---- */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
void orc_sample_uniform(uint64_t* out, const uint64_t* q, uint32_t nLimbs, uint32_t batch, size_t N, uint64_t seed, uint32_t stream);
uint32_t orc_dgg_table(double sigma, double* vals, uint32_t cap, double* a);
void orc_sample_gaussian(uint64_t* out, int64_t* ints, const uint64_t* q, uint32_t nLimbs, uint32_t batch, size_t N, double sigma,
                         uint64_t seed, uint32_t stream);
void orc_sample_ternary(uint64_t* out, int64_t* ints, const uint64_t* q, uint32_t nLimbs, uint32_t batch, size_t N, uint64_t seed,
                        uint32_t stream);

#ifdef __cplusplus
}
#endif
#endif
