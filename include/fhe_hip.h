/*
 * fhe_hip.h — C ABI of the MI355X (gfx950) RNS polynomial-arithmetic backend for OpenFHE's DCRTPoly
 * hot path.  This is the drop-in boundary: a thin `lattice/hal/hip/` HAL shim implementing
 * lbcrypto::DCRTPolyInterface and the standalone parity/benchmark harness both
 * bind exactly these entry points.  The reference has no FFI of its own — its "operator API" is the C++
 * class surface selected at compile time by src/core/include/lattice/hal/lat-backend.h:39-61 — so each
 * entry point cites the reference member function(s) it replaces (paths relative to
 * openfhe-development/).
 *
 * Conventions
 *  - plain C types only; every call returns fhe_status (0 = OK), no exceptions cross the ABI;
 *    fhe_last_error() returns the message of the last failure on the calling thread
 *    (the C++ shim re-throws it with OPENFHE_THROW so callers see the reference's behaviour);
 *  - a tower lives in DEVICE memory as uint64_t[batch][nLimbs][N], limb-major, N contiguous, every word
 *    a canonical residue in [0, q_limb) (reference: std::vector<PolyImpl<NativeVector>>,
 *    src/core/include/lattice/hal/default/dcrtpoly.h:395-397); `limbIdx[r]` maps tower row r to a limb
 *    of the context (NULL = identity), which is how towers at lower levels, digits and the Q∪P
 *    extension share one context;
 *  - the caller owns all tower memory (fhe_malloc/fhe_free or any other device allocation, e.g. a
 *    torch tensor's data_ptr); the library owns its context/plan tables behind opaque handles;
 *  - all work is asynchronous on `stream` (a hipStream_t passed as void*, NULL = default stream);
 *    fhe_stream_sync() or the caller's own hip sync makes results visible;
 *  - there is NO CPU fallback: if no gfx950 device is usable every entry point fails with
 *    FHE_ERR_DEVICE.
 */
#ifndef FHE_HIP_H
#define FHE_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int fhe_status;
enum {
    FHE_OK          = 0,
    FHE_ERR_ARG     = 1, /* bad argument (the reference would OPENFHE_THROW) */
    FHE_ERR_DEVICE  = 2, /* HIP runtime failure / no device */
    FHE_ERR_ALLOC   = 3,
    FHE_ERR_UNSUPPORTED = 4
};

typedef struct fhe_ctx fhe_ctx;          /* ring + modulus tower + device twiddle tables */
typedef struct fhe_conv fhe_conv;        /* one CRT basis-conversion plan (tables on device) */
typedef struct fhe_ks_plan fhe_ks_plan;  /* HYBRID key-switching plan for a (Q, P, dnum) */
typedef struct fhe_ks_key fhe_ks_key;    /* an evaluation key resident on the device */

const char* fhe_last_error(void);
const char* fhe_version(void);
/* number of usable devices (hipGetDeviceCount); 0 if none */
int fhe_device_count(void);

/* ---- context -------------------------------------------------------------------------------------
 * Replaces ILDCRTParams + the lazily built static twiddle cache
 * (src/core/include/lattice/hal/default/ildcrtparams.h:70-372,
 *  ChineseRemainderTransformFTTNat::PreCompute, src/core/include/math/hal/intnat/transformnat-impl.h:714-756).
 * q[i] prime < 2^60 with q[i] = 1 mod 2N, psi[i] a primitive 2N-th root of unity mod q[i] (the reference
 * uses RootOfUnity(), the minimum one). Tables are built eagerly and are immutable afterwards.
 * logN in [4, 17]; nLimbs <= 256 (rows of one tower and limbs of one context; round 6, until then 128). */
fhe_status fhe_ctx_create(uint32_t logN, uint32_t nLimbs, const uint64_t* q, const uint64_t* psi, int device,
                          fhe_ctx** out);
void       fhe_ctx_destroy(fhe_ctx* ctx);
uint32_t   fhe_ctx_logn(const fhe_ctx* ctx);
uint32_t   fhe_ctx_limbs(const fhe_ctx* ctx);
int        fhe_ctx_device(const fhe_ctx* ctx);

/* ---- memory / streams ---------------------------------------------------------------------------- */
fhe_status fhe_malloc(fhe_ctx* ctx, size_t bytes, void** devPtr);
fhe_status fhe_free(fhe_ctx* ctx, void* devPtr);
/* free / total bytes of the context's device (hipMemGetInfo): a host runtime that caches released buffers (the HAL backend of
 * DCRTPoly) keeps a reserve for kernel launches with it */
fhe_status fhe_mem_info(fhe_ctx* ctx, size_t* freeBytes, size_t* totalBytes);
fhe_status fhe_memcpy_h2d(fhe_ctx* ctx, void* dst, const void* src, size_t bytes, void* stream);
fhe_status fhe_memcpy_d2h(fhe_ctx* ctx, void* dst, const void* src, size_t bytes, void* stream);
fhe_status fhe_memcpy_d2d(fhe_ctx* ctx, void* dst, const void* src, size_t bytes, void* stream);
fhe_status fhe_stream_sync(fhe_ctx* ctx, void* stream);
/* Streams and graphs.  Every entry point only enqueues work on the caller's stream, so a sequence of calls (an EvalMult, a
 * BFV EvalMult, a rotation ... 20-60 kernel launches) can be recorded ONCE into a HIP graph and replayed with a single
 * launch: fhe_graph_begin(stream) ... calls on that stream ... fhe_graph_end(stream, &graph); fhe_graph_launch(graph,
 * stream).  Run the sequence once before capturing (tables are built on first use; building is not capturable). */
fhe_status fhe_stream_create(fhe_ctx* ctx, void** stream);
fhe_status fhe_stream_destroy(fhe_ctx* ctx, void* stream);
/* `stream` waits ON THE DEVICE (the host is not blocked) for everything enqueued so far on `other`: the hand-over of a tower
 * from one host thread's stream to another's (the HAL backend gives every host thread a stream of its own). */
fhe_status fhe_stream_wait(fhe_ctx* ctx, void* stream, void* other);
/* Completion marks: an event recorded on `stream` NOW; another stream waits for it on the device later.  Exact where fhe_stream_wait is
 * not (that one makes the waiter follow everything the other stream has enqueued by the time of the call): the HAL backend marks every
 * released buffer it caches, so that a host thread reusing another thread's buffer waits for that buffer's last use only. */
fhe_status fhe_event_create(fhe_ctx* ctx, void** event);
fhe_status fhe_event_record(fhe_ctx* ctx, void* event, void* stream);
fhe_status fhe_stream_wait_event(fhe_ctx* ctx, void* stream, void* event);
fhe_status fhe_event_destroy(fhe_ctx* ctx, void* event);
fhe_status fhe_memset_zero(fhe_ctx* ctx, void* dst, size_t bytes, void* stream);
fhe_status fhe_graph_begin(fhe_ctx* ctx, void* stream);
fhe_status fhe_graph_end(fhe_ctx* ctx, void* stream, void** graph);
fhe_status fhe_graph_launch(fhe_ctx* ctx, void* graph, void* stream);
void       fhe_graph_destroy(void* graph);

/* ---- a4/a5/a6: NTT -------------------------------------------------------------------------------
 * Replaces DCRTPolyImpl::SwitchFormat (dcrtpoly-impl.h:1932-1940) -> PolyImpl::SwitchFormat
 * (poly-impl.h:420-440) -> ChineseRemainderTransformFTTNat::ForwardTransformToBitReverseInPlace /
 * InverseTransformFromBitReverseInPlace (transformnat-impl.h:648-657, 678-690; loops :303-374, :512-625).
 * In place on x[batch][nLimbs][N] (or out of place with xin != xout); forward = COEFFICIENT -> EVALUATION
 * (natural -> bit-reversed order), inverse = EVALUATION -> COEFFICIENT (1/N folded in). */
fhe_status fhe_ntt_fwd(fhe_ctx* ctx, uint64_t* x, const uint32_t* limbIdx, uint32_t nLimbs, uint32_t batch, void* stream);
fhe_status fhe_ntt_inv(fhe_ctx* ctx, uint64_t* x, const uint32_t* limbIdx, uint32_t nLimbs, uint32_t batch, void* stream);
fhe_status fhe_ntt_fwd_oop(fhe_ctx* ctx, const uint64_t* xin, uint64_t* xout, const uint32_t* limbIdx, uint32_t nLimbs,
                           uint32_t batch, void* stream);
fhe_status fhe_ntt_inv_oop(fhe_ctx* ctx, const uint64_t* xin, uint64_t* xout, const uint32_t* limbIdx, uint32_t nLimbs,
                           uint32_t batch, void* stream);

/* Negacyclic polynomial product of COEFFICIENT-format towers, c = a * b in Z_q[x]/(x^N + 1) per limb: what the reference
 * spells a.SetFormat(EVALUATION); b.SetFormat(EVALUATION); c = a * b; c.SetFormat(COEFFICIENT)
 * (dcrtpoly-impl.h:1932-1940 with transformnat-impl.h:303-374, 512-625; dcrtpoly.h:174-189).  a and b are not modified; out may
 * not alias them.  Two-pass rings run fused kernels (the forward row pass of b, the Hadamard product and the inverse row
 * pass are one kernel: SURVEY.md 8(d) "fused fwd o mul o inv").  ws: 2 towers of device workspace. */
size_t     fhe_poly_mul_workspace_bytes(const fhe_ctx* ctx, uint32_t nLimbs, uint32_t batch);
fhe_status fhe_poly_mul(fhe_ctx* ctx, const uint64_t* a, const uint64_t* b, uint64_t* out, const uint32_t* limbIdx,
                        uint32_t nLimbs, uint32_t batch, void* ws, size_t wsBytes, void* stream);

/* DCRTPolyImpl::Plus(const std::vector<Integer>&) (dcrtpoly-impl.h:520-527): limb i plus the constant consts[i] (reduced mod
 * q_i first) — PolyImpl::Plus(Integer) (poly-impl.h:211-218) adds it to every word in EVALUATION and to coefficient 0 only in
 * COEFFICIENT (coeff0Only != 0).  This is what LeveledSHECKKSRNS::EvalAddInPlace(ciphertext, double)
 * (ckksrns-leveledshe.cpp:60-68) does to element 0.  out may alias a; constants travel by value (asynchronous). */
fhe_status fhe_add_const(fhe_ctx* ctx, uint64_t* out, const uint64_t* a, const uint64_t* consts, const uint32_t* limbIdx,
                         uint32_t nLimbs, uint32_t batch, int coeff0Only, void* stream);
/* DCRTPolyImpl::Minus(const std::vector<Integer>&) (dcrtpoly-impl.h:541-548 -> poly-impl.h:221-225: ModSub on every word, in
 * both formats); EvalSubInPlace(ciphertext, double) (ckksrns-leveledshe.cpp:112-120) */
fhe_status fhe_sub_const(fhe_ctx* ctx, uint64_t* out, const uint64_t* a, const uint64_t* consts, const uint32_t* limbIdx,
                         uint32_t nLimbs, uint32_t batch, void* stream);

/* ---- a7: element-wise tower arithmetic -----------------------------------------------------------
 * Replaces DCRTPolyImpl::operator+= / -= / *= , Plus/Minus/Times, Negate
 * (dcrtpoly-impl.h:347-408, dcrtpoly.h:131-189) and NativeVectorT::ModAddEq/ModSubEq/ModMulEq
 * (src/core/lib/math/hal/intnat/mubintvecnat.cpp:229-339). out may alias a or b. */
fhe_status fhe_add(fhe_ctx* ctx, uint64_t* out, const uint64_t* a, const uint64_t* b, const uint32_t* limbIdx,
                   uint32_t nLimbs, uint32_t batch, void* stream);
fhe_status fhe_sub(fhe_ctx* ctx, uint64_t* out, const uint64_t* a, const uint64_t* b, const uint32_t* limbIdx,
                   uint32_t nLimbs, uint32_t batch, void* stream);
fhe_status fhe_mul(fhe_ctx* ctx, uint64_t* out, const uint64_t* a, const uint64_t* b, const uint32_t* limbIdx,
                   uint32_t nLimbs, uint32_t batch, void* stream);
fhe_status fhe_neg(fhe_ctx* ctx, uint64_t* out, const uint64_t* a, const uint32_t* limbIdx, uint32_t nLimbs,
                   uint32_t batch, void* stream);
/* acc += a * b per limb: the accumulation step of KeySwitchHYBRID::EvalFastKeySwitchCoreExt
 * (src/pke/lib/keyswitch/keyswitch-hybrid.cpp:419-430) on whole towers; exact, so the order of accumulation is free */
fhe_status fhe_mul_add(fhe_ctx* ctx, uint64_t* acc, const uint64_t* a, const uint64_t* b, const uint32_t* limbIdx,
                       uint32_t nLimbs, uint32_t batch, void* stream);
/* Times(const std::vector<NativeInteger>&) / operator*=(NativeInteger) (dcrtpoly-impl.h:582-620):
 * consts[r] (HOST array, one per tower row) multiplies limb r of every tower in the batch.  The constants travel by value
 * in the kernel arguments: the call is asynchronous like every other one and can be captured into a graph; the host
 * array may be released as soon as the call returns. */
fhe_status fhe_mul_const(fhe_ctx* ctx, uint64_t* out, const uint64_t* a, const uint64_t* consts,
                         const uint32_t* limbIdx, uint32_t nLimbs, uint32_t batch, void* stream);
/* out = sum_i consts[i] (.) x[i] per limb (+ out when accumulate != 0): pke's weighted sums — internalEvalLinearWSumMutable,
 * src/pke/lib/scheme/ckksrns/ckksrns-advancedshe.cpp:97-136: `EvalMultInPlace(ct_i, c_i); EvalAddInPlaceNoCheck(ct_0, ct_i)` for every
 * term, the inner loops of the Chebyshev evaluation of bootstrapping (ckksrns-fhe.cpp:691-692, :786) — as one launch per 16 terms:
 * every term is read once, the sum written once.  x: HOST array of nTerms DEVICE towers [batch][nLimbs][N] (each dense, allocated
 * on its own); consts: HOST array [nTerms][nLimbs], reduced modulo their limbs (their device table is cached by content).  Exact
 * modular arithmetic: the residues are the reference's whatever the order of the sum.  out may be one of the x[i] (with more than 16 terms the
 * terms that alias out are summed by the first launch, before out is overwritten; at most 16 terms may alias out then). */
fhe_status fhe_lincomb(fhe_ctx* ctx, uint64_t* out, const uint64_t* const* x, const uint64_t* consts, uint32_t nTerms,
                       const uint32_t* limbIdx, uint32_t nLimbs, uint32_t batch, int accumulate, void* stream);
/* The two elements of a ciphertext in ONE launch: element e of every operand is a tower allocated on its own (o_e = a_e op b_e,
 * o_e = a_e * consts per limb).  pke applies its operations element by element (base-leveledshe.cpp:562-606,
 * ckksrns-leveledshe.cpp:748-759); one ciphertext's tower alone leaves most of the chip idle.  In-place use (o_e == a_e) is fine. */
fhe_status fhe_add_pair(fhe_ctx* ctx, uint64_t* o0, uint64_t* o1, const uint64_t* a0, const uint64_t* a1, const uint64_t* b0,
                        const uint64_t* b1, const uint32_t* limbIdx, uint32_t nLimbs, void* stream);
fhe_status fhe_sub_pair(fhe_ctx* ctx, uint64_t* o0, uint64_t* o1, const uint64_t* a0, const uint64_t* a1, const uint64_t* b0,
                        const uint64_t* b1, const uint32_t* limbIdx, uint32_t nLimbs, void* stream);
fhe_status fhe_mul_const_pair(fhe_ctx* ctx, uint64_t* o0, uint64_t* o1, const uint64_t* a0, const uint64_t* a1, const uint64_t* consts,
                              const uint32_t* limbIdx, uint32_t nLimbs, void* stream);
/* DCRTPolyImpl::TimesQovert (dcrtpoly-impl.h:868-885): every word x of limb r becomes ((x * NegQModt) mod t) * tInvModq[r] mod q_r —
 * ModMulFastConst modulo the plaintext modulus, then the generalized Barrett product modulo q_r (BFV encryption scales the message
 * by Q/t with it).  out may alias a; tInvModq is a HOST array (by value in the kernel arguments). */
fhe_status fhe_times_q_over_t(fhe_ctx* ctx, uint64_t* out, const uint64_t* a, uint64_t t, uint64_t negQModt,
                              const uint64_t* tInvModq, const uint32_t* limbIdx, uint32_t nLimbs, uint32_t batch, void* stream);
/* DCRTPolyImpl::SetValuesModSwitch (dcrtpoly-impl.h:630-647): `words` COEFFICIENT words modulo qFrom scaled to the modulus qTo,
 * out[j] = uint64(floor(0.5 + double(x[j]) * (double(qTo) / double(qFrom)))) mod qTo, in the reference's double arithmetic. */
fhe_status fhe_mod_switch_round(fhe_ctx* ctx, const uint64_t* x, uint64_t qFrom, uint64_t qTo, uint64_t* out, size_t words,
                                void* stream);
/* NativeVectorT::MultAccEqNoCheck per limb (src/core/lib/math/hal/intnat/mubintvecnat.cpp:132-142; PolyImpl wrapper
 * poly.h:323): acc[r] += v[r] * consts[r]  (constant reduced mod q first, Shoup product, ModAddFast) */
fhe_status fhe_mult_acc(fhe_ctx* ctx, uint64_t* acc, const uint64_t* v, const uint64_t* consts,
                        const uint32_t* limbIdx, uint32_t nLimbs, uint32_t batch, void* stream);
/* LeveledSHEBase::EvalMultCore (src/pke/lib/schemebase/base-leveledshe.cpp:607-644):
 * d0 = a0*b0, d1 = a0*b1 + a1*b0, d2 = a1*b1 */
fhe_status fhe_tensor(fhe_ctx* ctx, const uint64_t* a0, const uint64_t* a1, const uint64_t* b0, const uint64_t* b1,
                      uint64_t* d0, uint64_t* d1, uint64_t* d2, const uint32_t* limbIdx, uint32_t nLimbs,
                      uint32_t batch, void* stream);

/* LeveledSHEBase::EvalSquareCore for 2-element ciphertexts (base-leveledshe.cpp:646-664):
 * d0 = a0*a0, d1 = a0*a1 + a0*a1, d2 = a1*a1 */
fhe_status fhe_tensor_square(fhe_ctx* ctx, const uint64_t* a0, const uint64_t* a1, uint64_t* d0, uint64_t* d1, uint64_t* d2,
                             const uint32_t* limbIdx, uint32_t nLimbs, uint32_t batch, void* stream);

/* ---- a8: automorphism ----------------------------------------------------------------------------
 * Replaces DCRTPolyImpl::AutomorphismTransform(k[, precomp]) (dcrtpoly-impl.h:314-333) ->
 * PolyImpl::AutomorphismTransform (poly-impl.h:310-376; table PrecomputeAutoMap, nbtheory2.cpp:264-275).
 * k odd. evalFormat != 0: EVALUATION gather; 0: COEFFICIENT signed permutation. out must not alias in. */
fhe_status fhe_automorph(fhe_ctx* ctx, uint64_t* out, const uint64_t* in, uint32_t k, int evalFormat,
                         const uint32_t* limbIdx, uint32_t nLimbs, uint32_t batch, void* stream);

/* ---- a9: centred modulus switch -------------------------------------------------------------------
 * Replaces PolyImpl::SwitchModulus / NativeVectorT::SwitchModulus (poly-impl.h:400-407,
 * mubintvecnat.cpp:109-122): limb `srcPos` (context limb srcCtxLimb) of every tower src[batch][srcLimbs][N]
 * is lifted, centred, into each of the nLimbs rows of out. */
fhe_status fhe_switch_modulus(fhe_ctx* ctx, uint64_t* out, const uint32_t* limbIdx, uint32_t nLimbs,
                              const uint64_t* src, uint32_t srcLimbs, uint32_t srcPos, uint32_t srcCtxLimb,
                              uint32_t batch, void* stream);
/* With srcLimbs = 1 this is also the "ModRaise" constructor DCRTPolyImpl(const PolyType&, params) (dcrtpoly-impl.h:87-93,
 * used by FHECKKSRNS::EvalBootstrap, ckksrns-fhe.cpp:592-601): one COEFFICIENT polynomial modulo q_0 lifted, centred, into
 * every limb of a tower. */

/* DCRTPolyImpl::CRTDecompose(baseBits) (dcrtpoly-impl.h:230-285; KeySwitchBV's digit decomposition, keyswitch-bv.cpp:254): x [nLimbs][N] in
 * COEFFICIENT format -> out [towers][nLimbs][N] in EVALUATION format, towers in the reference's order (limb 0's digits, least significant
 * first, then limb 1's ...; baseBits == 0: one tower per limb).  fhe_crt_decompose_towers returns the number of towers, 0 when the
 * selection is outside the device path (baseBits > 31, or a window that would leave the 64-bit word — where the reference itself is
 * undefined). */
uint32_t fhe_crt_decompose_towers(const fhe_ctx* ctx, const uint32_t* limbIdx, uint32_t nLimbs, uint32_t baseBits);
fhe_status fhe_crt_decompose(fhe_ctx* ctx, const uint64_t* x, const uint32_t* limbIdx, uint32_t nLimbs, uint32_t baseBits, uint64_t* out,
                             void* stream);

/* ---- a10/a16: CRT basis conversion ------------------------------------------------------------------
 * fhe_conv_create builds the device tables for converting from the basis {srcLimbIdx} to {dstLimbIdx}
 * (context limbs). The table VALUES are computed by the library the way CryptoParametersRNS /
 * CryptoParametersBFVRNS do (src/pke/lib/schemerns/rns-cryptoparameters.cpp:199-349):
 *   QHatInvModq[i] = [(Q/q_i)^-1]_{q_i},  QHatModp[i][j] = [Q/q_i]_{p_j},  mu_j = floor(2^128/p_j),
 *   and for the exact form alphaQModp[a][j] = [a*Q]_{p_j}, qInv[i] = 1.0/q_i.
 * fhe_approx_switch_basis replaces DCRTPolyImpl::ApproxSwitchCRTBasis (dcrtpoly-impl.h:888-932, fast path);
 * fhe_switch_basis_exact replaces DCRTPolyImpl::SwitchCRTBasis (:1008-1085; the overflow count is
 * accumulated in double in the reference's order).
 * in  = [batch][inStride][N]  COEFFICIENT format, source limb i at row inFirst+i;
 * out = [batch][outStride][N] COEFFICIENT format, target limb j at row outFirst+j. */
fhe_status fhe_conv_create(fhe_ctx* ctx, const uint32_t* srcLimbIdx, uint32_t nSrc, const uint32_t* dstLimbIdx,
                           uint32_t nDst, fhe_conv** out);
void       fhe_conv_destroy(fhe_conv* conv);
fhe_status fhe_approx_switch_basis(fhe_conv* conv, const uint64_t* in, uint32_t inStride, uint32_t inFirst,
                                   uint64_t* out, uint32_t outStride, uint32_t outFirst, uint32_t batch, void* stream);
fhe_status fhe_switch_basis_exact(fhe_conv* conv, const uint64_t* in, uint32_t inStride, uint32_t inFirst,
                                  uint64_t* out, uint32_t outStride, uint32_t outFirst, uint32_t batch, void* stream);
/* Plan with the CALLER's tables, laid out as the reference passes them to ApproxSwitchCRTBasis / SwitchCRTBasis
 * (dcrtpoly-impl.h:888-932, 1008-1085): QHatInvModq[nSrc], QHatModp[nSrc][nDst]; for the exact variant additionally
 * alphaQModp[nSrc+1][nDst] and qInv[nSrc] (doubles), else both NULL.  Needed where the tables are not the plain CRT
 * ones, e.g. FastExpandCRTBasisPloverQ's mPlQHatInvModq / qInvModp. */
fhe_status fhe_conv_create_custom(fhe_ctx* ctx, const uint32_t* srcLimbIdx, uint32_t nSrc, const uint32_t* dstLimbIdx,
                                  uint32_t nDst, const uint64_t* hatInv, const uint64_t* hatMod, const uint64_t* alphaMod,
                                  const double* qInv, fhe_conv** out);
/* DCRTPolyImpl::ExpandCRTBasis / ExpandCRTBasisReverseOrder (dcrtpoly-impl.h:1088-1148): x [batch][nSrc][N] over the plan's
 * source basis Q in format inEval -> out [batch][nSrc+nDst][N] over Q u P in resultEval (reverseOrder: P rows first).
 * ws (fhe_expand_crt_basis_workspace_bytes) is only needed for EVALUATION input. */
size_t     fhe_expand_crt_basis_workspace_bytes(const fhe_conv* plan, uint32_t batch);
fhe_status fhe_expand_crt_basis(fhe_conv* plan, const uint64_t* x, int inEval, uint64_t* out, int resultEval,
                                int reverseOrder, uint32_t batch, void* ws, size_t wsBytes, void* stream);
/* DCRTPolyImpl::ApproxModUp (dcrtpoly-impl.h:935-963): x [batch][nSrc][N] over the plan's source basis Q in format inEval ->
 * out [batch][nSrc+nDst][N] over Q u P in EVALUATION (the P part is ApproxSwitchCRTBasis of the coefficient form; the
 * EVALUATION copy of the Q limbs is reused when the input has one).  ws as for fhe_expand_crt_basis. */
fhe_status fhe_mod_up(fhe_conv* plan, const uint64_t* x, int inEval, uint64_t* out, uint32_t batch, void* ws, size_t wsBytes,
                      void* stream);
/* DCRTPolyImpl::ExpandCRTBasisQlHat (dcrtpoly-impl.h:1167-1187): limbs [0, sizeQl) of x [batch][sizeQl][N] times
 * QlHatModq[i] (HOST constants), limbs [sizeQl, sizeQ) zero -> out [batch][sizeQ][N] over context limbs limbIdx[0..sizeQ)
 * (NULL = identity); the format is unchanged. */
fhe_status fhe_expand_crt_basis_ql_hat(fhe_ctx* ctx, const uint64_t* x, uint32_t sizeQl, const uint64_t* QlHatModq,
                                       const uint32_t* limbIdx, uint32_t sizeQ, uint32_t batch, uint64_t* out, void* stream);
/* DCRTPolyImpl::FastExpandCRTBasisPloverQ (dcrtpoly-impl.h:1151-1164), COEFFICIENT format: toPl = custom-table plan
 * Q -> Pl (mPlQHatInvModq, qInvModp), toQl = plan Pl -> Ql; x [batch][nQ][N] -> out [batch][nQl+nPl][N] = [Ql | Pl]. */
fhe_status fhe_fast_expand_crt_basis_p_over_q(fhe_conv* toPl, fhe_conv* toQl, const uint64_t* x, uint64_t* out,
                                              uint32_t batch, void* stream);

/* ---- a11..a14: HYBRID key switching -----------------------------------------------------------------
 * The context must hold the Q limbs [0,sizeQ) followed by the P limbs [sizeQ, sizeQ+sizeP).
 * fhe_ks_plan_create replaces the HYBRID part of CryptoParametersRNS::PrecomputeCRTTables
 * (rns-cryptoparameters.cpp:80-350): digit partition (alpha = ceil(sizeQ/numPartQ)), complementary bases,
 * PartQlHatInvModq / PartQlHatModp per level, PInvModq, PHatInvModp, PHatModq, Barrett constants.
 * fhe_ks_key_upload copies an evaluation key (b and a vectors of EvalKeyRelin,
 * src/pke/include/key/evalkeyrelin.h:141,171), each given as HOST uint64_t[numPartQ][sizeQ+sizeP][N] in
 * EVALUATION format, to the device. With RCCL the caller broadcasts the device copy instead
 * (fhe_ks_key_alloc + fhe_ks_key_devptr). */
fhe_status fhe_ks_plan_create(fhe_ctx* ctx, uint32_t sizeQ, uint32_t sizeP, uint32_t numPartQ, fhe_ks_plan** out);
void       fhe_ks_plan_destroy(fhe_ks_plan* plan);
uint32_t   fhe_ks_plan_alpha(const fhe_ks_plan* plan);
fhe_status fhe_ks_key_alloc(fhe_ks_plan* plan, fhe_ks_key** out);
fhe_status fhe_ks_key_upload(fhe_ks_plan* plan, const uint64_t* keyB, const uint64_t* keyA, fhe_ks_key** out);
/* adopt key vectors that already live in device memory (not owned: destroy leaves them alone). This is the
 * multi-GPU path: rank 0 uploads, every rank receives the words with an RCCL broadcast into its own buffer
 * (e.g. a torch tensor) and wraps it — the only collective of the whole design. */
fhe_status fhe_ks_key_wrap(fhe_ks_plan* plan, uint64_t* devKeyB, uint64_t* devKeyA, fhe_ks_key** out);
void       fhe_ks_key_destroy(fhe_ks_key* key);
/* device pointers of the b (which=0) / a (which=1) vectors: uint64_t[numPartQ][sizeQ+sizeP][N] */
uint64_t*  fhe_ks_key_devptr(fhe_ks_key* key, int which);
size_t     fhe_ks_key_words(const fhe_ks_key* key);

/* workspace (device bytes) needed by the composite calls below for `batch` towers at level sizeQl */
size_t     fhe_ks_workspace_bytes(const fhe_ks_plan* plan, uint32_t sizeQl, uint32_t batch);

/* KeySwitchHYBRID::EvalKeySwitchPrecomputeCore + EvalFastKeySwitchCore
 * (src/pke/lib/keyswitch/keyswitch-hybrid.cpp:314-400), i.e. KeySwitchCore (:308-312):
 * c[batch][sizeQl][N] EVALUATION -> out0,out1[batch][sizeQl][N] EVALUATION. `ws` = device workspace. */
fhe_status fhe_keyswitch_hybrid(fhe_ks_plan* plan, const fhe_ks_key* key, const uint64_t* c, uint32_t sizeQl,
                                uint32_t batch, uint64_t* out0, uint64_t* out1, void* ws, size_t wsBytes,
                                void* stream);
/* acc0 += ks0(c), acc1 += ks1(c): the tail of LeveledSHEBase::EvalMult(ct, ct, key) (base-leveledshe.cpp:207-211), the two
 * additions fused into the last kernel of the key switch.  Same layout and workspace as fhe_keyswitch_hybrid. */
fhe_status fhe_keyswitch_hybrid_acc(fhe_ks_plan* plan, const fhe_ks_key* key, const uint64_t* c, uint32_t sizeQl,
                                    uint32_t batch, uint64_t* acc0, uint64_t* acc1, void* ws, size_t wsBytes, void* stream);
/* cc->EvalMult(ct1, ct2) for 2-element ciphertexts: EvalMultCore + KeySwitchCore + add
 * (src/pke/lib/schemebase/base-leveledshe.cpp:201-214, 607-644).  All towers [batch][sizeQl][N] EVALUATION.
 * c0/c1 may alias a0/a1. */
fhe_status fhe_ckks_eval_mult(fhe_ks_plan* plan, const fhe_ks_key* key, const uint64_t* a0, const uint64_t* a1,
                              const uint64_t* b0, const uint64_t* b1, uint32_t sizeQl, uint32_t batch, uint64_t* c0,
                              uint64_t* c1, void* ws, size_t wsBytes, void* stream);
/* Hoisting (Halevi-Shoup): LeveledSHEBase::EvalFastRotationPrecompute = EvalKeySwitchPrecomputeCore(c1)
 * (src/pke/lib/schemebase/base-leveledshe.cpp:425-430) once, then per rotation key EvalFastKeySwitchCore + add +
 * automorphism (:432-463).  The digits stay in the workspace `ws` between the calls (same sizeQl / batch / ws).
 * fhe_eval_automorphism = EvalAutomorphism (:381-422): out0 = Auto_k(c0 + ks0(c1)), out1 = Auto_k(ks1(c1)).
 * `key` must be the evaluation key of automorphism index k (FindAutomorphismIndex2nComplex for CKKS rotations). */
fhe_status fhe_ks_precompute(fhe_ks_plan* plan, const uint64_t* c1, uint32_t sizeQl, uint32_t batch, void* ws,
                             size_t wsBytes, void* stream);
fhe_status fhe_ks_fast_keyswitch(fhe_ks_plan* plan, const fhe_ks_key* key, const uint64_t* c1, uint32_t sizeQl,
                                 uint32_t batch, uint64_t* out0, uint64_t* out1, void* ws, size_t wsBytes, void* stream);
fhe_status fhe_eval_fast_rotation(fhe_ks_plan* plan, const fhe_ks_key* key, const uint64_t* c0, const uint64_t* c1,
                                  uint32_t k, uint32_t sizeQl, uint32_t batch, uint64_t* out0, uint64_t* out1, void* ws,
                                  size_t wsBytes, void* stream);
fhe_status fhe_eval_automorphism(fhe_ks_plan* plan, const fhe_ks_key* key, const uint64_t* c0, const uint64_t* c1,
                                 uint32_t k, uint32_t sizeQl, uint32_t batch, uint64_t* out0, uint64_t* out1, void* ws,
                                 size_t wsBytes, void* stream);
/* The inner product of HYBRID key switching (KeySwitchHYBRID::EvalFastKeySwitchCoreExt, keyswitch-hybrid.cpp:419-430) over
 * towers that live in separate allocations — what a DCRTPoly backend holds: every digit and every element of the evaluation
 * key is its own tower.  out_e[b][i] = sum_{t < nTerms} x[t][b][i] * k_e[t][keyRow[i]] mod q_{limbIdx[i]}, e = 0 (k0 -> out0)
 * and, when k1/out1 are given, e = 1.  x[t] is [batch][rows][N], k_e[t] is a tower of at least max(keyRow)+1 rows, keyRow NULL
 * = identity (the reference's idx(i) = i < sizeQl ? i : i + sizeQ - sizeQl, :425).  x, k0, k1 are HOST arrays of device
 * pointers; all operands are canonical residues; 1 <= nTerms <= 8 (FHE_ERR_UNSUPPORTED beyond). */
fhe_status fhe_inner_product(fhe_ctx* ctx, uint32_t nTerms, const uint64_t* const* x, const uint64_t* const* k0,
                             const uint64_t* const* k1, const uint32_t* keyRow, const uint32_t* limbIdx, uint32_t rows,
                             uint32_t batch, uint64_t* out0, uint64_t* out1, void* stream);

/* Double hoisting (ckksrns-fhe.cpp:1830-2000) works in the extended basis Q_l u P and mods down once:
 *   fhe_ks_ext                 = KeySwitchHYBRID::KeySwitchExt for one element (keyswitch-hybrid.cpp:217-243):
 *                                out [batch][sizeQl+sizeP][N], Q_l rows = c * [P]_{q_i}, P rows = 0;
 *   fhe_ks_fast_keyswitch_ext  = EvalFastKeySwitchCoreExt (:402-435) on the digits fhe_ks_precompute left in ws;
 *   fhe_eval_fast_rotation_ext = LeveledSHECKKSRNS::EvalFastRotationExt (ckksrns-leveledshe.cpp:534-582);
 *   fhe_ks_down                = KeySwitchHYBRID::KeySwitchDown (:245-278), ApproxModDown of both elements.
 * Extended towers are [batch][sizeQl+sizeP][N] over context limbs {0..sizeQl-1, sizeQ..sizeQ+sizeP-1}; the generic
 * element-wise entry points (fhe_add / fhe_mul / fhe_automorph with that limb list) cover EvalAddExt / EvalMultExt. */
fhe_status fhe_ks_ext(fhe_ks_plan* plan, const uint64_t* c, uint32_t sizeQl, uint32_t batch, uint64_t* outExt, void* stream);
fhe_status fhe_ks_fast_keyswitch_ext(fhe_ks_plan* plan, const fhe_ks_key* key, const uint64_t* c1, uint32_t sizeQl,
                                     uint32_t batch, uint64_t* out0Ext, uint64_t* out1Ext, void* ws, size_t wsBytes,
                                     void* stream);
fhe_status fhe_eval_fast_rotation_ext(fhe_ks_plan* plan, const fhe_ks_key* key, const uint64_t* c0, const uint64_t* c1,
                                      uint32_t k, int addFirst, uint32_t sizeQl, uint32_t batch, uint64_t* out0Ext,
                                      uint64_t* out1Ext, void* ws, size_t wsBytes, void* stream);
fhe_status fhe_ks_down(fhe_ks_plan* plan, const uint64_t* x0Ext, const uint64_t* x1Ext, uint32_t sizeQl, uint32_t batch,
                       uint64_t* out0, uint64_t* out1, void* ws, size_t wsBytes, void* stream);
/* Baby-step/giant-step plaintext-matrix x ciphertext product with double hoisting: FHECKKSRNS::EvalLinearTransform
 * (src/pke/lib/scheme/ckksrns/ckksrns-fhe.cpp:1832-1882) and one level of EvalCoeffsToSlots / EvalSlotsToCoeffs
 * (:1884-2198), the linear-transform loops of CKKS bootstrapping.
 *   rot_j   = inK[j] ? EvalFastRotationExt(ct, inK[j], digits(ct), true) : KeySwitchExt(ct, true)        j < nIn
 *   inner_i = sum_j rot_j * diag[i*nIn + j]            (EvalMultExt / EvalAddExtInPlace; NULL = term absent)
 *   outK[i] == 0: first += KeySwitchDownFirstElement(inner_i); outer[1] += inner_i[1]
 *   else        : d = KeySwitchDown(inner_i); first += Automorphism(d[0]); outer += EvalFastRotationExt(d, outK[i], digits(d), false)
 *   (out0, out1) = KeySwitchDown(outer), out0 += first
 * inK / outK are automorphism indices (FindAutomorphismIndex2nComplex of the rotation), 0 = no rotation; inKeys[j] /
 * outKeys[i] the matching evaluation keys (ignored where the index is 0).  diag: HOST array of nOut*nIn DEVICE pointers to
 * plaintext rows [sizeQl+sizeP][N] in EVALUATION format over limbs {0..sizeQl-1, sizeQ..sizeQ+sizeP-1}
 * (EvalLinearTransformPrecompute's aux plaintexts), shared by the whole batch.  c0,c1,out0,out1: [batch][sizeQl][N].
 * Every stage runs once over all outer steps (their accumulations are exact modular sums, so the order is free); the
 * first call with a new set of diagonals uploads their pointer table (not capturable), later calls are pure launches. */
size_t fhe_ckks_bsgs_workspace_bytes(const fhe_ks_plan* plan, uint32_t sizeQl, uint32_t batch, uint32_t nIn, uint32_t nOut);
fhe_status fhe_ckks_bsgs_transform(fhe_ks_plan* plan, const uint64_t* c0, const uint64_t* c1, uint32_t sizeQl, uint32_t batch,
                                   uint32_t nIn, const uint32_t* inK, const fhe_ks_key* const* inKeys, uint32_t nOut,
                                   const uint32_t* outK, const fhe_ks_key* const* outKeys, const uint64_t* const* diag,
                                   uint64_t* out0, uint64_t* out1, void* ws, size_t wsBytes, void* stream);
/* DCRTPolyImpl::ApproxModDown with t = 0 (dcrtpoly-impl.h:966-1005):
 * x[batch][sizeQl+sizeP][N] EVALUATION -> out[batch][sizeQl][N] EVALUATION */
fhe_status fhe_approx_mod_down(fhe_ks_plan* plan, const uint64_t* x, uint32_t sizeQl, uint32_t batch, uint64_t* out,
                               void* ws, size_t wsBytes, void* stream);
/* ApproxModDown with the BGV factors t^-1 (mod p_j) before and t (mod q_i) after the conversion (dcrtpoly-impl.h:966-1005
 * with t > 0; tables tInvModp / tModqPrecon of CryptoParametersBGVRNS are derived inside).  Same layout and workspace. */
fhe_status fhe_approx_mod_down_bgv(fhe_ks_plan* plan, const uint64_t* x, uint32_t sizeQl, uint64_t t, uint32_t batch,
                                   uint64_t* out, void* ws, size_t wsBytes, void* stream);

/* ---- a15: CKKS rescale ---------------------------------------------------------------------------
 * Replaces DCRTPolyImpl::DropLastElementAndScale (dcrtpoly-impl.h:693-712) with the tables of
 * CryptoParametersCKKSRNS (src/pke/lib/scheme/ckksrns/ckksrns-cryptoparameters.cpp:60-81) computed inside.
 * The tower uses context limbs [0,sizeQl). x[batch][sizeQl][N] EVALUATION -> out[batch][sizeQl-1][N].
 * ws: device workspace of fhe_rescale_workspace_bytes(). */
size_t     fhe_rescale_workspace_bytes(const fhe_ctx* ctx, uint32_t sizeQl, uint32_t batch);
fhe_status fhe_rescale(fhe_ctx* ctx, const uint64_t* x, uint32_t sizeQl, uint32_t batch, uint64_t* out, void* ws,
                       size_t wsBytes, void* stream);
/* The same over any limbs of the context (limbIdx[sizeQl], NULL = the leading ones; the last entry is the dropped limb) with the
 * CALLER's tables, host arrays of sizeQl-1 residues as DropLastElementAndScale receives them (dcrtpoly-impl.h:693-694).  When
 * QlQlInvModqlDivqlModq[i] == -qlInvModq[i] mod q_i (the reference's own tables) on a ring of two static passes, the call is 4
 * launches: the switched tower and its transform never go to HBM (fhe_hip.cpp rescale_run).  out must not alias x. */
fhe_status fhe_rescale_limbs(fhe_ctx* ctx, const uint64_t* x, const uint32_t* limbIdx, uint32_t sizeQl,
                             const uint64_t* QlQlInvModqlDivqlModq, const uint64_t* qlInvModq, uint32_t batch, uint64_t* out,
                             void* ws, size_t wsBytes, void* stream);
/* The two elements of one ciphertext — towers x0, x1 -> out0, out1, each allocated on its own, any distance apart — in the same
 * four launches (LeveledSHECKKSRNS::ModReduceInternalInPlace, ckksrns-leveledshe.cpp:172-191, applies the member to both elements
 * with the same tables); ws of fhe_rescale_workspace_bytes(ctx, sizeQl, 2). */
fhe_status fhe_rescale_limbs_pair(fhe_ctx* ctx, const uint64_t* x0, const uint64_t* x1, const uint32_t* limbIdx, uint32_t sizeQl,
                                  const uint64_t* QlQlInvModqlDivqlModq, const uint64_t* qlInvModq, uint64_t* out0, uint64_t* out1,
                                  void* ws, size_t wsBytes, void* stream);
/* DCRTPolyImpl::ModReduce (dcrtpoly-impl.h:736-755) — BGV modulus switching by the last limb with plaintext modulus t
 * (tables negtInvModq / qlInvModq / tModqPrecon of CryptoParametersBGVRNS are derived inside): x [batch][sizeQl][N] in
 * `evalFormat`, out [batch][sizeQl-1][N] in the same format; ws as for fhe_rescale. */
fhe_status fhe_mod_reduce(fhe_ctx* ctx, const uint64_t* x, uint32_t sizeQl, uint64_t t, int evalFormat, uint32_t batch,
                          uint64_t* out, void* ws, size_t wsBytes, void* stream);

/* ---- a17: ScaleAndRound family (BFV HPS) ----------------------------------------------------------------
 * fhe_sr_plan_create keeps the caller's tables on the device. They are the reference's own
 * (CryptoParametersBFVRNS getters, e.g. GettRSHatInvModsDivsModr / GettRSHatInvModsDivsFrac,
 * src/pke/include/schemerns/rns-cryptoparameters.h:829-848): tab is [sizeO][sizeI+1] exactly as the reference
 * indexes it, frac is [sizeI] (NULL selects ApproxScaleAndRound, which has no fractional part).
 * fhe_scale_and_round replaces DCRTPolyImpl::ScaleAndRound (dcrtpoly-impl.h:1513-1628) / ApproxScaleAndRound
 * (:1470-1510): x is [batch][sizeI+sizeO][N] COEFFICIENT; outputFirst != 0 means the output basis is the FIRST
 * sizeO limbs of x (the reference decides this by comparing the first moduli, :1527-1534), else the LAST sizeO.
 * out is [batch][sizeO][N].
 * fhe_scale_and_round_p_over_q replaces ScaleAndRoundPOverQ (:1674-1689): x [batch][sizeQ+1][N] over the context
 * limbs limbIdx[0..sizeQ] -> out [batch][sizeQ][N]. */
typedef struct fhe_sr_plan fhe_sr_plan;
fhe_status fhe_sr_plan_create(fhe_ctx* ctx, uint32_t sizeI, const uint32_t* outLimbIdx, uint32_t sizeO,
                              const uint64_t* tab, const double* frac, fhe_sr_plan** out);
void       fhe_sr_plan_destroy(fhe_sr_plan* plan);
fhe_status fhe_scale_and_round(fhe_sr_plan* plan, const uint64_t* x, int outputFirst, uint64_t* out, uint32_t batch,
                               void* stream);
fhe_status fhe_scale_and_round_p_over_q(fhe_ctx* ctx, const uint64_t* x, const uint32_t* limbIdx, uint32_t sizeQ,
                                        uint64_t* out, uint32_t batch, void* stream);

/* ScaleAndRound -> NativePoly modulo t, the BFV decryption step (dcrtpoly-impl.h:1190-1467): x [batch][sizeQ][N]
 * COEFFICIENT over the context limbs limbIdx (NULL: 0..sizeQ-1) -> out [batch][N] residues mod t.  Tables are the
 * reference's tQHatInvModqDivqModt, tQHatInvModqBDivqModt, tQHatInvModqDivqFrac, tQHatInvModqDivqBFrac (the two "B"
 * tables may be NULL when max(q_i) and sizeQ keep the reference on its unsplit branches); which of the reference's eight
 * branches runs is decided from (max q_i, t, sizeQ) exactly as there.  fhe_scale_and_round_behz_decrypt is the BEHZ
 * overload (:1631-1671) with tgammaQHatModq / negInvqModtgamma, gamma = 2^26. */
fhe_status fhe_scale_and_round_native(fhe_ctx* ctx, const uint64_t* x, const uint32_t* limbIdx, uint32_t sizeQ, uint64_t t,
                                      const uint64_t* tabModt, const uint64_t* tabBModt, const double* frac,
                                      const double* bfrac, uint32_t batch, uint64_t* out, void* stream);
fhe_status fhe_scale_and_round_behz_decrypt(fhe_ctx* ctx, const uint64_t* x, const uint32_t* limbIdx, uint32_t sizeQ,
                                            uint64_t tgamma, const uint64_t* tgammaQHatModq,
                                            const uint64_t* negInvqModtgamma, uint32_t batch, uint64_t* out, void* stream);

/* ---- a18: BEHZ base conversions (BFV) ---------------------------------------------------------------------
 * fhe_behz_create builds the tables of CryptoParametersBFVRNS's BEHZ block
 * (src/pke/lib/scheme/bfvrns/bfvrns-cryptoparameters.cpp:673-850) for the basis Q (context limbs qLimbIdx) and
 * Bsk = B ∪ {m_sk} (context limbs bskLimbIdx, numQ+1 of them, m_sk last), plaintext modulus t, m̃ = 2^16.
 * fhe_param_behz_bsk returns the Bsk moduli/roots the reference would pick (:682-711) so that a caller can put them
 * into the context; it returns numQ+1, or 0 if m_sk would need more than 60 bits.
 * Towers are x[batch][numQ+numBsk][N].  numQ <= 63 (a tower of Q and Bsk limbs has at most 127 rows): up to 15 Q limbs run on kernels that
 * keep one coefficient's residues in registers, 16 ... 63 on kernels that keep them in a per-lane array (same exact sums, same residues).
 *   fhe_behz_q_to_bsk  = FastBaseConvqToBskMontgomery (dcrtpoly-impl.h:1694-1786): Q rows in `evalFormat` on entry,
 *                        all rows EVALUATION on return (ws: fhe_behz_workspace_bytes, only needed for EVALUATION input);
 *   fhe_behz_floorq    = FastRNSFloorq (:1791-1840), COEFFICIENT, in place;
 *   fhe_behz_conv_sk   = FastBaseConvSK (:1845-1929), COEFFICIENT, result out[batch][numQ][N]. */
typedef struct fhe_behz fhe_behz;
uint32_t   fhe_param_behz_bsk(uint32_t logN, uint32_t numQ, const uint64_t* q, uint64_t t, uint64_t* bsk, uint64_t* psiBsk);
fhe_status fhe_behz_create(fhe_ctx* ctx, const uint32_t* qLimbIdx, uint32_t numQ, const uint32_t* bskLimbIdx, uint64_t t,
                           fhe_behz** out);
void       fhe_behz_destroy(fhe_behz* plan);
/* The CALLER's tables in place of the derived ones, one member at a time — the vectors the reference passes to that member
 * (dcrtpoly-interface.h: FastBaseConvqToBskMontgomery, FastRNSFloorq, FastBaseConvSK), flattened row-major in the reference's index
 * order: QHatModbsk / qInvModbsk [numQ][numBsk], BHatModq [numB][numQ], vectors over Q, Bsk or B.  With an override the member
 * computes with the caller's VALUES (whatever they are), as DCRTPolyImpl does; a DCRTPoly backend keeps one plan per member and
 * table content.  The Barrett constants are functions of the moduli and stay derived. */
fhe_status fhe_behz_override_q_to_bsk(fhe_behz* plan, const uint64_t* mtildeQHatInvModq, const uint64_t* QHatModbsk,
                                      const uint64_t* QHatModmtilde, const uint64_t* QModbsk, uint64_t negQInvModmtilde,
                                      const uint64_t* mtildeInvModbsk);
fhe_status fhe_behz_override_floorq(fhe_behz* plan, const uint64_t* tQHatInvModq, const uint64_t* QHatModbsk,
                                    const uint64_t* qInvModbsk, const uint64_t* tQInvModbsk);
fhe_status fhe_behz_override_conv_sk(fhe_behz* plan, const uint64_t* BHatInvModb, const uint64_t* BHatModmsk, uint64_t BInvModmsk,
                                     const uint64_t* BHatModq, const uint64_t* BModq);

size_t     fhe_behz_workspace_bytes(const fhe_behz* plan, uint32_t batch);
fhe_status fhe_behz_q_to_bsk(fhe_behz* plan, uint64_t* x, int evalFormat, uint32_t batch, void* ws, size_t wsBytes,
                             void* stream);
fhe_status fhe_behz_floorq(fhe_behz* plan, uint64_t* x, uint32_t batch, void* stream);
fhe_status fhe_behz_conv_sk(fhe_behz* plan, const uint64_t* x, uint64_t* out, uint32_t batch, void* stream);
/* LeveledSHEBFVRNS::EvalMult, BEHZ branch, no relinearisation (src/pke/lib/scheme/bfvrns/bfvrns-leveledshe.cpp:198-445;
 * config 5 of BASELINE.json): the four input elements [batch][numQ][N] EVALUATION -> three product elements
 * [batch][numQ][N], COEFFICIENT as the reference returns them, or EVALUATION when outEval != 0 (the SetFormat that
 * LeveledSHEBase::EvalMult(ct,ct,key) applies before KeySwitchCore, base-leveledshe.cpp:204-205). */
size_t     fhe_bfv_eval_mult_behz_workspace_bytes(const fhe_behz* plan, uint32_t batch);
fhe_status fhe_bfv_eval_mult_behz(fhe_behz* plan, const uint64_t* a0, const uint64_t* a1, const uint64_t* b0,
                                  const uint64_t* b1, uint64_t* d0, uint64_t* d1, uint64_t* d2, int outEval,
                                  uint32_t batch, void* ws, size_t wsBytes, void* stream);
/* The same with relinearisation — LeveledSHEBase::EvalMult(ct, ct, key) on BFV/BEHZ ciphertexts (base-leveledshe.cpp:201-214):
 * EvalMultNoRelin, SetFormat(EVALUATION), KeySwitchCore on the third element, c0 += ks0, c1 += ks1.  The context holds Q
 * as its leading limbs (the key-switch plan's Q), P, and the Bsk limbs; outputs [batch][numQ][N] EVALUATION. */
size_t     fhe_bfv_eval_mult_relin_workspace_bytes(const fhe_behz* plan, const fhe_ks_plan* ks, uint32_t batch);
fhe_status fhe_bfv_eval_mult_relin_behz(fhe_behz* plan, fhe_ks_plan* ks, const fhe_ks_key* key, const uint64_t* a0,
                                        const uint64_t* a1, const uint64_t* b0, const uint64_t* b1, uint64_t* c0, uint64_t* c1,
                                        uint32_t batch, void* ws, size_t wsBytes, void* stream);

/* ---- parity helper: whole-tower checksums ----------------------------------------------------------------
 * out[row] = { sum_i w_i, sum_i (2i + 1) * w_i } mod 2^64 over the row's N words, for every limb-row of x[rows][N] (rows = batch *
 * nLimbs; the second word depends on the ORDER of the words); out is DEVICE memory, uint64_t[rows][2].  One read of the batch: bench.py and the full-shape tests compare EVERY
 * tower of a resident batch with the oracle's words summed on the host, instead of sampling a few towers. */
fhe_status fhe_checksum(fhe_ctx* ctx, const uint64_t* x, uint32_t rows, uint64_t* out, void* stream);

/* Kernel launches issued by this library since it was loaded, by kernel: writes lines "<kernel> <launches>\n" (most frequent first)
 * into buf (at most cap bytes, NUL-terminated when cap > 0) and returns the length the full text needs; *total, when given, receives
 * the sum.  (Tuning aid: the launch count of one pke operation is the difference of two calls.) */
size_t fhe_launch_stats(char* buf, size_t cap, uint64_t* total);

/* ---- host-side parameter helpers (no device work) -------------------------------------------------
 * Number theory the reference uses to pick moduli and roots, restated with 64-bit arithmetic so that a
 * caller can reproduce the reference's (N, q_i, psi_i) without linking OpenFHE:
 * FirstPrime/LastPrime/NextPrime/PreviousPrime (src/core/include/math/nbtheory-impl.h:329-393),
 * RootOfUnity = the minimum primitive m-th root (:183-231), the ILDCRTParams(order, depth, bits) chain
 * (src/core/include/lattice/hal/default/ildcrtparams.h:100-117) and the HYBRID auxiliary basis P of
 * CryptoParametersRNS::PrecomputeCRTTables (src/pke/lib/schemerns/rns-cryptoparameters.cpp:128-176, with the
 * CKKS prime step 2N, src/pke/lib/scheme/ckksrns/ckksrns-cryptoparameters.cpp:185-188). */
uint64_t   fhe_param_first_prime(uint32_t bits, uint64_t m);
uint64_t   fhe_param_last_prime(uint32_t bits, uint64_t m);
uint64_t   fhe_param_next_prime(uint64_t q, uint64_t m);
uint64_t   fhe_param_previous_prime(uint64_t q, uint64_t m);
uint64_t   fhe_param_root_of_unity(uint64_t m, uint64_t q);
fhe_status fhe_param_dcrt_chain(uint32_t order, uint32_t nLimbs, uint32_t bits, uint64_t* q, uint64_t* psi);
/* returns sizeP (0 on error); p/psiP need capacity >= 64 */
uint32_t   fhe_param_select_p(uint32_t logN, uint32_t sizeQ, const uint64_t* q, uint32_t numPartQ, uint32_t auxBits,
                              uint64_t* p, uint64_t* psiP);
/* FindAutomorphismIndex2nComplex (src/core/lib/math/nbtheory2.cpp:243-262): automorphism index 5^index mod m of a CKKS
 * rotation by `index` slots (m = 2N, a power of two); 0 on error */
uint32_t   fhe_param_find_automorphism_index_2n_complex(int32_t index, uint32_t m);

/* ---- f3: sampled towers on the device (optional) ------------------------------------
 * The sampling constructors of DCRTPolyImpl (dcrtpoly-impl.h:126-205) as device kernels: out[batch][nLimbs][N], COEFFICIENT format,
 *   fhe_sample_uniform   every word uniform in [0, q_limb)          DiscreteUniformGeneratorImpl::GenerateVector (discreteuniformgenerator.h:55-77)
 *   fhe_sample_gaussian  ONE integer per coefficient, stored modulo every limb (negative k as q - |k|): Peikert's inversion over the
 *                        reference's table (discretegaussiangenerator-impl.h:75-115), 1 < sigma < 300
 *   fhe_sample_ternary   ONE value of {-1, 0, 1} per coefficient, uniform (TernaryUniformGeneratorImpl::GenerateVector, h = 0)
 * Generator: Philox4x32-10 keyed by `seed`, counter = (element, draw, streamId) — NOT the reference's sequential Blake2 stream: the
 * distributions are the reference's, the words are not (the survey marks this row "gives up bit-parity with Blake2; keep optional");
 * the oracle restates the same construction word for word.  Give every sampled tower set its own streamId. */
fhe_status fhe_sample_uniform(fhe_ctx* ctx, uint64_t* out, const uint32_t* limbIdx, uint32_t nLimbs, uint32_t batch, uint64_t seed,
                              uint32_t streamId, void* stream);
fhe_status fhe_sample_gaussian(fhe_ctx* ctx, uint64_t* out, const uint32_t* limbIdx, uint32_t nLimbs, uint32_t batch, double sigma,
                               uint64_t seed, uint32_t streamId, void* stream);
fhe_status fhe_sample_ternary(fhe_ctx* ctx, uint64_t* out, const uint32_t* limbIdx, uint32_t nLimbs, uint32_t batch, uint64_t seed,
                              uint32_t streamId, void* stream);

/* ---- measurement helper ----------------------------------------------------------------------------
 * Runs `iters` back-to-back launches of fwd (dir=0), inv (dir=1) or fwd+inv (dir=2) NTT on x — or of a single
 * pass kernel of a two-pass ring: 10/11 = column/row pass of the forward, 12/13 = row/column pass of the
 * inverse (timing only; the data is then not a transform) — and returns
 * the average wall time per launch-set in milliseconds measured with hipEvents on `stream`
 * (bench.py's roofline leg: torch.cuda.Event only sees torch's stream). */
fhe_status fhe_time_ntt(fhe_ctx* ctx, uint64_t* x, const uint32_t* limbIdx, uint32_t nLimbs, uint32_t batch, int dir,
                        int iters, void* stream, float* msPerIter);

#ifdef __cplusplus
}
#endif
#endif
