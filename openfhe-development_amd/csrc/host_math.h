// host_math.h — host-side number theory and table builders of the HIP backend (product code).
//
// Builds, with 64-bit modular arithmetic only, the constant operands the kernels consume:
//  * twiddle tables of ChineseRemainderTransformFTTNat::PreCompute
//    (src/core/include/math/hal/intnat/transformnat-impl.h:714-756),
//  * CRT conversion tables of CryptoParametersRNS::PrecomputeCRTTables
//    (src/pke/lib/schemerns/rns-cryptoparameters.cpp:199-349) — the reference forms them from BigInteger
//    products and quotients; every stored value is a residue of a product of moduli, so modular products
//    give identical numbers,
//  * CKKS rescale tables (src/pke/lib/scheme/ckksrns/ckksrns-cryptoparameters.cpp:60-81),
//  * modulus chains / roots for self-contained benchmarks (nbtheory-impl.h:183-231, 329-393;
//    ildcrtparams.h:100-117).
#ifndef FHE_HOST_MATH_H
#define FHE_HOST_MATH_H
#include <cstdint>
#include <vector>

namespace fhe {
namespace host {

typedef unsigned __int128 u128;

inline uint64_t mulmod(uint64_t a, uint64_t b, uint64_t q) { return (uint64_t)(((u128)a * b) % q); }
inline uint64_t powmod(uint64_t a, uint64_t e, uint64_t q) {
    uint64_t r = 1 % q;
    a %= q;
    for (; e; e >>= 1) {
        if (e & 1)
            r = mulmod(r, a, q);
        a = mulmod(a, a, q);
    }
    return r;
}
inline uint64_t invmod(uint64_t a, uint64_t q) { return powmod(a % q, q - 2, q); }  // q prime
inline uint32_t bitlen(uint64_t x) {
    uint32_t r = 0;
    for (; x; x >>= 1)
        ++r;
    return r;
}
inline uint64_t shoup(uint64_t w, uint64_t q) { return (uint64_t)((((u128)w) << 64) / q); }  // PrepModMulConst
inline uint64_t barrett_mu(uint64_t q) { return (uint64_t)(((u128)1 << (2 * bitlen(q) + 3)) / q); }  // ComputeMu
inline void mu128(uint64_t q, uint64_t* out2) {  // floor(2^128/q), q odd
    u128 m  = (~(u128)0) / q;
    out2[0] = (uint64_t)m;
    out2[1] = (uint64_t)(m >> 64);
}
inline uint32_t bitrev(uint32_t x, uint32_t nbits) {
    uint32_t r = 0;
    for (uint32_t i = 0; i < nbits; ++i)
        r |= ((x >> i) & 1u) << (nbits - 1 - i);
    return r;
}

bool is_prime(uint64_t n);
bool is_primitive_root_2n(uint64_t psi, uint64_t twoN, uint64_t q);
uint64_t first_prime(uint32_t bits, uint64_t m);
uint64_t last_prime(uint32_t bits, uint64_t m);
uint64_t previous_prime(uint64_t q, uint64_t m);
uint64_t next_prime(uint64_t q, uint64_t m);
uint64_t min_root_of_unity(uint64_t m, uint64_t q);

// product of mods[k] (k in sel, k != skip) reduced mod `mod`
uint64_t prod_mod(const std::vector<uint64_t>& mods, int skip, uint64_t mod);

}  // namespace host
}  // namespace fhe
#endif
