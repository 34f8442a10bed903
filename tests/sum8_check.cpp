// TEST: the 8-term column sums + single 64-bit Barrett reduction of modarith.h (sum8_add / sum8_reduce) against 128-bit
// integer arithmetic, over every modulus size the library admits (4..60 bits) with random and extreme operands.
// Host compile of the device header (-DFHE_EMU), no library needed.
#include <cstdio>
#include <random>

#include "modarith.h"

typedef unsigned __int128 u128;

int main() {
    std::mt19937_64 gen(7);
    long checked = 0;
    for (uint32_t k = 4; k <= 60; ++k) {
        for (int rep = 0; rep < 400; ++rep) {
            uint64_t q = (gen() >> (64 - k)) | ((uint64_t)1 << (k - 1)) | 1u;  // odd, exactly k bits
            if (rep == 0)
                q = ((uint64_t)1 << k) - 1;  // largest k-bit value (odd)
            if (rep == 1)
                q = ((uint64_t)1 << (k - 1)) + 1;  // smallest odd k-bit value
            const u128 mu = ~(u128)0 / q;  // floor((2^128 - 1)/q) = floor(2^128/q) for odd q > 1
            const uint64_t mulo = (uint64_t)mu, muhi = (uint64_t)(mu >> 64);
            const int n = 1 + (int)(gen() % 8);
            fhe::sum8 s;
            fhe::sum8_clear(s);
            fhe::sum8s s30;
            fhe::sum8s_clear(s30);
            u128 S = 0;
            for (int i = 0; i < n; ++i) {
                uint64_t a = gen() >> 4, b = gen() % q;  // a < 2^60, b < q
                if (rep % 3 == 0)
                    a = ((uint64_t)1 << 60) - 1, b = q - 1;
                if (rep % 7 == 1)
                    a = gen() % q;  // a product of two residues
                fhe::sum8_add(s, a, b);
                uint32_t a0, a1, b0, b1;
                fhe::split30(a, a0, a1);
                fhe::split30(b, b0, b1);
                fhe::sum8s_add(s30, a0, a1, b0, b1);
                S += (u128)a * b;
            }
            const uint64_t got = fhe::sum8_reduce(s, q, k, mulo, muhi), want = (uint64_t)(S % q);
            const uint64_t got30 = fhe::sum8s_reduce(s30, q, k, mulo, muhi);
            if (got30 != want) {
                std::printf("sum8s mismatch: k=%u q=%llu n=%d got=%llu want=%llu\n", k, (unsigned long long)q, n,
                            (unsigned long long)got30, (unsigned long long)want);
                return 2;
            }
            if (got != want) {
                std::printf("sum8 mismatch: k=%u q=%llu n=%d got=%llu want=%llu\n", k, (unsigned long long)q, n,
                            (unsigned long long)got, (unsigned long long)want);
                return 1;
            }
            ++checked;
        }
    }
    std::printf("sum8_check OK (%ld sums)\n", checked);
    return 0;
}
