// TEST INFRASTRUCTURE — main() of the reference's pke unit tests built on tests/hal/minigtest/gtest/gtest.h
#include "gtest/gtest.h"

#include <cstdio>
#include <cstdlib>

// switches of the reference's test driver (test/Main_TestAll.cpp: which math backends the templated core tests cover); all on
bool TestB2 = true, TestB4 = true, TestB6 = true, TestNative = true;

extern "C" void fhe_hal_stats(uint64_t out[4]) __attribute__((weak));
extern "C" int fhe_hal_available(void) __attribute__((weak));
extern "C" void fhe_hal_other_host_counts(uint64_t out[2]) __attribute__((weak));
extern "C" void fhe_hal_composite_stats(uint64_t out[3]) __attribute__((weak));

int main(int argc, char** argv) {
    ::testing::InitGoogleTest(&argc, argv);
    const int rc = RUN_ALL_TESTS();
    if (fhe_hal_stats) {
        uint64_t st[4];
        fhe_hal_stats(st);
        std::printf("hal: available %d deviceOps %llu hostOps %llu h2dBytes %llu d2hBytes %llu\n", fhe_hal_available ? fhe_hal_available() : -1,
                    (unsigned long long)st[0], (unsigned long long)st[1], (unsigned long long)st[2], (unsigned long long)st[3]);
        if (fhe_hal_other_host_counts) {  // not fall-backs: rings below the device library's domain (N < 16), words produced by the host
            uint64_t o[2];
            fhe_hal_other_host_counts(o);
            std::printf("hal-other: hostOpsOnRingsBelow16 %llu hostProducedWords(SetElementAtIndex etc.) %llu\n", (unsigned long long)o[0],
                        (unsigned long long)o[1]);
        }
        if (fhe_hal_composite_stats) {
            uint64_t c[3];
            fhe_hal_composite_stats(c);
            std::printf("halcomposite calls %llu checksIdentical %llu checksDiffered %llu\n", (unsigned long long)c[0], (unsigned long long)c[1],
                        (unsigned long long)c[2]);
        }
    }
    else
        std::printf("hal: stock backend\n");
    return rc;
}
