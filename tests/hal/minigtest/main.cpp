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
extern "C" size_t fhe_hal_member_stats(char* buf, size_t cap) __attribute__((weak));
extern "C" void fhe_hal_out_of_domain(uint64_t out[4]) __attribute__((weak));
extern "C" size_t fhe_hal_decline_stats(char* buf, size_t cap) __attribute__((weak));
#include <string>

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
        if (fhe_hal_member_stats) {  // "halmember <member> <device ops> <host-mirror executions> <host reads>" (the per-member allow-list of the tests)
            std::string buf(fhe_hal_member_stats(nullptr, 0), '\0');
            fhe_hal_member_stats(&buf[0], buf.size());
            size_t at = 0;
            while (at < buf.size() && buf[at]) {
                const size_t nl = buf.find('\n', at);
                std::printf("halmember %s\n", buf.substr(at, nl - at).c_str());
                at = nl + 1;
            }
        }
        if (fhe_hal_out_of_domain) {  // why operations left the device library's domain
            uint64_t o[4];
            fhe_hal_out_of_domain(o);
            std::printf("haldomain ringOutside16to2p17 %llu modulusOutside %llu moreThan128Moduli %llu otherRootOfUnity %llu\n", (unsigned long long)o[0],
                        (unsigned long long)o[1], (unsigned long long)o[2], (unsigned long long)o[3]);
        }
        if (fhe_hal_decline_stats) {  // device plans / contexts the library declined to build, with its message
            std::string buf(fhe_hal_decline_stats(nullptr, 0), '\0');
            fhe_hal_decline_stats(&buf[0], buf.size());
            size_t at = 0;
            while (at < buf.size() && buf[at]) {
                const size_t nl = buf.find('\n', at);
                std::printf("haldecline %s\n", buf.substr(at, nl - at).c_str());
                at = nl + 1;
            }
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
