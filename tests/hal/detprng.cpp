// TEST INFRASTRUCTURE ONLY — a deterministic engine for OpenFHE's external-PRNG hook
// (PseudoRandomNumberGenerator::InitPRNGEngine(path), src/core/lib/math/distributiongenerator.cpp:60-90): every engine
// instance starts from the same state, so that two processes (the stock backend and the HIP backend) draw identical keys,
// noise and ciphertext randomness and their limbs can be compared word for word.  Never used outside tests.
#include <cstdint>

#include "utils/prng/prng.h"

namespace {
class SplitMix final : public PRNG {
    uint64_t s = 0x243F6A8885A308D3ull;

public:
    result_type operator()() override {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z          = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z          = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return static_cast<result_type>((z ^ (z >> 31)) >> 32);
    }
};
}  // namespace

extern "C" PRNG* createEngineInstance() {
    return new SplitMix;
}
