// emu_runtime.h — TEST INFRASTRUCTURE ONLY.
//
// A minimal lane emulator so that the HIP kernels of openfhe-development_amd/csrc (index math, LDS
// exchange schedule, lazy-reduction ranges) can be executed and checked against the oracle on a machine
// with no GPU.  One OS thread per lane of a 256-thread workgroup, a pthread barrier for s_barrier,
// workgroups run one after another.  A kernel that never reaches a barrier needs no concurrent lanes: its lanes run one after
// the other on the launching thread (a launch site is tried that way first; the first barrier a lane meets — before any store to
// memory, every kernel of the library exchanges through LDS before it writes results — sends the site to the lane threads for good).  It is compiled ONLY into tests/emu/libfhe_emu.so; the product
// library never contains or falls back to it.
#ifndef FHE_EMU_RUNTIME_H
#define FHE_EMU_RUNTIME_H
#include <cstddef>
#include <cstdint>
#include <functional>

namespace fhe_emu {
struct Tls {
    uint32_t tid, bid, nblk;
    bool sequential = false;  // the lanes of the block are being run one after the other on the launching thread
};
extern thread_local Tls tls;
void block_sync();
void* block_shared(size_t bytes);
// needsLaneThreads: per launch site, set once a lane of the site's kernel met a barrier
void launch(uint32_t grid, uint32_t threads, const std::function<void()>& body, bool* needsLaneThreads = nullptr);
}  // namespace fhe_emu
#endif
