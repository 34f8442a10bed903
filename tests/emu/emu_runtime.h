// emu_runtime.h — TEST INFRASTRUCTURE ONLY.
//
// A minimal lane emulator so that the HIP kernels of openfhe-development_amd/csrc (index math, LDS
// exchange schedule, lazy-reduction ranges) can be executed and checked against the oracle on a machine
// with no GPU.  One OS thread per lane of a 256-thread workgroup, a pthread barrier for s_barrier,
// workgroups run one after another.  It is compiled ONLY into tests/emu/libfhe_emu.so; the product
// library never contains or falls back to it.
#ifndef FHE_EMU_RUNTIME_H
#define FHE_EMU_RUNTIME_H
#include <cstddef>
#include <cstdint>
#include <functional>

namespace fhe_emu {
struct Tls {
    uint32_t tid, bid, nblk;
};
extern thread_local Tls tls;
void block_sync();
void* block_shared(size_t bytes);
void launch(uint32_t grid, uint32_t threads, const std::function<void()>& body);
}  // namespace fhe_emu
#endif
