// emu_runtime.h — TEST INFRASTRUCTURE ONLY.
//
// A minimal lane emulator so that the HIP kernels of openfhe-development_amd/csrc (index math, LDS
// exchange schedule, lazy-reduction ranges) can be executed and checked against the oracle on a machine
// with no GPU.  A kernel launched with FHE_LAUNCH_BARRIER (its lanes exchange through LDS) gets one OS thread per lane of a 64- to 512-thread
// workgroup and a pthread barrier for s_barrier, workgroups one after another; a kernel launched with FHE_LAUNCH has no barrier, so its
// lanes run one after the other on the launching thread (a barrier met there aborts: the launch site declared the wrong kind).
// It is compiled ONLY into tests/emu/libfhe_emu.so; the product library never contains or falls back to it.
#ifndef FHE_EMU_RUNTIME_H
#define FHE_EMU_RUNTIME_H
#include <cstddef>
#include <cstdint>
#include <functional>

namespace fhe_emu {
struct Tls {
    uint32_t tid, bid, nblk;
    bool sequential = false;  // the lanes of the block are being run one after the other on the launching thread
    void* pool = nullptr;     // the lane-thread pool this lane belongs to (one per workgroup size)
};
extern thread_local Tls tls;
void block_sync();
void* block_shared(size_t bytes);
void launch(uint32_t grid, uint32_t threads, const std::function<void()>& body, bool laneThreads);
}  // namespace fhe_emu
#endif
