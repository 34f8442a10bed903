// emu_runtime.cpp — TEST INFRASTRUCTURE ONLY (see emu_runtime.h).
#include "emu_runtime.h"

#include <pthread.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <vector>

namespace fhe_emu {
thread_local Tls tls;

namespace {
// one pool of lane threads per workgroup size: 256 (every kernel but one) and 64 / 128 / 512 (the row pass of ntt_row8.h: tiles of 1 / 2 / 8 waves)
struct Pool {
    const uint32_t kLanes;
    pthread_barrier_t start, stop, sync;
    std::vector<std::thread> workers;
    const std::function<void()>* body = nullptr;
    uint32_t bid = 0, nblk = 0;
    std::atomic<bool> quit{false};
    alignas(64) unsigned char shared[160 * 1024];
    explicit Pool(uint32_t lanes) : kLanes(lanes) {
        pthread_barrier_init(&start, nullptr, kLanes + 1);
        pthread_barrier_init(&stop, nullptr, kLanes + 1);
        pthread_barrier_init(&sync, nullptr, kLanes);
        for (uint32_t t = 0; t < kLanes; ++t)
            workers.emplace_back([this, t] {
                for (;;) {
                    pthread_barrier_wait(&start);
                    if (quit.load())
                        return;
                    tls.tid  = t;
                    tls.pool = this;
                    tls.bid  = bid;
                    tls.nblk = nblk;
                    (*body)();
                    pthread_barrier_wait(&stop);
                }
            });
    }
    ~Pool() {
        quit.store(true);
        pthread_barrier_wait(&start);
        for (auto& w : workers)
            w.join();
    }
};
Pool& pool(uint32_t lanes) {  // (created on first use, under launchMutex)
    static Pool* pools[4] = {nullptr, nullptr, nullptr, nullptr};  // 64, 128, 256, 512 lanes
    const int i = lanes == 64 ? 0 : lanes == 128 ? 1 : lanes == 256 ? 2 : 3;
    if (!pools[i])
        pools[i] = new Pool(lanes);
    return *pools[i];
}
std::mutex launchMutex;
}  // namespace

void block_sync() {
    if (tls.sequential) {
        std::fprintf(stderr, "fhe_emu: a kernel launched with FHE_LAUNCH met a barrier (its launch site must use FHE_LAUNCH_BARRIER)\n");
        std::abort();
    }
    pthread_barrier_wait(&static_cast<Pool*>(tls.pool)->sync);
}
void* block_shared(size_t bytes) {
    if (bytes > sizeof(Pool::shared))
        std::abort();
    if (tls.sequential) {  // (scratch of a kernel without barriers: private to the launching thread)
        static thread_local std::vector<unsigned char> mine;
        if (mine.size() < bytes)
            mine.resize(bytes);
        return mine.data();
    }
    return static_cast<Pool*>(tls.pool)->shared;
}
void launch(uint32_t grid, uint32_t threads, const std::function<void()>& body, bool laneThreads) {
    if (threads != 64 && threads != 128 && threads != 256 && threads != 512)
        std::abort();
    const uint32_t kLanes = threads;
    // FHE_EMU_SKIP=1: kernels do nothing (results are garbage).  For measuring the HOST side of a call sequence — pke's and the
    // backend's own time per operation — on a machine without a GPU; never set by the tests.
    static const bool skip = std::getenv("FHE_EMU_SKIP") != nullptr;
    if (skip)
        return;
    if (!laneThreads) {  // lanes one after the other on this thread (several host threads may do so at once)
        const Tls saved = tls;
        tls.sequential  = true;
        tls.nblk        = grid;
        for (uint32_t b = 0; b < grid; ++b)
            for (uint32_t t = 0; t < kLanes; ++t) {
                tls.bid = b, tls.tid = t;
                body();
            }
        tls = saved;
        return;
    }
    std::lock_guard<std::mutex> lk(launchMutex);
    Pool& p = pool(threads);
    p.body  = &body;
    p.nblk  = grid;
    for (uint32_t b = 0; b < grid; ++b) {
        p.bid = b;
        pthread_barrier_wait(&p.start);
        pthread_barrier_wait(&p.stop);
    }
}
}  // namespace fhe_emu
