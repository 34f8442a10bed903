// emu_runtime.cpp — TEST INFRASTRUCTURE ONLY (see emu_runtime.h).
#include "emu_runtime.h"

#include <pthread.h>

#include <atomic>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <vector>

namespace fhe_emu {
thread_local Tls tls;

namespace {
constexpr uint32_t kLanes = 256;
struct Pool {
    pthread_barrier_t start, stop, sync;
    std::vector<std::thread> workers;
    const std::function<void()>* body = nullptr;
    uint32_t bid = 0, nblk = 0;
    std::atomic<bool> quit{false};
    alignas(64) unsigned char shared[160 * 1024];
    Pool() {
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
Pool& pool() {
    static Pool p;
    return p;
}
std::mutex launchMutex;
}  // namespace

namespace {
struct MetBarrier {};
}  // namespace
void block_sync() {
    if (tls.sequential)
        throw MetBarrier{};
    pthread_barrier_wait(&pool().sync);
}
void* block_shared(size_t bytes) {
    if (bytes > sizeof(pool().shared))
        std::abort();
    return pool().shared;
}
void launch(uint32_t grid, uint32_t threads, const std::function<void()>& body, bool* needsLaneThreads) {
    if (threads != kLanes)
        std::abort();
    // FHE_EMU_SKIP=1: kernels do nothing (results are garbage).  For measuring the HOST side of a call sequence — pke's and the
    // backend's own time per operation — on a machine without a GPU; never set by the tests.
    static const bool skip = std::getenv("FHE_EMU_SKIP") != nullptr;
    if (skip)
        return;
    uint32_t first = 0;
    if (needsLaneThreads && !*needsLaneThreads) {
        // lanes one after the other on this thread; the first barrier (lane 0 of the first block, before it stored anything) ends the attempt
        const Tls saved = tls;
        try {
            tls.sequential = true;
            tls.nblk       = grid;
            for (uint32_t b = 0; b < grid; ++b)
                for (uint32_t t = 0; t < kLanes; ++t) {
                    tls.bid = b, tls.tid = t;
                    body();
                }
            tls = saved;
            return;
        }
        catch (const MetBarrier&) {
            if (tls.bid != 0 || tls.tid != 0)
                std::abort();  // (a kernel whose lanes reach barriers data-dependently: not a kernel of this library)
            tls               = saved;
            *needsLaneThreads = true;
        }
    }
    std::lock_guard<std::mutex> lk(launchMutex);
    Pool& p = pool();
    p.body  = &body;
    p.nblk  = grid;
    for (uint32_t b = first; b < grid; ++b) {
        p.bid = b;
        pthread_barrier_wait(&p.start);
        pthread_barrier_wait(&p.stop);
    }
}
}  // namespace fhe_emu
