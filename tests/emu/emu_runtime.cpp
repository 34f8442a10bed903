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

void block_sync() { pthread_barrier_wait(&pool().sync); }
void* block_shared(size_t bytes) {
    if (bytes > sizeof(pool().shared))
        std::abort();
    return pool().shared;
}
void launch(uint32_t grid, uint32_t threads, const std::function<void()>& body) {
    if (threads != kLanes)
        std::abort();
    std::lock_guard<std::mutex> lk(launchMutex);
    Pool& p = pool();
    p.body  = &body;
    p.nblk  = grid;
    for (uint32_t b = 0; b < grid; ++b) {
        p.bid = b;
        pthread_barrier_wait(&p.start);
        pthread_barrier_wait(&p.stop);
    }
}
}  // namespace fhe_emu
