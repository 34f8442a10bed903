// rt.h — device runtime spellings (HIP in the product; heap + lane emulator under FHE_EMU for tests).
#ifndef FHE_RT_H
#define FHE_RT_H
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <string>

#include "launch.h"

#include <atomic>
namespace fhe {
namespace rt {
// one counter per FHE_LAUNCH site (a relaxed increment per launch): fhe_launch_stats() sums them by kernel name
struct LaunchSite {
    const char* kernel;
    std::atomic<uint64_t> n{0};
    LaunchSite* next;
    static std::atomic<LaunchSite*>& head() {
        static std::atomic<LaunchSite*> h{nullptr};
        return h;
    }
    explicit LaunchSite(const char* k) : kernel(k) {
        next = head().load();
        while (!head().compare_exchange_weak(next, this)) {
        }
    }
};
}  // namespace rt
}  // namespace fhe
#define FHE_COUNT_LAUNCH(kernel)                      \
    static fhe::rt::LaunchSite fhe_launch_site_(#kernel); \
    fhe_launch_site_.n.fetch_add(1, std::memory_order_relaxed)

#ifdef FHE_EMU
#include <chrono>
#include <cstdlib>
namespace fhe {
namespace rt {
typedef void* stream_t;
inline bool ok() { return true; }
inline int device_count() { return 1; }
inline uint32_t cu_count(int) { return 3; }
inline const char* set_device(int) { return nullptr; }
inline const char* dmalloc(void** p, size_t bytes) {
    *p = std::malloc(bytes ? bytes : 1);
    return *p ? nullptr : "malloc failed";
}
inline const char* dfree(void* p) {
    std::free(p);
    return nullptr;
}
inline const char* mem_info(size_t* freeBytes, size_t* totalBytes) {  // (the emulator's "device" is host memory: no bound reported)
    *freeBytes = *totalBytes = (size_t)1 << 46;
    return nullptr;
}
inline const char* h2d(void* d, const void* s, size_t n, stream_t) {
    std::memcpy(d, s, n);
    return nullptr;
}
inline const char* d2h(void* d, const void* s, size_t n, stream_t) {
    std::memcpy(d, s, n);
    return nullptr;
}
inline const char* d2d(void* d, const void* s, size_t n, stream_t) {
    std::memmove(d, s, n);
    return nullptr;
}
inline const char* d2d_2d(void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t height, stream_t) {
    for (size_t r = 0; r < height; ++r)
        std::memmove((char*)d + r * dpitch, (const char*)s + r * spitch, width);
    return nullptr;
}
inline const char* dzero_2d(void* d, size_t pitch, size_t width, size_t height, stream_t) {
    for (size_t r = 0; r < height; ++r)
        std::memset((char*)d + r * pitch, 0, width);
    return nullptr;
}
typedef void* graph_t;
inline const char* stream_create(stream_t* s) {
    *s = nullptr;
    return nullptr;
}
inline const char* stream_destroy(stream_t) { return nullptr; }
inline const char* stream_wait(stream_t, stream_t) { return nullptr; }  // (the emulator runs every launch synchronously)
typedef void* event_t;
inline const char* event_create(event_t* e) {
    *e = (void*)1;
    return nullptr;
}
inline const char* event_record(event_t, stream_t) { return nullptr; }
inline const char* stream_wait_event(stream_t, event_t) { return nullptr; }
inline const char* event_destroy(event_t) { return nullptr; }
inline const char* dzero(void* d, size_t n, stream_t) {
    std::memset(d, 0, n);
    return nullptr;
}
inline const char* capture_begin(stream_t) { return "stream capture needs the HIP build"; }
inline const char* capture_end(stream_t, graph_t*) { return "stream capture needs the HIP build"; }
inline const char* graph_launch(graph_t, stream_t) { return "stream capture needs the HIP build"; }
inline const char* graph_destroy(graph_t) { return nullptr; }
inline const char* sync(stream_t) { return nullptr; }
inline const char* device_sync() { return nullptr; }
inline const char* last_launch_error() { return nullptr; }
struct Timer {
    std::chrono::steady_clock::time_point t0;
    const char* start(stream_t) {
        t0 = std::chrono::steady_clock::now();
        return nullptr;
    }
    const char* stop(stream_t, float* ms) {
        *ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
        return nullptr;
    }
};
}  // namespace rt
}  // namespace fhe
// FHE_LAUNCH: a kernel without barriers (its lanes run one after the other on the launching thread); FHE_LAUNCH_BARRIER: a kernel whose
// lanes exchange through LDS (one OS thread per lane).  The same launch on the device.
#define FHE_LAUNCH(kernel, grid, stream, ...)                                                     \
    do {                                                                                          \
        FHE_COUNT_LAUNCH(kernel);                                                                 \
        fhe_emu::launch((uint32_t)(grid), fhe::kThreads, [=]() { kernel(__VA_ARGS__); }, false); \
    } while (0)
#define FHE_LAUNCH_BARRIER(kernel, grid, stream, ...)                                            \
    do {                                                                                         \
        FHE_COUNT_LAUNCH(kernel);                                                                \
        fhe_emu::launch((uint32_t)(grid), fhe::kThreads, [=]() { kernel(__VA_ARGS__); }, true); \
    } while (0)
// (a workgroup of `threads` lanes instead of fhe::kThreads: the 8-residues-per-lane row pass, ntt_row8.h)
#define FHE_LAUNCH_BARRIER_N(kernel, grid, threads, stream, ...)                                  \
    do {                                                                                         \
        FHE_COUNT_LAUNCH(kernel);                                                                \
        fhe_emu::launch((uint32_t)(grid), (uint32_t)(threads), [=]() { kernel(__VA_ARGS__); }, true); \
    } while (0)
#else
#include <hip/hip_runtime.h>
namespace fhe {
namespace rt {
typedef hipStream_t stream_t;
inline const char* err(hipError_t e) { return e == hipSuccess ? nullptr : hipGetErrorString(e); }
inline int device_count() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}
inline const char* set_device(int d) { return err(hipSetDevice(d)); }
inline uint32_t cu_count(int d) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, d) != hipSuccess)
        return 256;
    return (uint32_t)p.multiProcessorCount;
}
inline const char* dmalloc(void** p, size_t bytes) { return err(hipMalloc(p, bytes ? bytes : 1)); }
inline const char* dfree(void* p) { return err(hipFree(p)); }
inline const char* mem_info(size_t* freeBytes, size_t* totalBytes) { return err(hipMemGetInfo(freeBytes, totalBytes)); }
inline const char* h2d(void* d, const void* s, size_t n, stream_t st) {
    return err(hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, st));
}
inline const char* d2h(void* d, const void* s, size_t n, stream_t st) {
    return err(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, st));
}
inline const char* d2d(void* d, const void* s, size_t n, stream_t st) {
    return err(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToDevice, st));
}
inline const char* d2d_2d(void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t height,
                          stream_t st) {
    return err(hipMemcpy2DAsync(d, dpitch, s, spitch, width, height, hipMemcpyDeviceToDevice, st));
}
inline const char* dzero_2d(void* d, size_t pitch, size_t width, size_t height, stream_t st) {
    return err(hipMemset2DAsync(d, pitch, 0, width, height, st));
}
typedef hipGraphExec_t graph_t;
inline const char* stream_create(stream_t* s) { return err(hipStreamCreateWithFlags(s, hipStreamNonBlocking)); }
inline const char* stream_destroy(stream_t s) { return err(hipStreamDestroy(s)); }
// `waiter` waits (on the device, without blocking the host) for everything enqueued on `signaler` so far.  A waiting stream
// keeps the event's state at the time of the call, so a small per-thread ring of events is re-recorded freely.
inline const char* stream_wait(stream_t waiter, stream_t signaler) {
    constexpr int kRing = 32;
    static thread_local hipEvent_t ring[kRing] = {};
    static thread_local int at = 0;
    hipEvent_t& e = ring[at];
    at = (at + 1) % kRing;
    if (!e)
        if (auto m = err(hipEventCreateWithFlags(&e, hipEventDisableTiming)))
            return m;
    if (auto m = err(hipEventRecord(e, signaler)))
        return m;
    return err(hipStreamWaitEvent(waiter, e, 0));
}
// a completion mark: recorded on a stream NOW, waited for (on the device) by another stream later — exact, unlike stream_wait, which makes
// the waiter follow everything the signaler has enqueued by the time of the call
typedef hipEvent_t event_t;
inline const char* event_create(event_t* e) { return err(hipEventCreateWithFlags(e, hipEventDisableTiming)); }
inline const char* event_record(event_t e, stream_t s) { return err(hipEventRecord(e, s)); }
inline const char* stream_wait_event(stream_t s, event_t e) { return err(hipStreamWaitEvent(s, e, 0)); }
inline const char* event_destroy(event_t e) { return err(hipEventDestroy(e)); }
inline const char* dzero(void* d, size_t n, stream_t st) { return err(hipMemsetAsync(d, 0, n, st)); }
inline const char* capture_begin(stream_t s) { return err(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal)); }
inline const char* capture_end(stream_t s, graph_t* out) {
    hipGraph_t g = nullptr;
    if (auto e = err(hipStreamEndCapture(s, &g)))
        return e;
    auto e = err(hipGraphInstantiate(out, g, nullptr, nullptr, 0));
    (void)hipGraphDestroy(g);
    return e;
}
inline const char* graph_launch(graph_t g, stream_t s) { return err(hipGraphLaunch(g, s)); }
inline const char* graph_destroy(graph_t g) { return err(hipGraphExecDestroy(g)); }
inline const char* sync(stream_t st) { return err(hipStreamSynchronize(st)); }
inline const char* device_sync() { return err(hipDeviceSynchronize()); }
inline const char* last_launch_error() { return err(hipGetLastError()); }
struct Timer {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const char* start(stream_t st) {
        if (auto e = err(hipEventCreate(&e0)))
            return e;
        if (auto e = err(hipEventCreate(&e1)))
            return e;
        return err(hipEventRecord(e0, st));
    }
    const char* stop(stream_t st, float* ms) {
        if (auto e = err(hipEventRecord(e1, st)))
            return e;
        if (auto e = err(hipEventSynchronize(e1)))
            return e;
        auto e = err(hipEventElapsedTime(ms, e0, e1));
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        return e;
    }
};
}  // namespace rt
}  // namespace fhe
#define FHE_LAUNCH(kernel, grid, stream, ...)                                                                            \
    do {                                                                                                                 \
        FHE_COUNT_LAUNCH(kernel);                                                                                        \
        hipLaunchKernelGGL(kernel, dim3((uint32_t)(grid)), dim3(fhe::kThreads), 0, (hipStream_t)(stream), __VA_ARGS__); \
    } while (0)
#define FHE_LAUNCH_BARRIER FHE_LAUNCH  // (the distinction only matters to the lane emulator of the tests)
#define FHE_LAUNCH_BARRIER_N(kernel, grid, threads, stream, ...)                                                             \
    do {                                                                                                                    \
        FHE_COUNT_LAUNCH(kernel);                                                                                           \
        hipLaunchKernelGGL(kernel, dim3((uint32_t)(grid)), dim3((uint32_t)(threads)), 0, (hipStream_t)(stream), __VA_ARGS__); \
    } while (0)
#endif
#endif
