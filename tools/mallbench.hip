// mallbench.hip — measurement tool (not part of the product library).  Round 5, VERDICT item 1(ii): does the 256 MiB
// Infinity Cache (memory-side L3) hold a tower between the two passes of a transform?  Three questions, answered by time
// alone (the L2-side FETCH/WRITE counters include Infinity-Cache hits, MI355X_MICROARCH.md "HBM"):
//   rmw    one kernel rewrites S bytes in place (32 KiB tile per workgroup, 16 B per lane), repeated: GB/s moved (2S per pass)
//          against S — S <= 32 MiB sits in the L2s, S <= 256 MiB could sit in the Infinity Cache, S = 4 GiB is HBM;
//   read   the same, loads only;
//   chunk  the two-pass schedule with copy-speed kernels: a 4 GiB buffer, per chunk of C bytes kernel A (strided 16-row
//          access like the column pass) then kernel B (contiguous like the row pass), both in place; ms for the whole buffer
//          against C.  C = 4 GiB is today's schedule (A over everything, then B over everything).
// Build: hipcc --offload-arch=gfx950 -O3 tools/mallbench.hip -o tools/mallbench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <functional>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int kTileBytes = 32768;  // one workgroup: 256 lanes x 8 x 16 B

// contiguous tile: lane t touches 16-byte pieces t, t+256, ... of the tile
template <bool WRITE>
__global__ void __launch_bounds__(256) tile_rmw(uint4* __restrict__ x, uint4* sink) {
    uint4* p = x + (size_t)blockIdx.x * (kTileBytes / 16) + threadIdx.x;
    uint4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = p[k * 256];
    if (WRITE) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { v[k].x += 1; p[k * 256] = v[k]; }
    } else {
        uint4 a = v[0];
#pragma unroll
        for (int k = 1; k < 8; ++k) { a.x ^= v[k].x; a.y ^= v[k].y; a.z ^= v[k].z; a.w ^= v[k].w; }
        if (a.x == 0x12345 && a.y == 7 && a.z == 9) sink[0] = a;
    }
}
// "column" tile: a limb-row of 512 KiB = 16 rows of 32 KiB; workgroup c of the limb touches 2 KiB of each of the 16 rows
__global__ void __launch_bounds__(256) col_rmw(uint4* __restrict__ x) {
    const size_t limb = blockIdx.x >> 4, c = blockIdx.x & 15;
    uint4* p = x + limb * (524288 / 16) + c * (2048 / 16) + (threadIdx.x & 127);
    const int half = threadIdx.x >> 7;  // two half-workgroups take 8 rows each
    uint4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = p[(size_t)(half * 8 + k) * (32768 / 16)];
#pragma unroll
    for (int k = 0; k < 8; ++k) { v[k].x += 1; p[(size_t)(half * 8 + k) * (32768 / 16)] = v[k]; }
}

static float timeit(hipStream_t st, int reps, const std::function<void()>& f) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); f(); CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ms / reps;
}

int main(int argc, char** argv) {
    const size_t total = (size_t)4 << 30;
    uint4 *x, *sink;
    CK(hipMalloc(&x, total)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(x, 1, total));
    hipStream_t st; CK(hipStreamCreate(&st));
    printf("{\"tool\": \"mallbench\", \"rmw\": [\n");
    const size_t mib = 1 << 20;
    std::vector<size_t> sizes = {8, 16, 24, 32, 48, 64, 96, 128, 160, 192, 224, 256, 320, 384, 512, 1024, 4096};
    for (size_t s : sizes) {
        const size_t S = s * mib; const unsigned nb = (unsigned)(S / kTileBytes);
        const int reps = s >= 1024 ? 5 : 40;
        float w = timeit(st, reps, [&] { tile_rmw<true><<<nb, 256, 0, st>>>(x, sink); });
        float r = timeit(st, reps, [&] { tile_rmw<false><<<nb, 256, 0, st>>>(x, sink); });
        printf(" {\"MiB\": %zu, \"rmw_ms\": %.4f, \"rmw_GBps_moved\": %.0f, \"read_ms\": %.4f, \"read_GBps\": %.0f},\n", s, w, 2.0 * S / w / 1e6, r,
               (double)S / r / 1e6);
        fflush(stdout);
    }
    printf(" null],\n \"chunk\": [\n");
    std::vector<size_t> chunks = {32, 48, 64, 96, 128, 192, 256, 512, 1024, 4096};
    for (int variant = 0; variant < 2; ++variant)  // 0: A = contiguous tiles too, 1: A = column-pass pattern
        for (size_t c : chunks) {
            const size_t C = c * mib; const unsigned nb = (unsigned)(C / kTileBytes); const size_t nchunks = total / C;
            float ms = timeit(st, 3, [&] {
                for (size_t i = 0; i < nchunks; ++i) {
                    uint4* p = x + i * (C / 16);
                    if (variant) col_rmw<<<nb, 256, 0, st>>>(p); else tile_rmw<true><<<nb, 256, 0, st>>>(p, sink);
                    tile_rmw<true><<<nb, 256, 0, st>>>(p, sink);
                }
            });
            printf(" {\"A\": \"%s\", \"chunk_MiB\": %zu, \"ms_for_4GiB_two_passes\": %.3f, \"GBps_moved\": %.0f},\n", variant ? "column" : "tile", c, ms,
                   4.0 * total / ms / 1e6);
            fflush(stdout);
        }
    // two streams, alternate chunks: the tails of one chunk's launches are filled by the other's
    for (size_t c : {(size_t)32, (size_t)48, (size_t)64, (size_t)96}) {
        hipStream_t s2; CK(hipStreamCreate(&s2));
        const size_t C = c * mib; const unsigned nb = (unsigned)(C / kTileBytes); const size_t nchunks = total / C;
        hipEvent_t e0, e1, ej; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&ej));
        auto pass = [&] {
            CK(hipEventRecord(ej, st)); CK(hipStreamWaitEvent(s2, ej, 0));
            for (size_t i = 0; i < nchunks; ++i) {
                hipStream_t s = (i & 1) ? s2 : st;
                uint4* p = x + i * (C / 16);
                col_rmw<<<nb, 256, 0, s>>>(p);
                tile_rmw<true><<<nb, 256, 0, s>>>(p, sink);
            }
            CK(hipEventRecord(ej, s2)); CK(hipStreamWaitEvent(st, ej, 0));
        };
        pass(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < 3; ++r) pass();
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
        printf(" {\"A\": \"column, two streams\", \"chunk_MiB\": %zu, \"ms_for_4GiB_two_passes\": %.3f, \"GBps_moved\": %.0f},\n", c, ms, 4.0 * total / ms / 1e6);
        CK(hipStreamDestroy(s2));
    }
    printf(" null]}\n");
    return 0;
}
