// stridebench.hip — measurement tool (not part of the product library), round 6.
// HBM rate of the COLUMN-pass access pattern as a function of its depth T1: a limb of N = 2^16 words is read and written by tiles
// of 2^T1 rows x 2^(12-T1) consecutive columns (row stride N / 2^T1 words), 512 threads x 8 words per lane, the whole batch
// (4 GiB) per launch.  T1 = 4: 2 KiB segments (today's column pass) ... T1 = 8: 128-byte segments.
// Build: hipcc --offload-arch=gfx950 -O3 tools/stridebench.hip -o tools/stridebench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

template <int T1, int XCD>
__global__ void __launch_bounds__(512) colcopy(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, uint32_t batchLimbs) {
    constexpr uint32_t logN = 16, logC = 12 - T1, tilesPerLimb = 1u << (logN - 12);
    uint32_t tile = blockIdx.x;
    if (XCD) {  // an XCD keeps one (limb-tile) across the batch, like the NTT kernels
        const uint32_t xcd = tile & 7u, i = tile >> 3;
        const uint32_t b = i % batchLimbs, pairIdx = i / batchLimbs;
        tile = b * tilesPerLimb + (pairIdx * 8u + xcd);
    }
    const uint32_t limb = tile / tilesPerLimb, tr = tile % tilesPerLimb;
    const size_t base   = ((size_t)limb << logN) + ((size_t)tr << logC);
    const uint32_t t = threadIdx.x, col = t & ((1u << logC) - 1u), r0 = t >> logC;  // 512 >> logC rows per instruction
    constexpr uint32_t rowsPerInst = 512u >> logC;
    uint64_t r[8];
#pragma unroll
    for (int k = 0; k < 8; ++k)
        r[k] = in[base + (((size_t)(r0 + rowsPerInst * k)) << (logN - T1)) + col];
#pragma unroll
    for (int k = 0; k < 8; ++k)
        out[base + (((size_t)(r0 + rowsPerInst * k)) << (logN - T1)) + col] = r[k] + 1;
}

template <int T1, int XCD>
static float run(const uint64_t* in, uint64_t* out, uint32_t limbs) {
    const uint32_t blocks = limbs * 16;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((colcopy<T1, XCD>), dim3(blocks), dim3(512), 0, 0, in, out, limbs);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < 10; ++i)
        hipLaunchKernelGGL((colcopy<T1, XCD>), dim3(blocks), dim3(512), 0, 0, in, out, limbs);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / 10;
}

int main() {
    const uint32_t limbs = 8192;  // x 512 KiB = 4 GiB
    const size_t words = (size_t)limbs << 16;
    uint64_t *in, *out; CK(hipMalloc(&in, words * 8)); CK(hipMalloc(&out, words * 8));
    CK(hipMemset(in, 1, words * 8)); CK(hipMemset(out, 0, words * 8));
    printf("{\"bytes_moved\": %.0f, \"results\": [\n", 2.0 * words * 8);
#define ONE(T1, X, last) { float ms = run<T1, X>(in, out, limbs); printf("  {\"T1\": %d, \"segment_bytes\": %d, \"xcd_order\": %d, \"ms\": %.4f, \"GBps_moved\": %.1f}%s\n", T1, 8 << (12 - T1), X, ms, 2.0 * words * 8 / ms / 1e6, last ? "" : ","); }
    ONE(4, 0, 0) ONE(5, 0, 0) ONE(6, 0, 0) ONE(7, 0, 0) ONE(8, 0, 0)
    ONE(4, 1, 0) ONE(5, 1, 0) ONE(6, 1, 0) ONE(7, 1, 0) ONE(8, 1, 1)
    printf("]}\n");
    return 0;
}
