// microbench.hip — measurement tool (not part of the product library): establishes the two ceilings that
// bound the NTT on MI355X: (1) achievable HBM stream bandwidth, (2) 64-bit integer multiply issue rate.
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench.hip -o tools/microbench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

__global__ void copy16(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = in[i];
}
__global__ void read16(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    uint4 acc = {0, 0, 0, 0};
    for (; i < n; i += stride) { uint4 v = in[i]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
    if (acc.x == 0x12345 && acc.y == 7) out[0] = acc;
}
__global__ void write16(uint4* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    uint4 v = {1, 2, 3, (unsigned)i};
    for (; i < n; i += stride) out[i] = v;
}

template <int KIND>
__global__ void alu(uint64_t* out, uint64_t seed, int iters) {
    uint64_t a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = seed * (threadIdx.x + 1 + i * 977) + blockIdx.x;
    uint64_t w = seed | 1, wp = seed * 0x9E3779B97F4A7C15ull, q = (seed >> 4) | 1;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (KIND == 0) {  // 32x32+64 -> 64  (v_mad_u64_u32)
                a[i] = (uint64_t)(uint32_t)a[i] * (uint32_t)w + a[i];
            } else if (KIND == 1) {  // v_mul_lo_u32
                uint32_t x = (uint32_t)a[i]; x = x * (uint32_t)w + 1; a[i] = x;
            } else if (KIND == 2) {  // v_mul_hi_u32
                uint32_t x = (uint32_t)a[i]; x = __umulhi(x, (uint32_t)w) + 3; a[i] = x;
            } else if (KIND == 3) {  // 64x64 -> hi 64
                a[i] = __umul64hi(a[i], wp) + 5;
            } else if (KIND == 4) {  // 64x64 -> lo 64
                a[i] = a[i] * w + 7;
            } else if (KIND == 5) {  // lazy Shoup modmul (mulhi + 2 mullo + sub)
                uint64_t Q = __umul64hi(a[i], wp); a[i] = a[i] * w - Q * q;
            } else if (KIND == 6) {  // full Harvey butterfly on pairs
                if (i < 4) {
                    uint64_t twoq = q << 1;
                    uint64_t X = a[i] >= twoq ? a[i] - twoq : a[i];
                    uint64_t Q = __umul64hi(a[i + 4], wp); uint64_t T = a[i + 4] * w - Q * q;
                    a[i] = X + T; a[i + 4] = X - T + twoq;
                }
            } else if (KIND == 7) {  // 64-bit add (v_lshl_add_u64 / add_co pair)
                a[i] = a[i] + w + (a[i] >> 63);
            } else if (KIND == 8) {  // fp64 fma
                double d = __longlong_as_double(a[i]); d = d * 1.0000001 + 0.5; a[i] = __double_as_longlong(d);
            } else if (KIND == 9) {  // 24-bit multiply
                uint32_t x = (uint32_t)a[i]; x = __umul24(x, (uint32_t)w) + 1; a[i] = x;
            }
        }
    }
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s ^= a[i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static float timeit(hipStream_t st, int reps, const std::function<void()>& f) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}
#include <functional>

template <int KIND>
__global__ void alu_asm(uint64_t* out, uint64_t seed, int iters) {
    uint32_t a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (uint32_t)seed * (threadIdx.x + 1 + i * 977); b[i] = a[i] ^ 0x5555u; }
    uint64_t w64 = seed | 1;
    uint32_t w = (uint32_t)seed | 1;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (KIND == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(w));
            else if (KIND == 1) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b[i]));
            else if (KIND == 2) { uint64_t x = ((uint64_t)b[i] << 32) | a[i]; asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(x) : "v"(w64)); a[i] = (uint32_t)x; b[i] = (uint32_t)(x >> 32); }
            else if (KIND == 3) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(w) : );
            else if (KIND == 4) asm volatile("v_sub_co_u32 %0, vcc, %0, %2\n v_subb_co_u32 %1, vcc, %1, %2, vcc" : "+v"(a[i]), "+v"(b[i]) : "v"(w) : "vcc");
            else if (KIND == 5) { uint64_t x = ((uint64_t)b[i] << 32) | a[i]; asm volatile("v_cmp_ge_u64 vcc, %0, %1" : : "v"(x), "v"(w64) : "vcc"); }
            else if (KIND == 6) { uint64_t d; asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(d) : "v"(a[i]), "v"(w) : "vcc"); a[i] = (uint32_t)(d >> 32); }
            else if (KIND == 7) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(w));
            else if (KIND == 8) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(w));
            else if (KIND == 9) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(w));
        }
    }
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s ^= a[i] ^ ((uint64_t)b[i] << 32);
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main(int argc, char** argv) {
    size_t gb = argc > 1 ? atol(argv[1]) : 4;
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d,\n", p.name, p.multiProcessorCount, p.clockRate / 1000);
    size_t bytes = gb << 30, n = bytes / 16;
    uint4 *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
    hipStream_t st = 0;
    for (int blocks : {2048, 8192, 65536}) {
        float c = timeit(st, 5, [&] { hipLaunchKernelGGL(copy16, dim3(blocks), dim3(256), 0, st, a, b, n); });
        float r = timeit(st, 5, [&] { hipLaunchKernelGGL(read16, dim3(blocks), dim3(256), 0, st, a, b, n); });
        float w = timeit(st, 5, [&] { hipLaunchKernelGGL(write16, dim3(blocks), dim3(256), 0, st, b, n); });
        printf(" \"hbm_blocks_%d\": {\"copy_GBps_rw\": %.0f, \"read_GBps\": %.0f, \"write_GBps\": %.0f},\n", blocks,
               2.0 * bytes / c / 1e6, bytes / r / 1e6, bytes / w / 1e6);
    }
    uint64_t* out; const int blocks = 256 * 8, iters = 2000;
    CK(hipMalloc(&out, (size_t)blocks * 256 * 8));
    const char* names[] = {"mad_u64_u32", "mul_lo_u32", "mul_hi_u32", "mulhi64", "mullo64", "shoup_lazy", "butterfly",
                           "add64", "fma_f64", "mul_u24"};
    double opsPerIter[] = {8, 8, 8, 8, 8, 8, 4, 8, 8, 8};
#define RUN(K) { float ms = timeit(st, 3, [&] { hipLaunchKernelGGL(alu<K>, dim3(blocks), dim3(256), 0, st, out, 0x1234567ull, iters); }); \
      double ops = (double)blocks * 256 * iters * opsPerIter[K]; \
      printf(" \"alu_%s\": {\"Gops\": %.1f, \"cycles_per_wave_op_per_simd\": %.2f},\n", names[K], ops / ms / 1e6, \
             (double)p.multiProcessorCount * 4 * (p.clockRate * 1e3) * (ms * 1e-3) / (ops / 64)); }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9)
       const char* anames[] = {"v_add_u32", "v_mov_b32", "v_lshl_add_u64", "v_cndmask_b32", "v_sub_co+subb", "v_cmp_ge_u64",
                            "v_mad_u64_u32", "v_mul_u32_u24", "v_mad_u32_u24", "v_add3_u32"};
#define RUNA(K) { float ms = timeit(st, 3, [&] { hipLaunchKernelGGL(alu_asm<K>, dim3(blocks), dim3(256), 0, st, out, 0x1234567ull, iters); }); \
      double ops = (double)blocks * 256 * iters * 8; \
      printf(" \"asm_%s\": {\"Gops\": %.1f, \"cycles_per_wave_op_per_simd\": %.2f},\n", anames[K], ops / ms / 1e6, \
             (double)p.multiProcessorCount * 4 * (p.clockRate * 1e3) * (ms * 1e-3) / (ops / 64)); }
    RUNA(0) RUNA(1) RUNA(2) RUNA(3) RUNA(4) RUNA(5) RUNA(6) RUNA(7) RUNA(8) RUNA(9)
    // cache residency: write S bytes then read them back (separately timed), and rewrite the same S bytes repeatedly
    for (size_t mb : {16, 64, 128, 192, 512, 2048}) {
        size_t sb = mb << 20, sn = sb / 16;
        if (sb > bytes) break;
        float wt = 0, rt = 0;
        hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipEventRecord(e0, st));
            hipLaunchKernelGGL(write16, dim3(8192), dim3(256), 0, st, b, sn);
            CK(hipEventRecord(e1, st));
            hipLaunchKernelGGL(read16, dim3(8192), dim3(256), 0, st, b, a, sn);
            CK(hipEventRecord(e2, st)); CK(hipEventSynchronize(e2));
            float t1, t2; CK(hipEventElapsedTime(&t1, e0, e1)); CK(hipEventElapsedTime(&t2, e1, e2));
            if (rep >= 2) { wt += t1; rt += t2; }
        }
        printf(" \"cache_%zuMB\": {\"write_then_read_GBps\": [%.0f, %.0f]},\n", mb, sb / (wt / 4) / 1e6, sb / (rt / 4) / 1e6);
    }
    printf(" \"done\": true}\n");
    return 0;
}
