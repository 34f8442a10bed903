// energybench.hip — measurement tool (not part of the product library), round 6.
// The headline NTT leg runs at the 1400 W package cap (tools/power_probe.sh): time = energy / cap.  This tool prices instructions in
// that regime: each kernel keeps every SIMD busy with one instruction kind (8 waves per SIMD, register operands with random bits)
// for ~1.2 s in 100 ms launches; the issue rate of the LAST launches (after the power controller has settled) is what the cap
// leaves of it.  tools/energybench.sh samples rocm-smi beside it.
// Build: hipcc --offload-arch=gfx950 -O3 tools/energybench.hip -o tools/energybench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

#define CLOB "v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","s40","s41","s42","s43","s52","s53","s54","s55"
#define R4(x) x x x x
#define R16(x) R4(x) R4(x) R4(x) R4(x)
#define R64(x) R16(x) R16(x) R16(x) R16(x)
// 8 independent instructions per macro
#define I8(fmt) fmt(2,3,10,11,18) fmt(4,5,12,13,20) fmt(6,7,14,15,22) fmt(8,9,16,17,24) fmt(2,3,12,13,18) fmt(4,5,14,15,20) fmt(6,7,16,17,22) fmt(8,9,10,11,24)
#define S_(x) #x
#define MADVV(a,b,c,d,e)  "v_mad_u64_u32 v[" S_(a) ":" S_(b) "], s[40:41], v" S_(c) ", v" S_(d) ", v[" S_(a) ":" S_(b) "]\n\t"
#define MADVS(a,b,c,d,e)  "v_mad_u64_u32 v[" S_(a) ":" S_(b) "], s[40:41], v" S_(c) ", s52, v[" S_(a) ":" S_(b) "]\n\t"
#define MADV0(a,b,c,d,e)  "v_mad_u64_u32 v[" S_(a) ":" S_(b) "], s[40:41], v" S_(c) ", v" S_(d) ", 0\n\t"
#define MULLO(a,b,c,d,e)  "v_mul_lo_u32 v" S_(a) ", v" S_(c) ", v" S_(d) "\n\t"
#define MULHI(a,b,c,d,e)  "v_mul_hi_u32 v" S_(a) ", v" S_(c) ", v" S_(d) "\n\t"
#define ADD32(a,b,c,d,e)  "v_add_u32 v" S_(a) ", v" S_(c) ", v" S_(d) "\n\t"
#define ADD64(a,b,c,d,e)  "v_lshl_add_u64 v[" S_(a) ":" S_(b) "], v[" S_(c) ":" S_(d) "], 0, v[" S_(a) ":" S_(b) "]\n\t"
#define CNDM(a,b,c,d,e)   "v_cndmask_b32_e64 v" S_(a) ", v" S_(c) ", v" S_(d) ", s[42:43]\n\t"
#define SUBCO(a,b,c,d,e)  "v_sub_co_u32_e64 v" S_(a) ", s[40:41], v" S_(c) ", v" S_(d) "\n\t"
#define MUL24(a,b,c,d,e)  "v_mul_u32_u24_e32 v" S_(a) ", v" S_(c) ", v" S_(d) "\n\t"
#define FMA64(a,b,c,d,e)  "v_fma_f64 v[" S_(a) ":" S_(b) "], v[" S_(c) ":" S_(d) "], v[" S_(c) ":" S_(d) "], v[" S_(a) ":" S_(b) "]\n\t"
#define FMA32(a,b,c,d,e)  "v_fma_f32 v" S_(a) ", v" S_(c) ", v" S_(d) ", v" S_(a) "\n\t"
#define MOV32(a,b,c,d,e)  "v_mov_b32 v" S_(a) ", v" S_(c) "\n\t"

#define INIT \
    "s_mov_b32 s52, 0x6789abcd\n\ts_mov_b32 s53, 0x9E3779B1\n\ts_mov_b32 s54, 0x85EBCA6B\n\ts_mov_b32 s42, 0x55555555\n\ts_mov_b32 s43, 0x33333333\n\t" \
    "v_mov_b32 v2, %0\n\tv_mul_lo_u32 v3, v2, s53\n\tv_mul_lo_u32 v4, v3, s54\n\tv_mul_lo_u32 v5, v4, s53\n\t" \
    "v_mul_lo_u32 v6, v5, s54\n\tv_mul_lo_u32 v7, v6, s53\n\tv_mul_lo_u32 v8, v7, s54\n\tv_mul_lo_u32 v9, v8, s53\n\t" \
    "v_mul_lo_u32 v10, v9, s54\n\tv_mul_lo_u32 v11, v10, s53\n\tv_mul_lo_u32 v12, v11, s54\n\tv_mul_lo_u32 v13, v12, s53\n\t" \
    "v_mul_lo_u32 v14, v13, s54\n\tv_mul_lo_u32 v15, v14, s53\n\tv_mul_lo_u32 v16, v15, s54\n\tv_mul_lo_u32 v17, v16, s53\n\t" \
    "v_mov_b32 v18, v2\n\tv_mov_b32 v19, v3\n\tv_mov_b32 v20, v4\n\tv_mov_b32 v21, v5\n\tv_mov_b32 v22, v6\n\tv_mov_b32 v23, v7\n\tv_mov_b32 v24, v8\n\tv_mov_b32 v25, v9\n\t"

template <int KIND>
__global__ void __launch_bounds__(256) burn(unsigned* out, unsigned seed, int iters) {
    unsigned sd = seed + threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    asm volatile(INIT : : "v"(sd) : CLOB);
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) asm volatile(R16(I8(MADVV)) : : : CLOB);
        if (KIND == 1) asm volatile(R16(I8(MADVS)) : : : CLOB);
        if (KIND == 2) asm volatile(R16(I8(MADV0)) : : : CLOB);
        if (KIND == 3) asm volatile(R16(I8(MULLO)) : : : CLOB);
        if (KIND == 4) asm volatile(R16(I8(MULHI)) : : : CLOB);
        if (KIND == 5) asm volatile(R16(I8(ADD32)) : : : CLOB);
        if (KIND == 6) asm volatile(R16(I8(ADD64)) : : : CLOB);
        if (KIND == 7) asm volatile(R16(I8(CNDM)) : : : CLOB);
        if (KIND == 8) asm volatile(R16(I8(SUBCO)) : : : CLOB);
        if (KIND == 9) asm volatile(R16(I8(MUL24)) : : : CLOB);
        if (KIND == 10) asm volatile(R16(I8(FMA64)) : : : CLOB);
        if (KIND == 11) asm volatile(R16(I8(FMA32)) : : : CLOB);
        if (KIND == 12) asm volatile(R16(I8(MOV32)) : : : CLOB);
    }
    unsigned r;
    asm volatile("v_add_u32 %0, v2, v16" : "=v"(r) : : CLOB);
    if (r == 0x7fffffffu && seed == 1) out[0] = r;
}

// ---- data movement at full rate: HBM copy (4 GiB -> 4 GiB), LDS exchange (the row pass's X2 pattern), L2-resident 16-byte reads ----
__global__ void __launch_bounds__(512) hbmcopy(const uint64_t* __restrict__ in, uint64_t* __restrict__ out) {
    const uint32_t t = threadIdx.x;
    const uint64_t* s = in + ((size_t)blockIdx.x << 12);
    uint64_t* d       = out + ((size_t)blockIdx.x << 12);
    uint64_t r[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = s[t + 512 * k];
#pragma unroll
    for (int k = 0; k < 8; ++k) d[t + 512 * k] = r[k] + 1;
}
__global__ void __launch_bounds__(512) ldsloop(unsigned* out, int iters) {
    __shared__ uint64_t lds[8 * 576];
    const uint32_t t = threadIdx.x, w = t >> 6, l = t & 63;
    uint64_t* L = lds + w * 576;
    uint64_t r[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = (uint64_t)t * 0x9E3779B97F4A7C15ull + k;
    const uint32_t hi = l >> 3, lo = l & 7;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) L[8 * hi + lo + 72 * k] = r[k];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = L[72 * hi + lo + 8 * k] + 1;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    uint64_t s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += r[k];
    if (s == 0x123456789ull) out[0] = (unsigned)s;
}
__global__ void __launch_bounds__(256) l2read(const uint4* __restrict__ tab, unsigned* out, int iters) {
    // every workgroup re-reads the same 1 MiB table (L2 / L1 resident), 16 bytes per lane, 1 KiB contiguous per wave instruction
    const uint32_t t = threadIdx.x;
    uint4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint4 v = tab[((it * 8 + k) * 256 + t + blockIdx.x * 64) & 65535];
            acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = acc.x;
}

static double now() { struct timespec ts; clock_gettime(CLOCK_REALTIME, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

int main(int argc, char** argv) {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    unsigned* out; CK(hipMalloc(&out, 4096));
    const char* names[13] = {"v_mad_u64_u32 v*v+v64", "v_mad_u64_u32 v*s+v64", "v_mad_u64_u32 v*v+0", "v_mul_lo_u32", "v_mul_hi_u32", "v_add_u32",
                             "v_lshl_add_u64 (64-bit add)", "v_cndmask_b32_e64", "v_sub_co_u32_e64", "v_mul_u32_u24_e32", "v_fma_f64", "v_fma_f32", "v_mov_b32"};
    const int blocks = cus * 8;  // 8 waves per SIMD
    printf("{\"device\": \"%s\", \"results\": [\n", p.name);
    for (int kind = 0; kind < 13; ++kind) {
        // calibrate iterations for ~100 ms per launch
        int iters = 2000;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        auto launch = [&](int it) {
            switch (kind) {
#define C(K) case K: hipLaunchKernelGGL(burn<K>, dim3(blocks), dim3(256), 0, 0, out, 3u, it); break;
                C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12)
#undef C
            }
        };
        CK(hipEventRecord(e0)); launch(iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        iters = (int)(iters * 100.0 / ms);
        const double t0 = now();
        double lastRate = 0, firstRate = 0;
        for (int rep = 0; rep < 12; ++rep) {
            CK(hipEventRecord(e0)); launch(iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            // wave-instructions per second per SIMD: 8 waves x iters x 128
            const double rate = 8.0 * iters * 128.0 / (ms * 1e-3);
            if (rep == 0) firstRate = rate;
            if (rep >= 8) lastRate += rate / 4;
        }
        const double t1 = now();
        printf("  {\"instr\": \"%s\", \"t_start\": %.3f, \"t_end\": %.3f, \"ginstr_per_s_per_simd_first\": %.4f, \"ginstr_per_s_per_simd_settled\": %.4f}%s\n",
               names[kind], t0, t1, firstRate / 1e9, lastRate / 1e9, kind == 12 ? "" : ",");
        fflush(stdout);
    }
    printf("],\n \"memory\": [\n");
    {
        const size_t words = (size_t)1 << 29;
        uint64_t *in, *o2; CK(hipMalloc(&in, words * 8)); CK(hipMalloc(&o2, words * 8));
        CK(hipMemset(in, 0x5a, words * 8)); CK(hipMemset(o2, 0, words * 8));
        uint4* tab; CK(hipMalloc(&tab, 1 << 20)); CK(hipMemset(tab, 0x3c, 1 << 20));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int kind = 0; kind < 3; ++kind) {
            const double t0 = now();
            double bytes = 0, ms_total = 0;
            int reps = 0;
            while (now() - t0 < 1.5) {
                CK(hipEventRecord(e0));
                if (kind == 0) { for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(hbmcopy, dim3((unsigned)(words >> 12)), dim3(512), 0, 0, in, o2); }
                if (kind == 1) hipLaunchKernelGGL(ldsloop, dim3(cus * 4), dim3(512), 0, 0, out, 40000);
                if (kind == 2) hipLaunchKernelGGL(l2read, dim3(cus * 8), dim3(256), 0, 0, tab, out, 20000);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (now() - t0 > 0.7) {
                    ms_total += ms; ++reps;
                    bytes += kind == 0 ? 20.0 * 2 * words * 8 : kind == 1 ? (double)cus * 4 * 40000 * 65536.0 : (double)cus * 8 * 20000.0 * 8 * 256 * 16;
                }
            }
            const double t1 = now();
            const char* nm[3] = {"HBM copy 4 GiB -> 4 GiB (bytes read + written)", "LDS exchange X2 pattern (bytes written + read)", "L2-resident 16-byte reads of a 1 MiB table"};
            printf("  {\"what\": \"%s\", \"t_start\": %.3f, \"t_end\": %.3f, \"GBps\": %.1f}%s\n", nm[kind], t0, t1, bytes / (ms_total * 1e-3) / 1e9, kind == 2 ? "" : ",");
            fflush(stdout);
        }
    }
    printf("]}\n");
    return 0;
}
