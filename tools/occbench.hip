// occbench.hip — measurement tool (not part of the product library), round 6.
// Three questions behind the 8-residues-per-lane row pass:
//  (1) integer-issue rate of the butterfly's instruction mix as a function of waves per SIMD (1..8): does the
//      v_mad_u64_u32 pipe saturate at 4 waves per SIMD or does it take more?
//  (2) HBM copy rate of a 4096-word tile moved by a 512-thread workgroup as (a) 8 coalesced 8-byte accesses per lane
//      (512 B contiguous per wave instruction) or (b) 4 16-byte accesses per lane on 64 contiguous bytes per lane
//      (every wave instruction touches 32 lines) — whether the last exchange before a store can be skipped;
//  (3) the LDS bank model (ds_read/write_b64 conflict-free iff the 32 lanes of a half-wave hit 32 distinct
//      8-byte bank pairs): exchange patterns of the new kernel with and without their skews.
// Build: hipcc --offload-arch=gfx950 -O3 tools/occbench.hip -o tools/occbench ; run: tools/occbench > out.json
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

// ---- (1) issue rate ---------------------------------------------------------------------------------------------
// 8 independent v_mad_u64_u32 per repetition on v[2:17], scalar multiplier; <= 32 VGPRs so that 8 waves fit a SIMD
#define MAD8 \
    "v_mad_u64_u32 v[2:3], s[40:41], v2, s52, v[2:3]\n\t" \
    "v_mad_u64_u32 v[4:5], s[40:41], v4, s52, v[4:5]\n\t" \
    "v_mad_u64_u32 v[6:7], s[40:41], v6, s52, v[6:7]\n\t" \
    "v_mad_u64_u32 v[8:9], s[40:41], v8, s52, v[8:9]\n\t" \
    "v_mad_u64_u32 v[10:11], s[40:41], v10, s52, v[10:11]\n\t" \
    "v_mad_u64_u32 v[12:13], s[40:41], v12, s52, v[12:13]\n\t" \
    "v_mad_u64_u32 v[14:15], s[40:41], v14, s52, v[14:15]\n\t" \
    "v_mad_u64_u32 v[16:17], s[40:41], v16, s52, v[16:17]\n\t"
#define ADD8 \
    "v_add_u32 v2, v2, v3\n\t" "v_add_u32 v4, v4, v5\n\t" "v_add_u32 v6, v6, v7\n\t" "v_add_u32 v8, v8, v9\n\t" \
    "v_add_u32 v10, v10, v11\n\t" "v_add_u32 v12, v12, v13\n\t" "v_add_u32 v14, v14, v15\n\t" "v_add_u32 v16, v16, v17\n\t"
// one forward butterfly (truncated Shoup quotient, the text of tools/gen_ntt_asm.py fwd_stream) on a = v[2:3], b = v[4:5],
// temporaries v[6:15], twiddle in s[52:55], constants s[56:59]; single slot: the carry hazards are s_nop
#define BFLY1 \
    "v_mad_u64_u32 v[6:7], s[40:41], v5, s54, 0\n\t" \
    "v_mad_u64_u32 v[8:9], s[40:41], v4, s53, 0\n\t" \
    "v_mad_u64_u32 v[10:11], s[42:43], v4, s55, v[6:7]\n\t" \
    "v_mad_u64_u32 v[8:9], s[40:41], v5, s52, v[8:9]\n\t" \
    "v_mov_b32 v12, v11\n\t" \
    "v_cndmask_b32_e64 v13, 0, 1, s[42:43]\n\t" \
    "v_mad_u64_u32 v[14:15], s[40:41], v5, s55, v[12:13]\n\t" \
    "v_mad_u64_u32 v[6:7], s[40:41], v4, s52, v[2:3]\n\t" \
    "v_lshl_add_u64 v[4:5], v[2:3], 1, s[58:59]\n\t" \
    "v_mad_u64_u32 v[8:9], s[40:41], v14, s57, v[8:9]\n\t" \
    "v_mad_u64_u32 v[8:9], s[40:41], v15, s56, v[8:9]\n\t" \
    "v_add_u32 v7, v7, v8\n\t" \
    "v_mad_u64_u32 v[2:3], s[40:41], v14, s56, v[6:7]\n\t" \
    "v_sub_co_u32_e64 v4, s[42:43], v4, v2\n\t" \
    "s_nop 1\n\t" \
    "v_subb_co_u32_e64 v5, s[42:43], v5, v3, s[42:43]\n\t"
// the same butterfly twice on disjoint registers (a2 = v[16:17], b2 = v[18:19], temporaries v[20:29]), interleaved by hand
#define BFLY2 \
    "v_mad_u64_u32 v[6:7], s[40:41], v5, s54, 0\n\t" \
    "v_mad_u64_u32 v[20:21], s[40:41], v19, s54, 0\n\t" \
    "v_mad_u64_u32 v[8:9], s[40:41], v4, s53, 0\n\t" \
    "v_mad_u64_u32 v[22:23], s[40:41], v18, s53, 0\n\t" \
    "v_mad_u64_u32 v[10:11], s[42:43], v4, s55, v[6:7]\n\t" \
    "v_mad_u64_u32 v[24:25], s[44:45], v18, s55, v[20:21]\n\t" \
    "v_mad_u64_u32 v[8:9], s[40:41], v5, s52, v[8:9]\n\t" \
    "v_mad_u64_u32 v[22:23], s[40:41], v19, s52, v[22:23]\n\t" \
    "v_mov_b32 v12, v11\n\t" \
    "v_mov_b32 v26, v25\n\t" \
    "v_cndmask_b32_e64 v13, 0, 1, s[42:43]\n\t" \
    "v_cndmask_b32_e64 v27, 0, 1, s[44:45]\n\t" \
    "v_mad_u64_u32 v[14:15], s[40:41], v5, s55, v[12:13]\n\t" \
    "v_mad_u64_u32 v[28:29], s[40:41], v19, s55, v[26:27]\n\t" \
    "v_mad_u64_u32 v[6:7], s[40:41], v4, s52, v[2:3]\n\t" \
    "v_mad_u64_u32 v[20:21], s[40:41], v18, s52, v[16:17]\n\t" \
    "v_lshl_add_u64 v[4:5], v[2:3], 1, s[58:59]\n\t" \
    "v_lshl_add_u64 v[18:19], v[16:17], 1, s[58:59]\n\t" \
    "v_mad_u64_u32 v[8:9], s[40:41], v14, s57, v[8:9]\n\t" \
    "v_mad_u64_u32 v[22:23], s[40:41], v28, s57, v[22:23]\n\t" \
    "v_mad_u64_u32 v[8:9], s[40:41], v15, s56, v[8:9]\n\t" \
    "v_mad_u64_u32 v[22:23], s[40:41], v29, s56, v[22:23]\n\t" \
    "v_add_u32 v7, v7, v8\n\t" \
    "v_add_u32 v21, v21, v22\n\t" \
    "v_mad_u64_u32 v[2:3], s[40:41], v14, s56, v[6:7]\n\t" \
    "v_mad_u64_u32 v[16:17], s[40:41], v28, s56, v[20:21]\n\t" \
    "v_sub_co_u32_e64 v4, s[42:43], v4, v2\n\t" \
    "v_sub_co_u32_e64 v18, s[44:45], v18, v16\n\t" \
    "s_nop 0\n\t" \
    "v_subb_co_u32_e64 v5, s[42:43], v5, v3, s[42:43]\n\t" \
    "v_subb_co_u32_e64 v19, s[44:45], v19, v17, s[44:45]\n\t"

#define CLOB "v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21", \
             "v22","v23","v24","v25","v26","v27","v28","v29","s40","s41","s42","s43","s44","s45","s52","s53","s54","s55","s56","s57","s58","s59"
#define INIT \
    "v_mov_b32 v2, %0\n\tv_mov_b32 v3, %0\n\tv_mov_b32 v4, %0\n\tv_mov_b32 v5, %0\n\tv_mov_b32 v6, %0\n\tv_mov_b32 v7, %0\n\t" \
    "v_mov_b32 v8, %0\n\tv_mov_b32 v9, %0\n\tv_mov_b32 v10, %0\n\tv_mov_b32 v11, %0\n\tv_mov_b32 v12, %0\n\tv_mov_b32 v13, 0\n\t" \
    "v_mov_b32 v14, %0\n\tv_mov_b32 v15, %0\n\tv_mov_b32 v16, %0\n\tv_mov_b32 v17, %0\n\tv_mov_b32 v18, %0\n\tv_mov_b32 v19, %0\n\t" \
    "v_mov_b32 v20, %0\n\tv_mov_b32 v21, %0\n\tv_mov_b32 v22, %0\n\tv_mov_b32 v23, %0\n\tv_mov_b32 v24, %0\n\tv_mov_b32 v25, %0\n\t" \
    "v_mov_b32 v26, %0\n\tv_mov_b32 v27, 0\n\tv_mov_b32 v28, %0\n\tv_mov_b32 v29, %0\n\t" \
    "s_mov_b32 s52, 0x1234567\n\ts_mov_b32 s53, 0x2345671\n\ts_mov_b32 s54, 0x3456712\n\ts_mov_b32 s55, 0x4567123\n\t" \
    "s_mov_b32 s56, 0x5671234\n\ts_mov_b32 s57, 0x6712345\n\ts_mov_b32 s58, 0x7123456\n\ts_mov_b32 s59, 0x1234567\n\t"

#define R4(x) x x x x
#define R16(x) R4(x) R4(x) R4(x) R4(x)
template <int KIND>
__global__ void __launch_bounds__(256) issue(unsigned long long* out, unsigned seed, int iters) {
    unsigned sd = seed + threadIdx.x * 977u + blockIdx.x;
    asm volatile(INIT : : "v"(sd) : CLOB);
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) asm volatile(R16(MAD8) : : : CLOB);            // 128 mads
        if (KIND == 1) asm volatile(R16(ADD8) : : : CLOB);            // 128 adds
        if (KIND == 2) asm volatile(R4(BFLY1) R4(BFLY1) : : : CLOB);  // 8 butterflies, one slot
        if (KIND == 3) asm volatile(R4(BFLY2) : : : CLOB);            // 8 butterflies, two slots
    }
    unsigned r;
    asm volatile("v_add_u32 %0, v2, v16" : "=v"(r) : : CLOB);
    if (r == 0x7fffffffu && seed == 1) out[0] = r;
}

// ---- (2) tile copies ----------------------------------------------------------------------------------------------
// one 4096-word tile per 512-thread workgroup; LOADK / STOREK: 0 = 8 coalesced 8-byte accesses (word t + 512 k),
// 1 = 4 16-byte accesses on the lane's 8 consecutive words
template <int LOADK, int STOREK>
__global__ void __launch_bounds__(512) tilecopy(const uint64_t* __restrict__ in, uint64_t* __restrict__ out) {
    const uint32_t t = threadIdx.x;
    const uint64_t* s = in + ((size_t)blockIdx.x << 12);
    uint64_t* d       = out + ((size_t)blockIdx.x << 12);
    uint64_t r[8];
    if (LOADK == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = s[t + 512 * k];
    }
    else {
        const ulonglong2* s2 = reinterpret_cast<const ulonglong2*>(s + 8 * t);
#pragma unroll
        for (int k = 0; k < 4; ++k) { ulonglong2 v = s2[k]; r[2 * k] = v.x; r[2 * k + 1] = v.y; }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] += 1;
    if (STOREK == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) d[t + 512 * k] = r[k];
    }
    else {
        ulonglong2* d2 = reinterpret_cast<ulonglong2*>(d + 8 * t);
#pragma unroll
        for (int k = 0; k < 4; ++k) d2[k] = ulonglong2{r[2 * k], r[2 * k + 1]};
    }
}

// ---- (3) LDS exchange patterns ----------------------------------------------------------------------------------
// wave-private exchange: write in layout X, read in layout Y, `iters` times; per wave a region of 576 words.
// fields of the wave's 9-bit index: c = bits 6..8, k = bits 3..5, m = bits 0..2; lane l.
// MODE 0: exchange 2 with its skew (write reg c lanes (k,m) at 72c+8k+m; read reg k lanes (c,m))
// MODE 1: exchange 2 without skew (64c+8k+m)
// MODE 2: exchange 3 with its scheme (write reg k lanes (c,m) at Fc(c)+33k+m; read reg m lanes (c,k)), Fc = 8(c&3)+264(c>>2)
// MODE 3: exchange 3 without skew (64c+8k+m)
template <int MODE>
__global__ void __launch_bounds__(512) ldsx(unsigned long long* out, int iters) {
    __shared__ uint64_t lds[8 * 576];
    const uint32_t t = threadIdx.x, w = t >> 6, l = t & 63;
    uint64_t* L = lds + w * 576;
    uint64_t r[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = t * 8 + k;
    const uint32_t hi = l >> 3, lo = l & 7;
    uint32_t wb, rb;
    if (MODE == 0) { wb = 8 * hi + lo; rb = 72 * hi + lo; }
    if (MODE == 1) { wb = 8 * hi + lo; rb = 64 * hi + lo; }
    if (MODE == 2) { wb = 8 * (hi & 3) + 264 * (hi >> 2) + lo; rb = 8 * (hi & 3) + 264 * (hi >> 2) + 33 * lo; }
    if (MODE == 3) { wb = 64 * hi + lo; rb = 64 * hi + 8 * lo; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int wi = MODE == 0 ? 72 * k : MODE == 1 ? 64 * k : MODE == 2 ? 33 * k : 8 * k;
            L[wb + wi] = r[k];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int ri = MODE == 0 ? 8 * k : MODE == 1 ? 8 * k : MODE == 2 ? k : k;
            r[k] = L[rb + ri] + 1;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    uint64_t s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += r[k];
    if (s == 0x123456789ull) out[0] = s;
}

static float timeit(void (*f)(void*), void* p, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(p); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) f(p);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    unsigned long long* out; CK(hipMalloc(&out, 1 << 20));
    printf("{\"device\": \"%s\", \"cus\": %d,\n \"issue\": [\n", p.name, cus);
    const char* names[4] = {"v_mad_u64_u32 x128", "v_add_u32 x128", "fwd butterfly, one slot (8 per body)", "fwd butterfly, two slots interleaved (8 per body)"};
    const double perBody[4] = {128, 128, 8, 8};
    bool first = true;
    for (int kind = 0; kind < 4; ++kind)
        for (int W : {1, 2, 3, 4, 5, 6, 8}) {
            const int iters = 2000;
            struct A { int kind, blocks, iters; unsigned long long* out; } a{kind, cus * W, iters, out};
            auto f = [](void* v) {
                A* a = (A*)v;
                switch (a->kind) {
                    case 0: hipLaunchKernelGGL(issue<0>, dim3(a->blocks), dim3(256), 0, 0, a->out, 3u, a->iters); break;
                    case 1: hipLaunchKernelGGL(issue<1>, dim3(a->blocks), dim3(256), 0, 0, a->out, 3u, a->iters); break;
                    case 2: hipLaunchKernelGGL(issue<2>, dim3(a->blocks), dim3(256), 0, 0, a->out, 3u, a->iters); break;
                    case 3: hipLaunchKernelGGL(issue<3>, dim3(a->blocks), dim3(256), 0, 0, a->out, 3u, a->iters); break;
                }
            };
            const float ms = timeit(f, &a, 5);
            // per SIMD: W waves each run iters * perBody items; ns per item per SIMD
            const double ns = ms * 1e6 / ((double)W * iters * perBody[kind]);
            printf("%s  {\"seq\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"ns_per_item_per_simd\": %.4f}", first ? "" : ",\n", names[kind], W, ms, ns);
            first = false;
        }
    printf("\n ],\n \"tilecopy\": [\n");
    {
        const size_t words = (size_t)1 << 29;  // 4 GiB in, 4 GiB out
        uint64_t *in, *o2; CK(hipMalloc(&in, words * 8)); CK(hipMalloc(&o2, words * 8));
        CK(hipMemset(in, 1, words * 8)); CK(hipMemset(o2, 0, words * 8));
        struct A { int kind; const uint64_t* in; uint64_t* out; unsigned blocks; } a{0, in, o2, (unsigned)(words >> 12)};
        auto f = [](void* v) {
            A* a = (A*)v;
            switch (a->kind) {
                case 0: hipLaunchKernelGGL((tilecopy<0, 0>), dim3(a->blocks), dim3(512), 0, 0, a->in, a->out); break;
                case 1: hipLaunchKernelGGL((tilecopy<1, 0>), dim3(a->blocks), dim3(512), 0, 0, a->in, a->out); break;
                case 2: hipLaunchKernelGGL((tilecopy<0, 1>), dim3(a->blocks), dim3(512), 0, 0, a->in, a->out); break;
                case 3: hipLaunchKernelGGL((tilecopy<1, 1>), dim3(a->blocks), dim3(512), 0, 0, a->in, a->out); break;
            }
        };
        const char* nm[4] = {"load 8x8B coalesced, store 8x8B coalesced", "load 4x16B per-lane-contiguous, store coalesced",
                             "load coalesced, store 4x16B per-lane-contiguous", "load and store 4x16B per-lane-contiguous"};
        for (int k = 0; k < 4; ++k) {
            a.kind = k;
            const float ms = timeit(f, &a, 10);
            printf("%s  {\"variant\": \"%s\", \"ms\": %.4f, \"GBps_moved\": %.1f}", k ? ",\n" : "", nm[k], ms, 2.0 * words * 8 / ms / 1e6);
        }
        CK(hipFree(in)); CK(hipFree(o2));
    }
    printf("\n ],\n \"lds_exchange\": [\n");
    {
        const char* nm[4] = {"exchange 2 (B->C) with skew 72c+8k+m", "exchange 2 without skew", "exchange 3 (C->D) with scheme 8(c&3)+264(c>>2)+33k+m", "exchange 3 without skew"};
        for (int k = 0; k < 4; ++k) {
            struct A { int kind, blocks, iters; unsigned long long* out; } a{k, cus * 4, 2000, out};
            auto f = [](void* v) {
                A* a = (A*)v;
                switch (a->kind) {
                    case 0: hipLaunchKernelGGL(ldsx<0>, dim3(a->blocks), dim3(512), 0, 0, a->out, a->iters); break;
                    case 1: hipLaunchKernelGGL(ldsx<1>, dim3(a->blocks), dim3(512), 0, 0, a->out, a->iters); break;
                    case 2: hipLaunchKernelGGL(ldsx<2>, dim3(a->blocks), dim3(512), 0, 0, a->out, a->iters); break;
                    case 3: hipLaunchKernelGGL(ldsx<3>, dim3(a->blocks), dim3(512), 0, 0, a->out, a->iters); break;
                }
            };
            const float ms = timeit(f, &a, 5);
            // per CU: 4 workgroups x 2000 exchanges of 32 KiB written + 32 KiB read
            printf("%s  {\"pattern\": \"%s\", \"ms\": %.4f, \"ns_per_tile_exchange_per_cu\": %.2f}", k ? ",\n" : "", nm[k], ms, ms * 1e6 / (4.0 * 2000));
        }
    }
    printf("\n ]\n}\n");
    return 0;
}
