// launch.h — the few spellings that differ between the real gfx950 build (hipcc) and the test-only
// lane emulator (tests/emu/, g++: one OS thread per lane, a barrier for s_barrier).
// The product library is ALWAYS the hipcc build; FHE_EMU exists only so that kernel index logic can be
// unit-tested on a machine without a GPU.  There is no CPU fallback in the product.
#ifndef FHE_LAUNCH_H
#define FHE_LAUNCH_H

#ifdef FHE_EMU
#include "emu_runtime.h"  // tests/emu/emu_runtime.h
#define FHE_GLOBAL static
#define FHE_DEV static inline
#define FHE_LAUNCH_BOUNDS(n)
#define FHE_LAUNCH_BOUNDS2(n, w)
#define FHE_TID (fhe_emu::tls.tid)
#define FHE_BID (fhe_emu::tls.bid)
#define FHE_NBLK (fhe_emu::tls.nblk)
#define FHE_SYNC() fhe_emu::block_sync()
// exchange between the lanes of ONE wave: the device needs no instruction (LDS operations of a wave execute in order), the emulator's
// lanes are OS threads and meet at the workgroup barrier (every lane of the workgroup reaches the same FHE_WAVE_SYNC)
#define FHE_WAVE_SYNC() fhe_emu::block_sync()
#define FHE_SHARED_U64(name, n) uint64_t* name = reinterpret_cast<uint64_t*>(fhe_emu::block_shared(sizeof(uint64_t) * (n)))
#define FHE_UNIFORM(x) (x)
#define FHE_ULOAD64(p, i) ((p)[i])
#define FHE_ULOADF64(p, i) ((p)[i])
#else
#include <hip/hip_runtime.h>
#define FHE_GLOBAL __global__
#define FHE_DEV __device__ __forceinline__
#define FHE_LAUNCH_BOUNDS(n) __launch_bounds__(n)
#define FHE_LAUNCH_BOUNDS2(n, w) __launch_bounds__(n, w)
#define FHE_TID (threadIdx.x)
#define FHE_BID (blockIdx.x)
#define FHE_NBLK (gridDim.x)
#define FHE_SYNC() __syncthreads()
#define FHE_WAVE_SYNC()                                         \
    do {                                                        \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  \
        __builtin_amdgcn_wave_barrier();                        \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  \
    } while (0)
#define FHE_SHARED_U64(name, n) __shared__ uint64_t name[n]
#define FHE_UNIFORM(x) (__builtin_amdgcn_readfirstlane(x))
// Read-only, wave-uniform table word.  Reading it through the constant address space tells the compiler that the
// table cannot alias the kernel's own stores, so the access becomes an s_load (scalar cache, wide, hoistable)
// instead of a per-lane global_load that is waited for on the spot.
#define FHE_ULOAD64(p, i) (((const __attribute__((address_space(4))) uint64_t*)(p))[i])
#define FHE_ULOADF64(p, i) (((const __attribute__((address_space(4))) double*)(p))[i])
#endif

#endif
