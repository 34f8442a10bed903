// What CMake would generate from /root/reference/configure/config_core.in for the default
// build (MATHBACKEND 4, NATIVE_SIZE 64, OpenMP on, no NATIVEOPT, no REDUCED_NOISE).
#ifndef __CMAKE_GENERATED_CONFIG_CORE_H__
#define __CMAKE_GENERATED_CONFIG_CORE_H__
#define WITH_BE2
#define WITH_BE4
#define WITH_OPENMP
#define CKKS_M_FACTOR 1
#define HAVE_INT128 1
#define HAVE_INT64 1
#define MATHBACKEND 4
#define NATIVEINT 64
#endif
