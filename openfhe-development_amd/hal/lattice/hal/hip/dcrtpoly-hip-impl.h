// dcrtpoly-hip-impl.h — what lattice.cpp includes as DCRTPOLY_IMPLEMENTATION: the reference's member definitions of the host
// mirror class (DCRTPolyImpl) and the HIP backend's own (all in dcrtpoly-hip.h, header-only templates).
#ifndef LBCRYPTO_INC_LATTICE_HAL_HIP_DCRTPOLY_HIP_IMPL_H
#define LBCRYPTO_INC_LATTICE_HAL_HIP_DCRTPOLY_HIP_IMPL_H
#include "lattice/hal/default/dcrtpoly-impl.h"
#include "lattice/hal/hip/dcrtpoly-hip.h"
#endif
