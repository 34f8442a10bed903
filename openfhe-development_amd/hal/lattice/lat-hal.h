// lattice/lat-hal.h — backend selection for a build of OpenFHE against the MI355X HIP backend.
//
// This file shadows the reference's src/core/include/lattice/lat-hal.h (which is a single line,
// `#include "lattice/hal/lat-backend.h"`): put  -I <this repo>/openfhe-development_amd/hal  IN FRONT of the reference's
// include directories (CMake: WITH_HIP) and every translation unit of core / binfhe / pke picks the HIP backend below
// instead of lattice/hal/lat-backend.h:39-61.  Nothing else in the reference changes.
#ifndef LBCRYPTO_INC_LATTICE_LAT_HAL_H
#define LBCRYPTO_INC_LATTICE_LAT_HAL_H

#include "lattice/hal/hip/lat-backend-hip.h"

#endif
