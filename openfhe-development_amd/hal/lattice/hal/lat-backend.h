// lattice/hal/lat-backend.h — shadows the reference's file of the same name (src/core/include/lattice/hal/lat-backend.h:39-100).
// Headers that sit directly in src/core/include (openfhecore.h) reach the reference's own lattice/lat-hal.h through the
// includer-relative lookup; its `#include "lattice/hal/lat-backend.h"` then resolves here (this directory is first on the
// include path), so every translation unit gets the HIP backend's aliases whichever lat-hal.h it came through.
#ifndef LBCRYPTO_INC_LATTICE_HAL_LAT_BACKEND_H
#define LBCRYPTO_INC_LATTICE_HAL_LAT_BACKEND_H
#include "lattice/hal/hip/lat-backend-hip.h"
#endif
