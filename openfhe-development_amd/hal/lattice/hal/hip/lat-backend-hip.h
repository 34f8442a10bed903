// lat-backend-hip.h — the aliases of lattice/hal/lat-backend.h (reference: src/core/include/lattice/hal/lat-backend.h:39-100)
// with lbcrypto::DCRTPoly bound to the device-resident implementation DCRTPolyHipImpl (dcrtpoly-hip.h).  ILParams,
// ILDCRTParams, Poly and NativePoly stay the reference's own classes: DCRTPolyInterface fixes the tower type to
// PolyImpl<NativeVector>, and pke exchanges NativePoly objects with DCRTPoly all over its encoding and decryption code.
#ifndef LBCRYPTO_INC_LATTICE_HAL_LAT_BACKEND_HIP_H
#define LBCRYPTO_INC_LATTICE_HAL_LAT_BACKEND_HIP_H

#define ILPARAMS_IMPLEMENTATION     "lattice/hal/default/ilparams.h"
#define ILDCRTPARAMS_IMPLEMENTATION "lattice/hal/default/ildcrtparams.h"
#define POLY_IMPLEMENTATION         "lattice/hal/default/poly-impl.h"
// the host mirror inside DCRTPolyHipImpl is the reference's DCRTPolyImpl: its member definitions are needed as well
#define DCRTPOLY_IMPLEMENTATION     "lattice/hal/hip/dcrtpoly-hip-impl.h"

#define MAKE_ILPARAMS_TYPE(T)     template class ILParamsImpl<T>;
#define MAKE_ILDCRTPARAMS_TYPE(T) template class ILDCRTParams<T>;
#define MAKE_POLY_TYPE(T)         template class PolyImpl<T>;
#define MAKE_DCRTPOLY_TYPE(T)     \
    template class DCRTPolyImpl<T>; \
    template class DCRTPolyHipImpl<T>;

#include "lattice/hal/default/ilparams.h"
#include "lattice/hal/default/ildcrtparams.h"
#include "lattice/hal/default/poly.h"
#include "lattice/hal/default/dcrtpoly.h"
#include "lattice/hal/hip/dcrtpoly-hip.h"

namespace lbcrypto {

using ILNativeParams = ILParamsImpl<NativeInteger>;
using ILParams       = ILParamsImpl<BigInteger>;
using Poly           = PolyImpl<BigVector>;
using NativePoly     = PolyImpl<NativeVector>;
using DCRTPoly       = DCRTPolyHipImpl<BigVector>;

#ifdef WITH_BE2
using M2Params     = ILParamsImpl<M2Integer>;
using M2DCRTParams = ILDCRTParams<M2Integer>;
using M2Poly       = PolyImpl<M2Vector>;
using M2DCRTPoly   = DCRTPolyHipImpl<M2Vector>;
#else
using M2Params     = void;
using M2DCRTParams = void;
using M2Poly       = void;
using M2DCRTPoly   = void;
#endif

#ifdef WITH_BE4
using M4Params     = ILParamsImpl<M4Integer>;
using M4DCRTParams = ILDCRTParams<M4Integer>;
using M4Poly       = PolyImpl<M4Vector>;
using M4DCRTPoly   = DCRTPolyHipImpl<M4Vector>;
#else
using M4Params     = void;
using M4DCRTParams = void;
using M4Poly       = void;
using M4DCRTPoly   = void;
#endif

#ifdef WITH_NTL
using M6Params     = ILParamsImpl<M6Integer>;
using M6DCRTParams = ILDCRTParams<M6Integer>;
using M6Poly       = PolyImpl<M6Vector>;
using M6DCRTPoly   = DCRTPolyHipImpl<M6Vector>;
#else
using M6Params     = void;
using M6DCRTParams = void;
using M6Poly       = void;
using M6DCRTPoly   = void;
#endif

}  // namespace lbcrypto

#endif
