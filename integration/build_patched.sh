#!/bin/bash
# Applies integration/with_hip.patch to a COPY of the reference's sources and builds OpenFHE from that copy against the HIP backend of
# DCRTPoly with plain compiler flags (-DWITH_HIP -DFHE_HIP_PATCHED_PKE, the backend's include directory in front): no objcopy, no
# -fno-inline-functions — what an upstream tree carrying the patch would do.  Then builds the shim test program against the result.
#   integration/build_patched.sh [/root/reference]   -> integration/_build/{tree/, lib/libOPENFHE*_hip.so, lib/libfhe_boot_batch_hip.so,
#                                                       shim_ckks_hip_patched}  (what bench.py and the GPU suite prefer when present)
set -e
REF="${1:-/root/reference}"
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
ROOT="$(cd "$HERE/.." && pwd)"
B="$HERE/_build"
rm -rf "$B/tree"
mkdir -p "$B/tree/src" "$B/lib"
for d in core binfhe pke; do
  mkdir -p "$B/tree/src/$d"
  cp -r "$REF/src/$d/include" "$REF/src/$d/lib" "$B/tree/src/$d/"
done
# (the patch also carries the build option: the three CMakeLists.txt it edits travel with the sources; this script builds with make)
cp "$REF/CMakeLists.txt" "$B/tree/CMakeLists.txt"
for d in core pke; do cp "$REF/src/$d/CMakeLists.txt" "$B/tree/src/$d/CMakeLists.txt"; done
(cd "$B/tree" && patch -p1 -s < "$HERE/with_hip.patch")
make -s -j"$(nproc)" -C "$ROOT/openfhe-development_amd/hal" PATCHED=1 REF="$B/tree" OUT="$B/lib" "$B/lib/libOPENFHEpke_hip.so" "$B/lib/libfhe_boot_batch_hip.so"
HAL="$ROOT/openfhe-development_amd/hal"
STUB="$ROOT/third_party_stubs"
T="$B/tree"
g++ -std=c++17 -O2 -DNDEBUG -fopenmp -fPIC -DPARALLEL -DMATHBACKEND=4 -DOPENFHE_VERSION=1.5.1 -Wno-parentheses -w -DWITH_HIP -DFHE_HIP_PATCHED_PKE \
    -I"$HAL" -I"$ROOT/include" -I"$STUB/stub" -I"$STUB/gen" -I"$T/src/core/include" -I"$T/src/core/lib" -I"$T/src/binfhe/include" -I"$T/src/binfhe/lib" \
    -I"$T/src/pke/include" -I"$T/src/pke/lib" "$ROOT/tests/hal/shim_ckks.cpp" -o "$B/shim_ckks_hip_patched" \
    -L"$B/lib" -lOPENFHEpke_hip -lOPENFHEbinfhe_hip -lOPENFHEcore_hip -Wl,-rpath,"$B/lib"
echo "integration: built $B/shim_ckks_hip_patched from the patched tree"
