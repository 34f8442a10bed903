#!/bin/bash
# Builds the product library (hipcc, gfx950) and the test-side native pieces.
#   ./build.sh hip     -> openfhe-development_amd/csrc/libfhe_hip.so   (THE product; no CPU fallback exists)
#   ./build.sh emu     -> tests/emu/libfhe_emu.so   (TEST ONLY: lane emulator build of the same sources)
#   ./build.sh oracle  -> oracle/libfhe_oracle.so   (TEST ONLY: C restatement of the reference)
#   ./build.sh ref     -> oracle/_ref/*.so          (TEST ONLY: the reference itself, needs /root/reference)
#   ./build.sh hal     -> openfhe-development_amd/hal/_build/*.so: the reference's sources compiled against the HIP backend
#                         of lbcrypto::DCRTPoly (lattice/hal/hip/), + the test programs of tests/hal incl. the reference's own pke unit
#                         tests on both backends (needs /root/reference)
#   ./build.sh all     -> hip + emu + oracle (+ ref + hal when /root/reference exists)
set -e
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
CSRC="$ROOT/openfhe-development_amd/csrc"
what="${1:-all}"
build_hip() {
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fopenmp -Wno-unused-result \
        -x hip "$CSRC/fhe_hip.cpp" -o "$CSRC/libfhe_hip.so"
}
build_emu() {
  g++ -std=c++17 -O2 -g -DFHE_EMU -fopenmp -fPIC -shared -ffp-contract=off -Wall -Wno-unknown-pragmas \
      -I"$ROOT/tests/emu" -I"$CSRC" "$CSRC/fhe_hip.cpp" "$ROOT/tests/emu/emu_runtime.cpp" \
      -o "$ROOT/tests/emu/libfhe_emu.so" -lpthread
}
build_oracle() { make -s -C "$ROOT/oracle" oracle; }
build_ref() { make -s -j"$(nproc)" -C "$ROOT/oracle" ref; }
build_hal() {
  make -s -j"$(nproc)" -C "$ROOT/openfhe-development_amd/hal" && make -s -j3 -C "$ROOT/tests/hal" &&
    make -s -j"$(nproc)" -C "$ROOT/tests/hal" -f Makefile.ut &&  # the reference's own pke unit tests on both backends
    "$ROOT/integration/build_patched.sh" > /dev/null  # the same backend bound at SOURCE level (integration/with_hip.patch): what bench.py prefers
}
case "$what" in
  hip) build_hip ;;
  emu) build_emu ;;
  oracle) build_oracle ;;
  ref) build_ref ;;
  hal) build_ref; build_hal ;;
  all) build_hip; build_emu; build_oracle; if [ -d /root/reference/src ]; then build_ref; build_hal; fi ;;
  *) echo "unknown target $what"; exit 2 ;;
esac
