#!/bin/bash
# static instruction census of the NTT pass kernels (gfx950 assembly of the library's device code): total / VALU per kernel
# usage: tools/count_valu.sh [extra hipcc flags]
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
TMP=$(mktemp -d)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-result "$@" -x hip --cuda-device-only -S \
      "$ROOT/openfhe-development_amd/csrc/fhe_hip.cpp" -o "$TMP/dev.s"
python3 - "$TMP/dev.s" <<'PY'
import re, sys
from collections import Counter
cur, body = None, {}
for l in open(sys.argv[1]):
    m = re.match(r'^(_ZN3fhe\S+):', l)
    if m:
        cur = m.group(1)
        body[cur] = []
        continue
    if cur and re.match(r'^\s*s_endpgm', l):
        body[cur].append('s_endpgm')
        cur = None
        continue
    if cur:
        t = l.split(';')[0].strip()
        if t and not t.startswith('.') and not t.endswith(':'):
            body[cur].append(t.split()[0])
for name, ops in body.items():
    if 'ntt_static_kernel' not in name and 'poly_mul_row' not in name:
        continue
    valu = [o for o in ops if o.startswith('v_')]
    c = Counter(valu)
    print(f"{name}: {len(ops)} instructions, {len(valu)} VALU; " + ", ".join(f"{k} {v}" for k, v in c.most_common(10)))
PY
rm -rf "$TMP"
