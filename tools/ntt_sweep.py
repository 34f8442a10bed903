#!/usr/bin/env python3
"""tuning helper: per-pass and total NTT times for the current FHE_NTT_T1 / FHE_NTT_CHUNK environment."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from openfhe_amd import fhe_hip as fh
sys.path.insert(0, ROOT)
import bench

logN = int(os.environ.get("SWEEP_LOGN", "16")); L = int(os.environ.get("SWEEP_L", "30")); B = int(os.environ.get("SWEEP_B", "256"))
lib = fh.Lib()
q, psi = lib.dcrt_chain(logN, L, 60)
ctx = fh.Context(lib, logN, q, psi)
x = bench.fill_random_tower(ctx, q, B, 1)
ms = C.c_float()
res = {"T1": os.environ.get("FHE_NTT_T1", "default"), "chunk": os.environ.get("FHE_NTT_CHUNK", "0"), "B": B}
for d, nm in ((10, "fwd_col"), (11, "fwd_row"), (12, "inv_row"), (13, "inv_col"), (0, "fwd"), (1, "inv"), (2, "fwd_inv")):
    if d >= 10 and res["chunk"] != "0":
        continue
    lib.check(lib.L.fhe_time_ntt(ctx.h, x, None, L, B, d, 5, None, C.byref(ms)))
    res[nm + "_ms"] = round(ms.value, 3)
alg = 4.0 * 8 * (1 << logN) * L * B
res["fwd_inv_GBps"] = round(alg / (res["fwd_inv_ms"] * 1e-3) / 1e9, 1)
print(json.dumps(res))
